#!/bin/bash
# round 5, seventh GPU session: GPU test tier, operator profile with the library-made gray copy, n = 15 / 16 / 23 guarded vs exact
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/op_profile.py > $O/op_profile.txt 2>&1 || { tail -20 $O/op_profile.txt; exit 1; }
cat $O/op_profile.txt
timeout -k 10 300 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || { tail -20 $O/pcie_rate.txt; exit 1; }
grep -v amdgpu $O/pcie_rate.txt | sed -n 3,24p
for cfg in "--frames 600 --n-ac 15 --delta 20" "--frames 600 --n-ac 16 --delta 20" "--frames 600 --n-ac 15 --delta 8" "--frames 600 --n-ac 16 --delta 8" "--frames 600 --n-ac 23 --delta 8"; do
  echo "== $cfg: guarded, then exact" >> $O/ab_n15_n16.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode guarded --rounds 7 $V/libsvsdct.so 2>&1 | grep -E "embed +med" >> $O/ab_n15_n16.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode exact --rounds 7 $V/libsvsdct.so 2>&1 | grep -E "embed +med" >> $O/ab_n15_n16.txt
done
cut -c1-120 $O/ab_n15_n16.txt
