#!/bin/bash
# round 5, sixth GPU session: staging mode variance (fresh process x3, then after a heavy GPU job), operator profile, n = 15 vs 16
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for i in 1 2 3; do echo "-- fresh process $i" >> $O/stage_mode.txt; timeout -k 10 200 python tools/stage_mode_probe.py >> $O/stage_mode.txt 2>&1 || { tail $O/stage_mode.txt; exit 1; }; done
timeout -k 10 300 python tools/placement_ab.py --pairs 3 > $O/heavy.txt 2>&1
echo "-- after tools/placement_ab.py (30 GB allocated and freed, 1 min of kernels)" >> $O/stage_mode.txt
timeout -k 10 200 python tools/stage_mode_probe.py >> $O/stage_mode.txt 2>&1
grep -v amdgpu $O/stage_mode.txt
timeout -k 10 300 python tools/op_profile.py > $O/op_profile.txt 2>&1 || { tail -20 $O/op_profile.txt; exit 1; }
grep -E "==|proses|raw" $O/op_profile.txt
for cfg in "--frames 600 --n-ac 15 --delta 20" "--frames 600 --n-ac 16 --delta 20" "--frames 600 --n-ac 15 --delta 8" "--frames 600 --n-ac 16 --delta 8" "--frames 600 --n-ac 23 --delta 8"; do
  echo "== ab $cfg guarded vs exact" >> $O/ab_n15_n16.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode guarded --rounds 7 $V/libsvsdct.so 2>&1 | grep -E "embed med" | sed 's/ | extract.*| / | /' >> $O/ab_n15_n16.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode exact --rounds 7 $V/libsvsdct.so 2>&1 | grep -E "embed med" | sed 's/ | extract.*| / | /;s/^base/exact/' >> $O/ab_n15_n16.txt
done
cat $O/ab_n15_n16.txt
