#!/bin/bash
# round 5, fourth GPU session: staging without the rings (tests first), operator profile, placement x kernel-configuration A/B
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/op_profile.py > $O/op_profile.txt 2>&1 || { tail -20 $O/op_profile.txt; exit 1; }
cat $O/op_profile.txt
timeout -k 10 300 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || { tail -20 $O/pcie_rate.txt; exit 1; }
sed -n 3,25p $O/pcie_rate.txt
for i in 1 2; do timeout -k 10 400 python tools/placement_ab.py >> $O/placement_ab.txt 2>&1 || { tail -20 $O/placement_ab.txt; exit 1; }; done
cat $O/placement_ab.txt
SVSDCT_LIB=secure-video-steganography-using-ecc-and-dct_amd/lib/variants/libsvsdct_exp.so timeout -k 10 300 python tools/placement_probe.py --pairs 3 --contig --tag contig > $O/placement_contig.txt 2>&1 || tail -5 $O/placement_contig.txt
cat $O/placement_contig.txt
