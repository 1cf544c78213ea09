#!/bin/bash
# round 6, first GPU session: parity tier, then old-vs-new one-row embed kernel in one process, then the bench line
set -u
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $E/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 11 --env-sweep SVS_ROW1_OLD=0,1 $V/variants/libsvsdct_exp.so > $E/ab_row1.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $E/ab_row1.txt | tail -12
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_r02.so > $E/ab_vs_r02.txt 2>&1; echo "ab2 rc=$?"; grep -v amdgpu.ids $E/ab_vs_r02.txt | tail -8
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 7 --delta 20 --rounds 7 --env-sweep SVS_ROW1_OLD=0,1 $V/variants/libsvsdct_exp.so > $E/ab_row1_n7.txt 2>&1; grep -v amdgpu.ids $E/ab_row1_n7.txt | tail -6
timeout -k 10 400 python bench.py > $E/bench.json 2> $E/bench.err; echo "bench rc=$?"; cat $E/bench.json
