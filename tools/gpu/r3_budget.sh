#!/bin/bash
set -u
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for cfg in "--frames 600 --n-ac 10" "--frames 200 --n-ac 63" "--frames 200 --n-ac 20" "--frames 200 --n-ac 31" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== $cfg"
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_prev.so 2>&1 | grep -E "^base|^_prev"
done
echo "== guarded n = 10, this build"; timeout -k 10 250 python tools/gpu/r3_sustained.py 10 8 noise,natural 2>&1 | grep -v amdgpu.ids
echo "== guarded n = 10, previous build"; SVSDCT_LIB=$V/variants/libsvsdct_prev.so timeout -k 10 250 python tools/gpu/r3_sustained.py 10 8 noise,natural 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
