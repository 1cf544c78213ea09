#!/bin/bash
set -u
mkdir -p gpurun_out/r4ab2
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== $cfg guarded: base (one block per lane) | two adjacent blocks per lane, joint replay (150 VGPRs, 3 waves) | same with a 4-wave register target (100 B scratch)"
  timeout -k 10 400 python tools/ab_bench.py $cfg --mode guarded --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_u2bpl2.so $V/variants/libsvsdct_u2bpl2w4.so 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*| / | /'
done > gpurun_out/r4ab2/ab.txt 2>&1
cat gpurun_out/r4ab2/ab.txt
