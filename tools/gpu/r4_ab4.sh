#!/bin/bash
set -u
mkdir -p gpurun_out/r4ab4
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for order in "$V/libsvsdct.so $V/variants/libsvsdct_noinplace.so" "$V/variants/libsvsdct_noinplace.so $V/libsvsdct.so"; do
for cfg in "--frames 600 --n-ac 10" "--frames 600 --n-ac 10 --delta 20" "--frames 600 --n-ac 15 --delta 20"; do
  echo "== $cfg guarded: base = rows parked in LDS, phase 1 in place (68 VGPRs, 5 waves) | noinplace (103 VGPRs, 4 waves); order: $order"
  timeout -k 10 400 python tools/ab_bench.py $cfg --mode guarded --rounds 15 $order 2>&1 | grep -E "embed med" | sed 's/ | extract.*| / | /'
done; done > gpurun_out/r4ab4/ab.txt 2>&1
cat gpurun_out/r4ab4/ab.txt
