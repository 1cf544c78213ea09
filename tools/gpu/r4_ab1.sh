#!/bin/bash
# A/B of the two-row kernel's replay scope and register target (variants built with -D flags into lib/variants)
set -u
mkdir -p gpurun_out/r4ab1
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== ab $cfg guarded: base (pool, natural regs) | nopool | w5 | w6 | nopool_w5"
  timeout -k 10 400 python tools/ab_bench.py $cfg --mode guarded --rounds 7 $V/libsvsdct.so $V/variants/libsvsdct_nopool.so $V/variants/libsvsdct_w5.so $V/variants/libsvsdct_w6.so $V/variants/libsvsdct_nopool_w5.so 2>&1 | grep -E "frames|embed med"
done > gpurun_out/r4ab1/ab.txt 2>&1
cat gpurun_out/r4ab1/ab.txt
