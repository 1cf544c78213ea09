#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for cfg in "--frames 600 --n-ac 3" "--frames 2400 --h 1080 --w 1920 --n-ac 3" "--frames 300 --h 1080 --w 1920 --n-ac 3" "--frames 600 --n-ac 7 --delta 20"; do
  echo "== ab $cfg" >> $O/ab_vs_r02.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_r02.so $V/variants/libsvsdct_xbpl1.so 2>&1 | grep -E "^base|^_r02|^_xbpl1|Error|error" >> $O/ab_vs_r02.txt
done
cut -c1-230 $O/ab_vs_r02.txt
