#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
V=secure-video-steganography-using-ecc-and-dct_amd/lib
for rep in 1 2 3; do
  echo "== process $rep: --frames 600 --n-ac 3, order: xbpl1 base" >> $O/ab_xbpl.txt
  timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 15 $V/variants/libsvsdct_xbpl1.so $V/libsvsdct.so 2>&1 | grep -E "^base|^_xbpl1" | sed 's/embed med.*| extract/extract/' >> $O/ab_xbpl.txt
done
echo "== --frames 150 --h 4320 --w 7680 --n-ac 3" >> $O/ab_xbpl.txt
timeout -k 10 300 python tools/ab_bench.py --frames 150 --h 4320 --w 7680 --n-ac 3 --rounds 15 $V/variants/libsvsdct_xbpl1.so $V/libsvsdct.so 2>&1 | grep -E "^base|^_xbpl1" | sed 's/embed med.*| extract/extract/' >> $O/ab_xbpl.txt
echo "== --frames 300 --h 1080 --w 1920 --n-ac 3 (sustained is what matters at this size: 41 rounds)" >> $O/ab_xbpl.txt
timeout -k 10 300 python tools/ab_bench.py --frames 300 --h 1080 --w 1920 --n-ac 3 --rounds 41 $V/variants/libsvsdct_xbpl1.so $V/libsvsdct.so 2>&1 | grep -E "^base|^_xbpl1" | sed 's/embed med.*| extract/extract/' >> $O/ab_xbpl.txt
cut -c1-150 $O/ab_xbpl.txt
