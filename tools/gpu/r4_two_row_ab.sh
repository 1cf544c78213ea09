#!/bin/bash
# Round 4: A/Bs of the two-row (n = 8..15) embed kernel - replay scope, rows in registers vs parked in LDS, blocks per lane,
# register target.  Build the variants first, in the build container:  make -C <package>/csrc ab-variants
# Results of the runs this script reproduces: profiles/r04_ab_two_row.txt
set -u
mkdir -p gpurun_out/r4ab
V=secure-video-steganography-using-ecc-and-dct_amd/lib
libs="$V/libsvsdct.so"
for v in noinplace inplace_all pool u2bpl2 w5; do [ -f $V/variants/libsvsdct_$v.so ] && libs="$libs $V/variants/libsvsdct_$v.so"; done
for cfg in "--frames 600 --n-ac 10" "--frames 600 --n-ac 10 --delta 20" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== $cfg guarded"
  timeout -k 10 500 python tools/ab_bench.py $cfg --mode guarded --rounds 11 $libs 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*| / | /'
done > gpurun_out/r4ab/ab.txt 2>&1
cat gpurun_out/r4ab/ab.txt
