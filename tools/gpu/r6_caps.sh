#!/bin/bash
# round 6: occupancy cap of the integer-domain one-row kernel at other operating points: n = 7 general delta, one block per lane (odd block count per row), 1080p
set -u
mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6e
for cfg in "--frames 600 --n-ac 7 --delta 20" "--frames 600 --n-ac 3 --w 3848" "--frames 2400 --h 1080 --w 1920 --n-ac 3" "--frames 300 --h 1080 --w 1920 --n-ac 3" "--frames 600 --n-ac 1 --delta 10"; do
  echo "== $cfg" >> $E/caps.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,4,5,6 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|rror" >> $E/caps.txt
done
cat $E/caps.txt
