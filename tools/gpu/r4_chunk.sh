#!/bin/bash
# tile map of the two-row embed kernel (VALU-bound): identity / runs of 32 / contiguous eighth per XCD; experiments library
set -u
mkdir -p gpurun_out/r4chunk
V=secure-video-steganography-using-ecc-and-dct_amd/lib/variants
for cfg in "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 600 --n-ac 10" "--frames 120 --h 480 --w 640 --n-ac 10 --delta 20"; do
  echo "== $cfg guarded, SVS_EMBED_XCD_CHUNK sweep (0 = identity, 4294967295 = contiguous eighth)"
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode guarded --rounds 11 --env-sweep SVS_EMBED_XCD_CHUNK=4294967295,0,8,32 $V/libsvsdct_exp.so 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*//'
done > gpurun_out/r4chunk/chunk.txt 2>&1
cat gpurun_out/r4chunk/chunk.txt
