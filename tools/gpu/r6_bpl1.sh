#!/bin/bash
set -u
mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6e
: > $E/bpl1.txt
echo "== new kernel, one block per lane (W = 3848)" >> $E/bpl1.txt
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --w 3848 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,5,6,7,8 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|^pattern|^copy16_nt |rror" >> $E/bpl1.txt
echo "== round-5 kernel" >> $E/bpl1.txt
SVS_ROW1_OLD=1 timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --w 3848 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,6,8 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|^pattern|rror" >> $E/bpl1.txt
echo "== new kernel forced to one block per lane at W = 3840 (SVS_EMBED_BPL=1)" >> $E/bpl1.txt
SVS_EMBED_BPL=1 timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,4,6,8 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|^pattern|rror" >> $E/bpl1.txt
cat $E/bpl1.txt
