#!/bin/bash
# round 6: occupancy cap sweep of the integer-domain one-row embed kernel (experiments library knob SVS_EMBED_WG_PER_CU), and old vs new at the best cap
set -u
mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6b
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 --env-sweep SVS_EMBED_WG_PER_CU=0,3,4,5,6,7 $V/variants/libsvsdct_exp.so > $E/occ_new.txt 2>&1; echo "rc=$?"; grep -E "^SVS|^pattern" $E/occ_new.txt
SVS_ROW1_OLD=1 timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 9 --env-sweep SVS_EMBED_WG_PER_CU=0,3,4,5 $V/variants/libsvsdct_exp.so > $E/occ_old.txt 2>&1; echo "rc=$?"; grep -E "^SVS|^pattern" $E/occ_old.txt
