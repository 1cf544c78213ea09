#!/bin/bash
# round 6: the persistent, software-pipelined one-row embed kernel (SVS_ROW1_STREAM = workgroups per CU) against the one-shot kernel capped at 4, per placement
set -u
mkdir -p gpurun_out/r6d
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6d
timeout -k 10 300 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 5 --env-sweep SVS_ROW1_STREAM=0,1,2,3 $V/variants/libsvsdct_exp.so > $E/ab_stream.txt 2>&1; echo "rc=$?"; grep -E "^SVS|^pattern|rror|ssert" $E/ab_stream.txt
timeout -k 10 600 python tools/placement_ab.py --pairs 5 --rounds 3 \
  --cfg new4=exp:SVS_EMBED_WG_PER_CU=4 --cfg s1=exp:SVS_ROW1_STREAM=1 --cfg s2=exp:SVS_ROW1_STREAM=2 --cfg s3=exp:SVS_ROW1_STREAM=3 \
  --cfg pc4=exp:PATCOPY=1,SVS_EMBED_WG_PER_CU=4 --cfg pcs1=exp:PATCOPY=1,SVS_ROW1_STREAM=1 --cfg pcs2=exp:PATCOPY=1,SVS_ROW1_STREAM=2 --cfg pcs3=exp:PATCOPY=1,SVS_ROW1_STREAM=3 > $E/place_stream.txt 2>&1
echo rc=$?; grep -v amdgpu.ids $E/place_stream.txt
