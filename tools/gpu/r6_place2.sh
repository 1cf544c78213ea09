#!/bin/bash
# round 6: tile maps under the integer-domain kernel capped at 4 waves per SIMD, per placement
set -u
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
E=gpurun_out/r6c
timeout -k 10 600 python tools/placement_ab.py --pairs 6 --rounds 3 \
  --cfg new4=exp:SVS_EMBED_WG_PER_CU=4 --cfg m0=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=0 --cfg m1=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=1 \
  --cfg m4=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=4 --cfg m32=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=32 --cfg m256=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=256 \
  --cfg m2k=exp:SVS_EMBED_WG_PER_CU=4,SVS_EMBED_XCD_CHUNK=2048 --cfg m0w7=exp:SVS_EMBED_XCD_CHUNK=0 --cfg new3=exp:SVS_EMBED_WG_PER_CU=3 > $E/place2.txt 2>&1
echo rc=$?; grep -v amdgpu.ids $E/place2.txt
