#!/bin/bash
# round 6: the one-row embed launch on several buffer placements x {new kernel uncapped / capped at 4, 5 waves per SIMD, round-5 kernel, round-2 library,
# the new and the round-5 kernel with the arithmetic skipped, linear copy}
set -u
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
E=gpurun_out/r6c
timeout -k 10 600 python tools/placement_ab.py --pairs 6 --rounds 3 \
  --cfg new=exp: --cfg new4=exp:SVS_EMBED_WG_PER_CU=4 --cfg new5=exp:SVS_EMBED_WG_PER_CU=5 --cfg new6=exp:SVS_EMBED_WG_PER_CU=6 \
  --cfg old=exp:SVS_ROW1_OLD=1 --cfg r02=libsvsdct_r02.so \
  --cfg pcnew=exp:PATCOPY=1 --cfg pcnew4=exp:PATCOPY=1,SVS_EMBED_WG_PER_CU=4 --cfg pcold=exp:PATCOPY=1,SVS_ROW1_OLD=1 > $E/place.txt 2>&1
echo rc=$?; grep -v amdgpu.ids $E/place.txt
