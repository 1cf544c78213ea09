#!/bin/bash
# round 5, fifth GPU session: block-row aligned tiles A/B (same process, identical bytes checked), across placements; operator after the rework
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
V=secure-video-steganography-using-ecc-and-dct_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "drop_in or string_payload or integration_md or host_pointer or default_mode" > $O/pytest_gpu.txt 2>&1; rc=$?
tail -4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
for cfg in "--frames 600 --n-ac 3" "--frames 2400 --h 1080 --w 1920 --n-ac 3" "--frames 150 --h 4320 --w 7680 --n-ac 3" "--frames 600 --n-ac 7 --delta 20"; do
  echo "== ab $cfg  SVS_EMBED_ROW_TILES=0,1" >> $O/ab_rowtiles.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode guarded --rounds 9 --env-sweep SVS_EMBED_ROW_TILES=0,1 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*| / | /' >> $O/ab_rowtiles.txt
done
for cfg in "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 2400 --h 1080 --w 1920 --n-ac 10" "--frames 600 --n-ac 10 --delta 20"; do
  echo "== ab $cfg  SVS_EMBED_ROW_TILES=0,2" >> $O/ab_rowtiles.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --mode guarded --rounds 9 --env-sweep SVS_EMBED_ROW_TILES=0,2 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "frames|embed med" | sed 's/ | extract.*| / | /' >> $O/ab_rowtiles.txt
done
cat $O/ab_rowtiles.txt
timeout -k 10 500 python tools/placement_ab.py --pairs 5 > $O/placement_ab.txt 2>&1 || { tail -20 $O/placement_ab.txt; exit 1; }
cat $O/placement_ab.txt
timeout -k 10 500 python tools/placement_ab.py --pairs 4 --n-ac 10 > $O/placement_ab_n10.txt 2>&1 || { tail -20 $O/placement_ab_n10.txt; exit 1; }
cat $O/placement_ab_n10.txt
timeout -k 10 300 python tools/op_profile.py > $O/op_profile.txt 2>&1 || { tail -20 $O/op_profile.txt; exit 1; }
grep -E "==|proses|embed_frames_str|extract_frames_str" $O/op_profile.txt
