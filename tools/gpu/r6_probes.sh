#!/bin/bash
# Round-6 working probes as run (each wrote the profiles/r06_* file named beside it).  usage: bash tools/gpu/r6_probes.sh <step>
#   place     embed launch x 6 placements: new kernel uncapped / capped at 4, 5, 6 waves per SIMD, round-5 kernel, round-2 library, the kernel with
#             the arithmetic skipped, linear copy (as run: round-5 kernel = commit a377b8d's experiments library with SVS_ROW1_OLD=1)   -> r06_place_new_vs_old.txt
#   maps      tile maps under the new kernel capped at 4                                                                      -> r06_place_tile_maps.txt
#   bpl       two blocks per lane capped at 4 / 5 against one block per lane capped at 5 - 8, round-5 and round-2 libraries    -> r06_place_bpl.txt
#   pitch     the access pattern (empty payload) at nine frame geometries, four placements, capped at 4 and uncapped           -> r06_pitch_probe_cap{4,0}.txt
#   stream    tools/probes/stream_probe: pure copies, R rows per lane x cap, tile map, order                                   -> r06_stream_probe.txt
#   split     tools/probes/stream_probe 5 0 (every configuration): the R8 pattern's read and write streams apart, workgroup sizes, staggered XCD streams -> r06_stream_probe_split.txt
#             (and `stream_probe 4 <kind> brief` for kind = 0..3: allocation kinds -> r06_alloc_kinds.txt)
#   counters  tools/placement_probe.py under rocprofv3 --pmc (DRAM credit stalls, tag stalls; wait / VALU cycles), joined per pair            -> r06_placement_counters.txt
#   caps      occupancy cap at n = 7 / general delta, odd block count per row, 1080p, n = 1; one block per lane                 -> r06_caps.txt
# (the persistent, software-pipelined variant of the kernel - profiles/r06_stream_pipeline.txt - lived in the working tree between commits ac1b1d8 and a377b8d)
set -u
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/r6
C4=SVS_EMBED_WG_PER_CU=4
case "${1:-}" in
place) python tools/placement_ab.py --pairs 6 --rounds 3 --cfg new=exp:SVS_EMBED_WG_PER_CU=0 --cfg new4=exp:$C4 --cfg new5=exp:SVS_EMBED_WG_PER_CU=5 --cfg new6=exp:SVS_EMBED_WG_PER_CU=6 \
         --cfg r05=libsvsdct_r05.so --cfg r02=libsvsdct_r02.so --cfg pcnew=exp:PATCOPY=1,SVS_EMBED_WG_PER_CU=0 --cfg pcnew4=exp:PATCOPY=1,$C4 > $E/place.txt 2>&1; grep -v amdgpu.ids $E/place.txt ;;
maps)  python tools/placement_ab.py --pairs 6 --rounds 3 --cfg new4=exp:$C4 --cfg m0=exp:$C4,SVS_EMBED_XCD_CHUNK=0 --cfg m1=exp:$C4,SVS_EMBED_XCD_CHUNK=1 --cfg m4=exp:$C4,SVS_EMBED_XCD_CHUNK=4 \
         --cfg m32=exp:$C4,SVS_EMBED_XCD_CHUNK=32 --cfg m256=exp:$C4,SVS_EMBED_XCD_CHUNK=256 --cfg m2k=exp:$C4,SVS_EMBED_XCD_CHUNK=2048 > $E/maps.txt 2>&1; grep -v amdgpu.ids $E/maps.txt ;;
bpl)   python tools/placement_ab.py --pairs 6 --rounds 3 --cfg b2c4=exp:$C4 --cfg b2c5=exp:SVS_EMBED_WG_PER_CU=5 --cfg b1c5=exp:SVS_EMBED_BPL=1,SVS_EMBED_WG_PER_CU=5 --cfg b1c6=exp:SVS_EMBED_BPL=1,SVS_EMBED_WG_PER_CU=6 \
         --cfg b1c7=exp:SVS_EMBED_BPL=1,SVS_EMBED_WG_PER_CU=7 --cfg b1c8=exp:SVS_EMBED_BPL=1,SVS_EMBED_WG_PER_CU=8 --cfg r05=libsvsdct_r05.so --cfg r02=libsvsdct_r02.so > $E/bpl.txt 2>&1; grep -v amdgpu.ids $E/bpl.txt ;;
pitch) for c in 4 0; do python tools/pitch_probe.py --pairs 4 --cap $c > $E/pitch_cap$c.txt 2>&1; grep -v amdgpu.ids $E/pitch_cap$c.txt; done ;;
stream) tools/probes/stream_probe 4 > $E/stream_probe.txt 2>&1; cat $E/stream_probe.txt ;;
split) tools/probes/stream_probe 5 0 > $E/stream_probe_split.txt 2>&1; cat $E/stream_probe_split.txt
       for k in 0 1 2 3; do tools/probes/stream_probe 4 $k brief; done > $E/alloc_kinds.txt 2>&1 ;;
counters) : > $E/placement_counters.txt
       for pmc in "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
         rm -rf $E/pc_run
         rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $E/pc_run -- python tools/placement_probe.py --pairs 3 --reps 6 > $E/pc_probe.log 2>&1
         grep -E "pair|#" $E/pc_probe.log | grep -v amdgpu >> $E/placement_counters.txt
         python tools/placement_counters.py $E/pc_run 6 embed_row1_kernel >> $E/placement_counters.txt
       done
       rm -rf $E/pc_run; cat $E/placement_counters.txt ;;
caps)  : > $E/caps.txt
       for cfg in "--frames 600 --n-ac 7 --delta 20" "--frames 600 --n-ac 3 --w 3848" "--frames 2400 --h 1080 --w 1920 --n-ac 3" "--frames 300 --h 1080 --w 1920 --n-ac 3" "--frames 600 --n-ac 1 --delta 10"; do
         echo "== $cfg" >> $E/caps.txt
         python tools/ab_bench.py $cfg --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,4,5,6 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|rror" >> $E/caps.txt
       done
       echo "== one block per lane forced at W = 3840 (SVS_EMBED_BPL=1)" >> $E/caps.txt
       SVS_EMBED_BPL=1 python tools/ab_bench.py --frames 600 --n-ac 3 --rounds 7 --env-sweep SVS_EMBED_WG_PER_CU=0,4,6,8 $V/variants/libsvsdct_exp.so 2>&1 | grep -E "^SVS|^pattern|rror" >> $E/caps.txt
       cat $E/caps.txt ;;
*) echo "usage: $0 place|maps|bpl|pitch|stream|split|counters|caps"; exit 2 ;;
esac
