#!/bin/bash
# Round-6 evidence at one commit (run through gpurun from the repo root; tools/collect_evidence.sh r06 copies the results into profiles/):
# bench line x 3 processes, kernel stats (n = 3; n = 10 at 4K and 1080p), PMC traffic passes, SQ counters (n = 3, 10), the other BASELINE
# shapes with a CPU parity sample each, same-process A/B against the round-5 and round-2 libraries (make -C csrc r05 r02), the embed launch
# on six buffer placements, GPU test tier.
set -u
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/ev
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; }
PART=${1:-all}    # a = bench lines, kernel stats, counters; b = other shapes, A/Bs, placements, GPU tier (one gpurun call holds 20 minutes)
if [ "$PART" != b ]; then
step 400 bench.json python bench.py
for i in 2 3; do timeout -k 10 200 python bench.py --cpu-frames 0 2>/dev/null | grep '^{' > $E/bench_process_$i.json; done
step 400 rocprof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0
cp $E/prof_stats/*/*_kernel_stats.csv $E/kernel_stats.csv 2>/dev/null
step 300 rocprof_stats_g10.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10
cp $E/prof_g10/*/*_kernel_stats.csv $E/kernel_stats_g10.csv 2>/dev/null
step 300 rocprof_stats_g10_1080.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10_1080 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10 --frames 300 --height 1080 --width 1920
cp $E/prof_g10_1080/*/*_kernel_stats.csv $E/kernel_stats_g10_1080.csv 2>/dev/null
BENCH_ARGS="" bash tools/gpu_pmc.sh > $E/pmc_run.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out r06 > $E/pmc_summary_print.txt 2>&1; echo "pmc summary rc=$?"
cp profiles/r06_pmc_summary.json profiles/hbm_traffic.json $E/ 2>/dev/null
TAG=r6n3 BENCH_ARGS="" bash tools/gpu_pmc_sq.sh > $E/sq_run_n3.log 2>&1; python tools/sq_summary.py r6n3 > $E/sq_counters_n3.txt 2>&1
TAG=r6n10 BENCH_ARGS="--frames 600 --n-ac 10" bash tools/gpu_pmc_sq.sh > $E/sq_run_n10.log 2>&1; python tools/sq_summary.py r6n10 > $E/sq_counters_n10.txt 2>&1
fi
if [ "$PART" != a ]; then
: > $E/other_configs_bench.jsonl
for cfg in "--frames 300 --height 1080 --width 1920 --n-ac 10 --delta 8" "--frames 600 --n-ac 10" "--frames 600 --n-ac 10 --delta 20" "--frames 150 --height 4320 --width 7680 --delta 4" "--frames 150 --height 4320 --width 7680 --delta 8" "--frames 150 --height 4320 --width 7680 --delta 16" "--frames 120 --height 480 --width 640 --n-ac 10 --delta 20" "--frames 600 --n-ac 7 --delta 20" "--frames 200 --n-ac 63" "--frames 200 --n-ac 20" "--frames 600 --mode exact"; do
  timeout -k 10 400 python bench.py $cfg --cpu-frames 16 --steps 20 2>/dev/null | grep '^{' >> $E/other_configs_bench.jsonl
done
: > $E/ab_vs_earlier.txt
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 7 --delta 20" "--frames 2400 --h 1080 --w 1920 --n-ac 3"; do
  echo "== ab $cfg (flags 0): this build | the round-5 library (float-domain one-row kernel, same bytes) | the round-2 library (contract-level arithmetic: no guard, no replay, different bytes)" >> $E/ab_vs_earlier.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 11 $V/libsvsdct.so $V/variants/libsvsdct_r05.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "amdgpu.ids" >> $E/ab_vs_earlier.txt
done
step 600 placements.txt python tools/placement_ab.py --pairs 6 --rounds 3 --cfg this=base: --cfg r05=libsvsdct_r05.so --cfg r02=libsvsdct_r02.so --cfg this_copy=base:PATCOPY=1
step 600 guarded_probe_n3.txt python tools/guarded_probe.py --frames 200 --n-ac 3 --classes noise,natural,flat128,letterbox25,dark
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $E/pytest_gpu.log
cp gpurun_out/parity_report.json gpurun_out/pipeline_overlap.json $E/ 2>/dev/null
fi
rm -rf $E/prof_stats $E/prof_g10 $E/prof_g10_1080
ls $E
