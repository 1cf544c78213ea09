#!/bin/bash
# Round-4 evidence at one commit: bench line, kernel stats (headline + guarded n = 10 at 4K and 1080p + fused colour), PMC
# traffic passes, SQ counters, other configs, content-class probe, A/B against the round-2 library, GPU test tier.
set -u
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
V=secure-video-steganography-using-ecc-and-dct_amd/lib
E=gpurun_out/ev
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; }
step 400 bench.json python bench.py
step 400 rocprof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0
cp $E/prof_stats/*/*_kernel_stats.csv $E/kernel_stats.csv 2>/dev/null
step 300 rocprof_stats_g10.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10
cp $E/prof_g10/*/*_kernel_stats.csv $E/kernel_stats_g10.csv 2>/dev/null
step 300 rocprof_stats_g10_1080.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10_1080 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10 --frames 300 --height 1080 --width 1920
cp $E/prof_g10_1080/*/*_kernel_stats.csv $E/kernel_stats_g10_1080.csv 2>/dev/null
step 300 rocprof_stats_colour.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_colour -- python tools/aux_rates.py 10 8
cp $E/prof_colour/*/*_kernel_stats.csv $E/kernel_stats_colour_n10.csv 2>/dev/null
BENCH_ARGS="" bash tools/gpu_pmc.sh > $E/pmc_run.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out r04 > $E/pmc_summary_print.txt 2>&1; echo "pmc summary rc=$?"
cp profiles/r04_pmc_summary.json profiles/hbm_traffic.json $E/ 2>/dev/null
TAG=r4n3 BENCH_ARGS="--frames 600" bash tools/gpu_pmc_sq.sh > $E/sq_run.log 2>&1; python tools/sq_summary.py r4n3 > $E/sq_counters_n3.txt 2>&1
TAG=r4n10 BENCH_ARGS="--frames 600 --n-ac 10" bash tools/gpu_pmc_sq.sh >> $E/sq_run.log 2>&1; python tools/sq_summary.py r4n10 > $E/sq_counters_n10.txt 2>&1
TAG=r4n63 BENCH_ARGS="--frames 200 --n-ac 63 --mode fast" bash tools/gpu_pmc_sq.sh >> $E/sq_run.log 2>&1; python tools/sq_summary.py r4n63 > $E/sq_counters_n63.txt 2>&1
: > $E/other_configs_bench.jsonl
for cfg in "--frames 300 --height 1080 --width 1920 --n-ac 10 --delta 8" "--frames 600 --n-ac 10" "--frames 150 --height 4320 --width 7680 --delta 4" "--frames 150 --height 4320 --width 7680 --delta 8" "--frames 150 --height 4320 --width 7680 --delta 16" "--frames 120 --height 480 --width 640 --n-ac 10 --delta 20" "--frames 200 --n-ac 63" "--frames 200 --n-ac 63 --mode fast" "--frames 200 --n-ac 20 --mode fast" "--frames 600 --mode exact" "--frames 600 --n-ac 10 --delta 20"; do
  timeout -k 10 300 python bench.py $cfg --cpu-frames 0 --steps 20 2>/dev/null | grep '^{' >> $E/other_configs_bench.jsonl
done
step 600 guarded_probe_n3.txt python tools/guarded_probe.py --frames 200 --json $E/guarded_probe_n3.json
step 300 guarded_probe_n7.txt python tools/guarded_probe.py --frames 200 --n-ac 7 --classes noise,natural,flat128,letterbox25
step 400 guarded_probe_n10.txt python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,flat128,letterbox25,rows,checker8,dark,bright
step 300 guarded_probe_n15.txt python tools/guarded_probe.py --frames 200 --n-ac 15 --delta 20 --classes noise,natural
step 300 tie_fallback_new.txt python tools/tie_fallback_rate.py
SVSDCT_LIB=$V/variants/libsvsdct_r02.so SVS_SKIP_ABI_CHECK=1 timeout -k 10 300 python tools/tie_fallback_rate.py > $E/tie_fallback_r02.txt 2>&1
: > $E/ab_vs_r02.txt
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10" "--frames 2400 --h 1080 --w 1920 --n-ac 3"; do
  echo "== ab $cfg (flags 0): this build vs the round-2 library" >> $E/ab_vs_r02.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 9 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "amdgpu.ids" >> $E/ab_vs_r02.txt
done
SVS_WRITE_PIPELINE_TIMING=1 timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $E/pytest_gpu.log
cp gpurun_out/parity_report.json gpurun_out/pipeline_overlap.json $E/ 2>/dev/null
ls $E
