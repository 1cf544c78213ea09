#!/bin/bash
# Round-4 baseline evidence for the rigorous two-row embed kernel (GUARDED, n = 10), taken before any change:
# kernel stats under rocprofv3, SQ counters, content-class probe, A/B against the round-2 library.
set -u
mkdir -p gpurun_out/r4base
export TMPDIR=/tmp
E=gpurun_out/r4base
V=secure-video-steganography-using-ecc-and-dct_amd/lib
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; }
step 300 stats_g10.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10 --mode guarded
cp $E/prof_g10/*/*_kernel_stats.csv $E/kernel_stats_g10.csv 2>/dev/null
step 300 stats_g10_1080.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10_1080 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10 --mode guarded --frames 300 --height 1080 --width 1920
cp $E/prof_g10_1080/*/*_kernel_stats.csv $E/kernel_stats_g10_1080.csv 2>/dev/null
TAG=r4g10 BENCH_ARGS="--frames 600 --n-ac 10 --mode guarded" bash tools/gpu_pmc_sq.sh > $E/sq_run.log 2>&1; python tools/sq_summary.py r4g10 > $E/sq_counters_g10.txt 2>&1
step 400 guarded_probe_n10.txt python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,flat128,dark
: > $E/ab_vs_r02.txt
for cfg in "--frames 600 --n-ac 3" "--frames 600 --n-ac 10" "--frames 300 --h 1080 --w 1920 --n-ac 10"; do
  echo "== ab $cfg: this build vs the round-2 library" >> $E/ab_vs_r02.txt
  timeout -k 10 300 python tools/ab_bench.py $cfg --rounds 7 $V/libsvsdct.so $V/variants/libsvsdct_r02.so 2>&1 | grep -v "amdgpu.ids" >> $E/ab_vs_r02.txt
done
ls $E
