#!/bin/bash
# round 6: GPU tier, bench line, kernel stats and SQ counters of the integer-domain one-row embed kernel at its default occupancy cap
set -u
mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
E=gpurun_out/r6f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $E/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $E/bench.json 2> $E/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6f/bench.json'))
print(d['value'], d['kernel_ms']['embed'], d['kernel_ms']['extract'], d['roofline']['frac'], d['parity_sample']['pixels_differing_from_reference'])
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 > $E/bench_under_rocprof.json 2> $E/rocprof.err; echo "rocprof rc=$?"
cp $E/prof_stats/*/*_kernel_stats.csv $E/kernel_stats.csv 2>/dev/null; head -5 $E/kernel_stats.csv
TAG=r6n3 BENCH_ARGS="" bash tools/gpu_pmc_sq.sh > $E/sq_run.log 2>&1; python tools/sq_summary.py r6n3 > $E/sq_counters_n3.txt 2>&1; cat $E/sq_counters_n3.txt
rm -rf $E/prof_stats
