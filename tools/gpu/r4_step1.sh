#!/bin/bash
# Round 4, step 1: the rewritten two-row kernel (integer first stages, float-domain QIM, workgroup-shared replay) - parity
# tests that touch it, content-class probe, kernel stats and SQ counters at n = 10.
set -u
mkdir -p gpurun_out/r4s1
export TMPDIR=/tmp
E=gpurun_out/r4s1
step() { local secs=$1 log=$2; shift 2; echo "== $*"; timeout -k 10 "$secs" "$@" > "$E/$log" 2>&1; local rc=$?; echo "   rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; exit $rc; fi; return $rc; }
step 120 smoke.log python -c "import __graft_entry__ as g; g.smoke()" || { tail -20 $E/smoke.log; exit 1; }
step 900 pytest_sel.log python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guarded or exact_mode or fast_mode or golden or pitched or fused or random_geometries or stateless" || { tail -30 $E/pytest_sel.log; exit 1; }
tail -3 $E/pytest_sel.log
step 400 guarded_probe_n10.txt python tools/guarded_probe.py --frames 200 --n-ac 10 --classes noise,natural,flat128,dark
grep -v amdgpu $E/guarded_probe_n10.txt
step 300 stats_g10.log rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_g10 -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --n-ac 10 --mode guarded
cp $E/prof_g10/*/*_kernel_stats.csv $E/kernel_stats_g10.csv 2>/dev/null; head -3 $E/kernel_stats_g10.csv
TAG=r4s1 BENCH_ARGS="--frames 600 --n-ac 10 --mode guarded" bash tools/gpu_pmc_sq.sh > $E/sq_run.log 2>&1; python tools/sq_summary.py r4s1 > $E/sq_counters_g10.txt 2>&1; head -8 $E/sq_counters_g10.txt
