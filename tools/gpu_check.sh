#!/bin/bash
# Run on the GPU box via gpurun: smoke, GPU test tier, a short bench, and a rocprofv3 kernel-stats pass.
# A step that times out (124/137) stops the script: no further GPU step after a hang.
set -u
mkdir -p gpurun_out
step() {  # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s)" | tee -a gpurun_out/steps.log
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "   rc=$rc" | tee -a gpurun_out/steps.log
    tail -n 5 "gpurun_out/$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout - stopping"; exit $rc; fi
    return 0
}
: > gpurun_out/steps.log
step 300 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 900 pytest_gpu.log python -m pytest tests -m gpu -x -q -s
step 400 bench_n1.log python bench.py ${BENCH_ARGS:-}
if [ "${PROFILE:-1}" = "1" ]; then
    export TMPDIR=/tmp
    step 400 rocprof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0
fi
echo done
