#!/bin/bash
# HBM traffic counters for the bench workload: separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), each with --kernel-trace only.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-frames 0 ${BENCH_ARGS:-}"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo "$pmc" | tr ' ' '+')
    echo "== pmc $pmc"
    timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "gpurun_out/pmc_$tag" -- python bench.py $ARGS > "gpurun_out/pmc_$tag.log" 2>&1
    rc=$?; echo "   rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout - stopping"; exit $rc; fi
done
rocprofv3 -L > gpurun_out/rocprof_counters_list.txt 2>&1
ls gpurun_out/pmc_*/*/ | head -40
