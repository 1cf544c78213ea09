#!/bin/bash
# SQ occupancy/issue counters for one bench configuration (BENCH_ARGS), one rocprofv3 --pmc pass each
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="--steps 3 --warmup 1 --cpu-frames 0 ${BENCH_ARGS:-}"
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "gpurun_out/sq_${TAG:-x}_$i" -- python bench.py $ARGS > "gpurun_out/sq_${TAG:-x}_$i.log" 2>&1
    rc=$?; echo "pass $i rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
