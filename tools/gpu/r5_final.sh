#!/bin/bash
# round 5, last session at the final sources: GPU test tier, smoke, the PMC traffic passes (hbm_traffic.json is tied to the kernel-source hash),
# then bench.py so that its line carries roofline.traffic
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/final
E=gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $E/smoke.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $E/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $E/pytest_gpu.log
BENCH_ARGS="" bash tools/gpu_pmc.sh > $E/pmc_run.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out r05 > $E/pmc_summary_print.txt 2>&1; echo "pmc summary rc=$?"
cp profiles/r05_pmc_summary.json profiles/hbm_traffic.json $E/ 2>/dev/null
timeout -k 10 400 python bench.py > $E/bench.json 2> $E/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 > $E/rocprof_stats.log 2>&1; echo "rocprof rc=$?"
cp $E/prof_stats/*/*_kernel_stats.csv $E/kernel_stats.csv 2>/dev/null; rm -rf $E/prof_stats
cp gpurun_out/parity_report.json $E/ 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/final/bench.json") if l.startswith("{")][-1])
print(round(d["value"]/1e6,3), d["kernel_ms"]["embed"], d["kernel_ms"]["extract"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity_sample"]["pixels_differing_from_reference"])
PY
