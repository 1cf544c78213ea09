#!/bin/bash
# round 5, second GPU session: tests, staging pipeline rates after the up/down-stream fix, placement probe, counter list, bench x3
set -o pipefail
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -6 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || { tail -20 $O/pcie_rate.txt; exit 1; }
cat $O/pcie_rate.txt
timeout -k 10 200 python tools/stage_chunk_sweep.py > $O/chunk_sweep.txt 2>&1 || { tail -20 $O/chunk_sweep.txt; exit 1; }
cat $O/chunk_sweep.txt
for i in 1 2 3; do timeout -k 10 300 python tools/placement_probe.py --tag "process $i" >> $O/placement.txt 2>&1 || { tail -20 $O/placement.txt; exit 1; }; done
cat $O/placement.txt
for i in 1 2 3; do timeout -k 10 200 python bench.py --cpu-frames 0 --steps 40 > $O/bench_$i.json 2>> $O/bench.err || exit 1; done
python - <<'PY'
import json
for i in (1,2,3):
    d=json.loads(open(f"gpurun_out/r5b/bench_{i}.json").read().strip().splitlines()[-1])
    print("bench", i, round(d["value"]/1e6,3), "Tpix/s embed", round(d["kernel_ms"]["embed"],4), "extract", round(d["kernel_ms"]["extract"],4))
PY
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -ciE "TCP_UTCL1|TCC_EA0_RDREQ|TCC_TAG_STALL|GRBM_GUI_ACTIVE|UTCL2" $O/counters.txt
