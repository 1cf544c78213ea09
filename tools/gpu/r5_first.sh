#!/bin/bash
# round 5, first GPU session: smoke, GPU test tier, host-pointer boundary rates, one bench line
set -o pipefail
O=gpurun_out/r5a; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -15 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/pcie_rate.py > $O/pcie_rate.txt 2>&1 || { tail -20 $O/pcie_rate.txt; exit 1; }
cat $O/pcie_rate.txt
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","kernel_ms")}, d["roofline"]["frac"], d["config"]["mode"], d.get("parity_sample",{}).get("pixels_differing_from_reference"))
PY
