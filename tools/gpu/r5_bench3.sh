#!/bin/bash
# three bench.py processes back to back on one box (the headline is one draw from a placement distribution)
set -u
O=gpurun_out/bench3_$(date +%s); mkdir -p $O
for i in 1 2 3; do timeout -k 10 200 python bench.py --cpu-frames 0 2>/dev/null | grep '^{' > $O/bench_$i.json; done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.loads(open(f).read())
    print(f"{d['value']/1e6:.3f} Tpixel/s   embed {d['kernel_ms']['embed']:.4f} ms  extract {d['kernel_ms']['extract']:.4f} ms   roofline.frac {d['roofline']['frac']:.3f}  traffic {d['roofline']['traffic']}")
PY
