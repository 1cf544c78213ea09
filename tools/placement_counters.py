#!/usr/bin/env python3
"""Joins a `rocprofv3 --kernel-trace --pmc ... -- python tools/placement_probe.py` run with the probe's own output: the k-th
(cover, stego) pair of the probe issued embed launches [k * (3 + reps), (k + 1) * (3 + reps)); per pair -> mean duration of its
timed launches and mean counter values.  usage: placement_counters.py <rocprof output dir> <reps> [kernel substring]"""
import collections, csv, glob, os, sys
d, reps = sys.argv[1], int(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else "embed_kernel"
per = 3 + reps
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
trace = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r.get("Dispatch_Id")] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
disp = collections.OrderedDict()
for r in rows:
    if sub not in r["Kernel_Name"]:
        continue
    e = disp.setdefault(int(r["Dispatch_Id"]), {"c": {}})
    e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        e["t"] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
ids = sorted(disp)
names = sorted({c for e in disp.values() for c in e["c"]})
print(f"# {len(ids)} launches of *{sub}*; counters: {', '.join(names)}")
print("pair  launches  mean_ms   " + "  ".join(f"{n:>34s}" for n in names))
for k in range(len(ids) // per):
    mine = ids[k * per + 3:(k + 1) * per]      # the timed launches (the first three warm up)
    ts = []
    for i in mine:
        t = disp[i].get("t") or trace.get(str(i))
        if t:
            ts.append((t[1] - t[0]) / 1e6)
    vals = [sum(disp[i]["c"].get(n, 0.0) for i in mine) / len(mine) for n in names]
    print(f"{k:4d}  {len(mine):8d}  {sum(ts) / max(len(ts), 1):7.4f}   " + "  ".join(f"{v:34.1f}" for v in vals))
