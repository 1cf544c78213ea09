#!/usr/bin/env python3
import collections, csv, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/sq_{tag}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svs::", "")
        if "embed" in k or "extract" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ks = sorted({k for k, _ in agg})
for k in ks:
    m = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
    us = sum(dur[k]) / len(dur[k])
    w = m.get("SQ_WAVES", 1)
    print(f"{k}: {us:.0f} us/launch (profiled), waves {w:.0f}")
    print(f"   per wave: VALU {m.get('SQ_INSTS_VALU',0)/w:.0f} SALU {m.get('SQ_INSTS_SALU',0)/w:.0f} VMEM_RD {m.get('SQ_INSTS_VMEM_RD',0)/w:.1f} VMEM_WR {m.get('SQ_INSTS_VMEM_WR',0)/w:.1f} LDS {m.get('SQ_INSTS_LDS',0)/w:.1f}")
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(f"   of wave-cycles: wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active_any {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} active_valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f}; busy_cycles {m.get('SQ_BUSY_CYCLES',0):.3g}")
    if "GRBM_GUI_ACTIVE" in m:
        print(f"   eff clock {m['GRBM_GUI_ACTIVE']/8/us/1e3:.2f} GHz")
    if "SQC_ICACHE_REQ" in m:
        req = max(m["SQC_ICACHE_REQ"], 1)
        print(f"   instruction cache: requests {m['SQC_ICACHE_REQ']:.3g} ({m['SQC_ICACHE_REQ']/w:.1f} per wave), hits {m.get('SQC_ICACHE_HITS',0)/req:.4f}, "
              f"misses {m.get('SQC_ICACHE_MISSES',0)/req:.5f}, duplicate misses {m.get('SQC_ICACHE_MISSES_DUPLICATE',0)/req:.5f}")
    if "SQ_IFETCH" in m:
        print(f"   instruction fetches per wave {m['SQ_IFETCH']/w:.1f}; mean fetch latency {m.get('SQ_IFETCH_LEVEL',0)/max(m['SQ_IFETCH'],1):.1f} cycles; "
              f"VALU_CVT per wave {m.get('SQ_INSTS_VALU_CVT',0)/w:.0f}, VALU_INT32 per wave {m.get('SQ_INSTS_VALU_INT32',0)/w:.0f}")
