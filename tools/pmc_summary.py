#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles/<tag>_pmc_summary.json and
profiles/hbm_traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic).

Corrections applied, as MI355X_MICROARCH.md (HBM section) prescribes:
  * FETCH_SIZE / WRITE_SIZE are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced stream -> doubled.  The factor is
    CALIBRATED in the same run on frame_sse_kernel, which reads a known 2*F*H*W bytes with the same 8-byte-per-lane
    row accesses (TCC_EA0_RDREQ_sum * 128 B is reported beside it as an independent count);
  * WRITE_SIZE is exact for these full-line streaming stores (checked on fill_synthetic_kernel's known F*H*W bytes).
"""
import collections, csv, glob, json, os, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, repo)
from bench import kernel_source_sha      # what bench.py compares the capture against
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(repo, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
F, H, W, n = 600, 2160, 3840, int(sys.argv[3]) if len(sys.argv) > 3 else 3     # n != 3: summary only, hbm_traffic.json is left alone
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svs::", "")
        k = k.split("<")[0]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
px = F * H * W
known_read = 2 * px                     # frame_sse_kernel
fetch_scale = known_read / (mean[("frame_sse_kernel", "FETCH_SIZE")] * 1024)
write_scale = px / (mean[("fill_synthetic_kernel", "WRITE_SIZE")] * 1024)
out = {"workload": {"frames": F, "height": H, "width": W, "n_ac": n},
       "calibration": {"FETCH_SIZE_scale_from_frame_sse_kernel": fetch_scale,
                       "WRITE_SIZE_scale_from_fill_synthetic_kernel": write_scale}, "kernels": {}}
extract_name = "extract_kernel" if ("extract_kernel", "FETCH_SIZE") in mean else "extract_exact_kernel"   # n <= 7 uses the exact one
embed_name = "embed_row1_kernel" if ("embed_row1_kernel", "FETCH_SIZE") in mean else "embed_kernel"   # n <= 7: the integer-domain one-row kernel (round 6)
kerns = [(embed_name, 2 * px + px * n // 512), (extract_name, px + px * n // 512)]
if ("embed_exact_kernel", "FETCH_SIZE") in mean:
    kerns.append(("embed_exact_kernel", 2 * px + px * n // 512))
for kern, alg in kerns:
    rd = mean[(kern, "FETCH_SIZE")] * 1024 * 2.0
    wr = mean[(kern, "WRITE_SIZE")] * 1024
    out["kernels"][kern] = {
        "algorithmic_bytes_per_launch": alg,
        "hbm_read_bytes_FETCH_SIZE_x2": rd, "hbm_read_bytes_RDREQ_x128": mean.get((kern, "TCC_EA0_RDREQ_sum"), 0) * 128,
        "hbm_write_bytes_WRITE_SIZE": wr, "hbm_write_bytes_WRREQ_x64": mean.get((kern, "TCC_EA0_WRREQ_sum"), 0) * 64,
        "hbm_bytes_per_launch": rd + wr, "traffic_over_algorithmic": (rd + wr) / alg,
        "SQ_INSTS_VALU_per_wave": mean.get((kern, "SQ_INSTS_VALU"), 0) / max(mean.get((kern, "SQ_WAVES"), 1), 1),
        "effective_clock_GHz_note": "GRBM_GUI_ACTIVE/8/kernel time; see raw counters",
        "raw_means": {c: v for (k, c), v in mean.items() if k == kern}}
import subprocess, datetime
try:
    rev = subprocess.run(["git", "-C", repo, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    rev = "?"
captured = f"{datetime.date.today().isoformat()}, commit {rev}, tag {tag}"
out["captured"] = captured
with open(os.path.join(repo, "profiles", f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
if n == 3:
  with open(os.path.join(repo, "profiles", "hbm_traffic.json"), "w") as fh:
    json.dump({"frames": F, "height": H, "width": W, "n_ac": n,
               "embed_bytes_per_launch": out["kernels"][embed_name]["hbm_bytes_per_launch"],
               "captured": captured, "kernel_source_sha256": kernel_source_sha(),
               "extract_bytes_per_launch": out["kernels"][extract_name]["hbm_bytes_per_launch"],
               "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/gpu_pmc.sh)"},
              fh, indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw_means"} for k, v in out["kernels"].items()}, indent=1))
