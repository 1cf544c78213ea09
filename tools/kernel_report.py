#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage."""
import os, re, subprocess, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "secure-video-steganography-using-ecc-and-dct_amd", "csrc", "svs_capi.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
       "-I" + os.path.join(repo, "include"), "-Rpass-analysis=kernel-resource-usage",
       "-o", "/tmp/libsvsdct_report.so", src] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +([A-Za-z ]+(?:\[bytes/[a-z]+\])?): *(\S+)", line) or \
        re.search(r"remark: +([A-Za-z ]+?(?: \[bytes/[a-z]+\])?): +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name:48s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs', r.get('SGPRs','?')):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} lds={r.get('LDS Size [bytes/block]','?')}")
