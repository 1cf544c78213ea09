#!/usr/bin/env python3
"""Instruction mix per kernel from hipcc -S output (device only). usage: isa_count.py [filter-substring]"""
import collections, os, re, subprocess, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "secure-video-steganography-using-ecc-and-dct_amd", "csrc", "svs_capi.hip")
out = "/tmp/svs_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950",
                "-I" + os.path.join(repo, "include"), "-S", "--cuda-device-only", "-o", out, src]
               + [a for a in sys.argv[2:]], check=True, stderr=subprocess.DEVNULL)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
text = open(out).read()
for m in re.finditer(r"^(_Z\w+):\s*;[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0]
    if flt not in name:
        continue
    ins = []
    for l in m.group(2).splitlines():
        l = l.strip()
        if not l or l[0] in ";./" or l.endswith(":"):
            continue
        ins.append(l.split()[0])
    c = collections.Counter(ins)
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print(f"{name}: total {len(ins)} valu {grp('v_')} salu {grp('s_')} vmem {grp(('global_','buffer_','flat_'))} ds {grp('ds_')}")
    print("   ", ", ".join(f"{k}:{n}" for k, n in sorted(c.items(), key=lambda kv: -kv[1])[:45]))
