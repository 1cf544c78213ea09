#!/usr/bin/env python3
"""Static instruction mix and register / LDS use of the kernels whose demangled name contains a substring.

usage: kernel_isa.py <substring> [--dump FILE] [extra hipcc flags...]
Compiles csrc/svs_capi.hip to device assembly (gfx950) and prints, per matching kernel: VGPRs, SGPRs, LDS, scratch,
occupancy, and the instruction histogram with the v_mov_b32 / s_waitcnt / address-arithmetic counts singled out.
"""
import collections, os, re, subprocess, sys

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "secure-video-steganography-using-ecc-and-dct_amd", "csrc", "svs_capi.hip")
args = sys.argv[1:]
flt = args[0] if args else ""
dump = None
if "--dump" in args:
    i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
if "--reuse" in args:
    args.remove("--reuse")
    reuse = True
else:
    reuse = False
out = "/tmp/svs_isa.s"
if not (reuse and os.path.exists(out)):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "--offload-arch=gfx950",
                    "-I" + os.path.join(repo, "include"), "-S", "--cuda-device-only", "-o", out, src] + args[1:],
                   check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    body = m.group(2)
    get = lambda k: (re.search(r"\." + k + r" (\S+)", body) or [None, "?"])[1]
    meta[m.group(1)] = dict(vgpr=get("amdhsa_next_free_vgpr"), sgpr=get("amdhsa_next_free_sgpr"), lds=get("amdhsa_group_segment_fixed_size"),
                            scratch=get("amdhsa_private_segment_fixed_size"), accum=get("amdhsa_accum_offset"))
for m in re.finditer(r"^(_Z\w+):\s*;[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    sym = m.group(1)
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.split("(")[0]
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
    md = meta.get(sym, {})
    occ = re.search(r"; Occupancy: (\d+)", text[m.end():m.end() + 4000])
    print(f"{name}\n    vgpr {md.get('vgpr')} (accum_offset {md.get('accum')}) sgpr {md.get('sgpr')} lds {md.get('lds')} scratch {md.get('scratch')}"
          f" occupancy {occ.group(1) if occ else '?'}")
    print(f"    total {len(ins)} valu {grp('v_')} salu {grp('s_')} vmem {grp(('global_', 'buffer_', 'flat_'))} ds {grp('ds_')}"
          f" | v_mov {c['v_mov_b32_e32'] + c['v_mov_b32_e64'] + 2 * c['v_mov_b64_e32']} s_waitcnt {c['s_waitcnt']} v_cndmask {grp('v_cndmask')}"
          f" lshl_add_u64 {c['v_lshl_add_u64']} s_cbranch {grp('s_cbranch')}")
    print("    " + ", ".join(f"{k}:{n}" for k, n in sorted(c.items(), key=lambda kv: -kv[1])[:40]))
    if dump:
        open(dump, "w").write(m.group(0))
