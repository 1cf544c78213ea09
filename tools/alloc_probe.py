#!/usr/bin/env python3
"""Round 6: does it matter HOW a fresh process allocates its (cover, stego) pair?  One pair per process, allocated first thing:
  separate   two svs_malloc (hipMalloc) allocations                         (what two torch.empty calls amount to)
  arena      ONE svs_malloc for cover + stego, the stego plane 2 MB aligned behind the cover
  arena+gap  the same with <gap> MB between the two planes
Prints the embed launch's median / min over `reps` (sustained) and the extract median; with SVSDCT_LIB = the experiments library also the
same pair under other embed / extract tile maps.  usage: alloc_probe.py <mode> [gap_mb] [reps]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
import torch
from svsdct import batch, native
from svsdct.native import Planes

mode = sys.argv[1] if len(sys.argv) > 1 else "arena"
gap = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = native.load(); native.ensure_device(0); torch.cuda.set_device(0)
F, H, W, n, delta = 600, 2160, 3840, 3, 8.0
planes = Planes.contiguous(F, H, W)
size, cap = F * H * W, batch.capacity_bits(F, H, W, n)
st = torch.cuda.current_stream().cuda_stream


def svs_malloc(nb):
    p = C.c_void_p(); native.check(lib.svs_malloc(C.byref(p), nb), "malloc"); return p


al = 2 << 20
sz = (size + al - 1) // al * al
if mode == "separate":
    g, s = svs_malloc(size).value, svs_malloc(size).value
else:
    a = svs_malloc(2 * sz + gap + al)
    g = (a.value + al - 1) // al * al
    s = g + sz + gap
pay = torch.zeros((cap + 7) // 8 + 16, dtype=torch.uint8, device="cuda")
ext = torch.zeros_like(pay)
native.check(lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st), "bits")
native.check(lib.svs_fill_synthetic_dev(g, C.byref(planes), 20250620, 0, 16, 224, st), "fill")
for _ in range(3):
    batch.embed_device(g, s, planes, delta, n, pay.data_ptr(), 0, cap, st)
    batch.extract_device(s, planes, delta, n, ext.data_ptr(), ext.numel(), st)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
for e in ev:
    e[0].record(); batch.embed_device(g, s, planes, delta, n, pay.data_ptr(), 0, cap, st)
    e[1].record(); batch.extract_device(s, planes, delta, n, ext.data_ptr(), ext.numel(), st)
    e[2].record()
torch.cuda.synchronize()
te = [e[0].elapsed_time(e[1]) for e in ev]; tx = [e[1].elapsed_time(e[2]) for e in ev]
print(f"{mode:9s} gap {gap >> 20:5d} MB: embed {np.median(te):.4f} / {min(te):.4f}  extract {np.median(tx):.4f}   {g:#x} {s:#x} (stego - cover = {s - g:#x})")
if "exp" in os.environ.get("SVSDCT_LIB", ""):      # experiments library: the same pair under other tile maps (0xFFFFFFFF = one eighth of the batch per XCD)
    for em, xm in (("4294967295", "4294967295"), ("32", "32"), ("32", "4294967295")):
        os.environ["SVS_EMBED_XCD_CHUNK"], os.environ["SVS_EXTRACT_XCD_CHUNK"] = em, xm
        for e in ev:
            e[0].record(); batch.embed_device(g, s, planes, delta, n, pay.data_ptr(), 0, cap, st)
            e[1].record(); batch.extract_device(s, planes, delta, n, ext.data_ptr(), ext.numel(), st)
            e[2].record()
        torch.cuda.synchronize()
        te = [e[0].elapsed_time(e[1]) for e in ev]; tx = [e[1].elapsed_time(e[2]) for e in ev]
        print(f"    embed map {em:>10s}: {np.median(te):.4f}   extract map {xm:>10s}: {np.median(tx):.4f}")
