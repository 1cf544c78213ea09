#!/usr/bin/env python3
"""Round 5: is the per-process bimodality of the headline embed launch (1.58 vs 1.68 ms per 600 x 4K, profiles/r04_box_variance.txt)
a property of WHERE the two 5 GB buffers landed (physical pages / fragment sizes behind the allocation) or of the process / clock?

In ONE process: several (cover, stego) pairs allocated in different ways, all kept alive so that every pair sits on different
memory, the same embed launch timed on each with HIP events in sustained bursts.  If the time differs between pairs of one
process, placement decides it; if every pair of a process runs at the same rate and processes differ, it is not the buffers.
  torch      two torch.empty allocations per pair (what bench.py does)
  hipmalloc  two svs_malloc (hipMalloc) allocations per pair
  arena      ONE svs_malloc for cover + stego + payload, 2 MB aligned sub-buffers
"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
import torch

from svsdct import batch, native
from svsdct.native import Planes

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--tag", default="")
ap.add_argument("--contig", action="store_true", help="also pairs of physically contiguous allocations (hipDeviceMallocContiguous; "
                                                      "needs SVSDCT_LIB = the experiments library, which has the knob)")
a = ap.parse_args()
lib = native.load()
native.ensure_device(0)
torch.cuda.set_device(0)
F, H, W, n, delta = a.frames, 2160, 3840, 3, 8.0
planes = Planes.contiguous(F, H, W)
size = F * H * W
cap = batch.capacity_bits(F, H, W, n)
nbytes = (cap + 7) // 8 + 8
st = torch.cuda.current_stream().cuda_stream
payload = torch.zeros(nbytes + 8, dtype=torch.uint8, device="cuda")
native.check(lib.svs_fill_bits_dev(payload.data_ptr(), cap, 1, 0, st), "bits")
ext = torch.zeros(nbytes + 8, dtype=torch.uint8, device="cuda")


def time_pair(gray, stego, pay=None):
    pay = pay or payload.data_ptr()
    native.check(lib.svs_fill_synthetic_dev(gray, C.byref(planes), 20250620, 0, 16, 224, st), "fill")
    for _ in range(3):
        batch.embed_device(gray, stego, planes, delta, n, pay, 0, cap, st)
        batch.extract_device(stego, planes, delta, n, ext.data_ptr(), ext.numel(), st)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.reps)]
    for e in ev:       # the bench's own pattern: embed, extract, embed, ...
        e[0].record()
        batch.embed_device(gray, stego, planes, delta, n, pay, 0, cap, st)
        e[1].record()
        batch.extract_device(stego, planes, delta, n, ext.data_ptr(), ext.numel(), st)
        e[2].record()
    torch.cuda.synchronize()
    te = [e[0].elapsed_time(e[1]) for e in ev]
    tx = [e[1].elapsed_time(e[2]) for e in ev]
    return float(np.median(te)), float(np.min(te)), float(np.median(tx))


def svs_malloc(nb):
    p = C.c_void_p()
    native.check(lib.svs_malloc(C.byref(p), nb), "malloc")
    return p


keep = []
print(f"# {a.tag} pid {os.getpid()}  {F} x 4K, n = 3, delta = 8; embed median / min, extract median (ms); address of cover, stego")
for k in range(a.pairs):
    g, s = torch.empty(size, dtype=torch.uint8, device="cuda"), torch.empty(size, dtype=torch.uint8, device="cuda")
    keep.append((g, s))
    me, mn, mx = time_pair(g.data_ptr(), s.data_ptr())
    print(f"torch     pair {k}: embed {me:.4f} / {mn:.4f}  extract {mx:.4f}   {g.data_ptr():#x} {s.data_ptr():#x}")
for k in range(a.pairs):
    g, s = svs_malloc(size), svs_malloc(size)
    keep.append((g, s))
    me, mn, mx = time_pair(g.value, s.value)
    print(f"hipmalloc pair {k}: embed {me:.4f} / {mn:.4f}  extract {mx:.4f}   {g.value:#x} {s.value:#x}")
for k in range(a.pairs):
    al = 2 << 20
    sz = (size + al - 1) // al * al
    arena = svs_malloc(2 * sz + nbytes + 8 + al)
    keep.append(arena)
    base = (arena.value + al - 1) // al * al
    pay = base + 2 * sz
    native.check(lib.svs_fill_bits_dev(pay, cap, 1, 0, st), "bits")
    me, mn, mx = time_pair(base, base + sz, pay)
    print(f"arena     pair {k}: embed {me:.4f} / {mn:.4f}  extract {mx:.4f}   {base:#x} {base + sz:#x}")
if a.contig:
    for k in range(a.pairs):
        os.environ["SVS_MALLOC_CONTIG"] = "2"          # fail rather than fall back
        try:
            g, s = svs_malloc(size), svs_malloc(size)
        except native.SvsNativeError as exc:
            print(f"contig    pair {k}: allocation failed ({exc})")
            break
        finally:
            os.environ["SVS_MALLOC_CONTIG"] = "0"
        keep.append((g, s))
        me, mn, mx = time_pair(g.value, s.value)
        print(f"contig    pair {k}: embed {me:.4f} / {mn:.4f}  extract {mx:.4f}   {g.value:#x} {s.value:#x}")
# the first pair again, now that 12 more pairs are resident: does the SAME memory still run at its earlier rate?
g, s = keep[0]
me, mn, mx = time_pair(g.data_ptr(), s.data_ptr())
print(f"torch     pair 0 again: embed {me:.4f} / {mn:.4f}  extract {mx:.4f}")
