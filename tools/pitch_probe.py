#!/usr/bin/env python3
"""Round 6: does the rate of the one-row embed kernel's ACCESS PATTERN (the launch with an empty payload: every lane copies its
rows, arithmetic skipped) depend on the frame geometry?  Same buffers, same byte count (to 0.1 %), different row lengths.
Experiments library (knobs): SVS_EMBED_WG_PER_CU from the command line.  Prints median ms per launch and GB/s per geometry and placement."""
import argparse, ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
sys.path.insert(0, PKG); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from svsdct import native
from svsdct.native import Planes
from testlib import EXPERIMENT_HOOKS

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=3); ap.add_argument("--burst", type=int, default=8); ap.add_argument("--cap", default="4")
ap.add_argument("--full", action="store_true", help="full payload (arithmetic on) instead of the empty one")
a = ap.parse_args()
lib = C.CDLL(os.path.join(PKG, "lib", "variants", "libsvsdct_exp.so"))
for nm, (res, args) in {**native.SIGNATURES, **EXPERIMENT_HOOKS}.items():
    fn = getattr(lib, nm); fn.restype, fn.argtypes = res, args
torch.cuda.set_device(0)
assert lib.svs_init(0) == 0
os.environ["SVS_EMBED_WG_PER_CU"] = a.cap
st = torch.cuda.current_stream().cuda_stream
TOTAL = 600 * 2160 * 3840
geoms = [(3840, 2160), (4096, 2024), (7680, 4320), (1920, 1080), (2048, 1080), (3584, 2160), (512, 512), (15360, 2160), (3840 + 256, 2160, 3840)]
pairs = []
for k in range(a.pairs):
    g, s = C.c_void_p(), C.c_void_p()
    assert lib.svs_malloc(C.byref(g), TOTAL + (1 << 22)) == 0 and lib.svs_malloc(C.byref(s), TOTAL + (1 << 22)) == 0
    pairs.append((g, s))
pay = torch.zeros(TOTAL // 64 * 3 // 8 + 64, dtype=torch.uint8, device="cuda")
assert lib.svs_fill_bits_dev(pay.data_ptr(), pay.numel() * 8 - 256, 1, 0, st) == 0
done = C.c_uint64()
print(f"# cap {a.cap} waves per SIMD; {'full payload' if a.full else 'empty payload (pattern copy)'}; median ms per launch (GB/s) per placement")
for geo in geoms:
    pitch, H = geo[0], geo[1]
    W = geo[2] if len(geo) > 2 else pitch
    F = TOTAL // (pitch * H)
    planes = Planes(F, H, W, 0, pitch, pitch * H)
    nbits = F * (H // 8) * (W // 8) * 3 if a.full else 0
    row = []
    for g, s in pairs:
        if a.full:
            assert lib.svs_fill_synthetic_dev(g, C.byref(planes), 20250620, 0, 16, 224, st) == 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.burst + 1)]
        ev[0].record()
        for i in range(a.burst):
            assert lib.svs_embed_dev(g, s, C.byref(planes), 8.0, 3, pay.data_ptr(), 0, nbits, 2, C.byref(done), st) == 0, lib.svs_last_error()
            ev[i + 1].record()
        torch.cuda.synchronize()
        t = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(1, a.burst)]))
        row.append((t, 2 * F * H * W / t / 1e6))
    print(f"W {W:6d} pitch {pitch:6d} H {H:5d} F {F:6d}: " + "  ".join(f"{t:.4f} ({gb:6.0f})" for t, gb in row))
# linear copy on the same pairs
row = []
for g, s in pairs:
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.burst + 1)]
    ev[0].record()
    for i in range(a.burst):
        assert lib.svs_ref_copy_dev(g, s, TOTAL, 3, st) == 0
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(1, a.burst)]))
    row.append((t, 2 * TOTAL / t / 1e6))
print("linear copy (16 B per lane)            : " + "  ".join(f"{t:.4f} ({gb:6.0f})" for t, gb in row))
