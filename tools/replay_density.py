#!/usr/bin/env python3
"""FAST embed time against the share of blocks that go through the exact replay pass (flat areas: letterbox bars, panels).
Frames: hash noise with the top `share` of every frame replaced by a flat bar.  Prints FAST and EXACT kernel times."""
import ctypes as C, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes

F, H, W, n, delta = int(os.environ.get("FRAMES", "200")), 2160, 3840, int(os.environ.get("N_AC", "3")), 8.0
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
planes = Planes.contiguous(F, H, W)
cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8
gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev); stego = torch.empty_like(gray)
pay = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st)
print(f"{F} x {W}x{H}, n = {n}, delta = {delta:g}: ms per launch (median of 7)")
for share in (0.0, 0.01, 0.05, 0.125, 0.25, 0.5, 1.0):
    lib.svs_fill_synthetic_dev(gray.data_ptr(), C.byref(planes), 7, 0, 16, 224, st)
    rows = int(H * share) // 8 * 8
    if rows:
        gray[:, :rows, :] = 16
    torch.cuda.synchronize()
    out = {}
    for mode in ("fast", "exact"):
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            batch.embed_device(gray.data_ptr(), stego.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode=mode)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out[mode] = statistics.median(ts[2:])
        if mode == "fast":
            keep = stego.clone()
    same = bool(torch.equal(keep[:, :rows], stego[:, :rows])) if rows else True
    print(f"  flat share {share:5.3f}: fast {out['fast']:7.3f}  exact {out['exact']:7.3f}   flat area identical in both modes: {same}")
