#!/usr/bin/env python3
"""FAST extraction (n >= 8) of stego frames vs never-embedded frames: how often the in-kernel exact fallback runs
(quantiser input within the forward error bound of a rounding tie) and what it costs.  Prints ms per launch."""
import ctypes as C, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes

F, H, W = 200, 2160, 3840
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
planes = Planes.contiguous(F, H, W)
gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev); stego = torch.empty_like(gray)
st = torch.cuda.current_stream().cuda_stream
lib.svs_fill_synthetic_dev(gray.data_ptr(), C.byref(planes), 7, 0, 16, 224, st)
for n, delta in ((10, 8.0), (10, 20.0), (10, 4.0), (20, 8.0), (63, 16.0)):
    cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8
    pay = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev); ext = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
    ext2 = torch.zeros_like(ext)
    lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st)
    batch.embed_device(gray.data_ptr(), stego.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode="fast")
    row = []
    for name, src in (("stego", stego), ("never embedded", gray)):
        for mode, out in (("fast", ext), ("exact", ext2)):
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                batch.extract_device(src.data_ptr(), planes, delta, n, out.data_ptr(), out.numel(), st, mode=mode)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            row.append(f"{name} {mode} {statistics.median(ts[2:]):.3f}")
        row.append("same bits" if torch.equal(ext[:nbytes], ext2[:nbytes]) else "BITS DIFFER")
    print(f"n = {n:2d} delta = {delta:4g}: " + " | ".join(row))
