#!/usr/bin/env python3
"""Embed / extract kernel times against the number of coefficients per block: the default (guarded) mode - streaming kernels
with the rigorous guard up to n = 15, the lane-per-block pocketfft kernel above - next to the exact mode, and the two extract
families.  200 x 4K device-resident frames, delta 8 and 20.  (Round 3's version compared a contract-level FAST arithmetic for
n >= 16 with the exact kernel; that arithmetic is gone, profiles/history/r03_many_coefficients.txt keeps its numbers.)"""
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


def timed(fn):
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])


print(f"{F} x {W}x{H}: ms per launch (median of 5)")
for delta in (8.0, 20.0):
    for n in (1, 3, 7, 8, 10, 15, 16, 24, 32, 48, 63):
        cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8
        pay = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev); ext = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
        lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st)
        t_fast = timed(lambda: batch.embed_device(gray.data_ptr(), stego.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode="guarded"))
        t_exact = timed(lambda: batch.embed_device(gray.data_ptr(), stego.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode="exact"))
        x_fast = timed(lambda: batch.extract_device(stego.data_ptr(), planes, delta, n, ext.data_ptr(), ext.numel(), st, mode="fast"))
        x_exact = timed(lambda: batch.extract_device(stego.data_ptr(), planes, delta, n, ext.data_ptr(), ext.numel(), st, mode="exact"))
        eb = 2 * F * H * W + nbytes
        print(f"  delta {delta:4g} n {n:2d} rows {n // 8 + 1}: embed guarded {t_fast:6.3f} ({eb / t_fast / 1e6:5.0f} GB/s)  exact {t_exact:6.3f} "
              f"({eb / t_exact / 1e6:5.0f} GB/s) | extract fast {x_fast:6.3f}  exact {x_exact:6.3f}")
