#!/usr/bin/env python3
"""Does the embed kernel's time depend on where the stego buffer lies relative to the cover buffer?  (bench.py showed two
states per PROCESS - 1.58 or 1.68 ms per 600 x 4K embed launch - with identical settings: the allocator's addresses are the
only thing that differs.)  One allocation, cover at its start, stego at start + size + pad; sustained bursts per pad.
usage: python tools/buffer_offset_probe.py [--frames 600] [--n-ac 3]"""
import argparse, ctypes as C, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=600); ap.add_argument("--n-ac", type=int, default=3)
ap.add_argument("--delta", type=float, default=8.0); ap.add_argument("--h", type=int, default=2160); ap.add_argument("--w", type=int, default=3840)
a = ap.parse_args()
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
F, H, W, n = a.frames, a.h, a.w, a.n_ac
size = F * H * W
planes = Planes.contiguous(F, H, W)
cap = batch.capacity_bits(F, H, W, n)
pads = [0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 3 << 20, 1 << 22,
        (1 << 22) + 4096, 1 << 23, 1 << 24, (1 << 24) + 8192, 1 << 25]
big = torch.empty(2 * size + max(pads) + (1 << 22), dtype=torch.uint8, device=dev)
base = big.data_ptr()
start = (-base) % (1 << 21)                       # cover at a 2 MB boundary
pay = torch.zeros((cap + 7) // 8 + 8, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
gray_ptr = base + start
lib.svs_fill_synthetic_dev(gray_ptr, C.byref(planes), 20250620, 0, 16, 224, st)
lib.svs_fill_bits_dev(pay.data_ptr(), cap, 20250620, 0, st)
torch.cuda.synchronize()
print(f"{F} x {W}x{H}, n = {n}: cover at {hex(gray_ptr)}, frame bytes {H * W}, batch bytes {size} (= {size / (1 << 21):.2f} x 2 MiB)")


def burst(stego_ptr, reps=20):
    for _ in range(5):
        batch.embed_device(gray_ptr, stego_ptr, planes, a.delta, n, pay.data_ptr(), 0, cap, st, mode="guarded")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        batch.embed_device(gray_ptr, stego_ptr, planes, a.delta, n, pay.data_ptr(), 0, cap, st, mode="guarded")
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


for rep in range(2):
    for pad in pads:
        stego_ptr = gray_ptr + size + pad
        ms = burst(stego_ptr)
        print(f"  pass {rep}  stego - cover = size + {pad:9d}  ((stego - cover) mod 2 MiB = {(size + pad) % (1 << 21):8d}, mod 64 KiB = {(size + pad) % 65536:6d})  {ms:.4f} ms")
