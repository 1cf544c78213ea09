#!/usr/bin/env python3
"""Bandwidth of the kernels around the operator (colour plumbing, metrics) on device-resident 4K batches.
usage: python tools/aux_rates.py [n_ac=3] [delta=8]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import native
from svsdct.native import Planes
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
F, H, W = 200, 2160, 3840
planes = Planes.contiguous(F, H, W)
bgr = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev); gray2 = torch.empty_like(gray)
back = torch.empty_like(bgr)
st = torch.cuda.current_stream().cuda_stream
work = torch.empty(int(lib.svs_ssim_workspace_bytes(C.byref(planes))) // 8 + 8, dtype=torch.float64, device=dev)
ssim = torch.empty(F, dtype=torch.float64, device=dev); sse = torch.empty(F, dtype=torch.int64, device=dev)
def timed(name, fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
    print(f"{name:24s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {F * H * W / ms / 1e6:7.0f} Gpixel/s")
px = F * H * W
timed("bgr_to_gray", lambda: native.check(lib.svs_bgr_to_gray_dev(bgr.data_ptr(), 3 * W, 3 * W * H, gray.data_ptr(), C.byref(planes), None, st), "x"), 4 * px)
timed("gray_to_bgr", lambda: native.check(lib.svs_gray_to_bgr_dev(gray.data_ptr(), C.byref(planes), back.data_ptr(), 3 * W, 3 * W * H, st), "x"), 4 * px)
lib.svs_fill_synthetic_dev(gray2.data_ptr(), C.byref(planes), 1, 0, 16, 224, st)
timed("frame_sse (PSNR)", lambda: native.check(lib.svs_frame_sse_dev(gray.data_ptr(), gray2.data_ptr(), C.byref(planes), sse.data_ptr(), st), "x"), 2 * px)
timed("frame_ssim", lambda: native.check(lib.svs_frame_ssim_dev(gray.data_ptr(), gray2.data_ptr(), C.byref(planes), None, ssim.data_ptr(), work.data_ptr(), st), "x"), 2 * px)

# fused colour path vs the three-step chain (convert, embed, convert) on the same frames
from svsdct import batch
n_ac = int(sys.argv[1]) if len(sys.argv) > 1 else 3
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
print(f"-- fused colour path, n = {n_ac}, delta = {delta:g}")
cap = batch.capacity_bits(F, H, W, n_ac)
bits = torch.empty((cap + 7) // 8 + 8, dtype=torch.uint8, device=dev)
lib.svs_fill_bits_dev(bits.data_ptr(), cap, 5, 0, st)
out_bits = torch.empty_like(bits)
for mode in ("fast", "guarded", "exact"):
    def chain():
        native.check(lib.svs_bgr_to_gray_dev(bgr.data_ptr(), 3 * W, 3 * W * H, gray.data_ptr(), C.byref(planes), None, st), "x")
        batch.embed_device(gray.data_ptr(), gray2.data_ptr(), planes, delta, n_ac, bits.data_ptr(), 0, cap, st, mode=mode)
        native.check(lib.svs_gray_to_bgr_dev(gray2.data_ptr(), C.byref(planes), back.data_ptr(), 3 * W, 3 * W * H, st), "x")
    timed(f"3-step chain {mode}", chain, 10 * px)
    timed(f"fused embed_bgr {mode}", lambda: batch.embed_bgr_device(bgr.data_ptr(), back.data_ptr(), 0, planes, delta, n_ac, bits.data_ptr(), 0, cap, st, mode=mode), 6 * px)
    timed(f"fused +gray ref {mode}", lambda: batch.embed_bgr_device(bgr.data_ptr(), back.data_ptr(), gray.data_ptr(), planes, delta, n_ac, bits.data_ptr(), 0, cap, st, mode=mode), 7 * px)
def chain_x():
    native.check(lib.svs_bgr_to_gray_dev(back.data_ptr(), 3 * W, 3 * W * H, gray.data_ptr(), C.byref(planes), None, st), "x")
    batch.extract_device(gray.data_ptr(), planes, delta, n_ac, out_bits.data_ptr(), out_bits.numel(), st)
timed("2-step extract chain", chain_x, 5 * px)
timed("fused extract_bgr", lambda: batch.extract_bgr_device(back.data_ptr(), planes, delta, n_ac, out_bits.data_ptr(), out_bits.numel(), st), 3 * px)
torch.cuda.synchronize()
print("fused round trip bit errors:", int((out_bits[: cap // 8] != bits[: cap // 8]).sum().item()))
