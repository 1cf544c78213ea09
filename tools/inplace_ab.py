#!/usr/bin/env python3
"""Embed out of place (separate stego buffer, what bench.py measures) vs in place (stego written over the cover; the C ABI
allows d_gray == d_stego).  Same frames, same payload, interleaved rounds, HIP-event timing on the launch stream."""
import ctypes as C, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
F, H, W = 600, 2160, 3840
for n_ac, delta in ((3, 8.0), (10, 20.0)):
    planes = Planes.contiguous(F, H, W)
    cap = batch.capacity_bits(F, H, W, n_ac)
    cover = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    work = torch.empty_like(cover); stego = torch.empty_like(cover)
    bits = torch.empty((cap + 7) // 8 + 8, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.svs_fill_synthetic_dev(cover.data_ptr(), C.byref(planes), 20250620, 0, 16, 224, st)
    lib.svs_fill_bits_dev(bits.data_ptr(), cap, 20250620, 0, st)
    t = {"out_of_place": [], "in_place": []}
    for r in range(11):
        for name in t:
            work.copy_(cover)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst = stego if name == "out_of_place" else work
            batch.embed_device(work.data_ptr(), dst.data_ptr(), planes, delta, n_ac, bits.data_ptr(), 0, cap, st)
            e1.record(); torch.cuda.synchronize()
            if r >= 2:
                t[name].append(e0.elapsed_time(e1))
        same = bool(torch.equal(stego, work))
    nbytes = 2 * F * H * W + cap / 8
    for name, v in t.items():
        print(f"n={n_ac} delta={delta}: {name:13s} med {statistics.median(v):.4f} min {min(v):.4f} ms -> {nbytes / statistics.median(v) / 1e6:7.1f} GB/s")
    print("   identical stego:", same)
