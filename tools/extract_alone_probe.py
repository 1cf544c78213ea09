#!/usr/bin/env python3
"""The one-row extract kernel on its own (as the extract pipeline runs it - no embed launch in front): one vs two blocks per lane,
(a) a sustained burst of launches, (b) single launches between synchronisations, (c) single launches after 20 ms of idle GPU."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
sys.path.insert(0, PKG)
import numpy as np, torch
from svsdct import batch, native
from svsdct.native import Planes

def load(path):
    lib = C.CDLL(path)
    for nm, (res, args) in native.SIGNATURES.items():
        fn = getattr(lib, nm); fn.restype, fn.argtypes = res, args
    return lib

libs = {"1 block/lane (default)": load(os.path.join(PKG, "lib", "libsvsdct.so")), "2 blocks/lane": load(os.path.join(PKG, "lib", "variants", "libsvsdct_xbpl2.so"))}
torch.cuda.set_device(0)
for (F, H, W, n) in ((600, 2160, 3840, 3), (300, 1080, 1920, 3), (32, 2160, 3840, 7)):
    planes = Planes.contiguous(F, H, W)
    cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8 + 16
    st = torch.cuda.current_stream().cuda_stream
    gray = torch.empty(F * H * W, dtype=torch.uint8, device="cuda"); out = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    l0 = next(iter(libs.values())); assert l0.svs_init(0) == 0
    assert l0.svs_fill_synthetic_dev(gray.data_ptr(), C.byref(planes), 7, 0, 16, 224, st) == 0
    got = C.c_uint64()
    res = {}
    for rnd in range(3):
        for name, lib in libs.items():
            def launch():
                assert lib.svs_extract_dev(gray.data_ptr(), C.byref(planes), 8.0, n, out.data_ptr(), out.numel(), 2, C.byref(got), st) == 0
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            ev[0].record()
            for i in range(10):
                launch(); ev[i + 1].record()
            torch.cuda.synchronize()
            res.setdefault((name, "burst"), []).extend(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 10))
            for mode, gap in (("single", 0.0), ("after idle", 0.02)):
                for _ in range(6):
                    torch.cuda.synchronize(); time.sleep(gap)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
                    res.setdefault((name, mode), []).append(e0.elapsed_time(e1))
    print(f"{F} x {W}x{H} n={n}: " + "  |  ".join(f"{name}: " + ", ".join(f"{m} {np.median(res[(name, m)]):.4f}" for m in ("burst", "single", "after idle")) for name in libs))
