#!/usr/bin/env python3
"""Host-pointer staging: the two-stream pipeline (upload stream + download stream joined by events) against everything on
one stream (experiments library, SVS_STAGE_MODE = 0 / 2), page-locked buffers both ways: median, 10th and 90th percentile
of the call time - the cross-stream event wait is the one step whose latency the host decides."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
os.environ["SVSDCT_LIB"] = os.path.join(PKG, "lib", "variants", "libsvsdct_exp.so")
sys.path.insert(0, PKG)
import numpy as np
from svsdct import batch, hostmem, native, synth
from svsdct.native import Planes
lib = native.load(); native.ensure_device(0)
for (F, H, W, n, d, reps) in ((1, 2160, 3840, 10, 20, 60), (2, 2160, 3840, 10, 20, 40), (32, 2160, 3840, 3, 8, 9), (1, 4320, 7680, 3, 8, 30)):
    src = hostmem.pinned_copy(np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (F, H, W))))
    dst = hostmem.pinned_empty(src.shape)
    bits = synth.synthetic_bits(batch.capacity_bits(F, H, W, n)); packed = batch.pack_bits(bits)
    planes = Planes.contiguous(F, H, W); used = C.c_uint64()
    out = []
    for rnd in range(2):
        for mode in (0, 2):
            os.environ["SVS_STAGE_MODE"] = str(mode)
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                rc = lib.svs_embed(src.ctypes.data, dst.ctypes.data, C.byref(planes), float(d), n, packed.ctypes.data, 0, bits.size, 2, C.byref(used))
                ts.append(time.perf_counter() - t); assert rc == 0
            ts = np.array(ts[3:]) * 1e3
            out.append(f"mode {mode}: {np.median(ts):.3f} [{np.percentile(ts, 10):.3f} .. {np.percentile(ts, 90):.3f}]")
    print(f"{F} x {W}x{H} n={n} (ms, median [p10 .. p90], two rounds): " + "   ".join(out))
