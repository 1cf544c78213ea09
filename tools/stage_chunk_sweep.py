#!/usr/bin/env python3
"""Chunk size of the host-pointer staging pipeline (csrc/svs_capi.hip stage_chunk_bytes): sweep of the experiments library's
SVS_STAGE_CHUNK_KB on one 4K frame, one 1080p frame and a 32-frame 4K batch, page-locked buffers both ways; 0 = the built-in rule."""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
os.environ["SVSDCT_LIB"] = os.path.join(PKG, "lib", "variants", "libsvsdct_exp.so")
sys.path.insert(0, PKG)
import numpy as np

from svsdct import batch, hostmem, native, synth
from svsdct.native import Planes

lib = native.load()
native.ensure_device(0)
for (F, H, W, n, d) in ((1, 2160, 3840, 10, 20), (1, 1080, 1920, 10, 20), (32, 2160, 3840, 3, 8), (8, 1080, 1920, 10, 20)):
    src = hostmem.pinned_copy(np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (F, H, W))))
    dst = hostmem.pinned_empty(src.shape)
    bits = synth.synthetic_bits(batch.capacity_bits(F, H, W, n))
    packed = batch.pack_bits(bits)
    planes = Planes.contiguous(F, H, W)
    used = C.c_uint64()
    line = []
    for kb in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 20):
        os.environ["SVS_STAGE_CHUNK_KB"] = str(kb)
        ts = []
        for _ in range(25 if F == 1 else 8):
            t = time.perf_counter()
            rc = lib.svs_embed(src.ctypes.data, dst.ctypes.data, C.byref(planes), float(d), n, packed.ctypes.data, 0, bits.size,
                               native.SVS_EXACT_GUARDED, C.byref(used))
            ts.append(time.perf_counter() - t)
            assert rc == 0
        line.append(f"{kb if kb < (1 << 20) else 'whole'}: {np.median(ts[2:]) * 1e3:.3f}")
    print(f"{F} x {W}x{H} n={n}: svs_embed ms by chunk KB (0 = built-in rule) ->  " + "   ".join(line))

