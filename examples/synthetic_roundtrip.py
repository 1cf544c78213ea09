#!/usr/bin/env python3
"""End-to-end example without OpenCV / cryptography: synthetic 1080p clip, a framed payload with the reference's
wire format (svsdct.framing), embed on the GPU, extract, parse the header, quality metrics on the device.

    python examples/synthetic_roundtrip.py [--mode guarded|fast|exact] [--frames 8] [--n-ac 10] [--delta 20]
"""
import argparse
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np  # noqa: E402

from svsdct import batch, framing, metrics, native, synth  # noqa: E402
from svsdct.native import Planes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="guarded", choices=["guarded", "fast", "exact"])
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--n-ac", type=int, default=10)
ap.add_argument("--delta", type=float, default=20)
a = ap.parse_args()

h, w = 1080, 1920
clip = synth.synthetic_frames(a.frames, h, w, seed=1)
secret = (np.add.outer(np.arange(64), np.arange(64)) * 2 % 256).astype(np.uint8)       # a 64x64 gray "image"
payload = framing.build_payload_bits(64, 64, b"\x02" + bytes(32), bytes(16), bytes(32), bytes(12), bytes(16),
                                     secret.tobytes())                                   # unencrypted, for the demo
print(f"payload {payload.size} bits; capacity {batch.capacity_bits(1, h, w, a.n_ac)} bits per frame")

t = time.perf_counter()
stego, used = batch.embed_frames(clip, a.delta, a.n_ac, payload, mode=a.mode)
packed, n_bits = batch.extract_frames(stego, a.delta, a.n_ac, mode=a.mode)
dt = time.perf_counter() - t
stream = np.unpackbits(packed, count=n_bits)
hdr = framing.parse_header(stream)
got = np.packbits(stream[hdr.bits:hdr.bits + 8 * hdr.ciphertext_len]).reshape(hdr.height, hdr.width)
print(f"mode {a.mode}: embedded {used} bits, extracted {n_bits} bits in {dt * 1e3:.1f} ms (host arrays in and out)")
print("header:", hdr.width, "x", hdr.height, "ciphertext bytes", hdr.ciphertext_len, "| image recovered:",
      bool(np.array_equal(got, secret)))

# quality of the stego frames, computed on the device
lib = native.load()
planes = Planes.contiguous(a.frames, h, w)
d_a, d_b = C.c_void_p(), C.c_void_p()
for ptr, arr in ((d_a, clip), (d_b, stego)):
    native.check(lib.svs_malloc(C.byref(ptr), arr.nbytes), "malloc")
    native.check(lib.svs_memcpy_h2d(ptr, arr.ctypes.data, arr.nbytes, None), "h2d")
psnr, ssim = metrics.psnr_ssim_device(d_a, d_b, planes, data_range=255.0)
lib.svs_free(d_a)
lib.svs_free(d_b)
changed = [k for k in range(a.frames) if np.isfinite(psnr[k])]
print("frames carrying payload:", changed, "| PSNR", [round(float(psnr[k]), 2) for k in changed],
      "dB | SSIM", [round(float(ssim[k]), 4) for k in changed])
