#!/usr/bin/env python3
"""Per-call latency of the drop-in operator (one frame per call, host arrays in and out)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
import config_and_setup as cs
from svsdct import synth
for (h, w), n, d in [((480, 640), 10, 20), ((1080, 1920), 10, 20), ((2160, 3840), 3, 8)]:
    g = synth.synthetic_frames(1, h, w)[0]
    bits = "".join(map(str, synth.synthetic_bits((h // 8) * (w // 8) * n)))
    cs.proses_frame_qim_dct(g, "embed", d, bits, num_ac_coeffs_to_use=n)
    t = time.perf_counter()
    for _ in range(20):
        _, s, used = cs.proses_frame_qim_dct(g, "embed", d, bits, num_ac_coeffs_to_use=n)
    te = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20):
        out = cs.proses_frame_qim_dct(s, "extract", d, num_ac_coeffs_to_use=n)
    tx = (time.perf_counter() - t) / 20
    print(f"{w}x{h} n={n} delta={d}: embed {te*1e3:.2f} ms/call, extract {tx*1e3:.2f} ms/call, round trip ok={out == bits}")
