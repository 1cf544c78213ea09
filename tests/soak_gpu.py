#!/usr/bin/env python3
"""One-off soak: many random geometries / parameters through the real kernels, fast mode vs the CPU build of the
kernel header, exact mode vs the oracle (bit for bit).  usage: python tests/soak_gpu.py [iterations] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
for p in (os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import qim_dct_oracle as orc
from svsdct import batch
from testlib import emu_embed, emu_extract
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
deltas = [1, 2, 3, 4, 5, 6.5, 8, 10, 16, 20, 25, 32, 0.5, 100, 0, -2]
for it in range(iters):
    f = int(rng.integers(1, 5)); h = 8 * int(rng.integers(1, 40)); w = 8 * int(rng.integers(1, 60))
    n_ac = int(rng.integers(0, 66)); delta = deltas[int(rng.integers(0, len(deltas)))]
    kind = int(rng.integers(0, 4))
    frames = (rng.integers(0, 256, (f, h, w)) if kind == 0 else rng.integers(110, 150, (f, h, w)) if kind == 1 else
              np.full((f, h, w), int(rng.integers(0, 256))) if kind == 2 else
              (np.add.outer(np.arange(h) * 2, np.arange(w))[None] + np.zeros((f, 1, 1), int)) % 256).astype(np.uint8)
    cap = batch.capacity_bits(f, h, w, n_ac)
    off = int(rng.integers(0, 130)); n_bits = int(rng.integers(0, cap + 30))
    bits = rng.integers(0, 2, off + n_bits).astype(np.uint8)
    stego, used = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="fast")
    want, want_used = emu_embed(frames, delta, n_ac, bits, bit_offset=off)
    assert used == want_used and np.array_equal(stego, want), ("fast embed", it, f, h, w, n_ac, delta)
    packed, n = batch.extract_frames(stego, delta, n_ac, mode="fast")
    assert np.array_equal(np.unpackbits(packed, count=n), emu_extract(stego, delta, n_ac)), ("fast extract", it)
    for src in (frames, stego):        # FAST extraction is the reference's on any frame (no tie masks)
        packed, n = batch.extract_frames(src, delta, n_ac, mode="fast")
        assert np.array_equal(np.unpackbits(packed, count=n), orc.batch_extract_bits(src, delta, n_ac)), ("fast vs oracle", it)
    stego_x, used_x = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="exact")
    if delta > 0 and min(n_ac, 63) > 0 or n_bits == 0:
        ref, ref_used = orc.batch_embed(frames, delta, bits[off:], n_ac)
        assert used_x == ref_used and np.array_equal(stego_x, ref), ("exact embed", it, f, h, w, n_ac, delta, n_bits)
    else:
        for k in range(f):
            assert np.array_equal(stego_x[k], orc.frame_embed(frames[k], delta, bits[off:], n_ac)[1]), ("exact rt", it)
    for src in (frames, stego_x):
        packed, n = batch.extract_frames(src, delta, n_ac, mode="exact")
        assert np.array_equal(np.unpackbits(packed, count=n), orc.batch_extract_bits(src, delta, n_ac)), ("exact extract", it)
    if it % 50 == 49:
        print(f"{it + 1} cases ok", flush=True)
print("soak ok:", iters, "cases")
