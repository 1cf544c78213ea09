#!/usr/bin/env python3
"""One-off soak of the GUARDED embed mode on the GPU: random geometries, coefficient counts, quantiser steps of every
evaluation kind over and beyond the guard's delta range, budgets, bit offsets and content kinds; the stego frames must equal the
lane-per-block pocketfft kernel's (which the parity tests pin to the oracle) byte for byte.
usage: python tests/soak_guarded_gpu.py [iterations] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
from svsdct import batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def content(kind, f, h, w):
    if kind == 0:
        return rng.integers(0, 256, (f, h, w))
    if kind == 1:
        lo = int(rng.integers(0, 230)); return rng.integers(lo, lo + int(rng.integers(2, 26)), (f, h, w))
    if kind == 2:
        return np.full((f, h, w), int(rng.integers(0, 256)))
    if kind == 3:
        a, b = rng.uniform(-1.5, 1.5, 2)
        return (np.add.outer(np.arange(h) * a, np.arange(w) * b)[None] + rng.integers(0, 256) + np.zeros((f, 1, 1))) % 256
    if kind == 4:
        return rng.integers(0, 4, (f, h, w)) + int(rng.integers(0, 2)) * 252
    if kind == 5:      # rows or columns constant
        v = rng.integers(0, 256, (f, h, 1)) if rng.integers(0, 2) else rng.integers(0, 256, (f, 1, w))
        return v + np.zeros((f, h, w), int)
    if kind == 6:      # checker of random pitch and levels
        p = int(rng.choice([1, 2, 4, 8, 16])); a, b = rng.integers(0, 256, 2)
        yy, xx = np.mgrid[0:h, 0:w]
        return np.where(((yy // p) + (xx // p)) % 2 == 0, a, b)[None] + np.zeros((f, 1, 1), int)
    if kind == 7:      # smooth: low-pass noise
        g = rng.integers(0, 256, (f, h // 8 + 2, w // 8 + 2)).astype(float)
        up = np.kron(g, np.ones((8, 8)))[:, 4:4 + h, 4:4 + w]
        for ax in (1, 2):
            up = (up + np.roll(up, 3, ax) + np.roll(up, -3, ax) + np.roll(up, 5, ax)) / 4
        return up + rng.integers(-2, 3, (f, h, w))
    parts = [content(int(rng.integers(0, 8)), f, h, w) for _ in range(2)]     # kind 8: two kinds side by side
    out = np.array(parts[0], float); out[:, :, w // 2:] = np.array(parts[1], float)[:, :, w // 2:]
    return out


def step():
    d = float(np.exp(rng.uniform(np.log(0.2), np.log(5000.0))))
    k = int(rng.integers(0, 4))
    if k == 0:
        return float(2.0 ** round(np.log2(d)))           # power of two
    if k == 1:
        return float(np.float32(d))                       # a float32
    if k == 2:
        return float(round(d)) if d >= 1 else d           # integers (the GUI's steps) / small arbitrary doubles
    return d                                              # not a float32: double requantisation


done = 0
for it in range(iters):
    f = int(rng.integers(1, 4)); h = 8 * int(rng.integers(1, 69)); w = 8 * int(rng.integers(1, 121))
    n_ac = int(rng.integers(1, 16)) if rng.uniform() < 0.9 else int(rng.integers(16, 64))
    delta = step()
    frames = np.clip(np.asarray(content(int(rng.integers(0, 9)), f, h, w)), 0, 255).astype(np.uint8)
    cap = batch.capacity_bits(f, h, w, n_ac)
    off = int(rng.integers(0, 130))
    n_bits = cap if rng.uniform() < 0.5 else int(rng.integers(0, cap + 30))
    bits = rng.integers(0, 2, off + n_bits).astype(np.uint8) if rng.uniform() < 0.8 else np.zeros(off + n_bits, np.uint8)
    a, used_a = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="guarded")
    b, used_b = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="exact")
    assert used_a == used_b and np.array_equal(a, b), ("guarded != exact", it, f, h, w, n_ac, delta, off, n_bits,
                                                      int((a != b).sum()))
    done += f * h * w
    if it % 200 == 199:
        print(f"{it + 1} cases ok ({done / 1e6:.0f} Mpixel)", flush=True)
print(f"guarded soak ok: {iters} cases, {done / 1e6:.0f} Mpixel, guarded == exact everywhere")
