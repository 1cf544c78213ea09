"""Synthetic cover frames and payload bits, identical on CPU (NumPy, here) and on
the GPU (`svs_fill_synthetic_dev` in csrc/svs_kernels.hip runs the same hash).

SURVEY.md section 8(d): gray uint8 planes, value = 16 + (hash32(seed, f, y, x) mod 224),
i.e. in [16, 240) so that n<=7, delta<=16 embedding can never clip at 0/255; payload is a
seeded Bernoulli(1/2) bit stream.  hash32 is the public-domain "lowbias32" integer
finaliser applied to a linear combination of the coordinates.
"""
from __future__ import annotations

import numpy as np

SEED_DEFAULT = 20250620
_M32 = np.uint64(0xFFFFFFFF)


def _lowbias32(h: np.ndarray) -> np.ndarray:
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & _M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash32(seed: int, f, y, x) -> np.ndarray:
    f = np.asarray(f, np.uint64)
    y = np.asarray(y, np.uint64)
    x = np.asarray(x, np.uint64)
    h = (np.uint64(seed & 0xFFFFFFFF)
         + f * np.uint64(0x9E3779B1) + y * np.uint64(0x85EBCA6B) + x * np.uint64(0xC2B2AE35))
    return _lowbias32(h)


def synthetic_frames(n_frames: int, height: int, width: int, seed: int = SEED_DEFAULT,
                     first_frame: int = 0, lo: int = 16, span: int = 224) -> np.ndarray:
    """uint8 [F, H, W]; frame index starts at `first_frame` (rank sharding)."""
    f = np.arange(first_frame, first_frame + n_frames, dtype=np.uint64)[:, None, None]
    y = np.arange(height, dtype=np.uint64)[None, :, None]
    x = np.arange(width, dtype=np.uint64)[None, None, :]
    h = hash32(seed, f, y, x)
    return (np.uint64(lo) + h % np.uint64(span)).astype(np.uint8)


def synthetic_bits(n_bits: int, seed: int = SEED_DEFAULT, first_bit: int = 0) -> np.ndarray:
    """uint8 0/1 array: bit i = lowbias32(seed*0x632BE5AB + i) >> 31 (counter based, so any
    rank can produce its own slice of the global stream)."""
    i = np.arange(first_bit, first_bit + n_bits, dtype=np.uint64)
    h = _lowbias32((np.uint64((seed * 0x632BE5AB) & 0xFFFFFFFF) + i) & _M32)
    return (h >> np.uint64(31)).astype(np.uint8)
