"""Run-time check of the device BGR->gray conversion against the OpenCV installed on this machine.

The reference converts with cv2.cvtColor(frame, COLOR_BGR2GRAY) (config_and_setup.py:112); OpenCV's fixed-point
table differs between generations (15-bit 3735/19235/9798 vs 14-bit 1868/9617/4899) and cv2 is not part of the
build image, so parity of the device conversion cannot be pinned at build time.
Instead the fused colour path is only used after `weights_matching_cv2` has found, on THIS machine, a table for
which the device kernel reproduces cv2 bit for bit on a probe frame that covers every (B, G, R) corner, grey ramp
and a seeded random field; if no table matches it raises and the callers keep converting on the host with cv2.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native
from .native import Planes

TABLES = {
    "15-bit (OpenCV >= 3.x)": (3735, 19235, 9798, 15),
    "14-bit (older OpenCV)": (1868, 9617, 4899, 14),
}
DEFAULT = TABLES["15-bit (OpenCV >= 3.x)"]


class ColourMismatch(RuntimeError):
    """cv2 is installed but none of the known fixed-point tables reproduces its BGR2GRAY."""


def probe_frame() -> np.ndarray:
    """uint8 [64, 96, 3]: all 8 corners of the colour cube, primary ramps, a grey ramp and random pixels."""
    rng = np.random.default_rng(20250620)
    f = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    corners = np.array([[b, g, r] for b in (0, 255) for g in (0, 255) for r in (0, 255)], np.uint8)
    f[0, :8] = corners
    ramp = np.arange(256, dtype=np.uint8)
    for ch in range(3):                                  # 256-step ramp of one channel, others 0 / 255
        block = np.zeros((256, 3), np.uint8)
        block[:, ch] = ramp
        f[8 + 3 * ch: 8 + 3 * ch + 3, :, :].reshape(-1, 3)[:256] = block
    grey = np.repeat(ramp[:, None], 3, axis=1)
    f[20:23].reshape(-1, 3)[:256] = grey
    return f


def device_gray(frames_bgr: np.ndarray, weights=None) -> np.ndarray:
    """BGR -> gray on the GPU (svs_bgr_to_gray_dev) for uint8 [F,H,W,3] with H, W multiples of 8."""
    lib = native.load()
    a = np.ascontiguousarray(frames_bgr, np.uint8)
    f, h, w, _ = a.shape
    planes = Planes.contiguous(f, h, w)
    out = np.empty((f, h, w), np.uint8)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    native.check(lib.svs_malloc(C.byref(d_in), a.nbytes), "svs_malloc")
    native.check(lib.svs_malloc(C.byref(d_out), out.nbytes), "svs_malloc")
    try:
        native.check(lib.svs_memcpy_h2d(d_in, a.ctypes.data, a.nbytes, None), "h2d")
        wt = None if weights is None else np.ascontiguousarray(weights, np.uint32)
        native.check(lib.svs_bgr_to_gray_dev(d_in, 3 * w, 3 * w * h, d_out, C.byref(planes),
                                             None if wt is None else wt.ctypes.data, None), "svs_bgr_to_gray_dev")
        native.check(lib.svs_memcpy_d2h(out.ctypes.data, d_out, out.nbytes, None), "d2h")
        native.check(lib.svs_stream_synchronize(None), "sync")
    finally:
        lib.svs_free(d_in)
        lib.svs_free(d_out)
    return out


_cache: dict = {}


def weights_matching_cv2(cv2_module=None, device: int = 0):
    """-> (wb, wg, wr, shift) for which the device conversion equals this machine's cv2 on the probe frame.
    Returns None when cv2 is not installed (nothing to compare with: the default table is used and parity stays
    unpinned); raises ColourMismatch when cv2 is installed and no table reproduces it."""
    if cv2_module is None:
        try:
            import cv2 as cv2_module  # noqa: WPS433
        except ImportError:
            return None
    key = (id(cv2_module), device)
    if key not in _cache:
        native.ensure_device(device)
        frame = probe_frame()
        want = cv2_module.cvtColor(frame, cv2_module.COLOR_BGR2GRAY)
        found = None
        for table in TABLES.values():
            if np.array_equal(device_gray(frame[None], table)[0], want):
                found = table
                break
        _cache[key] = found
    if _cache[key] is None:
        raise ColourMismatch("device BGR->gray does not reproduce this OpenCV build with any known table")
    return _cache[key]
