"""Device-side evaluation of stego quality for whole batches: per-frame PSNR
(cv2.PSNR definition, exact integer SSE) and mean SSIM (skimage defaults) without copying frames back.
The drop-in `evaluation` module keeps the reference's host functions for the GUI; this module is what a
parity / quality harness over thousands of frames uses."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import native
from .native import Planes


class _Buf:
    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        native.check(native.load().svs_malloc(C.byref(self.ptr), max(nbytes, 8)), "svs_malloc")

    def read(self, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype)
        lib = native.load()
        native.check(lib.svs_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "svs_memcpy_d2h")
        native.check(lib.svs_stream_synchronize(None), "svs_stream_synchronize")
        return out

    def __del__(self):
        try:
            native.load().svs_free(self.ptr)
        except Exception:
            pass


def psnr_ssim_device(d_a: int, d_b: int, planes: Planes, data_range: float | None = 255.0, stream: int = 0):
    """-> (psnr_db[F] float64, ssim[F] float64) for device-resident batches a (reference) and b.
    data_range None reproduces evaluation.calc_ssim's max(b) - min(b) per frame."""
    lib = native.load()
    f = planes.n_frames
    sse, ssim = _Buf(8 * f), _Buf(8 * f)
    work = _Buf(int(lib.svs_ssim_workspace_bytes(C.byref(planes))))
    rng = None
    if data_range is not None:
        rng = _Buf(8 * f)
        host = np.full(f, float(data_range), np.float64)
        native.check(lib.svs_memcpy_h2d(rng.ptr, host.ctypes.data, host.nbytes, stream or None), "svs_memcpy_h2d")
    native.check(lib.svs_frame_sse_dev(d_a, d_b, C.byref(planes), sse.ptr, stream or None), "svs_frame_sse_dev")
    native.check(lib.svs_frame_ssim_dev(d_a, d_b, C.byref(planes), rng.ptr if rng else None, ssim.ptr, work.ptr,
                                        stream or None), "svs_frame_ssim_dev")
    native.check(lib.svs_stream_synchronize(stream or None), "svs_stream_synchronize")
    sse_h = sse.read(f, np.uint64).astype(np.float64)
    n = planes.height * planes.width
    psnr = np.array([math.inf if s == 0 else 10.0 * math.log10(255.0 ** 2 * n / s) for s in sse_h])
    return psnr, ssim.read(f, np.float64)
