"""NumPy arrays in page-locked host memory, recycled.

The host-pointer entry points of the C ABI (svs_embed / svs_extract ...) move a page-locked buffer with the DMA engines
directly; pageable memory takes the HIP runtime's own staging (uploads at the same rate, downloads at half - a pageable
download also blocks the calling thread, so the two-stream overlap of the call needs page-locked outputs; a staging ring of the
library's own was measured and deleted, profiles/r05_stage_ring_ab.txt), and a FRESH pageable result array is worse still: every 4 KB page of it faults on first touch (an 8 MB stego frame = 2 025 faults per call of the reference's
frame loop, embed_process.py:117-121).  So the arrays this package hands back to its callers - the stego frames, the gray
reference copy the operator returns (config_and_setup.py:113-114,172) - come from `pinned_empty`: memory from svs_host_alloc
wrapped as an ordinary `numpy.ndarray`.  When the last view of such an array is garbage-collected the buffer goes back to a
free list keyed by size, so a frame loop cycles through two or three buffers and never allocates.

Arrays behave like any other NumPy array (the memory is ordinary cacheable host memory that happens to be pinned).
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref

import numpy as np

from . import native

_GRANULE = 1 << 16               # buffer sizes are rounded up to 64 KB
_KEEP_BYTES = 512 << 20          # free buffers kept for reuse; beyond that they are returned to the driver
_MAX_LIVE_BYTES = 4 << 30        # page-locked bytes in the hands of callers + in the pool; beyond that (a caller that keeps every
                                 # stego frame of a clip) arrays are ordinary pageable ones: slower downloads, same results
_lock = threading.Lock()
_free: dict[int, list[int]] = {}  # size -> pointers
_free_bytes = 0
_live_bytes = 0                   # allocated from the driver and not yet returned to it
stats = {"allocated": 0, "reused": 0, "released": 0, "pageable": 0}


def _give_back(ptr: int, size: int) -> None:
    global _free_bytes, _live_bytes
    try:
        with _lock:
            if _free_bytes + size <= _KEEP_BYTES:
                _free.setdefault(size, []).append(ptr)
                _free_bytes += size
                return
            _live_bytes -= size
        native.load().svs_host_free(C.c_void_p(ptr))
        stats["released"] += 1
    except Exception:       # interpreter shutdown: the process is going away with its memory
        pass


def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """Uninitialised array of `shape` / `dtype` in page-locked memory (needs the HIP library, like every product path: raises
    SvsNativeError when it cannot be loaded).  When the driver refuses more page-locked memory, or _MAX_LIVE_BYTES of it are out
    already, the array is an ordinary pageable one - the library handles both."""
    global _free_bytes, _live_bytes
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    size = max(_GRANULE, -(-nbytes // _GRANULE) * _GRANULE)
    ptr = None
    with _lock:
        bucket = _free.get(size)
        if bucket:
            ptr = bucket.pop()
            _free_bytes -= size
            stats["reused"] += 1
    if ptr is None:
        lib = native.load()
        with _lock:
            room = _live_bytes + size <= _MAX_LIVE_BYTES
            if room:
                _live_bytes += size
        p = C.c_void_p()
        if not room or lib.svs_host_alloc(C.byref(p), size) != native.SVS_OK or not p.value:
            if room:
                with _lock:
                    _live_bytes -= size
            stats["pageable"] += 1
            return np.empty(shape, dtype)
        ptr = p.value
        stats["allocated"] += 1
    raw = (C.c_uint8 * size).from_address(ptr)
    weakref.finalize(raw, _give_back, ptr, size)       # runs when the last array viewing `raw` is gone
    return np.frombuffer(raw, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


def pinned_copy(a: np.ndarray) -> np.ndarray:
    """C-contiguous copy of `a` in page-locked memory"""
    out = pinned_empty(a.shape, a.dtype)
    np.copyto(out, a)
    return out


def trim() -> None:
    """Return every free buffer to the driver (tests; long-lived processes after a large batch)."""
    global _free_bytes, _live_bytes
    with _lock:
        pending = [(p, s) for s, ps in _free.items() for p in ps]
        _free.clear()
        _free_bytes = 0
        _live_bytes -= sum(s for _, s in pending)
    for p, _ in pending:
        native.load().svs_host_free(C.c_void_p(p))
        stats["released"] += 1
