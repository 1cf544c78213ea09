"""Batched embed / extract over stacks of gray frames - the frame loops of the reference
(embed_process.py:108-128, extract_process.py:55-86,173-182) turned into one kernel launch per
batch.  Frame k of a batch takes stream bits [k*cap, (k+1)*cap) (cap = blocks per frame * n_ac);
the extracted stream is the frames' bits in order.

Two levels:
  * `embed_frames` / `extract_frames`      NumPy in, NumPy out (host entry points of the C ABI).
  * `embed_device` / `extract_device`      raw device pointers + stream (what bench.py and a
                                           device-resident pipeline use); nothing is copied.
Bits travel packed MSB-first (numpy.packbits order), so `unpack_to_str` of the extract output is
the reference operator's '0'/'1' string.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import native
from .hostmem import pinned_empty
from .native import Planes

MAX_AC = 63

# Transform mode (include/svsdct.h `flags`).  Every mode produces the reference's stego pixels and bits; the modes only
# differ in which kernels get there:
#   "guarded" n_ac <= 15 and 0.25 <= delta <= 4096: the streaming kernel (cheap sparse transform wherever a rigorous bound
#            on the reference's float32 round-trip noise proves it equals the reference's truncation, the pocketfft-identical
#            arithmetic inside the same launch for the blocks where it cannot; 8 tests per block at n_ac <= 7, 64 at
#            n_ac = 8..15); otherwise the "exact" kernels.  Extraction: n_ac <= 7 the pocketfft-identical forward, n_ac >= 8 the
#            FMA-factored forward with a proven per-block tie margin (the reference's bits for any input).
#   "fast"   flags = 0 of the C ABI: the same launches as "guarded" (kept for ABI compatibility).
#   "exact"  pocketfft-identical arithmetic on every block, one lane per block - the yardstick (VALU-bound).
# DEFAULT_MODE is THE product default: every entry point of this module, FramePipeline, the drop-in operator, both drop-in
# video loops and bench.py resolve an unspecified mode through resolve_mode() - nothing else chooses one
# (tests/test_capi_cpu.py::test_one_default_mode_everywhere).  The Python layer (not the shared library) lets
# SVS_DCT_MODE=fast|exact|guarded override it for a whole process.
DEFAULT_MODE = "guarded"
_MODE_FLAGS = {"fast": 0, "exact": native.SVS_EXACT_POCKETFFT, "guarded": native.SVS_EXACT_GUARDED}
_ENV_MODE = os.environ.get("SVS_DCT_MODE")


def resolve_mode(mode: str | None = None) -> str:
    """explicit argument > SVS_DCT_MODE > DEFAULT_MODE"""
    mode = mode or _ENV_MODE or DEFAULT_MODE
    if mode not in _MODE_FLAGS:
        raise ValueError(f"unknown transform mode {mode!r} (use 'fast', 'exact' or 'guarded')")
    return mode


def mode_flags(mode: str | None = None) -> int:
    """`flags` word of the C ABI for a transform mode (None = the product default)"""
    return _MODE_FLAGS[resolve_mode(mode)]


def host_level_mode() -> str:
    """transform mode of the drop-in operator and video pipelines: the product default"""
    return resolve_mode(None)


def clamp_ac(n_ac) -> int:
    return max(0, min(int(n_ac), MAX_AC))  # config_and_setup.py:138


def capacity_bits(n_frames: int, height: int, width: int, n_ac) -> int:
    return n_frames * (height // 8) * (width // 8) * clamp_ac(n_ac)


# ---- payload forms --------------------------------------------------------------------------
def str_to_bits(payload: str, limit: int | None = None) -> np.ndarray:
    """'0101...' -> uint8 0/1 array (vectorised; only the first `limit` characters are touched)."""
    if limit is not None:
        payload = payload[:limit]
    return np.frombuffer(payload.encode("ascii"), np.uint8) - np.uint8(48)


def bits_to_str(bits: np.ndarray) -> str:
    return (np.asarray(bits, np.uint8) + np.uint8(48)).tobytes().decode("ascii")


def pack_bits(bits: np.ndarray) -> np.ndarray:
    """0/1 array -> MSB-first packed bytes, zero padded to a multiple of 4 bytes (the C ABI reads
    the payload as aligned dwords)."""
    packed = np.packbits(np.asarray(bits, np.uint8))
    pad = (-packed.size) % 4
    if pad or packed.size == 0:
        packed = np.concatenate([packed, np.zeros(pad if packed.size else 4, np.uint8)])
    return packed


def _pack_window(bits: np.ndarray, bit_offset: int, n_bits: int):
    """Pack only the part of a long 0/1 stream one call reads: -> (packed bytes, bit offset rebased onto them).
    A frame loop that walks one stream by bit offset then packs O(batch) bits per call, not O(stream)."""
    start = (int(bit_offset) // 32) * 32
    return pack_bits(bits[start:int(bit_offset) + int(n_bits)]), int(bit_offset) - start


def unpack_to_str(packed: np.ndarray, n_bits: int) -> str:
    return bits_to_str(np.unpackbits(np.asarray(packed, np.uint8), count=n_bits))


# ---- host-array level -----------------------------------------------------------------------
def _as_stack(frames: np.ndarray) -> np.ndarray:
    a = np.asarray(frames)
    if a.dtype != np.uint8:
        raise TypeError("frames must be uint8")
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("frames must be [H,W] or [F,H,W] gray planes")
    if a.shape[1] % 8 or a.shape[2] % 8:
        raise ValueError("frame height and width must be multiples of 8")
    return np.ascontiguousarray(a)


def embed_frames(frames: np.ndarray, delta, n_ac, bits, bit_offset: int = 0, n_bits: int | None = None,
                 device: int = 0, mode: str | None = None):
    """Embed a bit stream into a stack of gray frames on the GPU.

    frames : uint8 [F,H,W] (or [H,W]);  bits : 0/1 array or '0'/'1' str (the stream; bit
    `bit_offset` is the first one used);  n_bits : bits available from bit_offset (default: rest).
    Returns (stego uint8 [F,H,W], n_embedded)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_stack(frames)
    f, h, w = stack.shape
    if isinstance(bits, str):
        bits = str_to_bits(bits)
    bits = np.asarray(bits, np.uint8)
    if n_bits is None:
        n_bits = max(0, bits.size - bit_offset)
    if bit_offset + n_bits > bits.size:
        raise ValueError("bit_offset + n_bits exceeds the payload length")
    packed, bit_offset = _pack_window(bits, bit_offset, n_bits)
    stego = pinned_empty(stack.shape)      # page-locked: the download lands in it by DMA, no staging copy, no page faults
    done = C.c_uint64(0)
    planes = Planes.contiguous(f, h, w)
    rc = lib.svs_embed(stack.ctypes.data, stego.ctypes.data, C.byref(planes), float(delta), int(n_ac),
                       packed.ctypes.data, int(bit_offset), int(n_bits), mode_flags(mode), C.byref(done))
    native.check(rc, "svs_embed")
    return stego, int(done.value)


def extract_frames(frames: np.ndarray, delta, n_ac, device: int = 0, mode: str | None = None):
    """Extract the packed bit stream of a stack of gray frames on the GPU.
    Returns (packed uint8 [ceil(n_bits/8)], n_bits)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_stack(frames)
    f, h, w = stack.shape
    cap = capacity_bits(f, h, w, n_ac)
    out = np.zeros(max(4, (cap + 7) // 8 + (-((cap + 7) // 8)) % 4), np.uint8)
    got = C.c_uint64(0)
    planes = Planes.contiguous(f, h, w)
    rc = lib.svs_extract(stack.ctypes.data, C.byref(planes), float(delta), int(n_ac), out.ctypes.data,
                         out.size, mode_flags(mode), C.byref(got))
    native.check(rc, "svs_extract")
    n = int(got.value)
    return out[: (n + 7) // 8], n


# ---- the reference operator's own payload types: '0' / '1' strings ------------------------------------------
_utf8_and_size = None


def _ascii_address(text: str):
    """(address, length) of the characters of an ASCII str WITHOUT copying them (CPython keeps ASCII strings as one byte per
    character; PyUnicode_AsUTF8AndSize hands out that very buffer).  The reference's frame loop passes the whole remaining
    payload to every call (embed_process.py:116) - slicing or encoding it per frame would copy it per frame."""
    global _utf8_and_size
    if not isinstance(text, str):
        raise TypeError("the payload must be a str of '0' / '1' characters (bit_payload_segment of the reference operator); "
                        "use embed_frames for arrays of bits")
    if _utf8_and_size is None:
        fn = C.pythonapi.PyUnicode_AsUTF8AndSize
        fn.restype, fn.argtypes = C.c_void_p, [C.py_object, C.POINTER(C.c_ssize_t)]
        _utf8_and_size = fn
    size = C.c_ssize_t(0)
    addr = _utf8_and_size(text, C.byref(size))
    if not addr or size.value != len(text):
        raise ValueError("the payload must be a string of '0' / '1' characters")
    return addr, size.value


def embed_frames_str(frames: np.ndarray, delta, n_ac, payload: str | None, device: int = 0, mode: str | None = None,
                     want_gray: bool = False):
    """`embed_frames` in the reference operator's own types (config_and_setup.py:106-109,172): the payload is a '0'/'1'
    string (bit_payload_segment) of which at most the capacity is read from the front - the characters go to the device as
    they are and are packed there (svs_embed_str); None / "" = nothing to embed.  want_gray: also return a copy of the input
    frames as an array of its own, the operator's first return value - the library makes it while the GPU works.
    Returns (stego uint8 [F,H,W], n_embedded) or (gray_copy, stego, n_embedded)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_stack(frames)
    f, h, w = stack.shape
    addr, n_chars = _ascii_address(payload) if payload else (None, 0)
    stego = pinned_empty(stack.shape)
    gray = np.empty_like(stack) if want_gray else None
    done = C.c_uint64(0)
    planes = Planes.contiguous(f, h, w)
    rc = lib.svs_embed_str(stack.ctypes.data, gray.ctypes.data if want_gray else None, stego.ctypes.data, C.byref(planes),
                           float(delta), int(n_ac), addr, n_chars, mode_flags(mode), C.byref(done))
    native.check(rc, "svs_embed_str")
    return (gray, stego, int(done.value)) if want_gray else (stego, int(done.value))


def extract_frames_str(frames: np.ndarray, delta, n_ac, device: int = 0, mode: str | None = None) -> str:
    """`extract_frames` returning the reference operator's own type: the '0'/'1' string of config_and_setup.py:173-174
    (expanded on the device, one decode on the host)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_stack(frames)
    f, h, w = stack.shape
    cap = capacity_bits(f, h, w, n_ac)
    if cap == 0:
        return ""
    out = pinned_empty(cap)
    got = C.c_uint64(0)
    planes = Planes.contiguous(f, h, w)
    rc = lib.svs_extract_str(stack.ctypes.data, C.byref(planes), float(delta), int(n_ac), out.ctypes.data, cap,
                             mode_flags(mode), C.byref(got))
    native.check(rc, "svs_extract_str")
    return str(memoryview(out)[: int(got.value)], "ascii")


# ---- device-pointer level -------------------------------------------------------------------
def embed_device(d_gray: int, d_stego: int, planes: Planes, delta, n_ac, d_bits_packed: int,
                 bit_offset: int, n_bits: int, stream: int = 0, mode: str | None = None) -> int:
    """Enqueue the embed kernel on `stream` (a hipStream_t handle as int); returns bits embedded."""
    done = C.c_uint64(0)
    rc = native.load().svs_embed_dev(d_gray, d_stego, C.byref(planes), float(delta), int(n_ac), d_bits_packed,
                                     int(bit_offset), int(n_bits), mode_flags(mode), C.byref(done),
                                     stream or None)
    native.check(rc, "svs_embed_dev")
    return int(done.value)


def extract_device(d_gray: int, planes: Planes, delta, n_ac, d_bits_out: int, out_capacity_bytes: int,
                   stream: int = 0, mode: str | None = None) -> int:
    """Enqueue the extract kernel on `stream`; returns the number of bits the batch yields."""
    got = C.c_uint64(0)
    rc = native.load().svs_extract_dev(d_gray, C.byref(planes), float(delta), int(n_ac), d_bits_out,
                                       int(out_capacity_bytes), mode_flags(mode), C.byref(got),
                                       stream or None)
    native.check(rc, "svs_extract_dev")
    return int(got.value)


# ---- fused colour path (BGR in, BGR out) ---------------------------------
def _weights_arg(weights):
    if weights is None:
        return None, None
    w = np.ascontiguousarray(weights, np.uint32)
    if w.shape != (4,):
        raise ValueError("weights must be (wb, wg, wr, shift)")
    return w, w.ctypes.data


def embed_bgr_device(d_bgr_in: int, d_bgr_out: int, d_gray_ref: int, planes: Planes, delta, n_ac,
                     d_bits_packed: int, bit_offset: int, n_bits: int, stream: int = 0, mode: str | None = None,
                     weights=None, in_pitches=None, out_pitches=None) -> int:
    """Enqueue the fused BGR -> gray -> embed -> BGR kernel over packed (or pitched) interleaved BGR frames;
    `d_gray_ref` (0 to skip) receives the gray frames before embedding.  Returns bits embedded."""
    irp, ifp = in_pitches or (3 * planes.width, 3 * planes.width * planes.height)
    orp, ofp = out_pitches or (3 * planes.width, 3 * planes.width * planes.height)
    keep, wptr = _weights_arg(weights)
    done = C.c_uint64(0)
    rc = native.load().svs_embed_bgr_dev(d_bgr_in, irp, ifp, d_bgr_out, orp, ofp, d_gray_ref or None,
                                         C.byref(planes), wptr, float(delta), int(n_ac), d_bits_packed,
                                         int(bit_offset), int(n_bits), mode_flags(mode), C.byref(done),
                                         stream or None)
    native.check(rc, "svs_embed_bgr_dev")
    return int(done.value)


def extract_bgr_device(d_bgr: int, planes: Planes, delta, n_ac, d_bits_out: int, out_capacity_bytes: int,
                       stream: int = 0, weights=None, pitches=None) -> int:
    """Enqueue extraction straight from interleaved BGR frames; returns the number of bits the batch yields."""
    rp, fp = pitches or (3 * planes.width, 3 * planes.width * planes.height)
    keep, wptr = _weights_arg(weights)
    got = C.c_uint64(0)
    rc = native.load().svs_extract_bgr_dev(d_bgr, rp, fp, C.byref(planes), wptr, float(delta), int(n_ac),
                                           d_bits_out, int(out_capacity_bytes), C.byref(got), stream or None)
    native.check(rc, "svs_extract_bgr_dev")
    return int(got.value)


def _as_bgr_stack(frames: np.ndarray) -> np.ndarray:
    a = np.asarray(frames)
    if a.dtype != np.uint8 or a.ndim != 4 or a.shape[3] != 3:
        raise TypeError("frames must be uint8 [F,H,W,3] (BGR)")
    if a.shape[1] % 8 or a.shape[2] % 8:
        raise ValueError("frame height and width must be multiples of 8")
    return np.ascontiguousarray(a)


def embed_bgr_frames(frames_bgr: np.ndarray, delta, n_ac, bits, bit_offset: int = 0, n_bits: int | None = None,
                     device: int = 0, mode: str | None = None, weights=None, want_gray: bool = True):
    """BGR frames in, stego BGR frames out (one fused pass on the GPU).
    Returns (stego_bgr uint8 [F,H,W,3], gray uint8 [F,H,W] or None, n_embedded)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_bgr_stack(frames_bgr)
    f, h, w, _ = stack.shape
    if isinstance(bits, str):
        bits = str_to_bits(bits)
    bits = np.asarray(bits, np.uint8)
    if n_bits is None:
        n_bits = max(0, bits.size - bit_offset)
    if bit_offset + n_bits > bits.size:
        raise ValueError("bit_offset + n_bits exceeds the payload length")
    packed, bit_offset = _pack_window(bits, bit_offset, n_bits)
    planes = Planes.contiguous(f, h, w)
    out = pinned_empty(stack.shape)
    gray = pinned_empty((f, h, w)) if want_gray else None
    keep, wptr = _weights_arg(weights)
    done = C.c_uint64(0)
    rc = lib.svs_embed_bgr(stack.ctypes.data, out.ctypes.data, gray.ctypes.data if want_gray else None,
                           C.byref(planes), wptr, float(delta), int(n_ac), packed.ctypes.data, int(bit_offset),
                           int(n_bits), mode_flags(mode), C.byref(done))
    native.check(rc, "svs_embed_bgr")
    used = int(done.value)
    return out, gray, used


def extract_bgr_frames(frames_bgr: np.ndarray, delta, n_ac, device: int = 0, weights=None):
    """Extract the packed bit stream straight from BGR frames.  Returns (packed uint8, n_bits)."""
    lib = native.load()
    native.ensure_device(device)
    stack = _as_bgr_stack(frames_bgr)
    f, h, w, _ = stack.shape
    cap = capacity_bits(f, h, w, n_ac)
    nbytes = max(4, (cap + 7) // 8 + (-((cap + 7) // 8)) % 4)
    out = np.zeros(nbytes, np.uint8)
    planes = Planes.contiguous(f, h, w)
    keep, wptr = _weights_arg(weights)
    got = C.c_uint64(0)
    rc = lib.svs_extract_bgr(stack.ctypes.data, C.byref(planes), wptr, float(delta), int(n_ac), out.ctypes.data,
                             nbytes, C.byref(got))
    native.check(rc, "svs_extract_bgr")
    n = int(got.value)
    return out[: (n + 7) // 8], n


# ---- frame sharding across ranks ----------------------------------------------
def shard_frames(n_frames: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous frame range [first, first+count) of `rank`, so that the global bit stream is the
    rank-order concatenation of the ranks' streams."""
    base, extra = divmod(n_frames, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)
