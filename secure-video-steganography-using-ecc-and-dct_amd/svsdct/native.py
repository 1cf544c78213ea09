"""ctypes binding of libsvsdct.so - the C ABI declared in include/svsdct.h.

There is deliberately no fallback: if the shared library has not been built (or cannot be
loaded) importing the symbols raises `SvsNativeError` with the build command.  The product path
never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SVSDCT_LIB", os.path.join(_PKG_DIR, "lib", "libsvsdct.so"))

SVS_OK = 0
SVS_ERR_INVALID_ARG = -1
SVS_ERR_HIP = -2
SVS_ERR_NO_DEVICE = -3
SVS_ERR_CAPACITY = -4
SVS_EXACT_POCKETFFT = 1      # flags bit: pocketfft-identical arithmetic (include/svsdct.h)
SVS_EXACT_GUARDED = 2        # flags bit: the same bit-identical result through the guarded kernel where it applies
ABI_VERSION = 4


class SvsNativeError(RuntimeError):
    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class Planes(C.Structure):
    """struct svs_planes"""
    _fields_ = [("n_frames", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("reserved", C.c_int32), ("row_pitch", C.c_int64), ("frame_pitch", C.c_int64)]

    @classmethod
    def contiguous(cls, n_frames: int, height: int, width: int) -> "Planes":
        return cls(n_frames, height, width, 0, width, height * width)


_u8p = C.c_void_p   # device or host byte pointers are passed as integers / c_void_p
_u64p = C.POINTER(C.c_uint64)
_PL = C.POINTER(Planes)

# name -> (restype, argtypes); every symbol include/svsdct.h declares
SIGNATURES = {
    "svs_abi_version": (C.c_int, []),
    "svs_last_error": (C.c_char_p, []),
    "svs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "svs_init": (C.c_int, [C.c_int]),
    "svs_device_arch": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "svs_shutdown": (C.c_int, []),
    "svs_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "svs_free": (C.c_int, [C.c_void_p]),
    "svs_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "svs_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "svs_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "svs_stream_synchronize": (C.c_int, [C.c_void_p]),
    "svs_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "svs_stream_destroy": (C.c_int, [C.c_void_p]),
    "svs_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "svs_host_free": (C.c_int, [C.c_void_p]),
    "svs_capacity_bits": (C.c_uint64, [_PL, C.c_int]),
    "svs_packed_bytes": (C.c_uint64, [C.c_uint64]),
    "svs_embed_dev": (C.c_int, [_u8p, _u8p, _PL, C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p, C.c_void_p]),
    "svs_embed": (C.c_int, [_u8p, _u8p, _PL, C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p]),
    "svs_embed_str": (C.c_int, [_u8p, _u8p, _u8p, _PL, C.c_double, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, _u64p]),
    "svs_extract_str": (C.c_int, [_u8p, _PL, C.c_double, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, _u64p]),
    "svs_extract_dev": (C.c_int, [_u8p, _PL, C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint32, _u64p, C.c_void_p]),
    "svs_extract": (C.c_int, [_u8p, _PL, C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint32, _u64p]),
    "svs_bgr_to_gray_dev": (C.c_int, [_u8p, C.c_int64, C.c_int64, _u8p, _PL, C.c_void_p, C.c_void_p]),
    "svs_gray_to_bgr_dev": (C.c_int, [_u8p, _PL, _u8p, C.c_int64, C.c_int64, C.c_void_p]),
    "svs_embed_bgr_dev": (C.c_int, [_u8p, C.c_int64, C.c_int64, _u8p, C.c_int64, C.c_int64, _u8p, _PL, C.c_void_p,
                                     C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p, C.c_void_p]),
    "svs_extract_bgr_dev": (C.c_int, [_u8p, C.c_int64, C.c_int64, _PL, C.c_void_p, C.c_double, C.c_int, _u8p,
                                       C.c_uint64, _u64p, C.c_void_p]),
    "svs_embed_bgr": (C.c_int, [_u8p, _u8p, _u8p, _PL, C.c_void_p, C.c_double, C.c_int, _u8p, C.c_uint64, C.c_uint64,
                                 C.c_uint32, _u64p]),
    "svs_extract_bgr": (C.c_int, [_u8p, _PL, C.c_void_p, C.c_double, C.c_int, _u8p, C.c_uint64, _u64p]),
    "svs_fill_synthetic_dev": (C.c_int, [_u8p, _PL, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "svs_fill_bits_dev": (C.c_int, [_u8p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]),
    "svs_frame_sse_dev": (C.c_int, [_u8p, _u8p, _PL, C.c_void_p, C.c_void_p]),
    "svs_ssim_workspace_bytes": (C.c_uint64, [_PL]),
    "svs_frame_ssim_dev": (C.c_int, [_u8p, _u8p, _PL, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "svs_bit_errors_dev": (C.c_int, [_u8p, _u8p, C.c_uint64, C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load libsvsdct.so once and attach prototypes.  Loading does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvsNativeError(
            f"libsvsdct.so not found at {LIB_PATH}; build it with "
            f"`python __graft_entry__.py` (or `make -C <package>/csrc`). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime etc.
        raise SvsNativeError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise SvsNativeError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.svs_abi_version() != ABI_VERSION and not os.environ.get("SVS_SKIP_ABI_CHECK"):   # (A/B runs against old builds)
        raise SvsNativeError(f"ABI version mismatch: library reports {lib.svs_abi_version()}, binding expects "
                             f"{ABI_VERSION}; rebuild libsvsdct.so")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != SVS_OK:
        msg = load().svs_last_error().decode("utf-8", "replace")
        raise SvsNativeError(f"{what} failed ({rc}): {msg}", rc)


_current = threading.local()      # hipSetDevice is per host thread: remember what THIS thread last selected


def ensure_device(device: int = 0) -> None:
    """Make `device` the calling thread's HIP device (svs_init = hipSetDevice + a usability check) unless this thread
    selected it already; raises when no GPU is usable.  A process that alternates devices, or a new thread, is
    switched explicitly instead of silently running on whatever device was current."""
    if getattr(_current, "device", None) == device:
        return
    check(load().svs_init(device), f"svs_init({device})")
    _current.device = device


def release_thread_context() -> None:
    """Give back the calling thread's staging context of the host-pointer entry points (svs_shutdown: two streams and the
    device buffers the thread's calls grew); the next such call builds a new one.  The frame pipelines call it when they
    close, so a worker thread that has processed a clip does not keep HBM for the rest of the process."""
    if _lib is not None:
        _lib.svs_shutdown()


def device_arch(device: int = 0) -> str:
    buf = C.create_string_buffer(128)
    check(load().svs_device_arch(device, buf, len(buf)), "svs_device_arch")
    return buf.value.decode()
