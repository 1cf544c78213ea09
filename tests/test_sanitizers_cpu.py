"""CPU tier: the per-block arithmetic and bit bookkeeping of csrc/svs_block.hpp under AddressSanitizer and
UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool; the device kernels execute this
same header).  Runs in a subprocess because ASan wants to be the first library loaded."""
import os
import subprocess
import sys
import textwrap

from testlib import CSRC, REPO


def test_block_header_is_clean_under_asan_and_ubsan(tmp_path):
    so = str(tmp_path / "libsvs_hostemu_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-w",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + CSRC,
                           os.path.join(REPO, "tests", "hostemu", "hostemu.cpp"), "-o", so])
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    script = textwrap.dedent(f"""
        import ctypes, numpy as np
        lib = ctypes.CDLL({so!r})
        lib.emu_embed.restype = ctypes.c_uint64
        lib.emu_embed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                  ctypes.c_void_p]
        lib.emu_extract.restype = ctypes.c_uint64
        lib.emu_extract.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        rng = np.random.default_rng(0)
        for (f, h, w), n_ac, delta, off in [((2, 24, 40), 3, 8, 0), ((1, 16, 16), 63, 7.5, 5), ((3, 8, 8), 10, 20, 31),
                                            ((1, 32, 8), 1, 0.1, 0), ((2, 16, 24), 40, -1, 0), ((1, 8, 16), 0, 8, 0)]:
            frames = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
            n = max(0, min(n_ac, 63)); cap = f * (h // 8) * (w // 8) * n
            bits = rng.integers(0, 2, off + max(cap - 3, 1)).astype(np.uint8)
            packed = np.packbits(bits); packed = np.concatenate([packed, np.zeros(-packed.size % 4, np.uint8)])  # exact fit: no slack
            for exact in (0, 1):
                out = np.empty_like(frames)
                lib.emu_embed(frames.ctypes.data, out.ctypes.data, f, h, w, float(delta), n_ac, packed.ctypes.data, packed.size,
                              off, bits.size - off, exact, None)
                flags = np.zeros(max(cap, 1), np.uint8)
                lib.emu_extract(out.ctypes.data, f, h, w, float(delta), n_ac, flags.ctypes.data, exact, None)
        print("sanitizers clean")
    """)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "sanitizers clean" in res.stdout, res.stdout + res.stderr
