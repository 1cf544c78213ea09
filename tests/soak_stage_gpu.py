#!/usr/bin/env python3
"""One-off soak of the host-pointer staging pipeline (csrc/svs_capi.hip embed_host / svs_stage.hpp): random pitched geometries,
chunk sizes (experiments library, SVS_STAGE_CHUNK_KB), budgets, bit offsets, coefficient counts, quantiser steps incl. <= 0,
page-locked or pageable buffers, packed-bit and string payloads - against ONE device-pointer call over the whole batch.
usage: python tests/soak_stage_gpu.py [iterations] [seed]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd")
os.environ["SVSDCT_LIB"] = os.path.join(PKG, "lib", "variants", "libsvsdct_exp.so")
for p in (PKG, REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
from svsdct import batch, hostmem, native
from svsdct.native import Planes

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
lib = native.load(); native.ensure_device(0)


def dev(nbytes):
    p = C.c_void_p(); native.check(lib.svs_malloc(C.byref(p), nbytes), "malloc"); return p


for it in range(iters):
    f = int(rng.integers(1, 7)); h = 8 * int(rng.integers(1, 30)); w = 8 * int(rng.integers(1, 50))
    rp = w + 8 * int(rng.integers(0, 4)) * int(rng.integers(0, 2)); fp = h * rp + 8 * int(rng.integers(0, 9)) * int(rng.integers(0, 2))
    n_ac = int(rng.choice([0, 1, 3, 7, 8, 10, 15, 16, 20, 63])); delta = float(rng.choice([8, 20, 4, 0.3, 7.5, 0, -2, 1000]))
    os.environ["SVS_STAGE_CHUNK_KB"] = str(int(rng.choice([0, 4, 8, 16, 64, 256])))
    planes = Planes(f, h, w, 0, rp, fp)
    cap = batch.capacity_bits(f, h, w, n_ac)
    n_bits = int(rng.choice([0, 1, max(cap // 3, 0), max(cap - 1, 0), cap, cap + 17]))
    off = int(rng.integers(0, 40))
    bits = rng.integers(0, 2, off + n_bits).astype(np.uint8)
    packed = batch.pack_bits(bits)
    host = (hostmem.pinned_empty(f * fp) if rng.integers(0, 2) else np.empty(f * fp, np.uint8))
    host[:] = rng.integers(0, 256, f * fp, dtype=np.uint8)
    out = (hostmem.pinned_empty(f * fp) if rng.integers(0, 2) else np.empty(f * fp, np.uint8))
    out[:] = 0xA5
    # yardstick: one device-pointer call over the whole pitched batch
    d_in, d_out, d_bits = dev(f * fp), dev(f * fp), dev(packed.size + 8)
    lib.svs_memcpy_h2d(d_in, host.ctypes.data, f * fp, None); lib.svs_memset(d_out, 0xA5, f * fp, None)
    lib.svs_memcpy_h2d(d_bits, packed.ctypes.data, packed.size, None)
    want_used = C.c_uint64()
    native.check(lib.svs_embed_dev(d_in, d_out, C.byref(planes), delta, n_ac, d_bits, off, n_bits, 2, C.byref(want_used), None), "dev")
    want = np.empty(f * fp, np.uint8)
    lib.svs_memcpy_d2h(want.ctypes.data, d_out, f * fp, None); lib.svs_stream_synchronize(None)
    for p in (d_in, d_out, d_bits):
        lib.svs_free(p)
    used = C.c_uint64()
    if off == 0 and rng.integers(0, 2):           # the string form
        text = batch.bits_to_str(bits).encode()
        ref = np.full(f * fp, 0x5A, np.uint8)
        native.check(lib.svs_embed_str(host.ctypes.data, ref.ctypes.data, out.ctypes.data, C.byref(planes), delta, n_ac, text, len(text), 2, C.byref(used)), "str")
    else:
        native.check(lib.svs_embed(host.ctypes.data, out.ctypes.data, C.byref(planes), delta, n_ac, packed.ctypes.data, off, n_bits, 2, C.byref(used)), "host")
    assert used.value == want_used.value, (it, used.value, want_used.value)
    # pixels equal, padding of the caller's buffer untouched (the device call leaves 0xA5 there as well: d_out was preset)
    assert np.array_equal(out, want), (it, f, h, w, rp, fp, n_ac, delta, n_bits, off, os.environ["SVS_STAGE_CHUNK_KB"], int((out != want).sum()))
    if (it + 1) % 100 == 0:
        print(f"{it + 1} cases ok", flush=True)
print(f"staging soak ok: {iters} cases")
