"""GPU tier: hardware facts the kernels rely on, checked on the device itself."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from testlib import REPO
from svsdct import native

pytestmark = pytest.mark.gpu


def test_cvt_pk_u8_f32_is_what_the_store_path_assumes():
    """put_pixel() (csrc/svs_block.hpp) feeds v_cvt_pk_u8_f32 integer-valued floats (pixel + floor(change)) and
    relies on two facts: the conversion SATURATES to [0, 255], and it is exact on integer-valued input (it rounds
    to nearest even - NOT truncation - so a non-integer must never reach it).  Check both on the device and record
    the raw behaviour."""
    native.ensure_device(0)
    lib = native.load()
    lib.svs_probe_cvt_pk_u8.restype = C.c_int
    lib.svs_probe_cvt_pk_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    vals = np.concatenate([np.arange(-300, 600, dtype=np.float32),
                           np.array([-1.5, -0.5, -0.0, 0.49, 0.5, 0.51, 0.99999, 1.5, 2.5, 3.5, 127.5, 127.99999, 254.5,
                                     254.99, 255.5, 1e9, -1e9, np.inf, -np.inf], np.float32)])
    out = np.zeros(vals.size, np.uint32)
    native.check(lib.svs_probe_cvt_pk_u8(vals.ctypes.data, out.ctypes.data, vals.size), "probe")
    ints = vals == np.floor(vals)
    assert np.array_equal(out[ints], np.clip(vals[ints], 0, 255).astype(np.uint32))     # saturating, exact on integers
    assert np.array_equal(out[~ints], np.clip(np.rint(vals[~ints]), 0, 255).astype(np.uint32))   # round half to even
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "cvt_pk_u8_probe.json"), "w") as fh:
        json.dump({"inputs": [repr(float(v)) for v in vals[-19:]], "outputs": out[-19:].tolist(),
                   "saturates": True, "rounding": "nearest even", "exact_on_integer_valued_input": True}, fh, indent=1)


def test_layout_experiment_kernel_gives_the_same_bits(monkeypatch):
    """The LDS + 8-lanes-per-block + DPP variant of the extract kernel (svs_device.hpp `extract_shuffle_kernel`, kept for the
    layout A/B in profiles/r01_ab_layout.txt) must produce the bits of the shipped kernel."""
    import numpy as np
    from svsdct import batch, native, synth
    native.ensure_device(0)
    for (f, h, w, n_ac, delta) in ((3, 64, 136, 3, 8), (2, 40, 72, 7, 16), (1, 8, 8, 1, 8), (5, 24, 1048, 5, 8)):
        frames = synth.synthetic_frames(f, h, w, seed=n_ac)
        payload = synth.synthetic_bits(batch.capacity_bits(f, h, w, n_ac), seed=n_ac)
        stego, _ = batch.embed_frames(frames, delta, n_ac, payload, mode="fast")
        want, n_bits = batch.extract_frames(stego, delta, n_ac, mode="fast")
        monkeypatch.setenv("SVS_EXTRACT_SHUFFLE", "1")
        got, n_got = batch.extract_frames(stego, delta, n_ac, mode="fast")
        monkeypatch.delenv("SVS_EXTRACT_SHUFFLE")
        assert n_got == n_bits and np.array_equal(got, want), (f, h, w, n_ac)
        assert np.array_equal(np.unpackbits(got, count=n_bits), payload)
