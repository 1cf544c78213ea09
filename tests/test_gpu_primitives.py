"""GPU tier: hardware facts the kernels rely on, checked on the device itself."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from testlib import REPO
from svsdct import native

pytestmark = pytest.mark.gpu


def test_cvt_pk_u8_f32_semantics_recorded():
    """v_cvt_pk_u8_f32 is only allowed in put_pixel() (build flag SVS_USE_CVT_PK_U8) if it clamps
    to [0,255] and truncates toward zero like np.uint8(np.clip(v,0,255)).  Record what gfx950 does."""
    native.ensure_device(0)
    lib = native.load()
    lib.svs_probe_cvt_pk_u8.restype = C.c_int
    lib.svs_probe_cvt_pk_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    vals = np.array([-300.0, -1.5, -0.5, -0.0, 0.0, 0.49, 0.5, 0.51, 0.99999, 1.0, 1.5, 2.5, 3.5, 127.5, 127.99999,
                     254.5, 254.99, 255.0, 255.5, 256.0, 300.0, 1e9, np.nan, np.inf, -np.inf], np.float32)
    out = np.zeros(vals.size, np.uint32)
    native.check(lib.svs_probe_cvt_pk_u8(vals.ctypes.data, out.ctypes.data, vals.size), "probe")
    finite = np.isfinite(vals)
    want = np.clip(vals[finite], 0, 255).astype(np.uint8).astype(np.uint32)
    truncates = bool(np.array_equal(out[finite], want))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "cvt_pk_u8_probe.json"), "w") as fh:
        json.dump({"inputs": [repr(float(v)) for v in vals], "outputs": out.tolist(),
                   "clamps_and_truncates": truncates}, fh, indent=1)
    print("v_cvt_pk_u8_f32:", dict(zip([float(v) for v in vals], out.tolist())), "trunc+clamp:", truncates)
    built_with = os.environ.get("SVS_EXPECT_CVT_PK_U8")
    if built_with == "1":
        assert truncates
