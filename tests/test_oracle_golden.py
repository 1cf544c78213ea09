"""Pin the CPU oracle (oracle/qim_dct_oracle.py) to the reference: every golden vector that
tests/golden/make_golden.py produced by running the reference's own proses_frame_qim_dct
(reference config_and_setup.py:106-174) must be reproduced bit for bit."""
import hashlib

import numpy as np
import pytest

from oracle import qim_dct_oracle as orc
from svsdct import synth


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _case_inputs(arrays, meta, name):
    info = meta["cases"][name]
    key = name + "/gray"
    if key in arrays.files:
        gray = arrays[key]
    else:
        h, w = info["shape"]
        gray = synth.synthetic_frames(1, h, w, seed=info["synth_seed"])[0]
    if name + "/payload" in arrays.files:
        payload = arrays[name + "/payload"]
    else:
        payload = synth.synthetic_bits(info["payload_len"], seed=info["synth_seed"])
    if info["payload_len"] is None:
        payload = None
    return info, gray, payload


def _single_frame_cases(meta):
    return [k for k in meta["cases"] if k != "G8_stream"]


def test_every_single_frame_vector(golden):
    arrays, meta = golden
    for name in _single_frame_cases(meta):
        info, gray, payload = _case_inputs(arrays, meta, name)
        g, stego, used = orc.frame_embed(gray, info["delta"], payload, info["n_ac"])
        assert used == info["used"], name
        assert (g == gray).all() and g is not gray
        assert _sha(stego) == info["stego_sha256"], name
        if name + "/stego" in arrays.files:
            assert np.array_equal(stego, arrays[name + "/stego"]), name
        assert abs(orc.psnr_u8(gray, stego) - info["psnr"]) < 1e-9 or info["psnr"] == float("inf")
        for src, tag in ((stego, "ext_stego"), (gray, "ext_cover")):
            bits = orc.frame_extract_bits(src, info["delta"], info["n_ac"])
            assert bits.size == info[tag + "_len"], name
            assert np.array_equal(np.packbits(bits), arrays[f"{name}/{tag}"]), (name, tag)


def test_loop_form_matches_vector_form(golden):
    arrays, meta = golden
    for name in _single_frame_cases(meta):
        info, gray, payload = _case_inputs(arrays, meta, name)
        if gray.size > 48 * 64:
            continue
        _, stego, used = orc.frame_operator_loops(gray, "embed", info["delta"], payload, info["n_ac"])
        assert used == info["used"], name
        assert _sha(stego) == info["stego_sha256"], name
        s = orc.frame_operator_loops(stego, "extract", info["delta"], None, info["n_ac"])
        assert np.array_equal(np.packbits(orc.bits_from_any(s)) if s else np.zeros(0, np.uint8),
                              arrays[name + "/ext_stego"]), name


def test_reference_ber_at_delta4(golden):
    """the reference itself loses payload bits at delta=4, n=3; the oracle must make
    exactly the same errors in exactly the same places."""
    arrays, meta = golden
    info, gray, payload = _case_inputs(arrays, meta, "G7_d4_n3")
    _, stego, _ = orc.frame_embed(gray, 4, payload, 3)
    assert np.array_equal(stego, arrays["G7_d4_n3/stego"])
    got = orc.frame_extract_bits(stego, 4, 3)
    errs = np.nonzero(got != payload)[0]
    assert errs.size == info["payload_errors"] > 0
    assert np.array_equal(errs, arrays["G7_d4_n3/error_positions"])


def test_stream_over_frames(golden):
    """Frame-loop bookkeeping of embed_process.py:108-128 / extract_process.py:64-76."""
    arrays, meta = golden
    info = meta["cases"]["G8_stream"]
    frames = synth.synthetic_frames(3, 32, 48, seed=info["synth_seed"])
    payload = arrays["G8_stream/payload"]
    stego, used = orc.batch_embed(frames, info["delta"], payload, info["n_ac"])
    assert used == info["used"] == payload.size
    for k in range(3):
        assert np.array_equal(stego[k], arrays[f"G8_stream/stego{k}"])
        assert np.array_equal(np.packbits(orc.frame_extract_bits(stego[k], info["delta"], info["n_ac"])),
                              arrays[f"G8_stream/ext{k}"])
    assert np.array_equal(stego[2], frames[2])          # untouched frame is byte-identical
    got = orc.batch_extract_bits(stego, info["delta"], info["n_ac"])
    assert np.array_equal(got[:payload.size], payload)


def test_payload_forms_and_errors():
    g = synth.synthetic_frames(1, 16, 16)[0]
    a = orc.frame_embed(g, 8, "1011", 3)
    b = orc.frame_embed(g, 8, np.array([1, 0, 1, 1], np.uint8), 3)
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] == 4
    assert orc.frame_operator(g, "nonsense", 8) is None
    with pytest.raises(ValueError):
        orc.frame_embed(np.zeros((8, 8, 3), np.uint8), 8, "1", 3)
    with pytest.raises(ValueError):
        orc.frame_extract(np.zeros((12, 8), np.uint8), 8, 3)
