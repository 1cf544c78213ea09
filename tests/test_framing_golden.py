"""CPU tier: the (f1) boundary - bit-string codecs, image <-> bit-string helpers, size metadata and the payload header -
against fixtures produced by the REFERENCE's own functions (tests/golden/make_framing_golden.py ran
config_and_setup.py:22-41, helpers.py:5-126,184-187 and the concatenation order of embed_process.py:62-74 in the build
container).  Results, exception types, exception messages and printed lines must be the reference's."""
import contextlib
import hashlib
import io
import json
import os

import numpy as np
import pytest

import config_and_setup as cs          # the drop-in modules
import helpers as hp
from svsdct import batch, framing
from testlib import REPO

GOLD = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "framing_golden.json")) as fh:
        meta = json.load(fh)
    return meta, np.load(os.path.join(GOLD, "framing_golden.npz"), allow_pickle=False)


def _same(case, fn, *args, convert=lambda x: x):
    """call fn(*args) and hold it against the fixture row: result or (exception type, message), and stdout"""
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            got = fn(*args)
    except Exception as exc:  # noqa: BLE001
        assert "error" in case, (args, exc)
        assert [type(exc).__name__, str(exc)] == case["error"], args
    else:
        assert "ok" in case, (args, got, case.get("error"))
        assert convert(got) == case["ok"], args
    return buf.getvalue()


def test_bit_string_codecs_match_the_reference(gold):
    meta, _ = gold
    c = meta["codecs"]
    for row in c["bytes_ke_bitstream"]:
        _same(row, cs.bytes_ke_bitstream, bytes.fromhex(row["in_hex"]))
    for row in c["bitstream_ke_bytes"]:
        _same(row, cs.bitstream_ke_bytes, row["in"], convert=lambda b: b.hex())
    for row in c["int_ke_bitstream"]:
        _same(row, cs.int_ke_bitstream, *row["in"])
    for row in c["bitstream_ke_int"]:
        _same(row, cs.bitstream_ke_int, *row["in"])


def test_image_helpers_match_the_reference(gold, tmp_path):
    meta, arr = gold
    h = meta["helpers"]
    for row in h["gambar_ke_bitstream"]:
        path = str(tmp_path / (row["name"] + ".png"))
        with open(path, "wb") as fh:
            fh.write(arr["png/" + row["name"]].tobytes())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            w, hh, bits = hp.gambar_ke_bitstream(path)
        assert (w, hh, len(bits)) == (row["width"], row["height"], row["n_bits"]), row["name"]
        assert np.array_equal(np.packbits(batch.str_to_bits(bits)), arr["bits/" + row["name"]]), row["name"]
        assert buf.getvalue().replace(path, "<PATH>") == row["stdout"], row["name"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            img = hp.bitstream_ke_gambar(bits, w, hh)
        assert img.mode == row["back_mode"] and np.array_equal(np.asarray(img), arr["back/" + row["name"]])
        assert buf.getvalue() == row["back_stdout"]
    path = str(tmp_path / "tidak_ada.png")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        got = hp.gambar_ke_bitstream(path)
    assert list(got) == h["gambar_ke_bitstream_missing"]["ok"]
    assert buf.getvalue().replace(path, "<PATH>") == h["gambar_ke_bitstream_missing"]["stdout"]
    row = h["bitstream_ke_gambar_wrong_length"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert (hp.bitstream_ke_gambar(*row["in"]) is None) == row["ok_is_none"]
    assert buf.getvalue() == row["stdout"]
    for row in h["buat_metadata_bitstream"]:
        _same(row, hp.buat_metadata_bitstream, *row["in"])
    for row in h["parse_metadata_bitstream"]:
        _same(row, hp.parse_metadata_bitstream, *row["in"], convert=list)
    for row in h["get_avi_path"]:
        _same(row, hp.get_avi_path, row["in"])


def test_payload_header_matches_the_reference_layout(gold):
    """svsdct.framing builds bit for bit what the reference's helpers concatenate (embed_process.py:62-74) and parses it
    back; the drop-in string helpers assemble the same stream."""
    meta, arr = gold
    for tag, info in meta["header"].items():
        f = {k: bytes.fromhex(v) for k, v in info["fields_hex"].items()}
        ct = arr[f"header/{tag}/ciphertext"].tobytes()
        want = np.unpackbits(arr[f"header/{tag}/payload_bits"], count=info["n_bits"])
        got = framing.build_payload_bits(info["width"], info["height"], f["eph_pub"], f["salt"], f["digest"], f["nonce"],
                                         f["tag"], ct)
        assert np.array_equal(got, want), tag
        assert info["header_bits"] == framing.HEADER_BITS_STANDARD == 976
        hdr = framing.parse_header(want)
        assert (hdr.width, hdr.height, hdr.eph_pub, hdr.salt, hdr.digest, hdr.nonce, hdr.tag, hdr.ciphertext_len, hdr.bits) == \
            (info["width"], info["height"], f["eph_pub"], f["salt"], f["digest"], f["nonce"], f["tag"], len(ct), 976)
        with pytest.raises(framing.HeaderIncomplete):
            framing.parse_header(want[:975])
        text = (hp.buat_metadata_bitstream(info["width"], info["height"]) +
                "".join(cs.int_ke_bitstream(len(f[k]), 8) + cs.bytes_ke_bitstream(f[k])
                        for k in ("eph_pub", "salt", "digest", "nonce", "tag")) +
                cs.int_ke_bitstream(len(ct), 32) + cs.bytes_ke_bitstream(ct))
        assert np.array_equal(batch.str_to_bits(text), want), tag
        assert cs.bitstream_ke_bytes(text[976:]) == ct


def test_shipped_known_answer_pair(gold):
    """media/input/image64.png -> convert('L') is pixel for pixel media/output/extracted_image_gui.png (one successful GUI
    round trip of the reference); its bit string as the reference's helper produces it"""
    meta, arr = gold
    m = meta["media"]
    assert m["image64_L_sha256"] == m["extracted_image_gui_sha256"]
    pixels = arr["media/image64_L"]
    assert hashlib.sha256(pixels.tobytes()).hexdigest() == m["image64_L_sha256"]
    bits = batch.bits_to_str(np.unpackbits(pixels.reshape(-1)))
    assert len(bits) == m["image64_bits"] and hashlib.sha256(bits.encode()).hexdigest() == m["image64_bitstream_sha256"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        img = hp.bitstream_ke_gambar(bits, 64, 64)
    assert np.array_equal(np.asarray(img), pixels)
