#!/usr/bin/env python3
"""Generate tests/golden/framing_golden.json + framing_golden.npz by running the REFERENCE's own host-side codecs in the
build container (run here only - needs /root/reference):     python tests/golden/make_framing_golden.py

What is called (all pure Python / NumPy / PIL, imported from /root/reference exactly as make_golden.py imports the
operator - placeholder modules for the absent cv2 / cryptography, nothing written to disk):
  config_and_setup.py:22-41    bytes_ke_bitstream, bitstream_ke_bytes, int_ke_bitstream, bitstream_ke_int
  helpers.py:5-126,184-187     gambar_ke_bitstream, bitstream_ke_gambar, buat_metadata_bitstream,
                               parse_metadata_bitstream, get_avi_path
and the payload header assembled with those helpers in the order of embed_process.py:62-74 from fixed field bytes.
The crypto itself (AES-GCM, ECDH, HKDF, SHA3) cannot run here (cryptography is absent), so the header's field VALUES are
fixed test bytes of the reference's field SIZES (33, 16, 32, 12, 16).
Also recorded: the shipped known-answer pair media/input/image64.png <-> media/output/extracted_image_gui.png (SURVEY
section 2 #13) as hashes, and the 'L' pixels of image64.png (4 KB of derived data) as a real secret image for round trips.
Fixtures are data: inputs and the reference's outputs, no reference source.
"""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _load_reference_operator  # noqa: E402  (registers the placeholder modules)


def call(fn, *args):
    """-> {"ok": result} or {"error": [type name, message]}; stdout of the reference is captured"""
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            out = fn(*args)
        return {"ok": out, "stdout": buf.getvalue()}
    except Exception as exc:  # noqa: BLE001 - the exception type and text ARE the fixture
        return {"error": [type(exc).__name__, str(exc)], "stdout": buf.getvalue()}


def main():
    _load_reference_operator()
    import config_and_setup as ref_cs   # the reference's module (already imported by the loader)
    sys.path.insert(0, REF)
    import helpers as ref_h
    assert os.path.dirname(os.path.abspath(ref_h.__file__)) == REF
    rng = np.random.default_rng(2718)
    out, arrays = {"codecs": {}, "helpers": {}, "header": {}, "media": {}}, {}

    # ---- config_and_setup.py:22-41 -------------------------------------------------------------------------------
    byte_cases = [b"", b"\x00", b"\xff", b"\x80\x01", bytes(range(256)), rng.integers(0, 256, 97, dtype=np.uint8).tobytes()]
    out["codecs"]["bytes_ke_bitstream"] = [{"in_hex": b.hex(), **call(ref_cs.bytes_ke_bitstream, b)} for b in byte_cases]
    bit_cases = ["", "1", "0101010", "01010101", "010101011", "1" * 16 + "0" * 7, "0" * 8, "1" * 24, "00000001" * 5 + "111",
                 "0101012", "01 10101"]
    rows = []
    for s in bit_cases:
        r = call(ref_cs.bitstream_ke_bytes, s)
        if "ok" in r:
            r["ok"] = r["ok"].hex()
        rows.append({"in": s, **r})
    out["codecs"]["bitstream_ke_bytes"] = rows
    out["codecs"]["int_ke_bitstream"] = [{"in": [v, w], **call(ref_cs.int_ke_bitstream, v, w)} for v, w in
                                         [(0, 8), (1, 8), (255, 8), (256, 8), (-1, 8), (33, 8), (0, 1), (1, 1), (2, 1),
                                          (65535, 16), (65536, 16), (4096 * 8, 32), (2 ** 32 - 1, 32), (2 ** 32, 32), (5, 3)]]
    out["codecs"]["bitstream_ke_int"] = [{"in": [s, w], **call(ref_cs.bitstream_ke_int, s, w)} for s, w in
                                         [("0", None), ("1", None), ("00100001", 8), ("00100001", 7), ("", None), ("", 8),
                                          ("1" * 32, 32), ("1" * 33, 32), ("0" * 5 + "101", 8), ("102", None), ("101", 0)]]

    # ---- helpers.py ----------------------------------------------------------------------------------------------
    import tempfile
    tmp = tempfile.mkdtemp()
    images = {
        "gray_7x5": Image.fromarray(rng.integers(0, 256, (5, 7), dtype=np.uint8), mode="L"),
        "rgb_12x9": Image.fromarray(rng.integers(0, 256, (9, 12, 3), dtype=np.uint8), mode="RGB"),
        "rgba_8x8": Image.fromarray(rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), mode="RGBA"),
        "lightgray_32": Image.new("L", (32, 32), "lightgray"),        # the reference's dummy secret (config_and_setup.py:225)
        "palette_6x4": Image.fromarray(rng.integers(0, 256, (4, 6, 3), dtype=np.uint8), mode="RGB").convert("P"),
    }
    g2b = []
    for name, img in images.items():
        path = os.path.join(tmp, name + ".png")
        img.save(path)
        png = open(path, "rb").read()
        arrays["png/" + name] = np.frombuffer(png, np.uint8)
        r = call(ref_h.gambar_ke_bitstream, path)
        w, h, bits = r.pop("ok")
        r["stdout"] = r["stdout"].replace(path, "<PATH>")
        arrays["bits/" + name] = np.packbits(np.frombuffer(bits.encode(), np.uint8) - 48)
        g2b.append({"name": name, "width": w, "height": h, "n_bits": len(bits), **r})
        back = call(ref_h.bitstream_ke_gambar, bits, w, h)
        img_back = back.pop("ok")
        arrays["back/" + name] = np.asarray(img_back)
        g2b[-1]["back_mode"] = img_back.mode
        g2b[-1]["back_stdout"] = back["stdout"]
    missing = call(ref_h.gambar_ke_bitstream, os.path.join(tmp, "tidak_ada.png"))
    missing["ok"] = list(missing["ok"])
    missing["stdout"] = missing["stdout"].replace(os.path.join(tmp, "tidak_ada.png"), "<PATH>")
    out["helpers"]["gambar_ke_bitstream"] = g2b
    out["helpers"]["gambar_ke_bitstream_missing"] = missing
    wrong = call(ref_h.bitstream_ke_gambar, "0" * 63, 4, 2)
    out["helpers"]["bitstream_ke_gambar_wrong_length"] = {"in": ["0" * 63, 4, 2], "ok_is_none": wrong["ok"] is None,
                                                          "stdout": wrong["stdout"]}
    out["helpers"]["buat_metadata_bitstream"] = [{"in": list(a), **call(ref_h.buat_metadata_bitstream, *a)} for a in
                                                 [(32, 32), (64, 64), (256, 256), (0, 0), (65535, 1), (65536, 1), (1, 65536),
                                                  (-1, 5), (7, 5, 4), (16, 3, 4), (1920, 1080, 12)]]
    pm = []
    for a in [("0" * 10 + "100000" + "0" * 10 + "100000",), ("1" * 32 + "0101",), ("0" * 31,), ("", ), ("01110101", 4),
              ("0111", 4), ("0000000001000000" + "0000000001000000" + "111",)]:
        r = call(ref_h.parse_metadata_bitstream, *a)
        if "ok" in r:
            r["ok"] = list(r["ok"])
        pm.append({"in": list(a), **r})
    out["helpers"]["parse_metadata_bitstream"] = pm
    out["helpers"]["get_avi_path"] = [{"in": p, **call(ref_h.get_avi_path, p)} for p in
                                      ["out/stego", "out/stego.mp4", "stego.video.mov", "a/b.c/d", "x.avi", ".hidden"]]

    # ---- payload header in the order of embed_process.py:62-74 --------------------------------------------------------
    fields = {"eph_pub": bytes([2]) + rng.integers(0, 256, 32, dtype=np.uint8).tobytes(),
              "salt": rng.integers(0, 256, 16, dtype=np.uint8).tobytes(),
              "digest": rng.integers(0, 256, 32, dtype=np.uint8).tobytes(),
              "nonce": rng.integers(0, 256, 12, dtype=np.uint8).tobytes(),
              "tag": rng.integers(0, 256, 16, dtype=np.uint8).tobytes()}
    for tag, (w, h) in {"secret_64x64": (64, 64), "secret_32x32": (32, 32), "secret_300x7": (300, 7)}.items():
        ct = rng.integers(0, 256, w * h, dtype=np.uint8).tobytes()          # AES-GCM ciphertext has the plaintext's length
        b2s, i2s = ref_cs.bytes_ke_bitstream, ref_cs.int_ke_bitstream
        total = (ref_h.buat_metadata_bitstream(w, h) + i2s(len(fields["eph_pub"]), 8) + b2s(fields["eph_pub"]) +
                 i2s(len(fields["salt"]), 8) + b2s(fields["salt"]) + i2s(len(fields["digest"]), 8) + b2s(fields["digest"]) +
                 i2s(len(fields["nonce"]), 8) + b2s(fields["nonce"]) + i2s(len(fields["tag"]), 8) + b2s(fields["tag"]) +
                 i2s(len(ct), 32) + b2s(ct))
        arrays[f"header/{tag}/payload_bits"] = np.packbits(np.frombuffer(total.encode(), np.uint8) - 48)
        arrays[f"header/{tag}/ciphertext"] = np.frombuffer(ct, np.uint8)
        out["header"][tag] = {"width": w, "height": h, "n_bits": len(total), "header_bits": len(total) - 8 * len(ct),
                              "fields_hex": {k: v.hex() for k, v in fields.items()}}

    # ---- shipped known-answer pair --------------------------------------------------------------------------------
    src = Image.open(os.path.join(REF, "media", "input", "image64.png"))
    got = Image.open(os.path.join(REF, "media", "output", "extracted_image_gui.png"))
    src_l = np.asarray(src.convert("L"))
    w, h, bits = call(ref_h.gambar_ke_bitstream, os.path.join(REF, "media", "input", "image64.png"))["ok"]
    out["media"] = {"image64_mode": src.mode, "image64_size": list(src.size), "extracted_mode": got.mode,
                    "extracted_size": list(got.size),
                    "image64_L_sha256": hashlib.sha256(src_l.tobytes()).hexdigest(),
                    "extracted_image_gui_sha256": hashlib.sha256(np.asarray(got).tobytes()).hexdigest(),
                    "image64_bitstream_sha256": hashlib.sha256(bits.encode()).hexdigest(), "image64_bits": len(bits)}
    arrays["media/image64_L"] = src_l

    np.savez_compressed(os.path.join(HERE, "framing_golden.npz"), **arrays)
    with open(os.path.join(HERE, "framing_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
