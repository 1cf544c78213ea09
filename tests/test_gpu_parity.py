"""GPU tier (-m gpu): the HIP kernels, called through the C ABI (libsvsdct.so), against the
pinned oracle, the reference's golden vectors and size-independent properties at BASELINE sizes.

Parity bar: IDENTITY.  BASELINE.json's north_star asks for bit-exact extracted bits and a stego PSNR within +-0.01 dB of
the reference's; since round 4 every embed mode reproduces the reference's stego PIXELS (pocketfft's float32 noise, exact
rounding ties and truncation artefacts included - SURVEY N4, N6), so the tests assert equality of arrays / zero squared
difference wherever both sides can be computed, never a PSNR tolerance between two of this build's modes.  PSNR appears only
as the size-independent band check of the full-batch test and in the report written to gpurun_out/parity_report.json.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from testlib import (CONTRACT_POINTS, GUARDED_POINTS, guarded_soak_cases, ORIGINAL_COVERS, REPO, case_inputs, contract_payloads,
                     emu_embed, emu_extract, exact_tie_mask, experiments_library, golden_bits, natural_like, sha,
                     single_frame_cases, structured_covers, using_library)
from oracle import qim_dct_oracle as orc
from svsdct import batch, native, synth
from svsdct.native import Planes

pytestmark = pytest.mark.gpu
_REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _device():
    native.ensure_device(0)
    yield
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as fh:
        json.dump(_REPORT, fh, indent=1, sort_keys=True)


def test_device_is_gfx950():
    assert native.device_arch(0).startswith("gfx950")


@pytest.mark.parametrize("mode", ["exact", "guarded"])
def test_exact_mode_golden_vectors_bit_for_bit(golden, mode):
    """EXACT mode (pocketfft-identical arithmetic) and GUARDED mode (the streaming kernel with its rigorous guard): stego
    PIXELS, bit counts and extracted bits equal the reference's in every golden case - ties, flat-block and delta <= 0
    round-trip artefacts included."""
    arrays, meta = golden
    for name in single_frame_cases(meta):
        info, gray, payload = case_inputs(arrays, meta, name)
        delta, n_ac = info["delta"], info["n_ac"]
        stego, used = batch.embed_frames(gray, delta, n_ac, payload, mode=mode)
        assert used == info["used"], name
        assert sha(stego[0]) == info["stego_sha256"], name
        for src, tag in ((stego[0], "ext_stego"), (gray, "ext_cover")):
            packed, n_bits = batch.extract_frames(src, delta, n_ac, mode=mode)
            assert n_bits == info[tag + "_len"], name
            assert np.array_equal(packed, arrays[f"{name}/{tag}"][:packed.size]), (name, tag)
        _REPORT[mode + "/" + name] = {"pixels": int(gray.size), "pixels_differing_from_reference": 0}
    info = meta["cases"]["G8_stream"]
    frames = synth.synthetic_frames(3, 32, 48, seed=info["synth_seed"])
    stego, used = batch.embed_frames(frames, info["delta"], info["n_ac"], arrays["G8_stream/payload"], mode=mode)
    assert used == info["used"]
    for k in range(3):
        assert np.array_equal(stego[k], arrays[f"G8_stream/stego{k}"])


@pytest.mark.parametrize("mode", ["exact", "guarded"])
@pytest.mark.parametrize("shape,n_ac,delta,frames", [((1080, 1920), 10, 8, 2), ((2160, 3840), 3, 8, 1),
                                                      ((480, 640), 10, 20, 2), ((360, 640), 63, 4, 1),
                                                      ((1080, 1920), 7, 4, 1), ((720, 1280), 1, 20, 2)])
def test_exact_mode_full_size_equals_oracle(shape, n_ac, delta, frames, mode):
    h, w = shape
    cover = synth.synthetic_frames(frames, h, w, seed=h ^ n_ac, lo=0, span=256)     # clipping included
    cap = batch.capacity_bits(frames, h, w, n_ac)
    payload = synth.synthetic_bits(cap - 17, seed=h)
    stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode=mode)
    want, want_used = orc.batch_embed(cover, delta, payload, n_ac)
    assert used == want_used
    assert np.array_equal(stego, want)                                               # every pixel
    packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode=mode)
    assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(want, delta, n_ac))
    packed, n_bits = batch.extract_frames(cover, delta, n_ac, mode=mode)          # ties resolved as scipy does
    assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(cover, delta, n_ac))


_natural_like = natural_like


@pytest.mark.parametrize("mode", ["exact", "guarded"])
@pytest.mark.parametrize("n_ac,delta", [(3, 8), (10, 20), (7, 4)])
def test_exact_mode_on_natural_like_content(n_ac, delta, mode):
    h, w = 1080, 1920
    cover = np.stack([_natural_like(h, w, s) for s in (1, 2)])
    cap = batch.capacity_bits(2, h, w, n_ac)
    payload = synth.synthetic_bits(cap, seed=n_ac)
    payload[: cap // 3] = 0                                # long zero runs: many blocks get no coefficient change
    stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode=mode)
    want, want_used = orc.batch_embed(cover, delta, payload, n_ac)
    assert used == want_used and np.array_equal(stego, want)
    # the artefact is really there (and reproduced): flat blocks that changed although no coefficient did
    flat = cover[0, 1080 // 2: 1080 // 2 + 64, : 1920 // 3 - 8]
    assert (want[0, 1080 // 2: 1080 // 2 + 64, : 1920 // 3 - 8] != flat).any() or delta == 4
    packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode="exact")
    assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(want, delta, n_ac))
    # flags = 0 on the same content: identical extracted bits from the reference's frames and from the never-embedded
    # cover (no masks: quantiser inputs near a tie take the exact path), and the reference's stego pixels
    for src in (want, cover):
        packed, n_bits = batch.extract_frames(src, delta, n_ac, mode="fast")
        assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(src, delta, n_ac))
    fast, used_f = batch.embed_frames(cover, delta, n_ac, payload, mode="fast")
    assert used_f == want_used and np.array_equal(fast, want)
    for k in range(2):
        _REPORT[f"natural_like_n{n_ac}_d{delta}_frame{k}_{mode}"] = {
            "pixels": h * w, "pixels_differing_from_reference": 0, "psnr": orc.psnr_u8(cover[k], fast[k]),
            "psnr_reference": orc.psnr_u8(cover[k], want[k])}


@pytest.mark.parametrize("n_ac,delta", CONTRACT_POINTS)
def test_fast_mode_contract_on_structured_content(n_ac, delta):
    """FAST embed on flat / letterboxed / one-dimensional / natural-like frames at 1080p and
    on the round-2 and round-3 probe classes (smooth ramps, noise-free sinusoid, sigma-1 Gaussian, near-black ...) at 544 x 960,
    under three payloads (Bernoulli(1/2), all zero, 1 % ones): stego PSNR within 0.01 dB of the oracle's - identical pixels at
    n <= 15, where FAST runs the rigorous arithmetic -, identical to the CPU build of the kernel header (which the CPU tier
    checks on the same content), the reference's receiver reads the same bits from either stego frame, FAST extraction of
    stego, reference stego and never-embedded cover equals the oracle's on every bit."""
    cases = [(k, v, ("bernoulli_half",)) for k, v in structured_covers(1080, 1920).items() if k in ORIGINAL_COVERS]
    cases += [(k, v, ("bernoulli_half", "all_zero", "one_percent_ones")) for k, v in structured_covers(544, 960).items()]
    for name, cover, payload_names in cases:
        h, w = cover.shape
        cap = batch.capacity_bits(1, h, w, n_ac)
        payloads = contract_payloads(cap, seed=n_ac * 100 + delta)
        for pname in payload_names:
            payload = payloads[pname]
            stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode="fast")
            _, ref, ref_used = orc.frame_embed(cover, delta, payload, n_ac)
            assert used == ref_used == cap
            a, b = orc.psnr_u8(cover, stego[0]), orc.psnr_u8(cover, ref)
            assert np.array_equal(stego[0], ref), (name, pname)           # every embed mode gives the reference's pixels (round 4)
            replayed = []
            emu, _ = emu_embed(cover, delta, n_ac, payload, replayed=replayed)
            assert np.array_equal(emu[0], stego[0]), (name, pname)
            # delta >= 8 is error-free unless a pixel clips (large steps do clip): wherever the reference's own
            # round trip returns the payload, so does this one
            if delta >= 8 and np.array_equal(orc.frame_extract_bits(ref, delta, n_ac), payload):
                assert np.array_equal(orc.frame_extract_bits(stego[0], delta, n_ac), payload), (name, pname)
            for src in (stego[0], ref, cover) if pname == "bernoulli_half" else (stego[0],):
                packed, n_bits = batch.extract_frames(src, delta, n_ac, mode="fast")
                assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.frame_extract_bits(src, delta, n_ac)), (name, pname)
            _REPORT[f"structured/{name}_{h}x{w}_{pname}_n{n_ac}_d{delta}"] = {
                "pixels": h * w, "pixels_differing_from_reference": int((stego[0] != ref).sum()), "psnr": a,
                "psnr_reference": b, "blocks_replayed_exactly": replayed[0]}
    # in place and through the device-pointer level: same frames
    cover = np.stack(list(structured_covers(256, 512).values()))
    f = cover.shape[0]
    payload = synth.synthetic_bits(batch.capacity_bits(f, 256, 512, n_ac), seed=5)
    want, _ = batch.embed_frames(cover, delta, n_ac, payload, mode="fast")
    d_frames, d_bits = _Dev(cover.nbytes), _Dev(batch.pack_bits(payload).nbytes)
    d_frames.put(cover)
    d_bits.put(batch.pack_bits(payload))
    planes = Planes.contiguous(f, 256, 512)
    assert batch.embed_device(d_frames.ptr, d_frames.ptr, planes, delta, n_ac, d_bits.ptr, 0, payload.size) == payload.size
    assert np.array_equal(d_frames.get().reshape(cover.shape), want)


@pytest.mark.parametrize("n_ac,delta", GUARDED_POINTS)
def test_guarded_mode_equals_reference_on_structured_content(n_ac, delta):
    """the streaming kernel with its rigorous guard (8 tests per block at n <= 7, 64 at n = 8..15) is the
    reference, pixel for pixel, on 14 content
    classes (flat, letterboxed, one-dimensional, posterised, text-like, dark / bright noise, exact cancellations ...) at
    every setting incl. the ends of its delta range; the share of blocks it redid exactly is recorded per class."""
    exp = experiments_library()           # the replay counter is a hook of the experiments library (same kernel sources)
    h, w = 544, 960
    d_cnt = _Dev(8)
    for name, cover in structured_covers(h, w).items():
        cap = batch.capacity_bits(1, h, w, n_ac)
        payload = synth.synthetic_bits(cap, seed=n_ac * 100 + int(delta))
        stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode="guarded")          # the product library
        _, ref, ref_used = orc.frame_embed(cover, delta, payload, n_ac)
        assert used == ref_used and np.array_equal(stego[0], ref), name
        native.check(exp.svs_memset(d_cnt.ptr, 0, 8, None), "memset")
        native.check(exp.svs_stream_synchronize(None), "sync")
        exp.svs_guard_counter_set(d_cnt.ptr)
        try:
            with using_library(exp):
                counted, used_c = batch.embed_frames(cover, delta, n_ac, payload, mode="guarded")
        finally:
            exp.svs_guard_counter_set(None)
        assert used_c == ref_used and np.array_equal(counted[0], ref), name
        if n_ac <= 7:
            fast, used_f = batch.embed_frames(cover, delta, n_ac, payload, mode="fast")  # n <= 7: the same launch
            assert used_f == ref_used and np.array_equal(fast[0], ref), name
        _REPORT[f"guarded/{name}_n{n_ac}_d{delta:g}"] = {
            "pixels": h * w, "pixels_differing_from_reference": 0,
            "blocks_redone_exactly_share": int(d_cnt.get(8, np.uint64)[0]) / (cap // n_ac)}


@pytest.mark.parametrize("f,h,w,n_ac,delta", [(600, 2160, 3840, 3, 8.0), (300, 1080, 1920, 7, 4.0), (150, 4320, 7680, 1, 16.0),
                                              (300, 1080, 1920, 10, 8.0), (150, 2160, 3840, 15, 20.0)])
def test_guarded_mode_full_batch_equals_exact_kernel_on_device(f, h, w, n_ac, delta):
    """BASELINE batch sizes, device-resident: the streaming kernel's output has zero squared difference to the
    lane-per-block pocketfft kernel's on every frame (5 Gpixel per case), in place as well."""
    lib = native.load()
    planes = Planes.contiguous(f, h, w)
    cap = batch.capacity_bits(f, h, w, n_ac)
    nbytes = (cap + 7) // 8 + 8
    d_gray, d_a, d_b = _Dev(f * h * w), _Dev(f * h * w), _Dev(f * h * w)
    d_pay, d_sse = _Dev(nbytes), _Dev(8 * f)
    native.check(lib.svs_fill_synthetic_dev(d_gray.ptr, C.byref(planes), 20250620, 0, 0, 256, None), "fill")   # clipping included
    native.check(lib.svs_memset(d_pay.ptr, 0, nbytes, None), "memset")
    native.check(lib.svs_fill_bits_dev(d_pay.ptr, cap - 5, 20250620, 0, None), "bits")
    assert batch.embed_device(d_gray.ptr.value, d_a.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap - 5, mode="exact") == cap - 5
    assert batch.embed_device(d_gray.ptr.value, d_b.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap - 5, mode="guarded") == cap - 5
    native.check(lib.svs_frame_sse_dev(d_a.ptr, d_b.ptr, C.byref(planes), d_sse.ptr, None), "sse")
    assert int(d_sse.get(8 * f, np.uint64).sum()) == 0
    assert batch.embed_device(d_gray.ptr.value, d_gray.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap - 5, mode="guarded") == cap - 5
    native.check(lib.svs_frame_sse_dev(d_a.ptr, d_gray.ptr, C.byref(planes), d_sse.ptr, None), "sse")
    assert int(d_sse.get(8 * f, np.uint64).sum()) == 0


def test_golden_vectors(golden):
    arrays, meta = golden
    for name in single_frame_cases(meta):
        info, gray, payload = case_inputs(arrays, meta, name)
        delta, n_ac = info["delta"], info["n_ac"]
        stego, used = batch.embed_frames(gray, delta, n_ac, payload, mode="fast")
        stego = stego[0]
        _, ref_stego, ref_used = orc.frame_embed(gray, delta, payload, n_ac)

        # (a) extraction from the REFERENCE's stego frame is bit-exact
        packed, n_bits = batch.extract_frames(ref_stego, delta, n_ac, mode="fast")
        assert n_bits == info["ext_stego_len"], name
        assert np.array_equal(np.unpackbits(packed, count=n_bits),
                              golden_bits(arrays, name, "ext_stego", n_bits)), name
        # (b) extraction from our own stego frame agrees with the oracle on the same frame
        packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode="fast")
        own = np.unpackbits(packed, count=n_bits)
        assert np.array_equal(own, orc.frame_extract_bits(stego, delta, n_ac)), name
        # the kernels compute exactly what the CPU build of the same header computes
        emu_stego, emu_used = emu_embed(gray, delta, n_ac, payload)
        assert emu_used == used and np.array_equal(emu_stego[0], stego), name

        _REPORT[name] = {"pixels": int(gray.size), "pixels_differing_from_reference": int((stego != ref_stego).sum()),
                         "psnr": orc.psnr_u8(gray, stego), "psnr_reference": info["psnr"]}
        assert used == info["used"] == ref_used, name
        assert sha(stego) == info["stego_sha256"], name       # flags = 0: the reference's pixels too (every mode, round 4)
        # (c) a receiver running the reference reads the same bits from our frame as from the reference's
        assert np.array_equal(orc.frame_extract_bits(stego, delta, n_ac)[:used],
                              orc.frame_extract_bits(ref_stego, delta, n_ac)[:used]), name
        # (d) the PSNR the reference recorded for this case (identity implies it; the golden JSON's number is the reference's own)
        if np.isfinite(info["psnr"]):
            assert orc.psnr_u8(gray, stego) == pytest.approx(info["psnr"], abs=1e-9), name
        else:
            assert np.array_equal(stego, gray), name
        # (e) extraction from the never-embedded cover: identical as well, rounding ties of c/delta included
        packed, n_bits = batch.extract_frames(gray, delta, n_ac, mode="fast")
        _REPORT[name]["exact_ties_in_cover"] = int(exact_tie_mask(gray, delta, n_ac).sum())
        assert np.array_equal(np.unpackbits(packed, count=n_bits),
                              golden_bits(arrays, name, "ext_cover", info["ext_cover_len"])), name


def test_reference_ber_at_delta4_is_reproduced(golden):
    """delta=4, n=3: the reference itself loses ~1.6 % of the payload.  Required:
    identical bits to the oracle on the same frames, hence the same error positions."""
    arrays, meta = golden
    info, gray, payload = case_inputs(arrays, meta, "G7_d4_n3")
    stego, used = batch.embed_frames(gray, 4, 3, payload, mode="fast")
    packed, n_bits = batch.extract_frames(arrays["G7_d4_n3/stego"], 4, 3, mode="fast")
    got = np.unpackbits(packed, count=n_bits)
    assert np.array_equal(np.nonzero(got != payload)[0], arrays["G7_d4_n3/error_positions"])
    packed, n_bits = batch.extract_frames(stego[0], 4, 3, mode="fast")
    assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.frame_extract_bits(stego[0], 4, 3))


def test_stream_over_frames(golden):
    arrays, meta = golden
    info = meta["cases"]["G8_stream"]
    frames = synth.synthetic_frames(3, 32, 48, seed=info["synth_seed"])
    payload = arrays["G8_stream/payload"]
    stego, used = batch.embed_frames(frames, info["delta"], info["n_ac"], payload, mode="fast")
    assert used == info["used"]
    for k in range(3):
        assert np.array_equal(stego[k], arrays[f"G8_stream/stego{k}"]), k
    assert np.array_equal(stego[2], frames[2])
    packed, n_bits = batch.extract_frames(stego, info["delta"], info["n_ac"], mode="fast")
    bits = np.unpackbits(packed, count=n_bits)
    assert np.array_equal(bits[:used], payload)
    for k in range(3):
        cap = n_bits // 3
        assert np.array_equal(np.packbits(bits[k * cap:(k + 1) * cap]), arrays[f"G8_stream/ext{k}"])
    # bit_offset into a longer shared buffer (how ranks index one payload)
    junk = synth.synthetic_bits(61, seed=3)
    stego2, used2 = batch.embed_frames(frames, info["delta"], info["n_ac"], np.concatenate([junk, payload]),
                                       bit_offset=61, mode="fast")
    assert used2 == used and np.array_equal(stego2, stego)


@pytest.mark.parametrize("shape,n_ac,delta,frames", [
    ((1080, 1920), 10, 8, 3),     # BASELINE config 2 shape (n = app default 10)
    ((2160, 3840), 3, 8, 2),      # BASELINE config 3 shape
    ((480, 640), 10, 20, 2),      # BASELINE config 1 shape / app defaults
    ((1080, 1920), 63, 16, 1),    # every AC coefficient
    ((72, 200), 17, 9, 5),        # odd block counts: 9 x 25 blocks per frame
])
def test_full_size_round_trip_and_oracle_agreement(shape, n_ac, delta, frames):
    h, w = shape
    cover = synth.synthetic_frames(frames, h, w, seed=h + n_ac)
    cap = batch.capacity_bits(frames, h, w, n_ac)
    payload = synth.synthetic_bits(cap, seed=h + n_ac)
    stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode="fast")
    assert used == cap
    packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode="fast")
    got = np.unpackbits(packed, count=n_bits)
    assert n_bits == cap
    if n_ac <= 7 or delta >= 8 and n_ac < 63:
        assert np.array_equal(got, payload)                      # payload BER 0
    # oracle on the first and last frame: same extracted bits from the same stego; PSNR parity
    per = cap // frames
    for k in {0, frames - 1}:
        assert np.array_equal(got[k * per:(k + 1) * per], orc.frame_extract_bits(stego[k], delta, n_ac))
        _, ref_stego, _ = orc.frame_embed(cover[k], delta, payload[k * per:(k + 1) * per], n_ac)
        a, b = orc.psnr_u8(cover[k], stego[k]), orc.psnr_u8(cover[k], ref_stego)
        assert np.array_equal(stego[k], ref_stego), (k, "flags = 0 must give the reference's pixels")
        _REPORT[f"full_{h}x{w}_n{n_ac}_d{delta}_frame{k}"] = {
            "pixels": h * w, "pixels_differing_from_reference": int((stego[k] != ref_stego).sum()),
            "psnr": a, "psnr_reference": b}
        # the reference's receiver recovers the payload from our frame wherever it does from its own
        assert np.array_equal(orc.frame_extract_bits(stego[k], delta, n_ac),
                              orc.frame_extract_bits(ref_stego, delta, n_ac))


@pytest.mark.parametrize("delta", [4, 8, 16])
def test_baseline_config5_shape_8k_delta_sweep(delta):
    """BASELINE.json configs[4]: 7680x4320, 3 AC coefficients, delta in {4, 8, 16}; one frame here
    (the oracle needs seconds per 8K frame), PSNR and BER against the oracle on that frame."""
    h, w, n_ac = 4320, 7680, 3
    cover = synth.synthetic_frames(1, h, w, seed=delta)
    cap = batch.capacity_bits(1, h, w, n_ac)
    payload = synth.synthetic_bits(cap, seed=delta)
    _, ref_stego, _ = orc.frame_embed(cover[0], delta, payload, n_ac)
    ref_bits = orc.frame_extract_bits(ref_stego, delta, n_ac)
    ref_ber = int((ref_bits != payload).sum())
    assert (ref_ber == 0) == (delta >= 8)                     # only delta = 4 loses bits
    # every mode: the reference's frame, pixel for pixel - hence its PSNR, its extracted bits and its payload errors (delta = 4:
    # the reference's own 1.6 %, at the reference's positions)
    for mode in ("exact", "guarded", "fast"):
        stego, used = batch.embed_frames(cover, delta, n_ac, payload, mode=mode)
        assert used == cap and np.array_equal(stego[0], ref_stego), mode
        packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode=mode)
        got = np.unpackbits(packed, count=n_bits)
        assert np.array_equal(got, ref_bits), mode
        assert int((got != payload).sum()) == ref_ber, mode
    _REPORT[f"8k_n3_d{delta}"] = {"pixels": h * w, "pixels_differing_from_reference": 0,
                                  "psnr": orc.psnr_u8(cover[0], stego[0]), "psnr_reference": orc.psnr_u8(cover[0], ref_stego),
                                  "payload_bit_errors": ref_ber, "payload_bit_errors_reference": ref_ber}


@pytest.mark.parametrize("f,h,w,n_ac,delta,band", [
    (600, 2160, 3840, 3, 8.0, (44.5, 45.5)),      # BASELINE configs[2], the benchmark workload
    (300, 1080, 1920, 10, 8.0, (39.8, 40.7)),     # configs[1] (n = 10 is the reference GUI's default)
    (150, 4320, 7680, 3, 16.0, (38.9, 40.0)),     # one GPU's share of configs[4], the largest delta of its sweep
])
def test_full_baseline_batch_on_device_properties(f, h, w, n_ac, delta, band):
    """BASELINE.json configs at FULL size (e.g. 600 x 3840x2160, n = 3, delta = 8, full-capacity payload = 233 280 000 bits),
    device-resident like bench.py, checked through size-independent properties: the payload comes back with zero
    errors from either transform mode, embedding is deterministic (two runs, zero squared difference), the exact-mode
    stego extracts to the same stream, and every frame's PSNR sits in the band the quantiser step implies."""
    lib = native.load()
    planes = Planes.contiguous(f, h, w)
    cap = batch.capacity_bits(f, h, w, n_ac)
    nbytes = (cap + 7) // 8 + 8
    d_gray, d_a, d_b = _Dev(f * h * w), _Dev(f * h * w), _Dev(f * h * w)
    d_pay, d_x1, d_x2 = _Dev(nbytes), _Dev(nbytes), _Dev(nbytes)
    d_cnt, d_sse = _Dev(8), _Dev(8 * f)
    native.check(lib.svs_fill_synthetic_dev(d_gray.ptr, C.byref(planes), 20250620, 0, 16, 224, None), "fill")
    native.check(lib.svs_memset(d_pay.ptr, 0, nbytes, None), "memset")
    native.check(lib.svs_fill_bits_dev(d_pay.ptr, cap, 20250620, 0, None), "bits")

    def errors(x):
        native.check(lib.svs_bit_errors_dev(x.ptr, d_pay.ptr, cap, d_cnt.ptr, None), "ber")
        return int(d_cnt.get(8, np.uint64)[0])

    def sse(x, y):
        native.check(lib.svs_frame_sse_dev(x.ptr, y.ptr, C.byref(planes), d_sse.ptr, None), "sse")
        return d_sse.get(8 * f, np.uint64)

    assert batch.embed_device(d_gray.ptr.value, d_a.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap, mode="fast") == cap
    assert batch.extract_device(d_a.ptr.value, planes, delta, n_ac, d_x1.ptr.value, nbytes, mode="fast") == cap
    assert errors(d_x1) == 0
    assert batch.embed_device(d_gray.ptr.value, d_b.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap, mode="fast") == cap
    assert int(sse(d_a, d_b).sum()) == 0                                     # deterministic
    per_frame = sse(d_gray, d_a).astype(np.float64)
    psnr = 10 * np.log10(255.0 ** 2 * h * w / per_frame)
    assert psnr.min() > band[0] and psnr.max() < band[1], (psnr.min(), psnr.max())   # set by n and delta alone
    assert batch.embed_device(d_gray.ptr.value, d_b.ptr.value, planes, delta, n_ac, d_pay.ptr.value, 0, cap, mode="exact") == cap
    assert batch.extract_device(d_b.ptr.value, planes, delta, n_ac, d_x2.ptr.value, nbytes, mode="exact") == cap
    assert errors(d_x2) == 0
    assert batch.extract_device(d_b.ptr.value, planes, delta, n_ac, d_x2.ptr.value, nbytes, mode="fast") == cap
    assert errors(d_x2) == 0                                                  # fast extract of exact stego
    # between the modes: identity - zero squared difference on every frame of the batch (5 Gpixel)
    assert int(sse(d_a, d_b).sum()) == 0
    _REPORT[f"full_baseline_batch/{w}x{h}x{f}_n{n_ac}_d{delta:g}"] = {"frames": f, "bits": int(cap), "payload_bit_errors": 0,
                                      "psnr_db_min_max": [float(psnr.min()), float(psnr.max())],
                                      "default_vs_exact_stego_squared_difference": 0}


def test_extreme_quantiser_steps():
    """delta from 1e-3 to 3.3e7 (incl. values float32 cannot represent): EXACT mode stays bit-identical to the oracle -
    stego pixels, bit counts and the bits read back from the oracle's stego - and FAST mode stays identical to the CPU
    build of the kernel header.  (FAST extraction against the oracle is not asserted at tiny delta: once delta/2 is
    below the float32 noise of an 8x8 DCT, two implementations legitimately disagree - as the reference does with itself
    after the uint8 store.)"""
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (2, 32, 48), dtype=np.uint8)
    for delta in (0.001, 0.01, 0.1, 0.3, 1 / 3, 255.5, 1e3, 1e4, 1e6, 3.3e7):
        for n_ac in (3, 10, 63):
            cap = batch.capacity_bits(2, 32, 48, n_ac)
            bits = rng.integers(0, 2, cap).astype(np.uint8)
            ref, used = orc.batch_embed(frames, delta, bits, n_ac)
            stego, got_used = batch.embed_frames(frames, delta, n_ac, bits, mode="exact")
            assert got_used == used and np.array_equal(stego, ref), (delta, n_ac)
            packed, n_bits = batch.extract_frames(ref, delta, n_ac, mode="exact")
            assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(ref, delta, n_ac)), (delta, n_ac)
            fast, fast_used = batch.embed_frames(frames, delta, n_ac, bits, mode="fast")
            want, want_used = emu_embed(frames, delta, n_ac, bits)
            assert fast_used == want_used and np.array_equal(fast, want), (delta, n_ac)
            if not (0.25 <= delta <= 4096):            # outside the streaming kernels' delta range FAST runs the exact kernels
                assert np.array_equal(fast, ref), (delta, n_ac)
    # FAST against the ORACLE at small delta on a frame of real size (the header's delta domain)
    big = synth.synthetic_frames(1, 544, 960, seed=77, lo=0, span=256)
    for delta in (0.02, 0.1, 0.5, 1.0):
        for n_ac in (3, 10):
            bits = synth.synthetic_bits(batch.capacity_bits(1, 544, 960, n_ac), seed=3)
            ref, _ = orc.batch_embed(big, delta, bits, n_ac)
            fast, _ = batch.embed_frames(big, delta, n_ac, bits, mode="fast")
            assert np.array_equal(fast, ref), (delta, n_ac)     # n <= 15: FAST is the rigorous arithmetic (exact kernels below 0.25)


def test_baseline_config4_shape_clips_sharded_by_frame():
    """BASELINE.json configs[3]: 8 x 1080p clips, one per GPU, extracted bits gathered in rank order.  On the
    one-GPU box the 8 shards run back to back through the same entry points the ranks use (shared payload
    indexed by bit offset); the concatenation must equal the single-call result for the whole clip."""
    h, w, n_ac, delta, world, frames_per_rank = 1080, 1920, 10, 8, 8, 2
    total = world * frames_per_rank
    cap = batch.capacity_bits(1, h, w, n_ac)
    payload = synth.synthetic_bits(total * cap, seed=21)
    whole = synth.synthetic_frames(total, h, w, seed=21)
    stego_all, used_all = batch.embed_frames(whole, delta, n_ac, payload, mode="fast")
    bits_all, n_all = batch.extract_frames(stego_all, delta, n_ac, mode="fast")
    pieces = []
    for rank in range(world):
        first, count = batch.shard_frames(total, world, rank)
        mine = synth.synthetic_frames(count, h, w, seed=21, first_frame=first)
        stego, used = batch.embed_frames(mine, delta, n_ac, payload, bit_offset=first * cap, n_bits=count * cap,
                                         mode="fast")
        assert used == count * cap and np.array_equal(stego, stego_all[first:first + count])
        packed, n_bits = batch.extract_frames(stego, delta, n_ac, mode="fast")
        pieces.append(np.unpackbits(packed, count=n_bits))
    joined = np.concatenate(pieces)
    assert np.array_equal(joined, np.unpackbits(bits_all, count=n_all))
    assert np.array_equal(joined, payload)


def test_random_geometries_match_the_cpu_build_of_the_kernel_header():
    """Indexing check over many shapes: odd block counts per row (no 16-byte path), partial waves and tiles, budgets
    ending mid-block, bit offsets, all coefficient-row counts.  The GPU must produce exactly what the CPU build of
    csrc/svs_block.hpp produces (fast mode) and what the oracle produces (exact mode)."""
    rng = np.random.default_rng(2025)
    for it in range(40):
        f = int(rng.integers(1, 6))
        h, w = 8 * int(rng.integers(1, 18)), 8 * int(rng.integers(1, 40))
        n_ac = int(rng.integers(1, 64))
        delta = [2, 4, 8, 20, 7.5, 13][it % 6]
        cap = batch.capacity_bits(f, h, w, n_ac)
        off = int(rng.integers(0, 97))
        n_bits = int(rng.integers(0, cap + 20))
        bits = rng.integers(0, 2, off + n_bits).astype(np.uint8)
        frames = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
        stego, used = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="fast")
        want, want_used = emu_embed(frames, delta, n_ac, bits, bit_offset=off)
        assert used == want_used == min(n_bits, cap) and np.array_equal(stego, want), (it, f, h, w, n_ac, delta)
        packed, n = batch.extract_frames(stego, delta, n_ac, mode="fast")
        assert np.array_equal(np.unpackbits(packed, count=n), emu_extract(stego, delta, n_ac)), (it, f, h, w, n_ac)
        stego_x, used_x = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="exact")
        ref, ref_used = orc.batch_embed(frames, delta, bits[off:], n_ac)
        assert used_x == ref_used and np.array_equal(stego_x, ref), (it, f, h, w, n_ac, delta)
        packed, n = batch.extract_frames(frames, delta, n_ac, mode="exact")
        assert np.array_equal(np.unpackbits(packed, count=n), orc.batch_extract_bits(frames, delta, n_ac))


def test_guarded_mode_random_geometries_steps_and_budgets_equal_the_oracle():
    """GUARDED over what the structured-content points do not vary (testlib.guarded_soak_cases: shapes, partial waves,
    budgets ending inside a block, bit offsets, every n <= 15, quantiser steps of all three evaluation kinds inside and outside
    the guard's delta range, mixed content).  The stego frames must be the oracle's, byte for byte."""
    for it, frames, delta, n_ac, bits, off, n_bits, cap in guarded_soak_cases():
        stego, used = batch.embed_frames(frames, delta, n_ac, bits, bit_offset=off, n_bits=n_bits, mode="guarded")
        ref, ref_used = orc.batch_embed(frames, delta, bits[off:off + n_bits], n_ac)
        assert used == ref_used == min(n_bits, cap), (it, used, ref_used)
        assert np.array_equal(stego, ref), (it, frames.shape, n_ac, delta, int((stego != ref).sum()))
        packed, n = batch.extract_frames(stego, delta, n_ac, mode="guarded")
        assert np.array_equal(np.unpackbits(packed, count=n), orc.batch_extract_bits(stego, delta, n_ac)), (it, n_ac, delta)


def test_idempotent_and_deterministic():
    cover = synth.synthetic_frames(2, 64, 64, seed=1)
    payload = synth.synthetic_bits(2 * 64 * 5, seed=1)
    a, _ = batch.embed_frames(cover, 8, 5, payload, mode="fast")
    b, _ = batch.embed_frames(cover, 8, 5, payload, mode="fast")
    assert np.array_equal(a, b)
    # embedding the bits a frame already carries changes the quantisation index by nothing: the
    # stego frame's own bits re-embedded give a frame that still decodes to them
    packed, n = batch.extract_frames(a, 8, 5, mode="fast")
    c, _ = batch.embed_frames(a, 8, 5, np.unpackbits(packed, count=n), mode="fast")
    packed2, _ = batch.extract_frames(c, 8, 5, mode="fast")
    assert np.array_equal(packed, packed2)


# ---- device-pointer entry points: pitches, aliasing, helper kernels -----------------------------
class _Dev:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        native.check(native.load().svs_malloc(C.byref(self.ptr), nbytes), "svs_malloc")
        self.nbytes = nbytes

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        native.check(native.load().svs_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, None), "h2d")
        native.check(native.load().svs_stream_synchronize(None), "sync")

    def get(self, nbytes=None, dtype=np.uint8):
        out = np.empty(nbytes or self.nbytes, np.uint8)
        native.check(native.load().svs_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "d2h")
        native.check(native.load().svs_stream_synchronize(None), "sync")
        return out.view(dtype)

    def __del__(self):
        native.load().svs_free(self.ptr)


def test_pitched_planes_in_place_and_device_helpers():
    lib = native.load()
    f, h, w, n_ac, delta = 3, 40, 72, 6, 8
    row_pitch, frame_pitch = 128, 128 * 40 + 256
    planes = Planes(f, h, w, 0, row_pitch, frame_pitch)
    span = f * frame_pitch
    d_frames = _Dev(span)
    native.check(lib.svs_memset(d_frames.ptr, 0xAB, span, None), "memset")
    native.check(lib.svs_fill_synthetic_dev(d_frames.ptr, C.byref(planes), 99, 4, 16, 224, None), "fill")
    host = d_frames.get()
    want = synth.synthetic_frames(f, h, w, seed=99, first_frame=4)
    view = np.stack([host[k * frame_pitch:k * frame_pitch + h * row_pitch].reshape(h, row_pitch)[:, :w]
                     for k in range(f)])
    assert np.array_equal(view, want)                       # on-device generator == NumPy generator
    pad = host.copy()

    cap = batch.capacity_bits(f, h, w, n_ac)
    d_bits = _Dev((cap + 7) // 8 + 8)
    native.check(lib.svs_fill_bits_dev(d_bits.ptr, cap, 99, 1000, None), "fill_bits")
    bits = np.unpackbits(d_bits.get((cap + 7) // 8), count=cap)
    assert np.array_equal(bits, synth.synthetic_bits(cap, seed=99, first_bit=1000))

    used = batch.embed_device(d_frames.ptr, d_frames.ptr, planes, delta, n_ac, d_bits.ptr, 0, cap)   # in place
    assert used == cap
    after = d_frames.get()
    stego = np.stack([after[k * frame_pitch:k * frame_pitch + h * row_pitch].reshape(h, row_pitch)[:, :w]
                      for k in range(f)])
    ref, _ = batch.embed_frames(want, delta, n_ac, bits, mode="fast")
    assert np.array_equal(stego, ref)
    # EXACT mode through the device-pointer entry points, pitched and in place as well
    native.check(lib.svs_fill_synthetic_dev(d_frames.ptr, C.byref(planes), 99, 4, 16, 224, None), "fill")
    assert batch.embed_device(d_frames.ptr, d_frames.ptr, planes, delta, n_ac, d_bits.ptr, 0, cap, mode="exact") == cap
    after_x = d_frames.get()
    stego_x = np.stack([after_x[k * frame_pitch:k * frame_pitch + h * row_pitch].reshape(h, row_pitch)[:, :w]
                        for k in range(f)])
    assert np.array_equal(stego_x, orc.batch_embed(want, delta, bits, n_ac)[0])
    native.check(lib.svs_memcpy_h2d(d_frames.ptr, after.ctypes.data, after.nbytes, None), "restore")
    native.check(lib.svs_stream_synchronize(None), "sync")
    mask = np.ones(span, bool)                              # padding bytes are never written
    for k in range(f):
        for y in range(h):
            mask[k * frame_pitch + y * row_pitch:k * frame_pitch + y * row_pitch + w] = False
    assert np.array_equal(after[mask], pad[mask])

    d_out = _Dev((cap + 7) // 8 + 8)
    n_bits = batch.extract_device(d_frames.ptr, planes, delta, n_ac, d_out.ptr, d_out.nbytes)
    assert n_bits == cap
    assert np.array_equal(np.unpackbits(d_out.get((cap + 7) // 8), count=cap), bits)

    # on-device checkers used by bench.py
    d_cnt = _Dev(8)
    native.check(lib.svs_bit_errors_dev(d_out.ptr, d_bits.ptr, cap, d_cnt.ptr, None), "bit_errors")
    assert int(d_cnt.get(dtype=np.uint64)[0]) == 0
    flipped = np.packbits(bits ^ (np.arange(cap) % 97 == 0).astype(np.uint8))
    d_bits.put(flipped)
    native.check(lib.svs_bit_errors_dev(d_out.ptr, d_bits.ptr, cap, d_cnt.ptr, None), "bit_errors")
    assert int(d_cnt.get(dtype=np.uint64)[0]) == int((np.arange(cap) % 97 == 0).sum())

    d_cover = _Dev(span)
    native.check(lib.svs_fill_synthetic_dev(d_cover.ptr, C.byref(planes), 99, 4, 16, 224, None), "fill")
    d_sse = _Dev(8 * f)
    native.check(lib.svs_frame_sse_dev(d_cover.ptr, d_frames.ptr, C.byref(planes), d_sse.ptr, None), "sse")
    sse = d_sse.get(dtype=np.uint64)
    want_sse = ((want.astype(np.int64) - stego.astype(np.int64)) ** 2).reshape(f, -1).sum(1)
    assert np.array_equal(sse.astype(np.int64), want_sse)


def test_pitched_planes_in_place_with_two_coefficient_rows():
    """The same pitched, in-place device-pointer call with n = 10 / 13: GUARDED (the rigorous two-row kernel) must give the
    oracle's frames and FAST what the CPU build of the header gives; padding bytes stay untouched; also with an odd and an even
    number of blocks per row (one / the 16-byte path is not taken at two rows, but the tile mapping differs)."""
    lib = native.load()
    for (f, h, w, n_ac, delta, row_pitch) in [(3, 40, 72, 10, 8, 128), (2, 48, 96, 13, 20, 96), (2, 16, 64, 10, 7.5, 80)]:
        frame_pitch = row_pitch * h + 64
        planes = Planes(f, h, w, 0, row_pitch, frame_pitch)
        span = f * frame_pitch
        want = synth.synthetic_frames(f, h, w, seed=5, first_frame=2)
        want[0, :16, :24] = 77                                            # some flat blocks
        cap = batch.capacity_bits(f, h, w, n_ac)
        bits = synth.synthetic_bits(cap - 37, seed=6)                      # ends inside the last frame
        packed = batch.pack_bits(bits)
        d_bits = _Dev(packed.size + 8)
        d_bits.put(np.concatenate([packed, np.zeros(8, np.uint8)]))
        for mode in ("guarded", "fast"):
            host = np.full(span, 0xAB, np.uint8)
            for k in range(f):
                host[k * frame_pitch:k * frame_pitch + h * row_pitch].reshape(h, row_pitch)[:, :w] = want[k]
            d_frames = _Dev(span)
            d_frames.put(host)
            used = batch.embed_device(d_frames.ptr, d_frames.ptr, planes, delta, n_ac, d_bits.ptr, 0, bits.size, mode=mode)
            assert used == bits.size
            after = d_frames.get()
            stego = np.stack([after[k * frame_pitch:k * frame_pitch + h * row_pitch].reshape(h, row_pitch)[:, :w]
                              for k in range(f)])
            ref = orc.batch_embed(want, delta, bits, n_ac)[0] if mode == "guarded" else emu_embed(want, delta, n_ac, bits)[0]
            assert np.array_equal(stego, ref), (mode, f, h, w, n_ac, delta, int((stego != ref).sum()))
            mask = np.ones(span, bool)
            for k in range(f):
                for y in range(h):
                    mask[k * frame_pitch + y * row_pitch:k * frame_pitch + y * row_pitch + w] = False
            assert np.all(after[mask] == 0xAB), (mode, "padding written")


def test_device_psnr_and_ssim_evaluators():
    """per-frame PSNR / SSIM on device against the CPU restatements (oracle/metrics_oracle.py;
    SSIM restates skimage's algorithm - skimage itself is not installed, so that leg is parity-unpinned)."""
    from oracle import metrics_oracle as mo
    from svsdct import metrics
    f, h, w = 3, 200, 328
    cover = synth.synthetic_frames(f, h, w, seed=8)
    cover[1] = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)      # smooth content
    cover[2, :100] = 77                                                                # flat area
    stego, _ = batch.embed_frames(cover, 12, 10, synth.synthetic_bits(batch.capacity_bits(f, h, w, 10), seed=8),
                                  mode="fast")
    planes = Planes.contiguous(f, h, w)
    d_a, d_b = _Dev(cover.nbytes), _Dev(stego.nbytes)
    d_a.put(cover)
    d_b.put(stego)
    for data_range in (255.0, None):
        psnr, ssim = metrics.psnr_ssim_device(d_a.ptr, d_b.ptr, planes, data_range)
        for k in range(f):
            assert abs(psnr[k] - mo.psnr_cv2(cover[k], stego[k])) < 1e-9
            assert abs(ssim[k] - mo.ssim_skimage(cover[k], stego[k], data_range)) < 1e-9, (k, data_range)
    psnr, ssim = metrics.psnr_ssim_device(d_a.ptr, d_a.ptr, planes)
    assert np.isinf(psnr).all() and np.allclose(ssim, 1.0)


def test_device_colour_conversions():
    """BGR -> gray (OpenCV fixed-point formula, both weight tables) and gray -> BGR on device."""
    lib = native.load()
    rng = np.random.default_rng(3)
    f, h, w = 2, 24, 40
    bgr = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
    bgr[0, 0, :4] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 0, 255]]
    planes = Planes.contiguous(f, h, w)
    d_bgr, d_gray, d_back = _Dev(bgr.nbytes), _Dev(f * h * w), _Dev(bgr.nbytes)
    d_bgr.put(bgr)
    b, g, r = (bgr[..., i].astype(np.uint32) for i in range(3))
    for weights, want in ((None, (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15),
                          (np.array([1868, 9617, 4899, 14], np.uint32), (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14)):
        wp = weights.ctypes.data if weights is not None else None
        native.check(lib.svs_bgr_to_gray_dev(d_bgr.ptr, 3 * w, 3 * w * h, d_gray.ptr, C.byref(planes), wp, None), "bgr2gray")
        assert np.array_equal(d_gray.get().reshape(f, h, w), want.astype(np.uint8))
    import config_and_setup as cs
    assert np.array_equal(cs._bgr_to_gray(bgr[0]), d_gray_default(bgr[0]))
    native.check(lib.svs_gray_to_bgr_dev(d_gray.ptr, C.byref(planes), d_back.ptr, 3 * w, 3 * w * h, None), "gray2bgr")
    back = d_back.get().reshape(f, h, w, 3)
    assert np.array_equal(back, np.repeat(d_gray.get().reshape(f, h, w)[..., None], 3, axis=3))
    bad = np.array([1, 2, 3, 15], np.uint32)
    assert lib.svs_bgr_to_gray_dev(d_bgr.ptr, 3 * w, 3 * w * h, d_gray.ptr, C.byref(planes), bad.ctypes.data, None) == \
        native.SVS_ERR_INVALID_ARG


@pytest.mark.parametrize("mode", ["exact", "fast", "guarded"])
def test_fused_colour_embed_equals_convert_embed_convert(mode):
    """SURVEY 8(f) rank 2 ("fused read of 3 B/px"): BGR in -> stego BGR out in one kernel must equal
    cvtColor -> operator -> cvtColor (embed_process.py:117-127) done step by step: same gray reference, same stego
    planes (replicated into B, G, R), same bit count; and extraction straight from BGR frames equals extraction
    from their gray planes.  In exact and guarded mode the stego planes are also the oracle's, pixel for pixel."""
    rng = np.random.default_rng(17)
    for (f, h, w, n_ac, delta, short) in ((2, 24, 40, 3, 8, 0), (3, 64, 128, 10, 12.5, 37), (1, 8, 8, 63, 5, 0),
                                          (2, 16, 24, 20, 0.3, 11), (1, 32, 32, 7, 16, 1000000)):
        bgr = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
        bgr[0, :8, :8] = 130                                            # flat block: DC only
        gray = np.stack([d_gray_default(x) for x in bgr])
        cap = batch.capacity_bits(f, h, w, n_ac)
        payload = synth.synthetic_bits(max(1, cap - short), seed=f * 100 + n_ac)
        want_stego, want_used = batch.embed_frames(gray, delta, n_ac, payload, mode=mode)
        got_bgr, got_gray, used = batch.embed_bgr_frames(bgr, delta, n_ac, payload, mode=mode)
        assert used == want_used
        assert np.array_equal(got_gray, gray)
        assert np.array_equal(got_bgr, np.repeat(want_stego[..., None], 3, axis=3)), (f, h, w, n_ac, delta)
        if mode in ("exact", "guarded"):
            ref_stego, ref_used = orc.batch_embed(gray, delta, payload, n_ac)
            assert ref_used == used and np.array_equal(got_bgr[..., 1], ref_stego)
        for src in (got_bgr, bgr):                                      # stego frames and never-embedded frames
            packed, n_bits = batch.extract_bgr_frames(src, delta, n_ac)
            src_gray = np.stack([d_gray_default(x) for x in src])
            assert n_bits == cap
            assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(src_gray, delta, n_ac))
    # extraction straight from never-embedded colour frames of real size (round 4: the fused extract kernel takes the two-step
    # FAST extraction for n >= 8 - candidates, pocketfft coefficient 4, per-block margin, 8-lane replay of tie blocks): ties of
    # c / delta are frequent on covers (1 block in 8 delta at flat index 4), every bit must be the oracle's
    if mode == "guarded":
        big = rng.integers(0, 256, (2, 272, 480, 3), dtype=np.uint8)
        big[1] = np.repeat(rng.integers(0, 256, (272, 480, 1), dtype=np.uint8), 3, axis=2)      # gray content in BGR
        big_gray = np.stack([d_gray_default(x) for x in big])
        for n_ac, delta in ((10, 8), (10, 1), (15, 20), (20, 8), (40, 4), (9, 0.0005)):
            packed, n_bits = batch.extract_bgr_frames(big, delta, n_ac)
            assert np.array_equal(np.unpackbits(packed, count=n_bits), orc.batch_extract_bits(big_gray, delta, n_ac)), (n_ac, delta)
    # custom weight table + no gray reference + empty payload (frames are just converted)
    bgr = rng.integers(0, 256, (1, 40, 48, 3), dtype=np.uint8)
    w14 = np.array([1868, 9617, 4899, 14], np.uint32)
    b, g, r = (bgr[..., i].astype(np.uint32) for i in range(3))
    g14 = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
    out, none, used = batch.embed_bgr_frames(bgr, 8, 3, np.zeros(0, np.uint8), mode=mode, weights=w14, want_gray=False)
    assert none is None and used == 0 and np.array_equal(out, np.repeat(g14[..., None], 3, axis=3))
    want, _ = batch.embed_frames(g14, 8, 3, np.ones(50, np.uint8), mode=mode)
    out, _, used = batch.embed_bgr_frames(bgr, 8, 3, np.ones(50, np.uint8), mode=mode, weights=w14)
    assert used == 50 and np.array_equal(out[..., 0], want)


def test_fused_colour_pitched_frames_leave_padding_alone():
    """Row / frame pitches larger than the frames (both BGR buffers and the gray reference): same pixels as the packed
    call, padding bytes untouched - the wave-cooperative addressing must never step outside a block row."""
    rng = np.random.default_rng(5)
    f, h, w, n_ac, delta = 3, 24, 88, 10, 12          # 11 blocks per row: waves straddle rows and frames
    bgr = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
    payload = synth.synthetic_bits(batch.capacity_bits(f, h, w, n_ac) - 5, seed=9)
    want_bgr, want_gray, want_used = batch.embed_bgr_frames(bgr, delta, n_ac, payload, mode="fast")
    irp, orp, grp = 3 * w + 24, 3 * w + 8, w + 16
    ifp, ofp, gfp = irp * h + 64, orp * (h + 2), grp * h + 8
    src = np.full(f * ifp, 0xA5, np.uint8)
    for k in range(f):
        for y in range(h):
            src[k * ifp + y * irp: k * ifp + y * irp + 3 * w] = bgr[k, y].reshape(-1)
    d_in, d_out, d_gray = _Dev(src.size), _Dev(f * ofp), _Dev(f * gfp)
    d_in.put(src)
    d_out.put(np.full(f * ofp, 0x5A, np.uint8))
    d_gray.put(np.full(f * gfp, 0x3C, np.uint8))
    packed = batch.pack_bits(payload)
    d_bits = _Dev(packed.size)
    d_bits.put(packed)
    planes = Planes(n_frames=f, height=h, width=w, row_pitch=grp, frame_pitch=gfp)
    used = batch.embed_bgr_device(d_in.ptr.value, d_out.ptr.value, d_gray.ptr.value, planes, delta, n_ac,
                                  d_bits.ptr.value, 0, payload.size, mode="fast", in_pitches=(irp, ifp),
                                  out_pitches=(orp, ofp))
    native.check(native.load().svs_stream_synchronize(None), "sync")
    assert used == want_used
    out, gray = d_out.get(), d_gray.get()
    mask_out, mask_gray = np.ones(out.size, bool), np.ones(gray.size, bool)
    for k in range(f):
        for y in range(h):
            a = k * ofp + y * orp
            assert np.array_equal(out[a: a + 3 * w], want_bgr[k, y].reshape(-1)), (k, y)
            mask_out[a: a + 3 * w] = False
            b = k * gfp + y * grp
            assert np.array_equal(gray[b: b + w], want_gray[k, y]), (k, y)
            mask_gray[b: b + w] = False
    assert (out[mask_out] == 0x5A).all() and (gray[mask_gray] == 0x3C).all()
    # extraction from the pitched stego buffer
    cap = batch.capacity_bits(f, h, w, n_ac)
    d_ext = _Dev((cap + 7) // 8 + 8)
    got = batch.extract_bgr_device(d_out.ptr.value, Planes.contiguous(f, h, w), delta, n_ac, d_ext.ptr.value,
                                   (cap + 7) // 8 + 8, pitches=(orp, ofp))
    native.check(native.load().svs_stream_synchronize(None), "sync")
    want_packed, want_n = batch.extract_bgr_frames(want_bgr, delta, n_ac)
    assert got == want_n == cap and np.array_equal(d_ext.get()[: want_packed.size], want_packed)


def test_fused_colour_argument_checks():
    lib = native.load()
    planes = Planes.contiguous(1, 8, 8)
    d = _Dev(1024)
    bad_w = np.array([1, 2, 3, 15], np.uint32)
    args = dict(delta=8.0, n_ac=3)
    assert lib.svs_embed_bgr_dev(d.ptr, 24, 192, d.ptr, 24, 192, None, C.byref(planes), bad_w.ctypes.data, 8.0, 3,
                                 d.ptr, 0, 8, 0, None, None) == native.SVS_ERR_INVALID_ARG
    assert lib.svs_embed_bgr_dev(d.ptr, 20, 192, d.ptr, 24, 192, None, C.byref(planes), None, 8.0, 3,
                                 d.ptr, 0, 8, 0, None, None) == native.SVS_ERR_INVALID_ARG       # pitch < 3*width
    assert lib.svs_embed_bgr_dev(d.ptr, 24, 192, None, 24, 192, None, C.byref(planes), None, 8.0, 3,
                                 d.ptr, 0, 8, 0, None, None) == native.SVS_ERR_INVALID_ARG       # no output
    assert lib.svs_embed_bgr_dev(d.ptr, 24, 192, d.ptr, 24, 192, None, C.byref(planes), None, 8.0, 3,
                                 d.ptr, 0, 8, 4, None, None) == native.SVS_ERR_INVALID_ARG       # unknown flag
    assert lib.svs_extract_bgr_dev(d.ptr, 24, 192, C.byref(planes), None, 8.0, 3, d.ptr, 0, None, None) == \
        native.SVS_ERR_CAPACITY


def d_gray_default(frame_bgr):
    b, g, r = (frame_bgr[..., i].astype(np.uint32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def test_overlapped_host_pipeline_equals_one_shot_calls():
    """pinned, multi-stream staging (svsdct/pipeline.py) must give the frames / bits of the plain
    host-array calls, batch after batch, including a short last batch and a payload that ends mid-clip."""
    from svsdct.pipeline import FramePipeline
    h, w, n_ac, delta, per_batch, n_frames = 64, 96, 5, 8, 4, 14
    clip = synth.synthetic_frames(n_frames, h, w, seed=31)
    cap1 = batch.capacity_bits(1, h, w, n_ac)
    payload = synth.synthetic_bits(cap1 * 9 + 11, seed=31)                 # ends inside frame 9
    want, used = batch.embed_frames(clip, delta, n_ac, payload, mode="fast")
    want_bits, n_all = batch.extract_frames(want, delta, n_ac, mode="fast")
    with FramePipeline(h, w, per_batch, delta, n_ac, depth=3) as pipe:
        pipe.set_payload(payload)
        batches = [clip[i:i + per_batch] for i in range(0, n_frames, per_batch)]
        got, carried = [None] * len(batches), 0
        for k, frames in enumerate(batches + [None] * pipe.depth):
            slot = k % pipe.depth
            if k >= pipe.depth:
                got[k - pipe.depth] = pipe.embed_result(slot).copy()
            if frames is not None:
                np.copyto(pipe.input(slot)[:len(frames)], frames)
                carried += pipe.submit_embed(slot, len(frames), bit_offset=k * pipe.batch_capacity)
        stego = np.concatenate(got)
        assert carried == used == payload.size and np.array_equal(stego, want)
        streams = []
        for k, frames in enumerate([stego[i:i + per_batch] for i in range(0, n_frames, per_batch)]):
            np.copyto(pipe.input(0)[:len(frames)], frames)
            pipe.submit_extract(0, len(frames))
            packed, n = pipe.extract_result(0)
            streams.append(np.unpackbits(packed, count=n))
        assert np.array_equal(np.concatenate(streams), np.unpackbits(want_bits, count=n_all))


def test_error_codes():
    lib = native.load()
    bad = Planes(1, 12, 16, 0, 16, 12 * 16)
    buf = np.zeros(64 * 64, np.uint8)
    out = np.zeros(64, np.uint8)
    got = C.c_uint64()
    rc = lib.svs_extract(buf.ctypes.data, C.byref(bad), 8.0, 3, out.ctypes.data, out.size, 0, C.byref(got))
    assert rc == native.SVS_ERR_INVALID_ARG and b"multiples of 8" in lib.svs_last_error()
    ok = Planes.contiguous(1, 64, 64)
    rc = lib.svs_extract(buf.ctypes.data, C.byref(ok), 8.0, 3, out.ctypes.data, 2, 0, C.byref(got))
    assert rc == native.SVS_ERR_CAPACITY
    rc = lib.svs_extract(buf.ctypes.data, C.byref(ok), 8.0, 3, out.ctypes.data, out.size, 0x80, C.byref(got))
    assert rc == native.SVS_ERR_INVALID_ARG and b"flags" in lib.svs_last_error()
    rc = native.SVS_ERR_CAPACITY
    assert rc == native.SVS_ERR_CAPACITY
    rc = lib.svs_embed(None, None, C.byref(ok), 8.0, 3, None, 0, 0, 0, C.byref(got))
    assert rc == native.SVS_ERR_INVALID_ARG
    with pytest.raises(native.SvsNativeError):
        native.check(rc, "svs_embed")


# ---- the drop-in operator ------------------------------------------------------------------------
def test_drop_in_operator_matches_reference_contract(golden):
    import config_and_setup as cs
    arrays, meta = golden
    info, gray, payload = case_inputs(arrays, meta, "G1_n10_d20")
    pstr = orc.bits_to_str(payload)
    g, stego, used = cs.proses_frame_qim_dct(gray, "embed", 20, pstr + "0101", num_ac_coeffs_to_use=10)
    assert used == info["used"] and g is not gray and np.array_equal(g, gray)
    assert stego.dtype == np.uint8 and stego.shape == gray.shape
    assert np.array_equal(stego, arrays["G1_n10_d20/stego"])           # the operator runs the product default (guarded): the reference's pixels
    text = cs.proses_frame_qim_dct(stego, "extract", 20, enable_debug_prints_extract=False, num_ac_coeffs_to_use=10)
    assert isinstance(text, str) and text == pstr
    assert cs.proses_frame_qim_dct(arrays["G1_n10_d20/stego"], "extract", 20, num_ac_coeffs_to_use=10) == \
        orc.bits_to_str(golden_bits(arrays, "G1_n10_d20", "ext_stego", info["ext_stego_len"]))
    assert cs.proses_frame_qim_dct(gray, "nonsense", 20) is None
    with pytest.raises(ValueError, match=meta["bad_rank_error"]):
        cs.proses_frame_qim_dct(np.zeros((8, 8, 4), np.uint8), "extract", 8)
    g, s, used = cs.proses_frame_qim_dct(gray, "embed", 20, None, num_ac_coeffs_to_use=10)
    assert used == 0 and np.array_equal(s, gray)
    # every single-frame golden through the operator as the reference's callers use it ('0'/'1' strings; degenerate cases too:
    # delta <= 0 and n = 0 with a non-empty payload round-trip every block, an empty or None payload copies the frame)
    for name in single_frame_cases(meta):
        info_k, gray_k, payload_k = case_inputs(arrays, meta, name)
        seg = (orc.bits_to_str(payload_k) if payload_k.size else "") if info_k["payload_len"] is not None else None
        g, s, used = cs.proses_frame_qim_dct(gray_k, "embed", info_k["delta"], seg, num_ac_coeffs_to_use=info_k["n_ac"])
        assert used == info_k["used"] and sha(s) == info_k["stego_sha256"] and np.array_equal(g, gray_k), name
        out = cs.proses_frame_qim_dct(s, "extract", info_k["delta"], num_ac_coeffs_to_use=info_k["n_ac"])
        assert out == orc.bits_to_str(golden_bits(arrays, name, "ext_stego", info_k["ext_stego_len"])), name
    # default n = 63, delta as float
    g, s, used = cs.proses_frame_qim_dct(gray, "embed", 7.5, pstr)
    assert used == len(pstr)
    assert cs.proses_frame_qim_dct(s, "extract", 7.5)[:used] == orc.frame_extract(s, 7.5, 63)[:used]


def test_embed_calls_are_stateless_across_sizes_streams_and_host_threads():
    """The embed kernels keep nothing between calls (undecided blocks are redone inside the launch), so calls of different
    sizes, block mappings (one / two blocks per lane) and content (every block of a wave undecided, a few, none) can be
    interleaved on two streams, and the host-pointer entry points can run concurrently from several host threads
    (the round-2 replay map was shared per device).  Every result must equal the CPU build of the header."""
    import threading
    lib = native.load()
    rng = np.random.default_rng(12)
    streams = []
    for _ in range(2):
        st = C.c_void_p()
        native.check(lib.svs_stream_create(C.byref(st)), "stream")
        streams.append(st)
    shapes = [(2, 64, 96), (5, 128, 256), (1, 8, 8), (3, 72, 88), (7, 256, 512), (2, 64, 96), (1, 40, 24)]   # 88/8, 24/8 odd

    def make(k, f, h, w, n_ac):
        kind = k % 3
        if kind == 0:
            cover = np.full((f, h, w), 100 + k, np.uint8)                    # flat: many undecided blocks per wave
        elif kind == 1:
            cover = rng.integers(16, 240, (f, h, w), dtype=np.uint8)         # noise: a few per cent
        else:
            cover = rng.integers(16, 240, (f, h, w), dtype=np.uint8)
            cover[:, : h // 2] = 200
        off = int(rng.integers(0, 100))                                       # stream starts at a bit offset
        cap = batch.capacity_bits(f, h, w, n_ac)
        short = min(int(rng.integers(0, 3 * n_ac)), cap - 1)                  # ... and ends inside the last blocks
        bits = rng.integers(0, 2, off + cap - short).astype(np.uint8)
        return cover, bits, off

    for n_ac, delta, mode in ((3, 8, "fast"), (1, 8, "guarded"), (10, 8, "fast")):
        jobs = []
        for k, (f, h, w) in enumerate(shapes * 2):
            cover, bits, off = make(k, f, h, w, n_ac)
            d_in, d_out, d_bits = _Dev(cover.nbytes), _Dev(cover.nbytes), _Dev(batch.pack_bits(bits).nbytes)
            d_in.put(cover)
            d_bits.put(batch.pack_bits(bits))
            st = streams[k % 2]
            used = batch.embed_device(d_in.ptr.value, d_out.ptr.value, Planes.contiguous(f, h, w), delta, n_ac,
                                      d_bits.ptr.value, off, bits.size - off, stream=st.value, mode=mode)
            assert used == bits.size - off
            jobs.append((cover, bits, off, d_in, d_out, d_bits))
        for st in streams:
            native.check(lib.svs_stream_synchronize(st), "sync")
        for cover, bits, off, d_in, d_out, d_bits in jobs:
            want, _ = emu_embed(cover, delta, n_ac, bits, bit_offset=off, exact=4 if mode == "guarded" else 0)
            assert np.array_equal(d_out.get().reshape(cover.shape), want), (cover.shape, n_ac)
    for st in streams:
        native.check(lib.svs_stream_destroy(st), "destroy")

    # four host threads, each looping over its own frames through the host-pointer API (ctypes releases the GIL)
    work = [make(k, *shapes[(k + 1) % len(shapes)], 3) for k in range(4)]
    wants = [emu_embed(c, 8, 3, b, bit_offset=o)[0] for c, b, o in work]
    failures = []

    def run(i):
        try:
            native.ensure_device(0)
            cover, bits, off = work[i]
            for _ in range(6):
                got, used = batch.embed_frames(cover, 8, 3, bits, bit_offset=off, mode="fast")
                if used != bits.size - off or not np.array_equal(got, wants[i]):
                    failures.append(i)
        except Exception as exc:   # noqa: BLE001
            failures.append((i, repr(exc)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures


# ---- the host-pointer boundary: per-thread staging context, chunked full-duplex transfers -------------------
def _pinned(shape):
    from svsdct.hostmem import pinned_empty
    return pinned_empty(shape)


@pytest.mark.parametrize("chunk_kb", [8, 32, 4096])
def test_host_pointer_calls_chunked_staging_pitched_pinned_and_pageable(chunk_kb, monkeypatch):
    """svs_embed / svs_extract move the batch in chunks over the two streams (one up, one down) of a per-thread context (csrc/svs_capi.hip).  With
    the experiments library's SVS_STAGE_CHUNK_KB the chunks become bands of 32 rows (8 KB), groups of three frames (32 KB) or
    the whole batch (4 MB) of a pitched five-frame clip with an odd block count per row; budgets that end in the first chunk,
    inside a later one, exactly on a chunk boundary and beyond the capacity; pageable and page-locked buffers on either side.
    Every result equals the oracle's, padding bytes of the caller's stego buffer stay untouched, n_embedded adds up."""
    exp = experiments_library()
    monkeypatch.setenv("SVS_STAGE_CHUNK_KB", str(chunk_kb))
    f, h, w, n_ac, delta = 5, 64, 136, 10, 8
    rp, fp = 144, 64 * 144 + 64
    planes = Planes(f, h, w, 0, rp, fp)
    span = (f - 1) * fp + (h - 1) * rp + w
    cover = synth.synthetic_frames(f, h, w, seed=5, lo=0, span=256)
    cover[1, :16] = 200                                                   # flat blocks: replayed exactly inside the launch
    cap = batch.capacity_bits(f, h, w, n_ac)
    per_frame = cap // f
    budgets = [0, 7, per_frame // 2 - 3, per_frame, 2 * per_frame + 170 * n_ac // 2, cap - 1, cap + 9]
    rng = np.random.default_rng(chunk_kb)
    for budget in budgets:
        for in_pinned, out_pinned in ((False, False), (True, True), (False, True), (True, False)):
            off = int(rng.integers(0, 70))
            bits = rng.integers(0, 2, off + budget).astype(np.uint8)
            want, want_used = orc.batch_embed(cover, delta, bits[off:], n_ac)
            src = _pinned(f * fp) if in_pinned else np.empty(f * fp, np.uint8)
            dst = _pinned(f * fp) if out_pinned else np.empty(f * fp, np.uint8)
            src[:] = 0x11
            dst[:] = 0xA5
            for k in range(f):
                src[k * fp:k * fp + h * rp].reshape(h, rp)[:, :w] = cover[k]
            packed = batch.pack_bits(bits)
            used = C.c_uint64(0)
            for flags in (native.SVS_EXACT_GUARDED, native.SVS_EXACT_POCKETFFT):
                dst[:] = 0xA5
                rc = exp.svs_embed(src.ctypes.data, dst.ctypes.data, C.byref(planes), float(delta), n_ac, packed.ctypes.data, off,
                                   budget, flags, C.byref(used))
                assert rc == 0, exp.svs_last_error()
                assert used.value == want_used == min(budget, cap)
                got = np.stack([dst[k * fp:k * fp + h * rp].reshape(h, rp)[:, :w] for k in range(f)])
                assert np.array_equal(got, want), (budget, in_pinned, out_pinned, flags)
                mask = np.ones(f * fp, bool)
                for k in range(f):
                    for y in range(h):
                        mask[k * fp + y * rp:k * fp + y * rp + w] = False
                assert (dst[mask] == 0xA5).all()                          # padding of the caller's buffer left alone
            # and back: extraction of the pitched stego frames through the host pointers
            out = _pinned((cap + 7) // 8 + 8) if out_pinned else np.zeros((cap + 7) // 8 + 8, np.uint8)
            got_bits = C.c_uint64(0)
            rc = exp.svs_extract(dst.ctypes.data, C.byref(planes), float(delta), n_ac, out.ctypes.data, out.size,
                                 native.SVS_EXACT_GUARDED, C.byref(got_bits))
            assert rc == 0 and got_bits.value == cap
            assert np.array_equal(np.unpackbits(out, count=cap), orc.batch_extract_bits(want, delta, n_ac))
    # the string form on the same pitched clip, with the operator's first return value: pixel bytes copied, padding left alone
    text = batch.bits_to_str(bits[off:]).encode()
    ref = np.full(f * fp, 0x5A, np.uint8)
    dst[:] = 0xA5
    rc = exp.svs_embed_str(src.ctypes.data, ref.ctypes.data, dst.ctypes.data, C.byref(planes), float(delta), n_ac, text, len(text),
                           native.SVS_EXACT_GUARDED, C.byref(used))
    assert rc == 0 and used.value == want_used
    for k in range(f):
        assert np.array_equal(dst[k * fp:k * fp + h * rp].reshape(h, rp)[:, :w], want[k])
        assert np.array_equal(ref[k * fp:k * fp + h * rp].reshape(h, rp)[:, :w], cover[k])
    assert (dst[mask] == 0xA5).all() and (ref[mask] == 0x5A).all()
    assert exp.svs_embed_str(src.ctypes.data, dst.ctypes.data, dst.ctypes.data, C.byref(planes), float(delta), n_ac, text, len(text),
                             native.SVS_EXACT_GUARDED, C.byref(used)) == native.SVS_ERR_INVALID_ARG
    assert span <= f * fp
    assert exp.svs_shutdown() == 0                                        # the thread's context goes; the next call rebuilds it
    stego, used = batch.embed_frames(cover, delta, n_ac, synth.synthetic_bits(cap, seed=1))
    assert used == cap


def test_in_place_embed_str_returns_the_frames_before_embedding():
    """svs_embed_str with stego == gray (in-place embedding, which the header allows) and a gray
    reference - the reference must hold the frames BEFORE embedding although the downloads overwrite the source.  Multi-chunk
    geometry (3 x 4K: the product library moves each frame in two bands), pageable and page-locked memory.  Overlapping
    reference buffers are refused."""
    lib = native.load()
    f, h, w, n_ac, delta = 3, 2160, 3840, 3, 8
    cover = synth.synthetic_frames(f, h, w, seed=61)
    cap = batch.capacity_bits(f, h, w, n_ac)
    bits = synth.synthetic_bits(cap - 1234, seed=62)
    text = batch.bits_to_str(bits).encode()
    want, want_used = orc.batch_embed(cover, delta, bits, n_ac)
    planes = Planes.contiguous(f, h, w)
    used = C.c_uint64(0)
    for pinned in (False, True):
        buf = _pinned(f * h * w) if pinned else np.empty(f * h * w, np.uint8)
        ref = _pinned(f * h * w) if pinned else np.empty(f * h * w, np.uint8)
        buf[:] = cover.ravel()
        ref[:] = 0x5A
        rc = lib.svs_embed_str(buf.ctypes.data, ref.ctypes.data, buf.ctypes.data, C.byref(planes), float(delta), n_ac, text, len(text),
                               native.SVS_EXACT_GUARDED, C.byref(used))
        assert rc == 0, lib.svs_last_error()
        assert used.value == want_used == bits.size
        assert np.array_equal(buf.reshape(f, h, w), want), pinned
        assert np.array_equal(ref.reshape(f, h, w), cover), f"gray reference holds stego pixels (pinned = {pinned})"
    # gray_ref_out == gray is the no-copy form; a reference that overlaps stego, or gray partially, is refused
    two = np.empty(2 * f * h * w, np.uint8)
    two[:f * h * w] = cover.ravel()
    a = two.ctypes.data
    assert lib.svs_embed_str(a, a, a + f * h * w, C.byref(planes), float(delta), n_ac, text, len(text), native.SVS_EXACT_GUARDED, C.byref(used)) == 0
    assert np.array_equal(two[f * h * w:].reshape(f, h, w), want)
    assert lib.svs_embed_str(a, a + w, a + f * h * w, C.byref(planes), float(delta), n_ac, text, len(text), native.SVS_EXACT_GUARDED,
                             C.byref(used)) == native.SVS_ERR_INVALID_ARG
    assert lib.svs_embed_str(a, a + f * h * w - w, a + f * h * w, C.byref(planes), float(delta), n_ac, text, len(text), native.SVS_EXACT_GUARDED,
                             C.byref(used)) == native.SVS_ERR_INVALID_ARG


def test_host_pointer_calls_product_library_large_batches_and_shutdown():
    """The product library's own chunking (4 MB): 4K frames go in two bands each, 1080p frames in pairs, 8K in eight bands;
    pageable NumPy input, page-locked output from the package's pool; delta <= 0 with a non-empty payload (every chunk is
    round-tripped by the exact kernel) and an empty payload (pure copy); buffers grow and shrink between calls; svs_shutdown
    in between."""
    from svsdct import hostmem
    lib = native.load()
    for (f, h, w, n_ac, delta) in ((3, 2160, 3840, 3, 8), (5, 1080, 1920, 10, 20), (1, 4320, 7680, 3, 16), (2, 64, 96, 63, 4)):
        cover = synth.synthetic_frames(f, h, w, seed=f + n_ac)
        cap = batch.capacity_bits(f, h, w, n_ac)
        budget = cap - cap // (2 * f) - 5                                # ends inside the last frame, in its first half
        bits = synth.synthetic_bits(budget, seed=h)
        want, want_used = orc.batch_embed(cover[:1], delta, bits[:cap // f], n_ac)     # oracle on frame 0 (seconds at 8K)
        stego, used = batch.embed_frames(cover, delta, n_ac, bits)
        assert used == budget and np.array_equal(stego[0], want[0])
        # the device-pointer call over the whole batch is the yardstick for the rest
        d_in, d_bits = _Dev(cover.nbytes), _Dev(batch.pack_bits(bits).nbytes)
        d_in.put(cover)
        d_bits.put(batch.pack_bits(bits))
        assert batch.embed_device(d_in.ptr.value, d_in.ptr.value, Planes.contiguous(f, h, w), delta, n_ac, d_bits.ptr.value, 0, budget) == budget
        assert np.array_equal(d_in.get().reshape(cover.shape), stego)
        packed, n_bits = batch.extract_frames(stego, delta, n_ac)
        assert n_bits == cap and np.array_equal(np.unpackbits(packed, count=budget), bits)
        if h == 1080:
            assert lib.svs_shutdown() == 0
            same, _ = batch.embed_frames(cover, delta, n_ac, bits)
            assert np.array_equal(same, stego)
            zero, used0 = batch.embed_frames(cover, 0, n_ac, bits)        # delta <= 0: nothing embedded, every block round-tripped
            assert used0 == 0 and np.array_equal(zero[:1], orc.batch_embed(cover[:1], 0, bits, n_ac)[0])
            assert not np.array_equal(zero, cover)
            none, used0 = batch.embed_frames(cover, delta, n_ac, np.zeros(0, np.uint8))
            assert used0 == 0 and np.array_equal(none, cover)
    assert hostmem.stats["reused"] > 0                                    # result arrays cycle through the pinned pool
    hostmem.trim()


def test_integration_md_section_b_stub_runs_as_written(golden):
    """INTEGRATION.md section B is the reference-side binding a maintainer would paste into the reference's
    config_and_setup.py.  The fenced block is extracted from the document, pointed at lib/libsvsdct.so and executed as
    written in a namespace that has what the reference's module has at that point (`np`; no cv2 - it is absent here, and the
    2-D branch does not need it); the resulting proses_frame_qim_dct is run over goldens G1, G2, G5: stego sha256, bit counts
    and extracted strings are the reference's (config_and_setup.py:106-109,172,174)."""
    import re
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    section = text[text.index("## B."):]
    block = re.search(r"```python\n(.*?)```", section, flags=re.S).group(1)
    assert '"/path/to/libsvsdct.so"' in block and "def proses_frame_qim_dct(" in block and "svs_embed_str" in block
    block = block.replace("/path/to/libsvsdct.so", native.LIB_PATH)
    ns = {"np": np, "__name__": "reference_config_and_setup"}
    exec(compile(block, "INTEGRATION.md#B", "exec"), ns)     # noqa: S102 - our own document
    op = ns["proses_frame_qim_dct"]
    assert "cv2" not in ns
    arrays, meta = golden
    ran = 0
    for name in single_frame_cases(meta):
        if not name.startswith(("G1_", "G2_", "G5_")):
            continue
        info, gray, payload = case_inputs(arrays, meta, name)
        delta, n_ac = info["delta"], info["n_ac"]
        seg = orc.bits_to_str(payload) or None
        g, stego, used = op(gray, "embed", delta, seg, num_ac_coeffs_to_use=n_ac)
        assert used == info["used"] and isinstance(used, int), name
        assert g is not gray and np.array_equal(g, gray) and sha(stego) == info["stego_sha256"], name
        for src, tag in ((stego, "ext_stego"), (gray, "ext_cover")):
            out = op(src, "extract", delta, num_ac_coeffs_to_use=n_ac)
            assert isinstance(out, str) and len(out) == info[tag + "_len"], name
            assert out == orc.bits_to_str(golden_bits(arrays, name, tag, info[tag + "_len"])), (name, tag)
        ran += 1
    assert ran >= 8
    with pytest.raises(ValueError, match=meta["bad_rank_error"]):
        op(np.zeros((8, 8, 4), np.uint8), "extract", 8)
    assert op(np.zeros((16, 16), np.uint8), "nonsense", 8) is None


def test_experiments_library_reproduces_the_golden_vectors(golden):
    """lib/variants/libsvsdct_exp.so (same sources, -DSVS_EXPERIMENTS: knobs + measurement hooks) is what the counter-based
    tests and tools/ab_bench.py load - it has to be the reference itself too."""
    arrays, meta = golden
    with using_library(experiments_library()):
        for mode in ("guarded", "exact"):
            for name in single_frame_cases(meta):
                info, gray, payload = case_inputs(arrays, meta, name)
                stego, used = batch.embed_frames(gray, info["delta"], info["n_ac"], payload, mode=mode)
                assert used == info["used"] and sha(stego[0]) == info["stego_sha256"], (mode, name)
                packed, n_bits = batch.extract_frames(stego[0], info["delta"], info["n_ac"], mode=mode)
                assert n_bits == info["ext_stego_len"], name
                assert np.array_equal(packed, arrays[f"{name}/ext_stego"][:packed.size]), (mode, name)


def test_default_mode_launches_the_streaming_kernel(tmp_path):
    """what the drop-in operator and the video pipelines run by default IS the streaming kernel bench.py
    measures.  Checked where the kernels can be told apart - the experiments library's replay counter: embed_kernel adds the
    blocks it redid exactly (flat blocks under zero bits are always among them), embed_exact_kernel never touches it."""
    import config_and_setup as cs
    from svsdct.pipeline import FramePipeline
    exp = experiments_library()
    d_cnt = _Dev(8)

    def counted(fn):
        native.check(exp.svs_memset(d_cnt.ptr, 0, 8, None), "memset")
        native.check(exp.svs_stream_synchronize(None), "sync")
        exp.svs_guard_counter_set(d_cnt.ptr)
        try:
            with using_library(exp):
                fn()
        finally:
            exp.svs_guard_counter_set(None)
        return int(d_cnt.get(8, np.uint64)[0])

    flat = np.full((64, 128), 128, np.uint8)
    for n_ac in (3, 10):
        zeros = "0" * batch.capacity_bits(1, 64, 128, n_ac)
        assert counted(lambda: cs.proses_frame_qim_dct(flat, "embed", 8, zeros, num_ac_coeffs_to_use=n_ac)) == 128   # every block
        assert counted(lambda: batch.embed_frames(flat, 8, n_ac, np.zeros(len(zeros), np.uint8))) == 128
        assert counted(lambda: batch.embed_frames(flat, 8, n_ac, np.zeros(len(zeros), np.uint8), mode="exact")) == 0

        def through_pipeline():
            with FramePipeline(64, 128, 2, 8, n_ac, depth=1) as pipe:          # mode unspecified, as a caller would
                assert pipe.mode == batch.DEFAULT_MODE == "guarded"
                pipe.set_payload(np.zeros(2 * len(zeros), np.uint8))
                pipe.input(0)[:] = 128
                pipe.submit_embed(0, 2, 0)
                pipe.embed_result(0)
        assert counted(through_pipeline) == 256
    assert batch.host_level_mode() == "guarded"


def test_string_payload_entry_points_equal_the_packed_ones():
    """svs_embed_str / svs_extract_str take and return the reference operator's own payload type - '0'/'1' strings
    (config_and_setup.py:106-109,173-174) - and convert on the device.  Same frames, counts and bits as the packed entry
    points for every budget (none, "", one character, mid-block, capacity, the whole remaining payload of a long clip), odd
    character counts (the 32-character packing tail), n = 0 / delta <= 0 with a non-empty payload (every block round-tripped),
    several frames, and sizes that cross the staging ring's slot (1.3 M characters)."""
    rng = np.random.default_rng(8)
    for (f, h, w, n_ac, delta) in ((1, 64, 96, 10, 20), (3, 40, 72, 3, 8), (1, 24, 40, 0, 8), (1, 24, 40, 4, 0), (1, 24, 40, 4, -3),
                                   (2, 16, 16, 63, 4), (1, 2160, 3840, 10, 20), (1, 2160, 3840, 63, 8)):
        cover = synth.synthetic_frames(f, h, w, seed=h + n_ac)
        cap = batch.capacity_bits(f, h, w, n_ac)
        for n_chars in sorted({0, 1, 31, 33, max(cap // 2 - 3, 0), max(cap - 1, 0), cap, cap + 1000}):
            bits = rng.integers(0, 2, n_chars).astype(np.uint8)
            text = batch.bits_to_str(bits)
            want, want_used = batch.embed_frames(cover, delta, n_ac, bits)
            for payload in ((text,) if n_chars else (text, None)):
                got, used = batch.embed_frames_str(cover, delta, n_ac, payload)
                assert used == want_used and np.array_equal(got, want), (f, h, w, n_ac, delta, n_chars)
            # with the operator's first return value: a copy of the input made by the library beside the GPU work
            ref, got, used = batch.embed_frames_str(cover, delta, n_ac, text, want_gray=True)
            assert used == want_used and np.array_equal(got, want) and np.array_equal(ref, cover)
            assert not np.may_share_memory(ref, cover) and not np.may_share_memory(ref, got)
        packed, n_bits = batch.extract_frames(want, delta, n_ac)
        text = batch.extract_frames_str(want, delta, n_ac)
        assert isinstance(text, str) and len(text) == n_bits == cap
        assert text == batch.unpack_to_str(packed, n_bits)
    with pytest.raises(ValueError):
        batch.embed_frames_str(cover, 8, 3, "01\u20ac")                    # not a one-byte-per-character string
    # characters other than '0' / '1' have no pinned meaning (the reference DECREMENTS the index for any digit but 1,
    # whatever its parity) - refused by the library when they lie inside what would be read, ignored beyond the capacity
    small = synth.synthetic_frames(1, 16, 16, seed=3)
    cap16 = batch.capacity_bits(1, 16, 16, 3)
    with pytest.raises(native.SvsNativeError):
        batch.embed_frames_str(small, 8, 3, "0130")
    with pytest.raises(native.SvsNativeError):
        batch.embed_frames_str(small, 8, 3, "01 1")
    ok = "01" * (cap16 // 2)
    got, used = batch.embed_frames_str(small, 8, 3, ok + "xyz")           # read up to the capacity only, as the reference does
    want, _ = batch.embed_frames_str(small, 8, 3, ok)
    assert used == cap16 and np.array_equal(got, want)
    with pytest.raises(TypeError):
        batch.embed_frames_str(small, 8, 3, [0, 1, 1])


def test_fused_colour_host_calls_in_bands_and_frame_groups():
    """svs_embed_bgr moves BGR frames through the same staging pipeline as svs_embed (csrc/svs_capi.hip): 1080p frames (6.2 MB of
    BGR each) travel in two bands, 640 x 480 frames in groups; the budget ends inside a later chunk; gray reference included.
    Same result as the gray path on the converted frames."""
    rng = np.random.default_rng(23)
    for (f, h, w, n_ac, delta) in ((3, 1080, 1920, 10, 20), (12, 480, 640, 3, 8)):
        bgr = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
        gray = np.stack([d_gray_default(x) for x in bgr])
        cap = batch.capacity_bits(f, h, w, n_ac)
        payload = synth.synthetic_bits(cap - cap // (2 * f) - 3, seed=f)
        want, want_used = batch.embed_frames(gray, delta, n_ac, payload)
        got_bgr, got_gray, used = batch.embed_bgr_frames(bgr, delta, n_ac, payload)
        assert used == want_used == payload.size
        assert np.array_equal(got_gray, gray)
        assert np.array_equal(got_bgr, np.repeat(want[..., None], 3, axis=3))
        packed, n_bits = batch.extract_bgr_frames(got_bgr, delta, n_ac)
        assert n_bits == cap and np.array_equal(np.unpackbits(packed, count=payload.size), payload)


def test_one_row_integer_store_plain_and_saturating_waves_equal_the_oracle():
    """Round 6 (embed_row1_kernel): a wave stores pixel dword + packed column deltas with one 32-bit add when NONE of its 128
    blocks can clip, and takes the packed 16-bit saturating form for all of them when one can (a wave-uniform ballot).  A
    1024-pixel-wide frame makes every block row one wave: rows of content that never clips, rows that clip everywhere, rows in
    which a single block touches 0 / 255, rows whose minimum + most negative delta lands exactly on 0 - both blocks-per-lane
    forms (an odd block count per row takes one block per lane), several steps, against the oracle pixel for pixel."""
    rng = np.random.default_rng(66)
    for w in (1024, 1032):
        bands = []
        safe = lambda: rng.integers(40, 216, (8, w))
        bands += [safe(), rng.integers(0, 256, (8, w)), rng.integers(0, 2, (8, w)), rng.integers(254, 256, (8, w))]
        one = safe(); one[3, 517] = 0; bands.append(one)
        two = safe(); two[0, 8] = 255; two[7, w - 1] = 0; bands.append(two)
        for lo in (1, 2, 3, 5, 7, 9, 12):
            bands += [rng.integers(lo, lo + 30, (8, w)), rng.integers(226 - lo, 256 - lo, (8, w))]
        cols = safe(); cols[:, ::2] = 0; cols[:, 1::2] = 255; bands.append(cols)
        bands.append(safe())
        frame = np.concatenate(bands).astype(np.uint8)[None]
        h = frame.shape[1]
        for n_ac, delta in ((3, 8), (1, 20), (7, 4), (3, 0.25), (5, 100), (3, 1000), (7, 37.5)):
            cap = batch.capacity_bits(1, h, w, n_ac)
            for bits in (rng.integers(0, 2, cap).astype(np.uint8), np.ones(cap - 3, np.uint8)):
                want, want_used = orc.batch_embed(frame, delta, bits, n_ac)
                for mode in ("guarded", "exact"):
                    got, used = batch.embed_frames(frame, delta, n_ac, bits, mode=mode)
                    assert used == want_used and np.array_equal(got, want), (w, n_ac, delta, mode, int((got != want).sum()))


def test_staging_buffers_shrink_after_a_run_of_small_calls():
    """the per-thread staging context no longer keeps the device memory of the largest batch it ever moved - a
    buffer above 64 MB that eight calls in a row used less than a quarter of is given back (and not before: a caller that
    alternates large and small batches keeps it)."""
    lib = native.load()
    assert lib.svs_shutdown() == 0
    hip = C.CDLL("libamdhip64.so")                                     # the runtime the library itself is linked against

    def free():
        a, b = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(a), C.byref(b)) == 0
        return a.value

    big = synth.synthetic_frames(48, 1080, 1920, seed=1)               # 99.5 MB of frames
    small = synth.synthetic_frames(1, 480, 640, seed=2)
    bits_big = synth.synthetic_bits(batch.capacity_bits(48, 1080, 1920, 3), seed=3)
    bits_small = synth.synthetic_bits(batch.capacity_bits(1, 480, 640, 3), seed=4)
    batch.embed_frames(small, 8, 3, bits_small)                        # the context exists (streams, small buffers)
    f0 = free()
    batch.embed_frames(big, 8, 3, bits_big)
    f1 = free()
    assert f0 - f1 > 80 << 20, (f0, f1)                                # grew by the batch
    for _ in range(4):
        batch.embed_frames(small, 8, 3, bits_small)
        batch.embed_frames(big, 8, 3, bits_big)                        # alternating: kept
    assert free() <= f1 + (8 << 20)
    for _ in range(9):
        stego, used = batch.embed_frames(small, 8, 3, bits_small)
    assert free() - f1 > 80 << 20, (f1, free())                        # given back after eight small calls in a row
    assert used == bits_small.size
