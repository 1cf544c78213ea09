"""CPU tier: the per-block arithmetic the gfx950 kernels execute (csrc/svs_block.hpp, compiled
for the host by tests/hostemu - test infrastructure, not a product path) against the pinned
oracle and the reference's golden vectors.  Catches logic and bit-bookkeeping errors without a
GPU; the -m gpu tier repeats the same checks through the real kernels."""
import numpy as np
import pytest
from scipy.fftpack import dct, idct

from testlib import (CONTRACT_POINTS, SMOOTH_COVERS, case_inputs, contract_payloads, emu_embed, emu_extract, golden_bits,
                     hostemu, psnr_gap, single_frame_cases, structured_covers)
from svsdct import batch
from oracle import qim_dct_oracle as orc
from svsdct import synth

PSNR_TOL_DB = 0.01   # BASELINE.json north_star: stego-frame PSNR within +-0.01 dB


def test_forward_and_inverse_transform_accuracy():
    lib = hostemu()
    rng = np.random.default_rng(0)
    for _ in range(300):
        blk = rng.integers(0, 256, (8, 8), dtype=np.uint8)
        got = np.zeros((8, 8), np.float32)
        lib.emu_forward_block(blk.ctypes.data, got.ctypes.data)
        want = dct(dct(blk.astype(np.float64), axis=0, norm="ortho"), axis=1, norm="ortho")
        assert np.abs(got - want).max() < 4e-4          # |D| <= 2040: a few float32 ulps
        vec = rng.normal(0, 50, 8).astype(np.float32)
        out = np.zeros(8, np.float32)
        lib.emu_idct8(vec.ctypes.data, out.ctypes.data)
        assert np.abs(out - idct(vec.astype(np.float64), norm="ortho")).max() < 1e-4


def test_reciprocal_quantiser_equals_ieee_division():
    """quant_index<QM_F32> replaces c/delta by c*(1/delta) with an exact-division fallback near rounding
    ties; it must give the division's result for every input, exact ties (k + 1/2) included."""
    lib = hostemu()
    rng = np.random.default_rng(5)
    total = 0
    for delta in [20, 1, 3, 7, 7.5, 9, 10, 12, 20.0, 25, 33, 100, 0.3, 1e-3, 1000, float(np.float32(6.1))]:
        d32 = float(np.float32(delta))
        if d32 != delta:
            continue                                   # non-representable deltas use the double mode
        parts = [rng.uniform(-2100, 2100, 2_000_000).astype(np.float32),
                 (rng.integers(-16320, 16321, 1_000_000) / 8.0).astype(np.float32)]       # exact multiples of 1/8
        k = np.arange(-3000, 3000, dtype=np.float64)
        ties = ((k + 0.5) * delta).astype(np.float32)                                     # c/delta exactly k + 1/2 ...
        parts += [ties, np.nextafter(ties, np.float32(np.inf)), np.nextafter(ties, np.float32(-np.inf))]
        parts += [np.float32([0.0, -0.0, 1e-30, -1e-30, 1e-45, 3e38, -3e38])]
        c = np.ascontiguousarray(np.concatenate(parts))
        assert lib.emu_quant_mismatches(c.ctypes.data, c.size, float(delta)) == 0, delta
        total += c.size
    assert total > 3e7


def test_golden_vectors(golden):
    arrays, meta = golden
    for name in single_frame_cases(meta):
        info, gray, payload = case_inputs(arrays, meta, name)
        delta, n_ac = info["delta"], info["n_ac"]
        stego, used = emu_embed(gray, delta, n_ac, payload)
        stego = stego[0]
        _, ref_stego, ref_used = orc.frame_embed(gray, delta, payload, n_ac)

        # (a) extraction from the REFERENCE's stego frame is bit-exact
        got = emu_extract(ref_stego, delta, n_ac)
        assert np.array_equal(got, golden_bits(arrays, name, "ext_stego", info["ext_stego_len"])), name
        # (b) extraction from our own stego frame agrees with the oracle on the same frame
        assert np.array_equal(emu_extract(stego, delta, n_ac), orc.frame_extract_bits(stego, delta, n_ac)), name

        assert used == info["used"] == ref_used, name
        # (c) what a receiver running the reference extracts from our frame == from the reference's frame
        assert np.array_equal(orc.frame_extract_bits(stego, delta, n_ac)[:used],
                              orc.frame_extract_bits(ref_stego, delta, n_ac)[:used]), name
        # (d) PSNR within tolerance
        if np.isfinite(info["psnr"]):
            assert abs(orc.psnr_u8(gray, stego) - info["psnr"]) <= PSNR_TOL_DB, name
        else:
            assert np.array_equal(stego, gray)
        # (e) extraction from the never-embedded cover: identical too, rounding ties of c/delta included
        cov = emu_extract(gray, delta, n_ac)
        assert np.array_equal(cov, golden_bits(arrays, name, "ext_cover", info["ext_cover_len"])), name


@pytest.mark.parametrize("n_ac,delta", CONTRACT_POINTS)
def test_fast_mode_psnr_contract_on_structured_content(n_ac, delta):
    """flags = 0 ("fast") against the oracle on content full of flat / one-dimensional / smooth blocks, under random, all-zero
    and sparse payloads: since round 4 every embed mode is BIT-IDENTICAL to the reference (n <= 15: the rigorous guard with
    its in-kernel exact replay; n >= 16: the pocketfft-identical arithmetic) - which is more than north_star's contract
    (stego PSNR within 0.01 dB, extracted bits exact), asserted as well.  FAST extraction of stego, reference stego and
    never-embedded cover equals the oracle's - no masks."""
    h, w = 256, 384
    for name, cover in structured_covers(h, w).items():
        cap = batch.capacity_bits(1, h, w, n_ac)
        for pname, payload in contract_payloads(cap, seed=n_ac * 100 + delta).items():
            replayed = []
            stego, used = emu_embed(cover, delta, n_ac, payload, replayed=replayed)
            _, ref, ref_used = orc.frame_embed(cover, delta, payload, n_ac)
            assert used == ref_used == cap
            assert np.array_equal(stego[0], ref), (name, pname)
            a, b = orc.psnr_u8(cover, stego[0]), orc.psnr_u8(cover, ref)
            assert psnr_gap(a, b) <= PSNR_TOL_DB, (name, pname, a, b)
            if n_ac <= 15 and pname == "bernoulli_half" and (name in ("flat_128", "half_letterbox", "checker_8") or
                                                             (name == "constant_rows" and n_ac <= 7)):
                assert replayed[0] > 0, name                 # the exact replay is really exercised
            for src in (stego[0], ref, cover) if pname == "bernoulli_half" else (stego[0],):
                assert np.array_equal(emu_extract(src, delta, n_ac), orc.frame_extract_bits(src, delta, n_ac)), (name, pname)


@pytest.mark.parametrize("n_ac,delta", [(8, 20), (10, 20), (10, 32), (10, 64), (11, 64), (15, 100), (16, 64), (63, 64)])
def test_fast_mode_on_smooth_content_with_zero_heavy_payloads(n_ac, delta):
    """VERDICT r03 weak #1, at the size of the review's probe (544 x 960): smooth ramps / sinusoid / sigma-1 Gaussian /
    near-black under all-zero and 1 %-ones payloads.  Round 3's FAST arithmetic put the horizontal ramp +0.59 dB off the
    oracle with no block replayed; flags = 0 now runs the bit-identical kernels."""
    h, w = 544, 960
    covers = structured_covers(h, w)
    for name in SMOOTH_COVERS:
        cover = covers[name]
        cap = batch.capacity_bits(1, h, w, n_ac)
        for pname, payload in contract_payloads(cap, seed=n_ac * 100 + delta).items():
            stego, used = emu_embed(cover, delta, n_ac, payload)
            _, ref, ref_used = orc.frame_embed(cover, delta, payload, n_ac)
            assert used == ref_used == cap
            a, b = orc.psnr_u8(cover, stego[0]), orc.psnr_u8(cover, ref)
            assert psnr_gap(a, b) <= PSNR_TOL_DB and np.array_equal(stego[0], ref), (name, pname, a, b)
            assert np.array_equal(emu_extract(stego[0], delta, n_ac), orc.frame_extract_bits(stego[0], delta, n_ac)), (name, pname)


def test_fast_extraction_takes_the_exact_path_only_near_ties():
    """On stego frames with delta >= 8 no quantiser input is near a tie, so the second path is never taken;
    on a never-embedded frame it is taken for the blocks with a coefficient on a tie, and the bits still are the
    reference's."""
    h, w, n_ac, delta = 128, 256, 10, 8
    cover = synth.synthetic_frames(2, h, w, seed=4)
    payload = synth.synthetic_bits(batch.capacity_bits(2, h, w, n_ac), seed=4)
    stego, _ = emu_embed(cover, delta, n_ac, payload)
    redone = []
    assert np.array_equal(emu_extract(stego, delta, n_ac, redone=redone), payload) and redone[0] == 0
    redone = []
    assert np.array_equal(emu_extract(cover, delta, n_ac, redone=redone), orc.batch_extract_bits(cover, delta, n_ac))
    assert 0 < redone[0] < 0.2 * 2 * (h // 8) * (w // 8)


def test_budget_tail_leaves_later_blocks_untouched(golden):
    arrays, meta = golden
    info, gray, payload = case_inputs(arrays, meta, "G2_budget7")
    stego, used = emu_embed(gray, 8, 5, payload)
    assert used == 7
    blocks = lambda a: a.reshape(2, 8, 4, 8).transpose(0, 2, 1, 3).reshape(8, 64)
    changed = (blocks(stego[0]) != blocks(gray)).any(1)
    assert changed[:2].all() and not changed[2:].any()


def test_bit_offset_and_multi_frame_stream(golden):
    arrays, meta = golden
    info = meta["cases"]["G8_stream"]
    frames = synth.synthetic_frames(3, 32, 48, seed=info["synth_seed"])
    payload = arrays["G8_stream/payload"]
    stego, used = emu_embed(frames, info["delta"], info["n_ac"], payload)
    assert used == payload.size == info["used"]
    for k in range(3):
        assert np.array_equal(stego[k], arrays[f"G8_stream/stego{k}"])
    # same payload reached through a non-zero bit offset into a longer buffer
    junk = synth.synthetic_bits(37, seed=9)
    stego2, used2 = emu_embed(frames, info["delta"], info["n_ac"], np.concatenate([junk, payload]), bit_offset=37)
    assert used2 == used and np.array_equal(stego2, stego)
    bits = emu_extract(stego, info["delta"], info["n_ac"])
    assert np.array_equal(bits[:used], payload)


@pytest.mark.parametrize("n_ac,delta", [(3, 8), (10, 8), (21, 12), (40, 16), (63, 8)])
def test_round_trip_is_error_free_for_delta_ge_8(n_ac, delta):
    """delta >= 8 with pixels away from 0/255 is provably error-free."""
    frames = synth.synthetic_frames(2, 64, 96, seed=n_ac, lo=64, span=128)
    cap = 2 * 8 * 12 * n_ac
    payload = synth.synthetic_bits(cap, seed=n_ac)
    stego, used = emu_embed(frames, delta, n_ac, payload)
    assert used == cap
    assert np.array_equal(emu_extract(stego, delta, n_ac), payload)
    assert np.array_equal(orc.batch_extract_bits(stego, delta, n_ac), payload)


def test_forward_transforms_differ_by_less_than_the_tie_bound():
    """SVS_TIE_SLOPE (csrc/svs_block.hpp; derived by tools/tie_bound.py): |c_fast - c_pocketfft| <= slope * c00 for every
    coefficient of every block of pixels in [0, 255].  Checked here on random, bright, dark, high-contrast and
    one-dimensional blocks, and the derivation itself is re-run (its slope must not exceed the constant in the header)."""
    import importlib.util
    import os
    import re
    from testlib import CSRC, REPO
    lib = hostemu()
    text = open(os.path.join(CSRC, "svs_block.hpp")).read()
    m = re.search(r"#define SVS_TIE_SLOPE \(([0-9.e+-]+) \* ([0-9.]+) \+ ([0-9.]+) \* ([0-9.e+-]+)\)", text)
    slope = float(m.group(1)) * float(m.group(2)) + float(m.group(3)) * float(m.group(4))
    spec = importlib.util.spec_from_file_location("tie_bound", os.path.join(REPO, "tools", "tie_bound.py"))
    tb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tb)
    assert tb.main() <= float(m.group(1)) * 1.0001                       # the header's constant covers the derived slope
    # the constants of the per-block margin: at least what tools/guard_bound.py --tie derives for all eight rows
    kdc = float(re.search(r"#define SVS_TIE2_KDC ([0-9.]+)", text).group(1))
    ke = float(re.search(r"#define SVS_TIE2_KE ([0-9.]+)", text).group(1))
    spec = importlib.util.spec_from_file_location("guard_bound", os.path.join(REPO, "tools", "guard_bound.py"))
    gb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gb)
    derived = gb.tie_constants(verbose=False, rows_list=(2, 8))      # monotone in the row count (tools/guard_bound.py --tie)
    assert max(v[0] for v in derived.values()) <= kdc and max(v[1] for v in derived.values()) <= ke <= 1.01 * max(v[1] for v in derived.values())
    ueff = gb.U_EFF
    rng = np.random.default_rng(77)
    blocks = [rng.integers(0, 256, (8, 8)) for _ in range(3000)]
    blocks += [rng.integers(200, 256, (8, 8)) for _ in range(500)] + [rng.integers(0, 8, (8, 8)) for _ in range(200)]
    blocks += [rng.choice([0, 255], (8, 8)) for _ in range(500)]
    blocks += [np.repeat(rng.integers(0, 256, (8, 1)), 8, 1) for _ in range(200)]
    blocks += [np.full((8, 8), v) for v in (1, 127, 128, 255)]
    worst = 0.0
    for blk in blocks:
        blk = np.ascontiguousarray(blk, np.uint8)
        fast = np.zeros((8, 8), np.float32)
        lib.emu_forward_block(blk.ctypes.data, fast.ctypes.data)
        pf = np.zeros((8, 8), np.float32)
        cols = np.zeros((8, 8), np.float32)
        for x in range(8):                                               # vertical first (axis 0), as the reference does
            src = np.ascontiguousarray(blk[:, x].astype(np.float32))
            out = np.zeros(8, np.float32)
            lib.emu_pf_dct2(src.ctypes.data, out.ctypes.data)
            cols[:, x] = out
        for u in range(8):
            src = np.ascontiguousarray(cols[u])
            out = np.zeros(8, np.float32)
            lib.emu_pf_dct2(src.ctypes.data, out.ctypes.data)
            pf[u] = out
        c00 = float(blk.sum()) / 8.0
        if c00 == 0:
            assert np.array_equal(fast, pf)
            continue
        err = np.abs(fast.astype(np.float64) - pf.astype(np.float64)).reshape(-1)[1:].max()
        assert err <= slope * c00, (err, slope * c00)
        worst = max(worst, err / c00)
        # round 3: the per-block margin the kernels actually test (SVS_TIE2_*), in the form they evaluate it
        S = float(blk.sum())
        margin = ueff * (kdc * S / 64.0 + ke * np.sqrt(max(S * (16320.0 - S), 0.0)) / 8.0)
        assert err <= margin, (err, margin)
    assert worst < slope / 10                                            # the proven bound is >= 10x what occurs
