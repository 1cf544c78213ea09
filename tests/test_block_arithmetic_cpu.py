"""CPU tier: the per-block arithmetic the gfx950 kernels execute (csrc/svs_block.hpp, compiled
for the host by tests/hostemu - test infrastructure, not a product path) against the pinned
oracle and the reference's golden vectors.  Catches logic and bit-bookkeeping errors without a
GPU; the -m gpu tier repeats the same checks through the real kernels."""
import numpy as np
import pytest
from scipy.fftpack import dct, idct

from testlib import (NOISE_ONLY_CASES, case_inputs, emu_embed, emu_extract, golden_bits, hostemu, near_tie_mask,
                     single_frame_cases)
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

        if name in NOISE_ONLY_CASES:
            assert used == 0 or delta > 0
            assert np.array_equal(stego, gray), name
            continue
        assert used == info["used"] == ref_used, name
        # (c) what a receiver running the reference extracts from our frame == from the reference's frame
        assert np.array_equal(orc.frame_extract_bits(stego, delta, n_ac)[:used],
                              orc.frame_extract_bits(ref_stego, delta, n_ac)[:used]), name
        # (d) PSNR within tolerance
        if np.isfinite(info["psnr"]):
            assert abs(orc.psnr_u8(gray, stego) - info["psnr"]) <= PSNR_TOL_DB, name
        else:
            assert np.array_equal(stego, gray)
        # (e) extraction from the cover: identical except within float32 rounding of a tie of c/delta
        cov = emu_extract(gray, delta, n_ac)
        want = golden_bits(arrays, name, "ext_cover", info["ext_cover_len"])
        ties = near_tie_mask(gray, delta, n_ac).reshape(-1)
        assert np.array_equal(cov[~ties], want[~ties]), name


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
