"""CPU tier: EXACT mode (csrc/svs_block.hpp, namespace pf + *_exact) - the bit-for-bit restatement of
pocketfft's float32 8-point DCT-II / DCT-III that scipy.fftpack uses.  With it the whole operator must
equal the reference BIT FOR BIT: stego pixels, extracted bits, rounding ties, round-trip artefacts."""
import numpy as np
from scipy.fftpack import dct, idct

from testlib import case_inputs, emu_embed, emu_extract, golden_bits, hostemu, sha, single_frame_cases
from oracle import qim_dct_oracle as orc
from svsdct import synth


def test_pocketfft_transforms_are_bit_identical_to_scipy():
    lib = hostemu()
    rng = np.random.default_rng(11)
    samples = np.concatenate([
        rng.integers(0, 256, (20000, 8)).astype(np.float32),                 # pixel columns
        rng.normal(0, 300, (20000, 8)).astype(np.float32),                   # coefficient-like
        (rng.integers(-16000, 16000, (20000, 8)) / 8).astype(np.float32),    # exact multiples of 1/8
        np.eye(8, dtype=np.float32) * 255, np.zeros((1, 8), np.float32), np.full((1, 8), 128, np.float32)])
    want2 = dct(samples, axis=1, norm="ortho")
    want3 = idct(samples, axis=1, norm="ortho")
    assert want2.dtype == np.float32
    got2, got3 = np.empty_like(samples), np.empty_like(samples)
    for i, row in enumerate(samples):
        row = np.ascontiguousarray(row)
        lib.emu_pf_dct2(row.ctypes.data, got2[i].ctypes.data)
        lib.emu_pf_dct3(row.ctypes.data, got3[i].ctypes.data)
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))       # bit patterns, signed zeros included
    assert np.array_equal(got3.view(np.uint32), want3.view(np.uint32))


def test_every_golden_vector_bit_for_bit(golden):
    arrays, meta = golden
    for name in single_frame_cases(meta):
        info, gray, payload = case_inputs(arrays, meta, name)
        delta, n_ac = info["delta"], info["n_ac"]
        stego, used = emu_embed(gray, delta, n_ac, payload, exact=True)
        assert used == info["used"], name
        assert sha(stego[0]) == info["stego_sha256"], name                   # pixels identical to the reference
        for src, tag in ((stego[0], "ext_stego"), (gray, "ext_cover")):      # ties included: no mask needed
            assert np.array_equal(emu_extract(src, delta, n_ac, exact=True),
                                  golden_bits(arrays, name, tag, info[tag + "_len"])), (name, tag)


def test_stream_and_partial_budgets_bit_for_bit(golden):
    arrays, meta = golden
    info = meta["cases"]["G8_stream"]
    frames = synth.synthetic_frames(3, 32, 48, seed=info["synth_seed"])
    stego, used = emu_embed(frames, info["delta"], info["n_ac"], arrays["G8_stream/payload"], exact=True)
    assert used == info["used"]
    for k in range(3):
        assert np.array_equal(stego[k], arrays[f"G8_stream/stego{k}"])


def test_larger_random_frames_against_oracle():
    rng = np.random.default_rng(2)
    for (h, w), n_ac, delta in [((96, 160), 10, 8), ((64, 64), 63, 5), ((40, 72), 36, 7.5), ((64, 96), 4, 3)]:
        gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
        bits = rng.integers(0, 2, (h // 8) * (w // 8) * n_ac - 5).astype(np.uint8)
        stego, used = emu_embed(gray, delta, n_ac, bits, exact=True)
        _, want, want_used = orc.frame_embed(gray, delta, bits, n_ac)
        assert used == want_used and np.array_equal(stego[0], want)
        assert np.array_equal(emu_extract(gray, delta, n_ac, exact=True), orc.frame_extract_bits(gray, delta, n_ac))


def test_pair_form_equals_the_one_block_form(golden):
    """embed_block_exact_pair (two adjacent blocks per lane, every transform instruction packed over the pair - what
    embed_exact_pair_kernel runs) gives the bytes of embed_block_exact on every golden case, on partial budgets that end
    inside the first or the second block of a pair, and on random frames."""
    arrays, meta = golden
    for name in single_frame_cases(meta):
        info, gray, payload = case_inputs(arrays, meta, name)
        if (gray.shape[1] // 8) % 2:
            continue
        a, ua = emu_embed(gray, info["delta"], info["n_ac"], payload, exact=2)
        assert ua == info["used"] and sha(a[0]) == info["stego_sha256"], name
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, (2, 32, 64), dtype=np.uint8)
    for n_ac, delta in ((3, 8), (10, 20), (63, 4), (7, 0.3)):
        cap = 2 * 4 * 8 * n_ac
        for n_bits in (cap, cap - 1, n_ac, n_ac + 1, 2 * n_ac, 2 * n_ac - 1, 5 * n_ac + 2, 0, 1):
            bits = rng.integers(0, 2, 11 + n_bits).astype(np.uint8)
            one, u1 = emu_embed(frames, delta, n_ac, bits, bit_offset=11, exact=1)
            two, u2 = emu_embed(frames, delta, n_ac, bits, bit_offset=11, exact=2)
            assert u1 == u2 and np.array_equal(one, two), (n_ac, delta, n_bits)


def test_constant_block_shortcut_equals_the_full_forward_transform():
    """forward_exact_paired_constant (two pocketfft lines instead of sixteen for a block of 64 equal pixels - what the replay
    pass of FAST embedding uses on letterbox bars) gives the bytes of the full transform for every pixel value, with payload
    bits that change nothing, one coefficient or all of them, and for delta <= 0 (round trip only)."""
    rng = np.random.default_rng(5)
    frames = np.repeat(np.repeat(np.arange(256, dtype=np.uint8).reshape(16, 16), 8, axis=0), 8, axis=1)[None]   # 256 constant blocks
    for n_ac, delta in ((3, 8), (3, 16), (7, 4), (10, 20), (63, 4), (5, 0.3), (4, 0), (0, 8)):
        for fill in (0, 1, None):
            cap = 256 * max(0, min(n_ac, 63))
            bits = (rng.integers(0, 2, max(cap, 1)) if fill is None else np.full(max(cap, 1), fill)).astype(np.uint8)
            full, u1 = emu_embed(frames, delta, n_ac, bits, exact=1)
            fast, u2 = emu_embed(frames, delta, n_ac, bits, exact=3)
            assert u1 == u2 and np.array_equal(full, fast), (n_ac, delta, fill)
            ref, _ = orc.batch_embed(frames, delta, bits, n_ac) if (delta > 0 and n_ac > 0) else (None, None)
            if ref is not None:
                assert np.array_equal(fast, ref), (n_ac, delta, fill)
