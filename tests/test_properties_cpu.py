"""CPU tier: randomised (hypothesis) comparison of the kernel arithmetic with the pinned oracle over
shapes, coefficient counts, quantisation steps, budgets and bit offsets the fixed vectors do not cover."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from testlib import emu_embed, emu_extract
from oracle import qim_dct_oracle as orc

DELTAS = st.sampled_from([1, 2, 3, 4, 5, 7.5, 8, 12, 16, 20, 33, 0.75, 100])
import os

# derandomize: the suite the driver runs must not depend on the day's random seed; SVS_HYPOTHESIS_EXAMPLES raises the
# example count (and re-randomises) for exploratory runs
_EXPLORE = int(os.environ.get("SVS_HYPOTHESIS_EXAMPLES", "0"))
COMMON = dict(deadline=None, max_examples=_EXPLORE or 200, derandomize=not _EXPLORE, database=None,
              suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@st.composite
def cases(draw):
    f = draw(st.integers(1, 3))
    h, w = 8 * draw(st.integers(1, 5)), 8 * draw(st.integers(1, 6))
    n_ac = draw(st.integers(0, 70))
    delta = draw(DELTAS)
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    kind = draw(st.sampled_from(["uniform", "narrow", "flat", "gradient"]))
    if kind == "uniform":
        frames = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
    elif kind == "narrow":
        frames = rng.integers(100, 140, (f, h, w), dtype=np.uint8)
    elif kind == "flat":
        frames = np.full((f, h, w), draw(st.integers(0, 255)), np.uint8)
    else:
        frames = ((np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[None] + np.arange(f)[:, None, None]) % 256).astype(np.uint8)
    cap = f * (h // 8) * (w // 8) * max(0, min(n_ac, 63))
    n_bits = draw(st.integers(0, cap + 9))
    off = draw(st.integers(0, 70))
    bits = rng.integers(0, 2, off + n_bits).astype(np.uint8)
    return frames, delta, n_ac, bits, off


@settings(**COMMON)
@given(cases())
def test_exact_mode_equals_oracle_bit_for_bit(case):
    frames, delta, n_ac, bits, off = case
    stego, used = emu_embed(frames, delta, n_ac, bits, bit_offset=off, exact=True)
    want, want_used = orc.batch_embed(frames, delta, bits[off:], n_ac)
    if delta > 0 and min(n_ac, 63) > 0 or bits.size == off:
        assert used == want_used
        assert np.array_equal(stego, want)
    else:
        # nothing can be consumed: the reference's frame loop would never advance; per frame every block is
        # round-tripped - which is what one operator call does
        for k in range(frames.shape[0]):
            assert np.array_equal(stego[k], orc.frame_embed(frames[k], delta, bits[off:], n_ac)[1])
    got = emu_extract(stego, delta, n_ac, exact=True)
    assert np.array_equal(got, orc.batch_extract_bits(stego, delta, n_ac))


@settings(**COMMON)
@given(cases())
def test_fast_mode_bits_equal_oracle(case):
    frames, delta, n_ac, bits, off = case
    stego, used = emu_embed(frames, delta, n_ac, bits, bit_offset=off, exact=False)
    n = max(0, min(n_ac, 63))
    assert used == (min(bits.size - off, frames.shape[0] * (frames.shape[1] // 8) * (frames.shape[2] // 8) * n)
                    if delta > 0 else 0)
    for src in (stego, frames):
        got = emu_extract(src, delta, n_ac, exact=False)
        want = orc.batch_extract_bits(src, delta, n_ac)
        assert np.array_equal(got, want)          # every bit, rounding ties included (exact path near ties)
