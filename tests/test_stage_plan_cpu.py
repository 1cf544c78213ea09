"""CPU tier: the chunk plan of the host-pointer entry points (csrc/svs_stage.hpp, compiled into tests/hostemu) - svs_embed,
svs_embed_str and svs_embed_bgr cut a batch into chunks that travel upload -> kernel -> download one behind the other.
Pure arithmetic, so its properties are checked here; the GPU tier checks the bytes (tests/test_gpu_parity.py::
test_host_pointer_calls_chunked_staging_pitched_pinned_and_pageable)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from testlib import hostemu, plan_chunks

MB = 1 << 20


def check_tiling(chunks, n_frames, height):
    """chunks tile the batch exactly once, in stream order; bands start on block rows; groups are whole frames"""
    seen = np.zeros((n_frames, height // 8), np.int32)
    last = (-1, -1)
    for f0, nf, r0, rows in chunks:
        assert nf >= 1 and rows >= 8 and r0 % 8 == 0 and rows % 8 == 0 and r0 + rows <= height and 0 <= f0 and f0 + nf <= n_frames
        if nf > 1:
            assert r0 == 0 and rows == height
        assert (f0, r0) > last                              # stream order: frames in order, rows in order
        last = (f0 + nf - 1, r0 + rows - 1) if nf > 1 else (f0, r0)
        seen[f0:f0 + nf, r0 // 8:(r0 + rows) // 8] += 1
    assert (seen == 1).all()


def test_the_built_in_rule_on_the_shapes_the_measurements_ran():
    """profiles/r05_pcie_rate.txt: one 4K frame = two bands of 1080 rows, one 1080p frame in one piece, 32 4K frames one frame
    per chunk, eight 1080p frames in pairs, one 8K frame in eight or nine bands; a 640x480 frame in one piece"""
    assert plan_chunks(1, 2160, 3840) == [(0, 1, 0, 1080), (0, 1, 1080, 1080)]
    assert plan_chunks(1, 1080, 1920) == [(0, 1, 0, 1080)]
    assert plan_chunks(32, 2160, 3840) == [(f, 1, 0, 2160) for f in range(32)]
    assert plan_chunks(8, 1080, 1920) == [(f, 2, 0, 1080) for f in range(0, 8, 2)]
    assert plan_chunks(1, 480, 640) == [(0, 1, 0, 480)]
    eight_k = plan_chunks(1, 4320, 7680)
    assert 8 <= len(eight_k) <= 9 and len({rows for _, _, _, rows in eight_k[:-1]}) == 1      # equal bands
    check_tiling(eight_k, 1, 4320)
    # the fused colour path counts three bytes per pixel
    assert plan_chunks(1, 1080, 3 * 1920) == [(0, 1, 0, 544), (0, 1, 544, 536)]
    # chunk sizes stay within the rule: between 4 and 8 MB for batches above 8 MB (a lone band or frame may be smaller)
    for (f, h, rb) in ((32, 2160, 3840), (600, 1080, 1920), (5, 4320, 7680), (3, 2160, 3 * 3840)):
        sizes = [nf * rows * rb for _, nf, _, rows in plan_chunks(f, h, rb)]
        assert max(sizes) <= 8 * MB + h * rb // 100 and sorted(sizes)[len(sizes) // 2] >= 3 * MB


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 40), st.integers(1, 60), st.integers(1, 700), st.integers(1, 64 * MB))
def test_any_geometry_and_target_tiles_the_batch(n_frames, block_rows, row_units, target):
    height, row_bytes = 8 * block_rows, 8 * row_units
    chunks = plan_chunks(n_frames, height, row_bytes, target_bytes=target)
    check_tiling(chunks, n_frames, height)
    frame_bytes = height * row_bytes
    for f0, nf, r0, rows in chunks:
        if nf > 1:
            assert nf * frame_bytes <= target                      # groups of whole frames never exceed the target
        elif frame_bytes > target and rows > 8:
            assert rows * row_bytes <= target + 8 * row_bytes      # a band exceeds it by less than a block row (equalised bands)


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 12), st.integers(1, 12), st.integers(1, 40), st.integers(0, 63), st.integers(0, 40000), st.integers(64, 4096))
def test_chunk_budgets_hand_out_the_stream_exactly_once(n_frames, block_rows, wb, n_ac, n_bits, target):
    """every chunk indexes the shared payload by the bit offset of its first block and sees what is left of the budget there:
    the per-chunk counts add up to min(n_bits, capacity), chunks past the end see an empty payload (pure copy, as the
    reference's loops break: config_and_setup.py:130,132,141) - except when nothing can be embedded at all (n_ac = 0):
    then a non-empty payload stays non-empty for EVERY chunk, which the reference round-trips block by block"""
    lib = hostemu()
    height, width = 8 * block_rows, 8 * wb
    chunks = plan_chunks(n_frames, height, width, target_bytes=target)
    bpf = block_rows * wb
    cap = n_frames * bpf * n_ac
    use = min(n_bits, cap)
    pass_bits = use if use else (1 if n_bits else 0)
    total = 0
    for f0, nf, r0, rows in chunks:
        g0 = f0 * bpf + (r0 // 8) * wb
        cap_c = nf * (rows // 8) * wb * n_ac
        got = lib.emu_chunk_budget(pass_bits, use, g0, n_ac)
        if use == 0:
            assert got == pass_bits
        else:
            assert got == max(0, use - g0 * n_ac)
            total += min(got, cap_c)
    if use:
        assert total == use
