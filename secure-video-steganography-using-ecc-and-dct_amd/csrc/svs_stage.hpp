// svs_stage.hpp - how the host-pointer entry points (svs_embed, svs_embed_str, svs_embed_bgr: csrc/svs_capi.hip) cut a batch
// into the chunks their upload / kernel / download pipeline works on.  Plain C++ (no HIP): tests/hostemu compiles it for the
// CPU-only test tier (tests/test_stage_plan_cpu.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef SVS_STAGE_CHUNK_BYTES
#define SVS_STAGE_CHUNK_BYTES (8u << 20)     // largest chunk of frames
#endif
#ifndef SVS_STAGE_CHUNK_MIN
#define SVS_STAGE_CHUNK_MIN (4u << 20)       // smallest (a batch below twice this travels in one piece)
#endif

namespace svs {

// A chunk of a batch: frames [f0, f0 + nf) x pixel rows [r0, r0 + rows); nf > 1 only with whole frames (r0 = 0, rows = H).
struct Chunk {
    int32_t f0, nf, r0, rows;
};

// Cuts a batch of n_frames frames of H rows (row_bytes bytes per row; H a multiple of 8) into chunks of about `target` bytes:
// bands of block rows when a frame is larger than that, groups of whole frames otherwise.  The chunks tile the batch in
// stream order (frames in order, rows in order), every band starts on a block row.
template <class F>
inline void for_each_chunk(int32_t n_frames, int32_t H, size_t row_bytes, size_t target, F &&fn) {
    const size_t frame_bytes = (size_t)H * row_bytes;
    if (frame_bytes > target && H > 8) {
        int32_t band = (int32_t)((target / (row_bytes ? row_bytes : 1)) & ~(size_t)7);
        if (band < 8) band = 8;
        // equal bands rather than a short last one
        const int32_t pieces = (H + band - 1) / band;
        band = (((H / 8) + pieces - 1) / pieces) * 8;
        for (int32_t f = 0; f < n_frames; ++f)
            for (int32_t r = 0; r < H; r += band) fn(Chunk{f, 1, r, r + band <= H ? band : H - r});
    } else {
        int32_t group = (int32_t)(target / (frame_bytes ? frame_bytes : 1));
        if (group < 1) group = 1;
        for (int32_t f = 0; f < n_frames; f += group) fn(Chunk{f, f + group <= n_frames ? group : n_frames - f, 0, H});
    }
}

// chunk size of a batch of `total` bytes: an eighth of it within [4 MB, 8 MB].  A chunk costs about 33 us of host time (two
// copies, the kernel launch, the event and its wait), so small chunks lose more than their overlap wins: one 4K frame is
// fastest as two bands of 4 MB (0.34 ms against 0.38 in one piece and 0.49 in eight), a 1080p frame in one piece, a batch of
// 32 4K frames in chunks of 8 MB (6.2 ms against 9.5 in one piece) - tools/stage_chunk_sweep.py, profiles/r05_pcie_rate.txt.
inline size_t stage_chunk_rule(uint64_t total) {
    const uint64_t eighth = total / 8;
    return (size_t)(eighth < SVS_STAGE_CHUNK_MIN ? SVS_STAGE_CHUNK_MIN : (eighth > SVS_STAGE_CHUNK_BYTES ? SVS_STAGE_CHUNK_BYTES : eighth));
}

// bits a chunk that starts at global block g0 sees of a budget of `pass_bits` bits (`use` = bits that can really be embedded;
// use == 0 with pass_bits == 1: a non-empty payload of which nothing can be embedded - every block of EVERY chunk is then
// round-tripped, as the reference does, config_and_setup.py:143-145,166-169)
inline uint64_t chunk_budget(uint64_t pass_bits, uint64_t use, uint64_t g0, uint32_t n) {
    if (use == 0) return pass_bits;
    const uint64_t before = g0 * (uint64_t)n;
    return pass_bits > before ? pass_bits - before : 0;
}

}  // namespace svs
