// svs_device.hpp - gfx950 kernels of the fused block-DCT / QIM frame operator.
// Per-block arithmetic lives in svs_block.hpp; this file maps blocks to lanes, moves bytes
// between HBM and registers, and packs the extracted bit stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svs_block.hpp"

namespace svs {

// ---------------------------------------------------------------------------------------
// geometry shared by all kernels; division by W/8 and by blocks-per-frame is done with
// host-computed multipliers (exact for dividends < 2^31)
// ---------------------------------------------------------------------------------------
struct FastDiv {
    uint32_t mul;
    uint32_t shift;  // 31..62
    uint32_t div;
    uint32_t pad;
};

struct Geometry {
    FastDiv by_wb;          // divide by blocks per block-row
    FastDiv by_bpf;         // divide by blocks per frame
    uint32_t total_blocks;  // n_frames * blocks per frame   (< 2^31)
    uint32_t n_ac;          // 1..63
    uint32_t xcd_chunk;     // tile_id() chunk (0 = identity)
    uint32_t pad;
    int64_t row_pitch;
    int64_t frame_pitch;
};

__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &d) {
    return (uint32_t)(((uint64_t)n * d.mul) >> d.shift);
}

// byte offset of the top-left pixel of global block `gblock` (frames in order, raster inside)
__device__ __forceinline__ int64_t block_offset(uint32_t gblock, const Geometry &g) {
    const uint32_t frame = fast_div(gblock, g.by_bpf);
    const uint32_t in_frame = gblock - frame * g.by_bpf.div;
    const uint32_t brow = fast_div(in_frame, g.by_wb);
    const uint32_t bcol = in_frame - brow * g.by_wb.div;
    return (int64_t)frame * g.frame_pitch + (int64_t)(brow * 8u) * g.row_pitch + (int64_t)(bcol * 8u);
}

#ifndef SVS_WG
#define SVS_WG 256  // threads per workgroup of the embed / extract kernels
#endif

// Workgroup -> tile mapping.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the
// group that shares an XCD and its L2).  With chunk C > 0, the workgroups of one XCD take C consecutive
// tiles at a time: tiles [g*8C + x*C, g*8C + (x+1)*C) go to XCD-group x in round g, so each XCD streams
// runs of C adjacent tiles.  C = 0 is the identity; C = 0xFFFFFFFF gives every XCD-group one contiguous eighth of the grid.  Placement only affects speed, never results.
__device__ __forceinline__ uint32_t tile_id(uint32_t chunk) {
    const uint32_t i = blockIdx.x;
    if (chunk == 0) return i;
    if (chunk == 0xFFFFFFFFu) {  // one contiguous eighth of the grid per XCD-group (bijective for any grid)
        const uint32_t n = gridDim.x, q = n / 8u, r = n % 8u, x = i % 8u;
        return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i / 8u;
    }
    const uint32_t span = 8u * chunk;
    const uint32_t full = (gridDim.x / span) * span;  // tiles covered by whole rounds
    if (i >= full) return i;
    const uint32_t x = i % 8u, j = i / 8u;
    return (j / chunk) * span + x * chunk + (j % chunk);
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Frames are streamed exactly once: non-temporal loads/stores keep them from displacing each other
// in L2 / Infinity Cache (measured on MI355X: embed +5 %, extract +11 %, profiles/history/r01_ab_variants.txt)
#if !defined(SVS_NO_NONTEMPORAL)
#define SVS_LD(p) __builtin_nontemporal_load(p)
#define SVS_ST(v, p) __builtin_nontemporal_store(v, p)
#else
#define SVS_LD(p) (*(p))
#define SVS_ST(v, p) (*(p) = (v))
#endif

// Lanes own one block (8-byte row accesses) or two horizontally adjacent blocks A|B (BPL = 2:
// one 16-byte access per row).  Rows travel as native 2- / 4-dword vectors; the per-block arithmetic
// works on plain scalar arrays filled from them (this exact shape is what hipcc scalarises fully -
// structs of rows updated in place ended up in LDS).
template <int BPL>
struct RowVec;
template <>
struct RowVec<1> { typedef u32x2 type; };
template <>
struct RowVec<2> { typedef u32x4 type; };

template <int BPL>
__device__ __forceinline__ void load_rows(const uint8_t *src, int64_t row_pitch, typename RowVec<BPL>::type (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = SVS_LD(reinterpret_cast<const typename RowVec<BPL>::type *>(src + r * row_pitch));
}

template <int BPL>
__device__ __forceinline__ void store_rows(uint8_t *dst, int64_t row_pitch, const typename RowVec<BPL>::type (&v)[8]) {
#if !defined(SVS_NO_STORE_SC1)
    // write-through (sc1) stores: the line is not kept in the XCD's L2 (measured against non-temporal stores,
    // profiles/history/r01_ab_quant_exact.txt: one-shot copy 6.74 vs 6.55 TB/s; embed +1.5 % at n = 3, +9.5 % at n = 10).  The data registers must not be reused before
    // the store has read them: s_nop 1 inside the string (cdna_hip_programming.md section 5.7 item 1).
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if constexpr (BPL == 1)
            asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + r * row_pitch), "v"(v[r]) : "memory");
        else
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + r * row_pitch), "v"(v[r]) : "memory");
    }
#else
#pragma unroll
    for (int r = 0; r < 8; ++r) SVS_ST(v[r], reinterpret_cast<typename RowVec<BPL>::type *>(dst + r * row_pitch));
#endif
}

// Tail of the extract kernels: a wavefront's 64*BPL consecutive blocks produce exactly n*BPL aligned
// 64-bit words of the packed stream.  The wave assembles them in a private LDS array of big-endian dwords
// (stream bit p of the wave's chunk = bit 31 - p%32 of dword p/32): every lane ORs its n-bit string in at bit
// offset lane*BPL*n (ds_or_b32 on at most three dwords), then lanes w < 2*BPL*n byte-swap one dword each and store
// it.  SVS_WAVE_BITS_DWORDS = dwords per wave incl. slack for the unconditional second / third OR.
#define SVS_WAVE_BITS_DWORDS(BPL) (2 * (BPL) * 63 + 4)

__device__ __forceinline__ void wave_lds_fence() {  // wave-private LDS: pins the compiler's order, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int U>
__device__ __forceinline__ void or_bit_string(uint32_t *mine, uint32_t bit_at, uint32_t hi, uint32_t lo) {
    // hi:lo = up to 8U-1 bits, MSB-first from bit 63; lands at stream bit `bit_at` of the wave's chunk
    const uint32_t d = bit_at >> 5, o = bit_at & 31u;
    atomicOr(&mine[d], hi >> o);
    atomicOr(&mine[d + 1], __builtin_amdgcn_alignbit(hi, lo, o));           // ({hi,lo} >> o) & 0xffffffff
    if constexpr (U > 4) atomicOr(&mine[d + 2], __builtin_amdgcn_alignbit(lo, 0u, o));  // o + n > 64 needs n > 32
}

template <int U, int BPL>
__device__ __forceinline__ void emit_wave_bits(uint32_t *mine, uint32_t lane, uint64_t wave_first_block, uint32_t n,
                                               uint32_t hi_a, uint32_t lo_a, uint32_t hi_b, uint32_t lo_b,
                                               uint8_t *__restrict__ out, uint64_t out_bytes) {
    const uint32_t words = 2u * BPL * n;  // 64*BPL*n bits
    for (uint32_t w = lane; w < words + 4u; w += 64u) mine[w] = 0u;
    wave_lds_fence();
    or_bit_string<U>(mine, lane * BPL * n, hi_a, lo_a);
    if constexpr (BPL == 2) or_bit_string<U>(mine, (lane * BPL + 1u) * n, hi_b, lo_b);
    wave_lds_fence();
    const uint64_t wave_byte0 = wave_first_block * n / 8u;  // multiple of 8 bytes
    for (uint32_t w = lane; w < words; w += 64u) {
        const uint32_t word = __builtin_bswap32(mine[w]);  // first stream bit -> MSB of the first byte in memory
        const uint64_t at = wave_byte0 + 4ull * w;
        if (at + 4 <= out_bytes) {
            *reinterpret_cast<uint32_t *>(out + at) = word;
        } else {
            for (uint32_t j = 0; j < 4 && at + j < out_bytes; ++j) out[at + j] = (uint8_t)(word >> (8 * j));
        }
    }
}

struct GuardEntry {
    uint32_t px[16];   // rows as (low dword, high dword) pairs: the original pixels in, the exact stego pixels out
    uint32_t hi, lo;   // payload window of the block
    uint32_t nb;       // bits the block takes
    uint32_t pad;
};
#define SVS_GUARD_TILE 72   // floats per block of the transposition tile: element (i, j) at 9 i + j; 72 = 8 (mod 64) keeps the
                            // eight groups of a wave on different LDS banks in both directions

// FAST extraction, second path: a block with a quantiser input inside the per-block tie margin (svs_block.hpp SVS_TIE2_*)
// gets the pocketfft-identical forward transform - by EIGHT LANES PER BLOCK, as in the embed kernels' replay: lane r of a
// group transforms column r, then row r, then takes coefficient column r and quantises its coefficients k = 8 u + r of the
// rows u the payload uses; the bits are ORed into the entry.  About 200 instructions per pass of 8 blocks, against 580 for
// every wave that had such a lane when the whole block was redone by its own lane (round 2).
template <int QM>
__device__ __forceinline__ void extract_replay8(GuardEntry *e, float *t, uint32_t r, uint32_t n, const QimParams &qp) {
    float a[8], b[8];
    {
        const uint32_t sh = 8u * (r & 3u), half = r >> 2;
#pragma unroll
        for (int y = 0; y < 8; ++y) a[y] = (float)((e->px[2 * y + half] >> sh) & 0xffu);
    }
    pf::dct2_8(a, b);                       // b[u] = V[u][r]
#pragma unroll
    for (int u = 0; u < 8; ++u) t[9 * u + r] = b[u];
    wave_lds_fence();
#pragma unroll
    for (int x = 0; x < 8; ++x) a[x] = t[9 * r + x];   // V[r][x]
    wave_lds_fence();
    pf::dct2_8(a, b);                       // b[v] = D[r][v]
#pragma unroll
    for (int v = 0; v < 8; ++v) t[9 * r + v] = b[v];
    wave_lds_fence();
    uint32_t hi = 0, lo = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t k = 8u * u + r;
        if (k >= 1u && k <= n) {            // no lane of the wave enters for rows the payload does not use
            const uint32_t bit = (uint32_t)quant_index<QM>(t[9 * u + r], qp) & 1u;
            const uint32_t i = k - 1u;
            if (i < 32u) hi |= bit << (31u - i);
            else lo |= bit << (63u - i);
        }
    }
    if (hi) atomicOr(&e->hi, hi);
    if (lo) atomicOr(&e->lo, lo);
}

// the wave's tie blocks through the worklist (CAP entries per round); on return hi/lo of those blocks hold the reference's bits
template <int QM, bool TWO, int CAP>
__device__ __forceinline__ void extract_phase2(GuardEntry *entries, float *tile, uint32_t lane, uint32_t n, const QimParams &qp,
                                               bool tie_a, const uint32_t (&ax)[8], const uint32_t (&ay)[8], uint32_t &hi_a, uint32_t &lo_a,
                                               bool tie_b, const uint32_t (&bx)[8], const uint32_t (&by)[8], uint32_t &hi_b, uint32_t &lo_b) {
    const uint64_t mask_a = __ballot(tie_a);
    const uint64_t mask_b = TWO ? __ballot(tie_b) : 0ull;
    if ((mask_a | mask_b) == 0) return;     // wave-uniform; always the case on stego frames
    const uint32_t n_a = (uint32_t)__popcll(mask_a), total = n_a + (uint32_t)__popcll(mask_b);
    const uint32_t rank_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_a, 0u));
    const uint32_t rank_b = n_a + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_b, 0u));
    for (uint32_t base = 0; base < total; base += (uint32_t)CAP) {   // wave-uniform
        const bool mine_a = tie_a && rank_a >= base && rank_a < base + (uint32_t)CAP;
        const bool mine_b = TWO && tie_b && rank_b >= base && rank_b < base + (uint32_t)CAP;
        if (mine_a) {
            GuardEntry *e = &entries[rank_a - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e->px[2 * r] = ax[r]; e->px[2 * r + 1] = ay[r]; }
            e->hi = 0; e->lo = 0;
        }
        if (mine_b) {
            GuardEntry *e = &entries[rank_b - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e->px[2 * r] = bx[r]; e->px[2 * r + 1] = by[r]; }
            e->hi = 0; e->lo = 0;
        }
        wave_lds_fence();
        const uint32_t todo = min(total - base, (uint32_t)CAP);
        for (uint32_t first = 0; first < todo; first += 8u) {
            const uint32_t idx = first + (lane >> 3);
            if (idx < todo) extract_replay8<QM>(&entries[idx], tile + (lane >> 3) * SVS_GUARD_TILE, lane & 7u, n, qp);
        }
        wave_lds_fence();
        if (mine_a) { hi_a = entries[rank_a - base].hi; lo_a = entries[rank_a - base].lo; }
        if (mine_b) { hi_b = entries[rank_b - base].hi; lo_b = entries[rank_b - base].lo; }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------
// EXTRACT: one lane = BPL adjacent blocks; a wavefront's 64*BPL blocks produce exactly n*BPL
// aligned 64-bit words of the packed stream (stream bit = global block * n + i), assembled through
// a wave-private LDS byte array and written with plain dword stores - no atomics, no pre-zeroed
// output.  HBM traffic per block: 64 B read + n bits written.
// ---------------------------------------------------------------------------------------
#define SVS_EXTRACT_CAP 16   // worklist entries per wave and round of the extract kernels
template <int U, int QM, int BPL, int NFIX = 0>
__global__ __launch_bounds__(SVS_WG) void extract_kernel(const uint8_t *__restrict__ gray, const Geometry g,
                                                      const QimParams qp, uint8_t *__restrict__ out,
                                                      const uint64_t out_bytes) {
    __shared__ uint32_t flags[SVS_WG / 64][SVS_WAVE_BITS_DWORDS(BPL)];
    __shared__ GuardEntry entries[SVS_WG / 64][SVS_EXTRACT_CAP];
    __shared__ float tiles[SVS_WG / 64][8 * SVS_GUARD_TILE];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = tile_id(g.xcd_chunk);
    const uint32_t gblock = (tile * (uint32_t)SVS_WG + threadIdx.x) * BPL;
    const uint32_t n = g.n_ac;

    // each block's bits, MSB first: bit i at position 63-i of hi:lo
    uint32_t hi_a = 0, lo_a = 0, hi_b = 0, lo_b = 0;
    uint32_t ax[8], ay[8], bx[8], by[8];
    bool tie_a = false, tie_b = false;
    float off_a = 0.0f, off_b = 0.0f;
    if (gblock < g.total_blocks) {
        typename RowVec<BPL>::type v[8];
        load_rows<BPL>(gray + block_offset(gblock, g), g.row_pitch, v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { ax[r] = v[r].x; ay[r] = v[r].y; }
        tie_a = extract_block_cheap<U, QM, NFIX>(ax, ay, n, qp, hi_a, lo_a, off_a);      // -> candidate
        if constexpr (BPL == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { bx[r] = v[r].z; by[r] = v[r].w; }
            tie_b = extract_block_cheap<U, QM, NFIX>(bx, by, n, qp, hi_b, lo_b, off_b);
        }
    }
    // Step two only in waves with a candidate (svs_block.hpp): pocketfft's own flat index 4 and the per-block tie margin.
    // Stego frames at delta >= 8 have none: one ballot.  (Round 3 computed both for every block: +3..7 % on stego frames.)
    if (__ballot(tie_a || tie_b) != 0) {
        if (gblock < g.total_blocks) {
            tie_a = extract_block_settle<QM>(ax, ay, n, qp, hi_a, off_a);
            if constexpr (BPL == 2) tie_b = extract_block_settle<QM>(bx, by, n, qp, hi_b, off_b);
        }
    }
    // A quantiser input within the per-block error bound of a rounding tie (svs_block.hpp, SVS_TIE2_*): those blocks get the
    // pocketfft-identical transform from eight lanes each (wave-private worklist).  Never taken on stego frames at delta >= 8.
    extract_phase2<QM, BPL == 2, SVS_EXTRACT_CAP>(&entries[wave][0], &tiles[wave][0], lane, n, qp, tie_a, ax, ay, hi_a, lo_a,
                                                  tie_b, bx, by, hi_b, lo_b);
    emit_wave_bits<U, BPL>(&flags[wave][0], lane, ((uint64_t)tile * (uint32_t)SVS_WG + wave * 64u) * BPL, n, hi_a, lo_a,
                           hi_b, lo_b, out, out_bytes);
}

#if defined(SVS_EXPERIMENTS)   // experiments library only (make variants -> lib/variants/libsvsdct_exp.so)
// ---------------------------------------------------------------------------------------
// Layout experiment (SVS_EXTRACT_SHUFFLE=1, one coefficient row, FAST arithmetic): the north-star's sketch taken
// literally - tiles staged in LDS, 8 lanes per block (one pixel row each), the vertical pass as cross-lane (DPP)
// butterflies, one coefficient per lane.  Kept for the A/B in profiles/history/r01_ab_layout.txt; the shipped kernels keep a
// whole block in one lane.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_add(float x, int ctrl_is) {  // x + x from the partner lane
    // ctrl_is: 0 = xor 1 (quad_perm 1,0,3,2), 1 = xor 2 (quad_perm 2,3,0,1), 2 = mirror inside the 8-lane group
    const int xi = __builtin_bit_cast(int, x);
    int yi;
    if (ctrl_is == 0) yi = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, false);
    else if (ctrl_is == 1) yi = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, false);
    else yi = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, false);
    return x + __builtin_bit_cast(float, yi);
}

template <int QM>
__global__ __launch_bounds__(SVS_WG) void extract_shuffle_kernel(const uint8_t *__restrict__ gray, const Geometry g,
                                                               const QimParams qp, uint8_t *__restrict__ out,
                                                               const uint64_t out_bytes) {
    __shared__ __attribute__((aligned(16))) u32x2 tiles[SVS_WG / 64][8][64];
    __shared__ uint32_t words[SVS_WG / 64][SVS_WAVE_BITS_DWORDS(1)];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = tile_id(g.xcd_chunk);
    const uint32_t gblock = tile * (uint32_t)SVS_WG + threadIdx.x, wave_first = gblock - lane;
    const uint32_t n = g.n_ac;
    if (gblock < g.total_blocks) {  // stage the wave's 64 blocks: coalesced 512-byte rows, as in the shipped kernels
        typename RowVec<1>::type v[8];
        load_rows<1>(gray + block_offset(gblock, g), g.row_pitch, v);
#pragma unroll
        for (int r = 0; r < 8; ++r) tiles[wave][r][lane] = v[r];
    }
    uint32_t *mine = &words[wave][0];
    for (uint32_t w = lane; w < 2u * n + 4u; w += 64u) mine[w] = 0u;
    wave_lds_fence();
    const uint32_t r = lane & 7u;
    // this lane's row of the orthonormal DCT-II matrix: k_r(x) = a(r) cos((2x+1) r pi/16), times a(0) of the vertical pass
    float kr[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
        kr[x] = (r == 0 ? 0.35355339059327373f : 0.5f) * __cosf((float)((2 * x + 1) * (int)r) * 0.19634954084936207f) *
                0.35355339059327373f;
#pragma unroll 1
    for (uint32_t p = 0; p < 8; ++p) {
        const uint32_t blk = 8u * p + (lane >> 3);
        const u32x2 v = tiles[wave][r][blk];
        float s[8] = {ubyte_to_float<0>(v.x), ubyte_to_float<1>(v.x), ubyte_to_float<2>(v.x), ubyte_to_float<3>(v.x),
                      ubyte_to_float<0>(v.y), ubyte_to_float<1>(v.y), ubyte_to_float<2>(v.y), ubyte_to_float<3>(v.y)};
#pragma unroll
        for (int x = 0; x < 8; ++x) s[x] = dpp_add(dpp_add(dpp_add(s[x], 0), 1), 2);  // column sums, in every lane
        float c = s[0] * kr[0];
#pragma unroll
        for (int x = 1; x < 8; ++x) c = fmaf(s[x], kr[x], c);
        // parity of every lane's coefficient as a wave mask (bit 8b + r), regrouped into the pass's 8n stream bits
        // (block-major, coefficient 1 first) by a second ballot, then ORed into the wave's big-endian words by lane 0
        const bool odd = r >= 1 && r <= n && wave_first + blk < g.total_blocks && ((uint32_t)quant_index<QM>(c, qp) & 1u);
        const uint64_t m = __ballot(odd);
        const uint32_t src = 8u * (lane / n) + (lane % n) + 1u;  // lane i < 8n picks block i / n, coefficient i % n + 1
        const uint64_t packed = __ballot(lane < 8u * n && ((m >> (src & 63u)) & 1ull));  // stream bit i of the pass = bit i
        if (lane == 0) {
            const uint64_t be = __builtin_bitreverse64(packed);  // stream bit 0 -> bit 63
            const uint32_t at = 8u * p * n, d = at >> 5, o = at & 31u;
            const uint32_t hi = (uint32_t)(be >> 32), lo = (uint32_t)be;
            atomicOr(&mine[d], hi >> o);
            atomicOr(&mine[d + 1], __builtin_amdgcn_alignbit(hi, lo, o));
            atomicOr(&mine[d + 2], __builtin_amdgcn_alignbit(lo, 0u, o));
        }
    }
    wave_lds_fence();
    const uint64_t wave_byte0 = ((uint64_t)tile * (uint32_t)SVS_WG + wave * 64u) * n / 8u;
    for (uint32_t w = lane; w < 2u * n; w += 64u) {
        const uint32_t word = __builtin_bswap32(mine[w]);
        const uint64_t at = wave_byte0 + 4ull * w;
        if (at + 4 <= out_bytes) {
            *reinterpret_cast<uint32_t *>(out + at) = word;
        } else {
            for (uint32_t j = 0; j < 4 && at + j < out_bytes; ++j) out[at + j] = (uint8_t)(word >> (8 * j));
        }
    }
}

#endif  // SVS_EXPERIMENTS

// ---------------------------------------------------------------------------------------
// EXACT-mode kernels (pocketfft-identical arithmetic, svs_block.hpp "EXACT mode"): one block per lane.
// The embed kernel transforms all 64 coefficients both ways (about 2 700 VALU instructions per block),
// so it is VALU-bound at roughly 40 % of the fast kernel's rate; it exists for bit-identical output.
// ---------------------------------------------------------------------------------------
#ifndef SVS_EXACT_MIN_WAVES
#define SVS_EXACT_MIN_WAVES 2  // waves per SIMD the exact embed kernel is register-allocated for (2: +1..3 % over 3; 4 spills 52 B and is 8 % slower)
#endif
template <int QM, int U = 8>  // U: coefficient rows the quantiser loop covers (flat indices 1..n lie in rows < U)
__global__ __launch_bounds__(SVS_WG, SVS_EXACT_MIN_WAVES) void embed_exact_kernel(const uint8_t *gray,  // may alias stego
                                                          uint8_t *stego, const Geometry g,
                                                          const QimParams qp,
                                                          const uint32_t *__restrict__ bits,
                                                          const uint64_t bit_offset, const uint64_t n_bits,
                                                          const uint32_t n_words) {
    const uint32_t gblock = tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x;
    if (gblock >= g.total_blocks) return;
    const int64_t off = block_offset(gblock, g);
    typename RowVec<1>::type v[8];
    load_rows<1>(gray + off, g.row_pitch, v);
    const uint32_t n = g.n_ac;  // 0 = round-trip every block without touching a coefficient
    const uint64_t first = (uint64_t)gblock * n;
    if (first >= n_bits) {
        if (stego != gray) store_rows<1>(stego + off, g.row_pitch, v);
        return;
    }
    uint32_t ax[8], ay[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { ax[r] = v[r].x; ay[r] = v[r].y; }
    uint32_t hi, lo;
    payload_window(bits, n_words, bit_offset + first, hi, lo);
    embed_block_exact<U, QM>(ax, ay, n, block_budget(first, n_bits, n), hi, lo, qp);
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r].x = ax[r]; v[r].y = ay[r]; }
    store_rows<1>(stego + off, g.row_pitch, v);
}

// EXACT embed, two adjacent blocks per lane (16-byte row accesses; needs an even number of blocks per row and 16-byte
// aligned rows - the host checks): the transforms run as packed-FP32 instructions over the pair (svs_block.hpp,
// embed_block_exact_pair).  Only the lane in which the payload budget ends can have its second block not entered; that
// block is then copied, not round-tripped (:130,:132).
#ifndef SVS_EXACT2_MIN_WAVES
#define SVS_EXACT2_MIN_WAVES 2
#endif
template <int QM, int U>
__global__ __launch_bounds__(SVS_WG, SVS_EXACT2_MIN_WAVES) void embed_exact_pair_kernel(const uint8_t *gray,  // may alias stego
                                                          uint8_t *stego, const Geometry g, const QimParams qp,
                                                          const uint32_t *__restrict__ bits, const uint64_t bit_offset,
                                                          const uint64_t n_bits, const uint32_t n_words) {
    const uint32_t gblock = (tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x) * 2u;
    if (gblock >= g.total_blocks) return;
    const int64_t off = block_offset(gblock, g);
    typename RowVec<2>::type v[8];
    load_rows<2>(gray + off, g.row_pitch, v);
    const uint32_t n = g.n_ac;  // 0 = round-trip every block without touching a coefficient
    const uint64_t first = (uint64_t)gblock * n;
    if (first >= n_bits) {
        if (stego != gray) store_rows<2>(stego + off, g.row_pitch, v);
        return;
    }
    const bool b_entered = first + n < n_bits || n == 0;   // n == 0: every block is entered (n_bits = 1 then)
    uint32_t ax[8], ay[8], bx[8], by[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { ax[r] = v[r].x; ay[r] = v[r].y; bx[r] = v[r].z; by[r] = v[r].w; }
    uint32_t hi_a, lo_a, hi_b, lo_b;
    payload_window(bits, n_words, bit_offset + first, hi_a, lo_a);
    payload_window(bits, n_words, bit_offset + first + n, hi_b, lo_b);
    embed_block_exact_pair<U, QM>(ax, ay, bx, by, n, block_budget(first, n_bits, n), block_budget(first + n, n_bits, n), hi_a,
                                  lo_a, hi_b, lo_b, qp);
    if (b_entered) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].x = ax[r]; v[r].y = ay[r]; v[r].z = bx[r]; v[r].w = by[r]; }
        store_rows<2>(stego + off, g.row_pitch, v);
    } else {  // the one lane the budget ends in: A is stored, B keeps (or gets a copy of) its original bytes
        typename RowVec<1>::type h[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { h[r].x = ax[r]; h[r].y = ay[r]; }
        store_rows<1>(stego + off, g.row_pitch, h);
        if (stego != gray) {
            load_rows<1>(gray + off + 8, g.row_pitch, h);
            store_rows<1>(stego + off + 8, g.row_pitch, h);
        }
    }
}

// ---------------------------------------------------------------------------------------
// EMBED, streaming kernel (launched for two coefficient rows, n = 8..15, since round 6 - one row: embed_row1_kernel below; flags 0 and SVS_EXACT_GUARDED, include/svsdct.h): one lane = BPL adjacent blocks, grid = ceil(total_blocks / (SVS_WG*BPL))
// workgroups of SVS_WG.  BPL = 2 (16-byte row accesses) needs an even number of blocks per row and 16-byte aligned rows
// (the host checks) and is instantiated for one coefficient row only.
// HBM traffic per block: 64 B read + 64 B written + n payload bits read - nothing else, whatever the content.
//   phase 1  lane = block: the cheap arithmetic (svs_block.hpp) - embed_block_guarded (n <= 7) / _guarded2 (n = 8..15):
//            pocketfft-identical payload coefficients, sparse inverse, rigorous per-block error bound - the result is the
//            reference's, bit for bit.  Returns "undecided" for the blocks whose truncation the reference's own float32
//            noise decides.  (n >= 16 does not come here: every mode runs embed_exact_kernel.)
//   phase 2  the wave's undecided blocks are compacted into a wave-private LDS worklist (ballot + mbcnt: no atomics, no
//            barrier) and redone with the pocketfft-identical arithmetic by EIGHT LANES PER BLOCK: lane r of a group
//            transforms column r, then row r, of its block - the four 1-D passes of the reference (vertical / horizontal
//            forward, vertical / horizontal inverse, config_and_setup.py:135,168) with a transposition through a
//            wave-private LDS tile in between.  Every value comes from the same svs::pf operation sequence as in
//            embed_block_exact, so the bits are the same; what changes is the shape: about 40 VGPRs and 300-450
//            instructions per pass of 8 blocks instead of 140 VGPRs and 2 100 per pass of 64 - affordable inside the
//            streaming kernel.  More than SVS_GUARD_CAP undecided blocks in a wave (flat content) take further rounds.
//            With two coefficient rows the worklist is shared by the workgroup (guard_phase2_wg).
//   phase 3  every lane stores its rows (its own result, or the one it collected from the worklist): the wave's stores
//            cover whole 512-byte row segments - no partial lines, no second launch, no scratch buffer in HBM.
// `gray` and `stego` may be the same buffer (in-place embedding, include/svsdct.h), so neither is __restrict__: every
// lane loads its own rows before it stores them and touches nobody else's.
// ---------------------------------------------------------------------------------------
#ifndef SVS_GUARD_CAP
#define SVS_GUARD_CAP 32    // worklist entries per wave and round (80 B each) of the one-row (rigorous guard) embed kernel
#endif

// px: the block's 16 row dwords (low, high per row) in LDS - original pixels in, exact stego pixels out
// UROWS: coefficient rows that can hold payload (n <= 8 UROWS - 1); the quantiser loop covers only those
template <int QM, int UROWS = 8>
__device__ __forceinline__ void guard_replay8(uint32_t *px, uint32_t hi, uint32_t lo, uint32_t nb, float *t, uint32_t r, uint32_t n,
                                              const QimParams &qp) {
    float a[8], b[8];
    // vertical forward transform of pixel column r
    {
        const uint32_t sh = 8u * (r & 3u), half = r >> 2;
#pragma unroll
        for (int y = 0; y < 8; ++y) a[y] = (float)((px[2 * y + half] >> sh) & 0xffu);
    }
    pf::dct2_8(a, b);                       // b[u] = V[u][r]
#pragma unroll
    for (int u = 0; u < 8; ++u) t[9 * u + r] = b[u];
    wave_lds_fence();
#pragma unroll
    for (int x = 0; x < 8; ++x) a[x] = t[9 * r + x];   // V[r][x]
    wave_lds_fence();
    pf::dct2_8(a, b);                       // b[v] = D[r][v]: coefficient row r
    // to coefficient COLUMNS (what the vertical inverse wants): lane r takes D[u][r], u = 0..7
#pragma unroll
    for (int v = 0; v < 8; ++v) t[9 * r + v] = b[v];
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = t[9 * u + r];
    wave_lds_fence();
    // QIM on flat indices k = 8 u + r in 1..n (config_and_setup.py:139-158): one coefficient per lane and row, so a wave
    // runs n / 8 + 1 quantiser sequences, each on all the lanes that have a coefficient
#pragma unroll
    for (int u = 0; u < UROWS; ++u) {
        const uint32_t k = 8u * u + r;
        if (k >= 1u && k <= n) {
            const int i = (int)k - 1;
            const int bit = (int)window_bit(hi, lo, i);
            const float c = a[u];
            int q = quant_index<QM>(c, qp);
            q = force_parity(q, bit);
            float cn;
            if constexpr (QM == QM_DOUBLE) cn = (float)((double)q * qp.delta_d);
            else cn = (float)q * qp.delta_f;
            a[u] = ((uint32_t)i < nb) ? cn : c;
        }
    }
    pf::dct3_8(a, b);                       // vertical inverse first (axis 0, :168): b[y] = P[y][r]
#pragma unroll
    for (int y = 0; y < 8; ++y) t[9 * y + r] = b[y];
    wave_lds_fence();
#pragma unroll
    for (int v = 0; v < 8; ++v) a[v] = t[9 * r + v];   // P[r][v]
    pf::dct3_8(a, b);                       // pixel row r
    uint32_t lo4, hi4;
    store_row_trunc(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], lo4, hi4);   // np.uint8(np.clip(.)) (:171)
    px[2 * r] = lo4;
    px[2 * r + 1] = hi4;
}

// phase 1 for one block whose rows are ax/ay, in place: stego pixels out - unless the block is undecided, then its original
// pixels are left untouched (svs_block.hpp decides before it writes).  -> undecided
template <int U, int QM, int NFIX = 0>
__device__ __forceinline__ bool guard_phase1(uint32_t (&ax)[8], uint32_t (&ay)[8], uint32_t n, uint64_t first,
                                             const QimParams &qp, const uint32_t *__restrict__ bits,
                                             uint64_t bit_offset, uint64_t n_bits, uint32_t n_words,
                                             uint32_t *keep_hi = nullptr) {
    const uint32_t hi = window32(payload_qword(bits, n_words, bit_offset + first), (uint32_t)((bit_offset + first) & 31u)), lo = 0;
    if (keep_hi) *keep_hi = hi;
    const uint32_t nb = block_budget(first, n_bits, n);
    static_assert(U <= 2, "n <= 15: with more coefficient rows every mode runs the lane-per-block pocketfft kernel");
    if constexpr (U == 1) return embed_block_guarded<QM>(ax, ay, n, nb, hi, lo, qp);          // n <= 7: rigorous, 8 tests
    else return embed_block_guarded2<QM, NFIX>(ax, ay, n, nb, hi, lo, qp);                    // n = 8..15: rigorous, 64 tests
}
// the same with the window handed in (two blocks per lane: both windows come from one payload_qword)
template <int U, int QM, int NFIX = 0>
__device__ __forceinline__ bool guard_phase1_window(uint32_t (&ax)[8], uint32_t (&ay)[8], uint32_t n, uint64_t first,
                                                    const QimParams &qp, uint64_t n_bits, uint32_t hi) {
    static_assert(U <= 2, "n <= 15");
    const uint32_t nb = block_budget(first, n_bits, n);
    if constexpr (U == 1) return embed_block_guarded<QM>(ax, ay, n, nb, hi, 0u, qp);
    else return embed_block_guarded2<QM, NFIX>(ax, ay, n, nb, hi, 0u, qp);
}

// what phase 2 needs to rebuild a block's payload window (kept out of the lanes' registers on the common path)
struct GuardPayload {
    const uint32_t *bits;
    uint64_t bit_offset, n_bits;
    uint32_t n_words;
};

// phase 2: the wave's undecided blocks (a: first block of every lane, b: second block when two blocks per lane; their rows
// still hold the ORIGINAL pixels, first_a / first_b are their first stream bits) through the worklist `entries` (CAP entries)
// and the transposition tile `tile` (8 * SVS_GUARD_TILE floats), both private to the wave.  On return the rows of undecided
// blocks hold the exact stego pixels.  Returns the number of blocks redone.
template <int QM, bool TWO, int CAP = SVS_GUARD_CAP, bool KEPT = false, int UROWS = 2>
__device__ __forceinline__ uint32_t guard_phase2(GuardEntry *entries, float *tile, uint32_t lane, uint32_t n,
                                                 const QimParams &qp, const GuardPayload &pl,
                                                 bool und_a, uint64_t first_a, uint32_t (&ax)[8], uint32_t (&ay)[8],
                                                 bool und_b, uint64_t first_b, uint32_t (&bx)[8], uint32_t (&by)[8],
                                                 uint32_t hi_a = 0, uint32_t hi_b = 0) {
    const uint64_t mask_a = __ballot(und_a);
    const uint64_t mask_b = TWO ? __ballot(und_b) : 0ull;
    if ((mask_a | mask_b) == 0) return 0;   // wave-uniform: the common case costs two ballots
    const uint32_t n_a = (uint32_t)__popcll(mask_a), total = n_a + (uint32_t)__popcll(mask_b);
    const uint32_t rank_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_a, 0u));
    const uint32_t rank_b = n_a + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_b, 0u));
    for (uint32_t base = 0; base < total; base += (uint32_t)CAP) {   // wave-uniform
        const bool mine_a = und_a && rank_a >= base && rank_a < base + (uint32_t)CAP;
        const bool mine_b = TWO && und_b && rank_b >= base && rank_b < base + (uint32_t)CAP;
        if (mine_a) {
            GuardEntry *e = &entries[rank_a - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e->px[2 * r] = ax[r]; e->px[2 * r + 1] = ay[r]; }
            uint32_t hi, lo;
            if constexpr (KEPT) { hi = hi_a; lo = 0; }   // n <= 15: the window's first word, still in a register from phase 1
            else payload_window(pl.bits, pl.n_words, pl.bit_offset + first_a, hi, lo);
            e->hi = hi; e->lo = lo; e->nb = block_budget(first_a, pl.n_bits, n);
        }
        if (mine_b) {
            GuardEntry *e = &entries[rank_b - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e->px[2 * r] = bx[r]; e->px[2 * r + 1] = by[r]; }
            uint32_t hi, lo;
            if constexpr (KEPT) { hi = hi_b; lo = 0; }
            else payload_window(pl.bits, pl.n_words, pl.bit_offset + first_b, hi, lo);
            e->hi = hi; e->lo = lo; e->nb = block_budget(first_b, pl.n_bits, n);
        }
        wave_lds_fence();
        const uint32_t todo = min(total - base, (uint32_t)CAP);
        for (uint32_t first = 0; first < todo; first += 8u) {
            const uint32_t idx = first + (lane >> 3);
            if (idx < todo) {
                GuardEntry *e = &entries[idx];
                guard_replay8<QM, UROWS>(e->px, e->hi, e->lo, e->nb, tile + (lane >> 3) * SVS_GUARD_TILE, lane & 7u, n, qp);
            }
        }
        wave_lds_fence();
        if (mine_a) {
            const GuardEntry *e = &entries[rank_a - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { ax[r] = e->px[2 * r]; ay[r] = e->px[2 * r + 1]; }
        }
        if (mine_b) {
            const GuardEntry *e = &entries[rank_b - base];
#pragma unroll
            for (int r = 0; r < 8; ++r) { bx[r] = e->px[2 * r]; by[r] = e->px[2 * r + 1]; }
        }
        wave_lds_fence();   // the next round overwrites the entries
    }
    return total;
}

// phase 2 over PARKED rows (round 4; the two-row kernel): every lane has written its block's original rows to its slot of a
// wave-private LDS array before phase 1 (slot = lane, SVS_SLOT_DWORDS apart), so phase 1 works in place and an undecided block
// needs no deposit: the worklist is a list of {slot, budget, payload window} words, the exact replay reads and writes the
// slots, and the owners of undecided blocks read theirs back.  No rounds: the worklist holds all 64 lanes if it must.
#define SVS_SLOT_DWORDS 18   // 16 row dwords + 2: 8-byte aligned, and 16 consecutive lanes hit 16 different even banks
template <int QM>
__device__ __forceinline__ uint32_t guard_phase2_slots(uint32_t *slots, u32x2 *meta, float *tile, uint32_t lane, uint32_t n,
                                                       const QimParams &qp, bool und, uint32_t nb, uint32_t hi,
                                                       uint32_t (&ax)[8], uint32_t (&ay)[8]) {
    const uint64_t mask = __ballot(und);
    if (mask == 0) return 0;   // wave-uniform
    const uint32_t total = (uint32_t)__popcll(mask);
    if (und) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        u32x2 m; m.x = lane | (nb << 8); m.y = hi;
        meta[rank] = m;
    }
    wave_lds_fence();
    for (uint32_t at = 0; at < total; at += 8u) {   // wave-uniform
        const uint32_t idx = at + (lane >> 3);
        if (idx < total) {
            const u32x2 m = meta[idx];
            guard_replay8<QM, 2>(slots + (m.x & 0xffu) * SVS_SLOT_DWORDS, m.y, 0u, m.x >> 8, tile + (lane >> 3) * SVS_GUARD_TILE, lane & 7u, n, qp);
        }
    }
    wave_lds_fence();
    if (und) {
        const u32x2 *mine = reinterpret_cast<const u32x2 *>(slots + lane * SVS_SLOT_DWORDS);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const u32x2 v = mine[r]; ax[r] = v.x; ay[r] = v.y; }
    }
    return total;
}

#ifndef SVS_U2_WGPOOL
#define SVS_U2_WGPOOL 0      // two rows: 1 = worklist shared by the workgroup (guard_phase2_wg, A/B build only): measured SLOWER (2.74 vs
                             // 2.65 ms per 600 x 4K, profiles/r04_ab_two_row.txt): the three barriers cost more than the passes saved
#endif
#if SVS_U2_WGPOOL
// phase 2 at WORKGROUP scope (round 4; the two-row kernel, where 5-13 % of the blocks are undecided and the kernel is bound
// by vector issue, not by HBM): the undecided blocks of all four waves share ONE worklist, and its passes of eight blocks
// are dealt round-robin to the waves.  A wave-private worklist runs ceil(k / 8) passes for its own k blocks - 0.97 passes
// per wave at 3 blocks (natural-like content, 5 %), 1.5 at 8.3 (noise, 13 %) - the shared one ceil(K / 8) for the workgroup's
// K: 0.5 and 1.15 per wave.  Price: one LDS atomic per wave and three workgroup barriers (the wave-private form has none),
// which is why the one-row kernel - HBM-bound, replay hidden - keeps the wave-private form.
// `counter` must have been zeroed (and a barrier passed) before the first wave gets here.  Every thread of the workgroup
// must call this (barriers inside).  Returns the workgroup's number of redone blocks.
template <int QM, int CAPWG, bool KEPT>
__device__ __forceinline__ uint32_t guard_phase2_wg(GuardEntry *entries, float *tile, uint32_t *counter, uint32_t lane, uint32_t wave,
                                                    uint32_t n, const QimParams &qp, const GuardPayload &pl, bool und,
                                                    uint64_t first, uint32_t (&ax)[8], uint32_t (&ay)[8], uint32_t hi_kept) {
    const uint64_t mask = __ballot(und);
    uint32_t base = 0;
    if (mask != 0) {   // wave-uniform
        if (lane == 0) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    __syncthreads();                       // every wave's count is in
    const uint32_t total = *counter;       // uniform over the workgroup; nobody writes it again
    if (total == 0) return 0;
    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    for (uint32_t round0 = 0; round0 < total; round0 += (uint32_t)CAPWG) {   // uniform over the workgroup
        const bool mine = und && slot >= round0 && slot < round0 + (uint32_t)CAPWG;
        if (mine) {
            GuardEntry *e = &entries[slot - round0];
#pragma unroll
            for (int r = 0; r < 8; ++r) { e->px[2 * r] = ax[r]; e->px[2 * r + 1] = ay[r]; }
            uint32_t hi, lo;
            if constexpr (KEPT) { hi = hi_kept; lo = 0; }
            else payload_window(pl.bits, pl.n_words, pl.bit_offset + first, hi, lo);
            e->hi = hi; e->lo = lo; e->nb = block_budget(first, pl.n_bits, n);
        }
        __syncthreads();
        const uint32_t todo = min(total - round0, (uint32_t)CAPWG);
        for (uint32_t at = 8u * wave; at < todo; at += 8u * (SVS_WG / 64)) {   // this wave's passes
            const uint32_t idx = at + (lane >> 3);
            if (idx < todo) {
                GuardEntry *e = &entries[idx];
                guard_replay8<QM, 2>(e->px, e->hi, e->lo, e->nb, tile + (lane >> 3) * SVS_GUARD_TILE, lane & 7u, n, qp);
            }
        }
        __syncthreads();
        if (mine) {
            const GuardEntry *e = &entries[slot - round0];
#pragma unroll
            for (int r = 0; r < 8; ++r) { ax[r] = e->px[2 * r]; ay[r] = e->px[2 * r + 1]; }
        }
        if (round0 + (uint32_t)CAPWG < total) __syncthreads();   // the next round overwrites the entries
    }
    return total;
}

#endif  // SVS_U2_WGPOOL

// Register targets (waves per SIMD) of the embed kernels: natural allocation.  (One row: 96 VGPRs, 5 waves; spills in the
// replay cost 1.90 vs 1.71 ms.  Two rows: 103 VGPRs, 4 waves; a target of 5 waves spills 68 B and costs 3.26 vs 2.74 ms
// per 600 x 4K, 6 waves 4.30 ms - profiles/r04_ab_two_row.txt.)
#ifndef SVS_KEEP_WINDOW
#define SVS_KEEP_WINDOW 1
#endif
#ifndef SVS_GUARD_CAP_WG
#define SVS_GUARD_CAP_WG 64  // entries of the shared worklist per round (noise content: 33 undecided blocks per workgroup)
#endif
#ifndef SVS_U2_MIN_WAVES
#define SVS_U2_MIN_WAVES 1   // natural allocation (about 100 VGPRs, 4 waves per SIMD)
#endif
template <int U>
constexpr int kEmbedMinWaves = U == 2 ? SVS_U2_MIN_WAVES : 1;
template <int BPL>
__device__ __forceinline__ uint32_t row2_shadow(uint32_t gblock, const Geometry &g, bool &live) {
    live = gblock < g.total_blocks;
    return live ? gblock : g.total_blocks - (uint32_t)BPL;   // the host launches with total_blocks >= BPL (a multiple of BPL)
}
// two rows, one block per lane: original rows parked in LDS and phase 1 in place (guard_phase2_slots: 68 instead of 103
// VGPRs, 5 instead of 4 waves per SIMD, 29.7 KB of LDS per workgroup).  The kernel is bound by vector issue either way, and
// which form the compiler schedules better depends on the quantiser: general delta (QM_F32, the GUI's default 20) 2.62 vs
// 2.78 ms per 600 x 4K at n = 10 and 2.82 vs 2.93 at n = 15 in favour of the parked form, power-of-two delta 2.74 vs 2.65
// against it - both orders of a 15-round A/B (profiles/r04_ab_two_row.txt).  1 = parked except for power-of-two delta,
// 2 = always, 0 = never.
#ifndef SVS_U2_INPLACE
#define SVS_U2_INPLACE 1
#endif
// The experiments library (-DSVS_EXPERIMENTS) counts the blocks a launch redid exactly into a device counter (measurement
// hook svs_guard_counter_set); the product kernels have no such parameter and the product library no such state.
#if defined(SVS_EXPERIMENTS)
#define SVS_REPLAY_COUNTER_PARAM , unsigned long long *__restrict__ replay_counter
#else
#define SVS_REPLAY_COUNTER_PARAM
#endif
template <int U, int QM, int BPL, int NFIX = 0>   // NFIX: compile-time n (two rows only; svs_capi.hip instantiates the GUI's default 10)
__global__ __launch_bounds__(SVS_WG, kEmbedMinWaves<U>) void embed_kernel(const uint8_t *gray,
                                                    uint8_t *stego, const Geometry g, const QimParams qp,
                                                    const uint32_t *__restrict__ bits, const uint64_t bit_offset,
                                                    const uint64_t n_bits, const uint32_t n_words SVS_REPLAY_COUNTER_PARAM) {
    static_assert(U == 2, "n = 8..15 (svs_capi.hip: one coefficient row runs embed_row1_kernel, more rows embed_exact_kernel in every mode)");
    constexpr bool WGPOOL = U == 2 && BPL == 1 && SVS_U2_WGPOOL;
    constexpr bool PARKED = U == 2 && BPL == 1 && !WGPOOL && (SVS_U2_INPLACE == 2 || (SVS_U2_INPLACE == 1 && QM != QM_POW2));
    constexpr int CAP = WGPOOL ? SVS_GUARD_CAP_WG : SVS_GUARD_CAP;
    // PARKED: per wave 64 slots of original rows + the worklist words; otherwise the worklist entries carry the rows
    __shared__ __attribute__((aligned(16))) uint32_t park[PARKED ? SVS_WG / 64 : 1][PARKED ? 64 * SVS_SLOT_DWORDS : 2];
    __shared__ u32x2 meta[PARKED ? SVS_WG / 64 : 1][PARKED ? 64 : 1];
    __shared__ GuardEntry entries[PARKED ? 1 : (WGPOOL ? 1 : SVS_WG / 64)][PARKED ? 1 : CAP];
    __shared__ float tiles[SVS_WG / 64][8 * SVS_GUARD_TILE];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#if SVS_U2_WGPOOL
    __shared__ uint32_t wg_undecided;
    if constexpr (WGPOOL) {
        if (threadIdx.x == 0) wg_undecided = 0u;
        __syncthreads();   // at the very start: no wave has work in flight yet
    }
#endif
    // (Round 5 tried block-row aligned tiles - a workgroup reads and writes ONE contiguous stretch, eight full pixel rows, at
    // the price of idle lanes - to bring the launch from the rate of a copy with this access pattern to that of a linear copy:
    // 1.73 vs 1.64 ms per 600 x 4K at n = 3, 3.15 vs 2.63 at n = 10, slower on every placement: profiles/r05_ab_row_tiles.txt.)
    // lanes past the end of the batch shadow its last block(s) - they load and compute like everybody else and never store - so
    // that the row registers are defined on one path only (round 6: no zero-initialised copies at the joins)
    bool live;
    const uint32_t gblock = row2_shadow<BPL>((tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x) * BPL, g, live);
    const uint32_t n = g.n_ac;
    bool und_a = false, und_b = false, write = false;
    // n <= 15: the payload window of a block is its first word - kept in a register from phase 1, because re-reading it for
    // the worklist is a global load in the life of every wave that replays
    constexpr bool KEPT = SVS_KEEP_WINDOW;
    uint32_t hi_a = 0, hi_b = 0, nb_a = 0;
    typename RowVec<BPL>::type v[8];
    uint32_t ax[8], ay[8], bx[8], by[8];
    const int64_t off = block_offset(gblock, g);
    load_rows<BPL>(gray + off, g.row_pitch, v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        ax[r] = v[r].x; ay[r] = v[r].y;
        if constexpr (BPL == 2) { bx[r] = v[r].z; by[r] = v[r].w; }
    }
    if (live) {
        const uint64_t first = (uint64_t)gblock * n;  // stream index of this lane's first bit
        write = stego != gray;                         // past the budget: byte-identical copy (the reference's loops `break`, :130,:132)
        if (first < n_bits) {
            write = true;
            if constexpr (BPL == 2) {
                // both blocks' windows from the two dwords at the lane's first stream bit (2 n <= 14 bits at one row)
                const uint64_t q = payload_qword(bits, n_words, bit_offset + first);
                const uint32_t sh = (uint32_t)((bit_offset + first) & 31u);
                hi_a = window32(q, sh);
                hi_b = window32(q, sh + n);
                und_a = guard_phase1_window<U, QM, NFIX>(ax, ay, n, first, qp, n_bits, hi_a);
                if constexpr (U == 2) SVS_SCHED_FENCE();   // one block at a time: interleaved, the two working sets need 150 VGPRs
                if (first + n < n_bits)   // a budget of 0 (only the lane the payload ends in can see it) leaves block B as it is
                    und_b = guard_phase1_window<U, QM, NFIX>(bx, by, n, first + n, qp, n_bits, hi_b);
            } else if constexpr (PARKED) {
                // park the original rows (the exact replay reads them there), then phase 1 in place
                u32x2 *slot = reinterpret_cast<u32x2 *>(&park[wave][lane * SVS_SLOT_DWORDS]);
#pragma unroll
                for (int r = 0; r < 8; ++r) slot[r] = v[r];
                hi_a = window32(payload_qword(bits, n_words, bit_offset + first), (uint32_t)((bit_offset + first) & 31u));
                nb_a = block_budget(first, n_bits, n);
                und_a = embed_block_guarded2<QM, NFIX, true>(ax, ay, n, nb_a, hi_a, 0u, qp);
            } else {
                und_a = guard_phase1<U, QM, NFIX>(ax, ay, n, first, qp, bits, bit_offset, n_bits, n_words, KEPT ? &hi_a : nullptr);
            }
        }
    }
    const GuardPayload pl{bits, bit_offset, n_bits, n_words};
    const uint64_t first_a = (uint64_t)gblock * n;
    uint32_t redone;
    if constexpr (PARKED) {
        redone = guard_phase2_slots<QM>(&park[wave][0], &meta[wave][0], &tiles[wave][0], lane, n, qp, und_a, nb_a, hi_a, ax, ay);
#if SVS_U2_WGPOOL
    } else if constexpr (WGPOOL) {
        redone = guard_phase2_wg<QM, CAP, KEPT>(&entries[0][0], &tiles[wave][0], &wg_undecided, lane, wave, n, qp, pl, und_a, first_a,
                                                ax, ay, hi_a);
        if (wave != 0) redone = 0;   // counted once per workgroup
#endif
    } else {
        redone = guard_phase2<QM, BPL == 2, CAP, KEPT>(&entries[wave][0], &tiles[wave][0], lane, n, qp, pl, und_a, first_a, ax, ay,
                                                       und_b, first_a + n, bx, by, hi_a, hi_b);
    }
#if defined(SVS_EXPERIMENTS)
    if (replay_counter != nullptr && redone != 0 && lane == 0) atomicAdd(replay_counter, (unsigned long long)redone);
#else
    (void)redone;
#endif
    if (write) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r].x = ax[r]; v[r].y = ay[r];
            if constexpr (BPL == 2) { v[r].z = bx[r]; v[r].w = by[r]; }
        }
        store_rows<BPL>(stego + off, g.row_pitch, v);
    }
}

// ---------------------------------------------------------------------------------------
// EMBED, one coefficient row (n <= 7), round 6: the streaming kernel above with the rows kept in ONE representation from
// the load to the store and the cheap result applied in the integer domain (svs_block.hpp, guard_decide_int).
//   * every lane loads (lanes past the end of the batch shadow its last block(s) and never store), so the row registers
//     are defined on one path only: no copies at the joins (round 5: 173 static v_mov_b32, 61 of them on every wave's path);
//   * a decided block costs 16 v_add_u32 instead of 192 conversion / add / saturating-conversion instructions; a WAVE in
//     which some block could clip at 0 / 255 (a wave-uniform ballot) takes a packed 16-bit saturating add for all its
//     lanes - same bytes, 10 instructions per row dword;
//   * the worklist deposit / collection moves rows as the 8-byte pairs they are held in.
// Phases 2 and 3 as in embed_kernel.  gray and stego may alias.
// ---------------------------------------------------------------------------------------
// Rows stay in the load / store vectors: block A = components x, y of v[r], block B (two blocks per lane) = z, w.
template <int BPL>
__device__ __forceinline__ void row1_block(const typename RowVec<BPL>::type (&v)[8], int which, uint32_t (&rx)[8], uint32_t (&ry)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if constexpr (BPL == 2) { rx[r] = which ? v[r].z : v[r].x; ry[r] = which ? v[r].w : v[r].y; }
        else { rx[r] = v[r].x; ry[r] = v[r].y; }
    }
}
// worklist deposit / collection of one block's rows as the 8-byte halves of the row vectors they live in (volatile: the
// load / store vectoriser would otherwise pair rows into 16-byte LDS accesses and gather their operands with v_mov)
typedef __attribute__((address_space(3))) volatile u32x2 lds_v64;
template <int BPL>
__device__ __forceinline__ void row1_deposit(GuardEntry *e, const typename RowVec<BPL>::type (&v)[8], int which) {
    lds_v64 *px = (lds_v64 *)(e->px);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        u32x2 t;
        if constexpr (BPL == 2) { if (which) { t.x = v[r].z; t.y = v[r].w; } else { t.x = v[r].x; t.y = v[r].y; } }
        else { t.x = v[r].x; t.y = v[r].y; }
        px[r] = t;
    }
}
template <int BPL>
__device__ __forceinline__ void row1_collect(const GuardEntry *e, typename RowVec<BPL>::type (&v)[8], int which) {
    const lds_v64 *px = (const lds_v64 *)(e->px);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u32x2 t = px[r];
        if constexpr (BPL == 2) { if (which) { v[r].z = t.x; v[r].w = t.y; } else { v[r].x = t.x; v[r].y = t.y; } }
        else { v[r].x = t.x; v[r].y = t.y; }
    }
}

// phase 2 of the one-row kernel: guard_phase2 on the row vectors (the payload windows are kept from phase 1)
template <int QM, int BPL, int CAP>
__device__ __forceinline__ uint32_t row1_phase2(GuardEntry *entries, float *tile, uint32_t lane, uint32_t n, const QimParams &qp,
                                                uint64_t n_bits, bool und_a, bool und_b, uint64_t first_a, uint32_t hi_a, uint32_t hi_b,
                                                typename RowVec<BPL>::type (&v)[8]) {
    const uint64_t mask_a = __ballot(und_a);
    const uint64_t mask_b = BPL == 2 ? __ballot(und_b) : 0ull;
    if ((mask_a | mask_b) == 0) return 0;   // wave-uniform: the common case costs two ballots
    const uint32_t n_a = (uint32_t)__popcll(mask_a), total = n_a + (uint32_t)__popcll(mask_b);
    const uint32_t rank_a = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_a, 0u));
    const uint32_t rank_b = n_a + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask_b, 0u));
    for (uint32_t base = 0; base < total; base += (uint32_t)CAP) {   // wave-uniform
        const bool mine_a = und_a && rank_a >= base && rank_a < base + (uint32_t)CAP;
        const bool mine_b = BPL == 2 && und_b && rank_b >= base && rank_b < base + (uint32_t)CAP;
        if (mine_a) {
            GuardEntry *e = &entries[rank_a - base];
            row1_deposit<BPL>(e, v, 0);
            e->hi = hi_a; e->lo = 0u; e->nb = block_budget(first_a, n_bits, n);
        }
        if (mine_b) {
            GuardEntry *e = &entries[rank_b - base];
            row1_deposit<BPL>(e, v, 1);
            e->hi = hi_b; e->lo = 0u; e->nb = block_budget(first_a + n, n_bits, n);
        }
        wave_lds_fence();
        const uint32_t todo = min(total - base, (uint32_t)CAP);
        for (uint32_t first = 0; first < todo; first += 8u) {
            const uint32_t idx = first + (lane >> 3);
            if (idx < todo) {
                GuardEntry *e = &entries[idx];
                guard_replay8<QM, 1>(e->px, e->hi, e->lo, e->nb, tile + (lane >> 3) * SVS_GUARD_TILE, lane & 7u, n, qp);
            }
        }
        wave_lds_fence();
        if (mine_a) row1_collect<BPL>(&entries[rank_a - base], v, 0);
        if (mine_b) row1_collect<BPL>(&entries[rank_b - base], v, 1);
        wave_lds_fence();   // the next round overwrites the entries
    }
    return total;
}

// what the lanes of a launch share (kernel arguments, in SGPRs)
struct Row1Args {
    const uint8_t *gray;
    uint8_t *stego;
    const uint32_t *bits;
    uint64_t bit_offset, n_bits;
    uint32_t n_words;
};

// lanes past the end of the batch shadow its last block(s): they load and compute like everybody else, and never store
template <int BPL>
__device__ __forceinline__ uint32_t row1_shadow(uint32_t gblock, const Geometry &g, bool &live) {
    live = gblock < g.total_blocks;
    return live ? gblock : g.total_blocks - (uint32_t)BPL;   // the host launches with total_blocks >= BPL (a multiple of BPL)
}

// phases 1-3 for the rows `v` of global block(s) gb (already loaded from gray + off): decide / apply, exact replay of the
// wave's undecided blocks, store.
template <int QM, int BPL>
__device__ __forceinline__ uint32_t row1_process(typename RowVec<BPL>::type (&v)[8], uint32_t gb, bool live, int64_t off, uint64_t q,
                                                 const Geometry &g, const QimParams &qp, const Row1Args &a, GuardEntry *entries,
                                                 float *tile, uint32_t lane) {
    const uint32_t n = g.n_ac;
    const uint64_t first = (uint64_t)gb * n;   // stream index of this lane's first bit
    bool und_a = false, und_b = false, clip = false;
    uint32_t hi_a = 0, hi_b = 0;
    ColumnDeltas ca = {0u, 0u, 0u, 0u}, cb = {0u, 0u, 0u, 0u};
    if (live && first < a.n_bits) {
        const uint32_t sh = (uint32_t)((a.bit_offset + first) & 31u);
        hi_a = window32(q, sh);
        uint32_t flags;
        {
            uint32_t rx[8], ry[8];
            row1_block<BPL>(v, 0, rx, ry);
            flags = guard_decide_int<QM>(rx, ry, n, block_budget(first, a.n_bits, n), hi_a, qp, ca);
            und_a = (flags & SVS_ROW1_UNDECIDED) != 0;
            if (und_a) { ca.e_lo = 0u; ca.o_lo = 0u; ca.e_hi = 0u; ca.o_hi = 0u; }   // keeps its original pixels for the replay
        }
        clip = flags == SVS_ROW1_MAY_CLIP;
        if constexpr (BPL == 2) {
            SVS_SCHED_FENCE();   // one block at a time
            hi_b = window32(q, sh + n);
            uint32_t rx[8], ry[8];
            row1_block<BPL>(v, 1, rx, ry);
            // a budget of 0 (only the lane the payload ends in can see it) leaves block B as it is: all its deltas are 0
            flags = guard_decide_int<QM>(rx, ry, n, block_budget(first + n, a.n_bits, n), hi_b, qp, cb);
            und_b = (flags & SVS_ROW1_UNDECIDED) != 0;
            if (und_b) { cb.e_lo = 0u; cb.o_lo = 0u; cb.e_hi = 0u; cb.o_hi = 0u; }
            clip = clip || flags == SVS_ROW1_MAY_CLIP;
        }
    }
    SVS_SCHED_FENCE();
    // The stores (every lane: the deltas of a lane without payload are 0).  Wave-uniform: when some block of the wave could
    // clip at 0 / 255, all of them take the saturating form - same bytes where nothing clips.
    const bool clip_wave = __ballot(clip) != 0;
    {
        const uint32_t keep = clip_wave ? 0u : 0xffffffffu;
        const uint32_t a0 = packed_addend(ca.e_lo, ca.o_lo) & keep, a1 = packed_addend(ca.e_hi, ca.o_hi) & keep;
        const uint32_t b0 = packed_addend(cb.e_lo, cb.o_lo) & keep, b1 = packed_addend(cb.e_hi, cb.o_hi) & keep;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r].x += a0; v[r].y += a1;
            if constexpr (BPL == 2) { v[r].z += b0; v[r].w += b1; }
        }
    }
    if (clip_wave) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r].x = add_clip_dword(v[r].x, ca.e_lo, ca.o_lo);
            v[r].y = add_clip_dword(v[r].y, ca.e_hi, ca.o_hi);
            if constexpr (BPL == 2) {
                v[r].z = add_clip_dword(v[r].z, cb.e_lo, cb.o_lo);
                v[r].w = add_clip_dword(v[r].w, cb.e_hi, cb.o_hi);
            }
            SVS_SCHED_FENCE();   // row by row: scheduled for latency, this rare path would set the kernel's register count
        }
    }
    const uint32_t redone = row1_phase2<QM, BPL, SVS_GUARD_CAP>(entries, tile, lane, n, qp, a.n_bits, und_a, und_b, first, hi_a, hi_b, v);
    // past the budget: byte-identical copy (the reference's loops `break`, :130,:132)
    if (live && (a.stego != a.gray || first < a.n_bits)) store_rows<BPL>(a.stego + off, g.row_pitch, v);
    return redone;
}

template <int QM, int BPL>
__global__ __launch_bounds__(SVS_WG) void embed_row1_kernel(const uint8_t *gray, uint8_t *stego, const Geometry g,
                                                          const QimParams qp, const uint32_t *__restrict__ bits,
                                                          const uint64_t bit_offset, const uint64_t n_bits,
                                                          const uint32_t n_words SVS_REPLAY_COUNTER_PARAM) {
    __shared__ GuardEntry entries[SVS_WG / 64][SVS_GUARD_CAP];
    __shared__ float tiles[SVS_WG / 64][8 * SVS_GUARD_TILE];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool live;
    const uint32_t gb = row1_shadow<BPL>((tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x) * BPL, g, live);
    const int64_t off = block_offset(gb, g);
    typename RowVec<BPL>::type v[8];
    load_rows<BPL>(gray + off, g.row_pitch, v);
    const Row1Args a{gray, stego, bits, bit_offset, n_bits, n_words};
    const uint32_t n = g.n_ac;
    uint64_t q = 0;
    if (live && (uint64_t)gb * n < n_bits) q = payload_qword(bits, n_words, bit_offset + (uint64_t)gb * n);
    const uint32_t redone = row1_process<QM, BPL>(v, gb, live, off, q, g, qp, a, &entries[wave][0], &tiles[wave][0], lane);
#if defined(SVS_EXPERIMENTS)
    if (replay_counter != nullptr && redone != 0 && lane == 0) atomicAdd(replay_counter, (unsigned long long)redone);
#else
    (void)redone;
#endif
}

// (Round 6 also ran this kernel as a PERSISTENT, software-pipelined loop - few workgroups per CU, each loading the rows of its
// next tile before it computes on the current one, so that the bytes in flight stay low and constant: correct, and slower on
// every placement, 1.77 - 1.95 ms per 600 x 4K against 1.58 - 1.62, even with the arithmetic skipped: profiles/r06_stream_pipeline.txt.
// What the one-shot launch has and the loop has not is the hardware's own pacing: a workgroup starts when another one ends.)

template <int U, int QM, int BPL = 1>
__global__ __launch_bounds__(SVS_WG) void extract_exact_kernel(const uint8_t *__restrict__ gray, const Geometry g,
                                                            const QimParams qp, uint8_t *__restrict__ out,
                                                            const uint64_t out_bytes) {
    __shared__ uint32_t flags[SVS_WG / 64][SVS_WAVE_BITS_DWORDS(BPL)];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = tile_id(g.xcd_chunk);
    const uint32_t gblock = (tile * (uint32_t)SVS_WG + threadIdx.x) * BPL;
    const uint32_t n = g.n_ac;
    uint32_t hi_a = 0, lo_a = 0, hi_b = 0, lo_b = 0;
    if (gblock < g.total_blocks) {
        typename RowVec<BPL>::type v[8];
        load_rows<BPL>(gray + block_offset(gblock, g), g.row_pitch, v);
        uint32_t ax[8], ay[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { ax[r] = v[r].x; ay[r] = v[r].y; }
        extract_block_exact<U, QM>(ax, ay, n, qp, hi_a, lo_a);
        if constexpr (BPL == 2) {   // two adjacent blocks per lane: 16-byte row loads (even block count per row, host-checked)
#pragma unroll
            for (int r = 0; r < 8; ++r) { ax[r] = v[r].z; ay[r] = v[r].w; }
            extract_block_exact<U, QM>(ax, ay, n, qp, hi_b, lo_b);
        }
    }
    emit_wave_bits<U, BPL>(&flags[wave][0], lane, ((uint64_t)tile * (uint32_t)SVS_WG + wave * 64u) * BPL, n, hi_a, lo_a, hi_b, lo_b,
                           out, out_bytes);
}

// ---------------------------------------------------------------------------------------
// measurement helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

// 8 pixels per thread per step; same hash as svsdct/synth.py
__global__ __launch_bounds__(256) void fill_synthetic_kernel(uint8_t *__restrict__ frames, int32_t n_frames,
                                                             int32_t height, int32_t width, int64_t row_pitch,
                                                             int64_t frame_pitch, uint32_t seed,
                                                             uint32_t first_frame, uint32_t lo, uint32_t span) {
    const uint32_t groups_per_row = (uint32_t)width / 8u;
    const uint64_t total = (uint64_t)n_frames * height * groups_per_row;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t xg = (uint32_t)(t % groups_per_row);
        const uint64_t r = t / groups_per_row;
        const uint32_t y = (uint32_t)(r % (uint32_t)height), f = (uint32_t)(r / (uint32_t)height);
        const uint32_t base = seed + (f + first_frame) * 0x9E3779B1u + y * 0x85EBCA6Bu;
        uint32_t px[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = xg * 8u + j;
            const uint32_t v = lo + lowbias32(base + x * 0xC2B2AE35u) % span;
            px[j >> 2] |= v << (8 * (j & 3));
        }
        u32x2 r8;
        r8.x = px[0];
        r8.y = px[1];
        *reinterpret_cast<u32x2 *>(frames + (int64_t)f * frame_pitch + (int64_t)y * row_pitch + xg * 8u) = r8;
    }
}

__global__ __launch_bounds__(256) void fill_bits_kernel(uint32_t *__restrict__ words, uint64_t n_words,
                                                        uint64_t n_bits, uint32_t seed, uint64_t first_bit) {
    const uint32_t s = seed * 0x632BE5ABu;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t be = 0;  // big-endian view: stream bit 32w+j at bit 31-j
        for (uint32_t j = 0; j < 32; ++j) {
            const uint64_t i = 32ull * w + j;
            if (i < n_bits) be |= (lowbias32(s + (uint32_t)(first_bit + i)) >> 31) << (31 - j);
        }
        words[w] = __builtin_bswap32(be);
    }
}

// ---------------------------------------------------------------------------------------
// The reference operator's payload types are Python strings of '0' / '1' characters (bit_payload_segment in, the joined
// string out: config_and_setup.py:106-109,124-126,173-174).  The host-pointer entry points svs_embed_str / svs_extract_str
// take / return exactly that; the conversion to and from the packed stream runs here, on the device, at the price of moving
// one byte per bit over the link (1.3 MB for a 4K frame at n = 10: 25 us) instead of three host passes over it.
// 32 characters -> one packed dword: the bit is the low bit of the character ('0' = 0x30, '1' = 0x31); four of them times
// 0x08040201 leave the nibble c0 c1 c2 c3 in bits 27..24 of the product (no carries: every cross term lands below bit 19).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ascii_to_packed_kernel(const uint8_t *__restrict__ ascii, uint64_t n_chars,
                                                              uint32_t *__restrict__ words, uint64_t n_words) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t base = 32ull * w;
        uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (base + 32 <= n_chars) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(ascii + base), b = *reinterpret_cast<const u32x4 *>(ascii + base + 16);
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        } else {
            for (uint32_t j = 0; j < 32 && base + j < n_chars; ++j) d[j >> 2] |= (uint32_t)ascii[base + j] << (8 * (j & 3));
        }
        uint32_t be = 0;   // stream bit 32 w + j at bit 31 - j
#pragma unroll
        for (int j = 0; j < 8; ++j) be |= ((((d[j] & 0x01010101u) * 0x08040201u) >> 24) & 0xfu) << (28 - 4 * j);
        words[w] = __builtin_bswap32(be);
    }
}

// one packed byte -> its eight characters, MSB first
__global__ __launch_bounds__(256) void packed_to_ascii_kernel(const uint8_t *__restrict__ packed, uint64_t n_bytes,
                                                              u32x2 *__restrict__ ascii8) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = packed[i];
        u32x2 o;
        o.x = 0x30303030u | ((v >> 7) & 1u) | (((v >> 6) & 1u) << 8) | (((v >> 5) & 1u) << 16) | (((v >> 4) & 1u) << 24);
        o.y = 0x30303030u | ((v >> 3) & 1u) | (((v >> 2) & 1u) << 8) | (((v >> 1) & 1u) << 16) | ((v & 1u) << 24);
        ascii8[i] = o;
    }
}

// per-frame sum of squared differences; blockIdx.y = frame, 8 pixels per thread per step,
// partial sums reduced in-wave, one 64-bit atomic per wave
__global__ __launch_bounds__(256) void frame_sse_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                        int32_t height, int32_t width, int64_t row_pitch,
                                                        int64_t frame_pitch, unsigned long long *__restrict__ sse) {
    const uint32_t f = blockIdx.y;
    const uint32_t groups_per_row = (uint32_t)width / 8u;
    const uint64_t total = (uint64_t)height * groups_per_row;
    unsigned long long acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t xg = (uint32_t)(t % groups_per_row), y = (uint32_t)(t / groups_per_row);
        const int64_t off = (int64_t)f * frame_pitch + (int64_t)y * row_pitch + xg * 8u;
        const u32x2 va = *reinterpret_cast<const u32x2 *>(a + off);
        const u32x2 vb = *reinterpret_cast<const u32x2 *>(b + off);
        const uint32_t wa[2] = {va.x, va.y}, wb[2] = {vb.x, vb.y};
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = (int)((wa[j >> 2] >> (8 * (j & 3))) & 0xff) - (int)((wb[j >> 2] >> (8 * (j & 3))) & 0xff);
            s += (uint32_t)(d * d);
        }
        acc += s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(&sse[f], acc);
}

__global__ __launch_bounds__(256) void bit_errors_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                         uint64_t n_bits, unsigned long long *__restrict__ count) {
    const uint64_t n_bytes = (n_bits + 7) / 8;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)(a[i] ^ b[i]);
        if (i == n_bytes - 1 && (n_bits & 7u)) x &= 0xFFu << (8 - (n_bits & 7u));
        acc += __popc(x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(count, acc);
}

// ---------------------------------------------------------------------------------------
// Colour plumbing around the operator: interleaved 8-bit BGR -> gray
// (cv2.COLOR_BGR2GRAY, config_and_setup.py:112) and gray -> BGR (cv2.COLOR_GRAY2BGR, embed_process.py:126).
// OpenCV's 8-bit BGR2GRAY is fixed point: (B*wb + G*wg + R*wr + 2^(shift-1)) >> shift; the weights are passed
// in because they differ between OpenCV generations (15-bit 3735/19235/9798 today, 14-bit 1868/9617/4899 in
// older builds) and cv2 is not available to pin either.  4 pixels per thread: 12 bytes in, one dword out.
// ---------------------------------------------------------------------------------------
// (kernels: bgr_to_gray_kernel / gray_to_bgr_kernel below, after the wave-cooperative row helpers they share with the
// fused colour path)
// ---------------------------------------------------------------------------------------
// Fused colour path ("fused read of 3 B/px"): the frames that carry payload go
// BGR -> gray -> embed -> BGR in ONE pass - 3 B/pixel read, 3 B/pixel written (+1 for the optional gray
// reference the operator returns) instead of the 10 B/pixel of convert / embed / convert.  One lane = one
// block = 8 rows x 24 bytes; stego pixels are written as B = G = R = gray (cv2.COLOR_GRAY2BGR).  Every block
// of these frames is converted; blocks past the payload budget carry their gray value unchanged.
// ---------------------------------------------------------------------------------------
struct ColourParams {
    int64_t in_row_pitch, in_frame_pitch;    // BGR input
    int64_t out_row_pitch, out_frame_pitch;  // BGR output
    uint32_t wb, wg, wr, shift;              // (B*wb + G*wg + R*wr + 2^(shift-1)) >> shift
};

__device__ __forceinline__ int64_t block_offset_bgr(uint32_t gblock, const Geometry &g, int64_t row_pitch,
                                                    int64_t frame_pitch) {
    const uint32_t frame = fast_div(gblock, g.by_bpf);
    const uint32_t in_frame = gblock - frame * g.by_bpf.div;
    const uint32_t brow = fast_div(in_frame, g.by_wb);
    const uint32_t bcol = in_frame - brow * g.by_wb.div;
    return (int64_t)frame * frame_pitch + (int64_t)(brow * 8u) * row_pitch + (int64_t)(bcol * 24u);
}

// One block row of interleaved BGR = 24 bytes at an 8-byte aligned address: three 8-byte accesses.  (A 16-byte +
// an 8-byte access is no faster for loads and 1.7x SLOWER for stores - the 16-byte half is misaligned half the time.)
__device__ __forceinline__ void load_bgr_row(const uint8_t *p, u32x2 &q0, u32x2 &q1, u32x2 &q2) {
    const u32x2 *row = reinterpret_cast<const u32x2 *>(p);
    q0 = SVS_LD(row); q1 = SVS_LD(row + 1); q2 = SVS_LD(row + 2);
}
__device__ __forceinline__ void store_bgr_row(uint8_t *p, const u32x2 &q0, const u32x2 &q1, const u32x2 &q2) {
    u32x2 *row = reinterpret_cast<u32x2 *>(p);
    SVS_ST(q0, row); SVS_ST(q1, row + 1); SVS_ST(q2, row + 2);
}

// 8 interleaved BGR pixels (6 dwords) -> 8 gray bytes (2 dwords)
__device__ __forceinline__ void bgr8_to_gray(const u32x2 &q0, const u32x2 &q1, const u32x2 &q2, const ColourParams &c,
                                             uint32_t &lo4, uint32_t &hi4) {
    const uint32_t w[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
    const uint32_t half = 1u << (c.shift - 1);
    uint32_t px[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b0 = 3 * j, b1 = 3 * j + 1, b2 = 3 * j + 2;
        const uint32_t B = (w[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu, G = (w[b1 >> 2] >> (8 * (b1 & 3))) & 0xffu,
                       R = (w[b2 >> 2] >> (8 * (b2 & 3))) & 0xffu;
        // B, G, R < 2^8 and the weights <= 2^16: 24-bit multiplies (full rate) are exact
        px[j] = (__umul24(B, c.wb) + __umul24(G, c.wg) + __umul24(R, c.wr) + half) >> c.shift;
    }
    lo4 = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    hi4 = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
}

// 8 gray bytes -> 8 interleaved BGR pixels with B = G = R: six byte permutes (v_perm_b32 selects bytes 0-3 from
// its second operand, 4-7 from its first)
__device__ __forceinline__ void gray8_to_bgr(uint32_t lo4, uint32_t hi4, u32x2 &q0, u32x2 &q1, u32x2 &q2) {
    q0.x = __builtin_amdgcn_perm(0u, lo4, 0x01000000u);  // a a a b
    q0.y = __builtin_amdgcn_perm(0u, lo4, 0x02020101u);  // b b c c
    q1.x = __builtin_amdgcn_perm(0u, lo4, 0x03030302u);  // c d d d
    q1.y = __builtin_amdgcn_perm(0u, hi4, 0x01000000u);
    q2.x = __builtin_amdgcn_perm(0u, hi4, 0x02020101u);
    q2.y = __builtin_amdgcn_perm(0u, hi4, 0x03030302u);
}

// A wave's 64 blocks are 64 x 24 = 1536 consecutive bytes per pixel row (except where the run of blocks wraps to the
// next block row / frame): seen as 192 units of 8 bytes, unit u = bytes [8*(u%3), +8) of the row of block u/3.
struct WaveUnits {
    uint32_t owner[3], part[3];  // of unit lane + 64 j
    bool live[3];
};
__device__ __forceinline__ WaveUnits wave_units(uint32_t lane, uint32_t wave_first, uint32_t total_blocks) {
    WaveUnits w;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t u = lane + 64u * j;
        w.owner[j] = (u * 171u) >> 9;  // u / 3 for u < 192
        w.part[j] = u - 3u * w.owner[j];
        w.live[j] = wave_first + w.owner[j] < total_blocks;
    }
    return w;
}

// Cooperative load of the wave's BGR rows (SVS_BGR_DIRECT_LOAD disables it): every load instruction covers 512
// contiguous bytes; the rows pass through a wave-private, double-buffered LDS row (2 x 192 units) from which each lane
// picks its own 24 bytes and converts them to gray.  +6 % in extract_bgr_kernel over lanes loading their own rows at a
// 24-byte stride (profiles/history/r01_aux_kernel_rates.txt).
__device__ __forceinline__ void wave_load_gray(const uint8_t *__restrict__ bgr, const Geometry &g, const ColourParams &c,
                                               const WaveUnits &wu, uint32_t wave_first, uint32_t lane, u32x2 *rowbuf,
                                               uint32_t (&ax)[8], uint32_t (&ay)[8]) {
    u32x2 raw[8][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint8_t *src = bgr + block_offset_bgr(wave_first + wu.owner[j], g, c.in_row_pitch, c.in_frame_pitch) +
                             8u * wu.part[j];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            u32x2 v; v.x = 0u; v.y = 0u;
            if (wu.live[j]) v = SVS_LD(reinterpret_cast<const u32x2 *>(src + r * c.in_row_pitch));
            raw[r][j] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        u32x2 *buf = rowbuf + (r & 1) * 192;
#pragma unroll
        for (int j = 0; j < 3; ++j) buf[lane + 64 * j] = raw[r][j];
        wave_lds_fence();  // also separates the reads of row r-1 from the writes of row r+1 into the same buffer
        bgr8_to_gray(buf[3 * lane], buf[3 * lane + 1], buf[3 * lane + 2], c, ax[r], ay[r]);
    }
    wave_lds_fence();
}

// Same, HALF rows at a time (two rounds of 4 rows): half the registers in flight, for kernels that are short of them
__device__ __forceinline__ void wave_load_gray_halves(const uint8_t *bgr, const Geometry &g,
                                                      const ColourParams &c, uint32_t wave_first, uint32_t lane,
                                                      u32x2 *rowbuf, uint32_t (&ax)[8], uint32_t (&ay)[8]) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const WaveUnits wu = wave_units(lane, wave_first, g.total_blocks);
        u32x2 raw[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint8_t *src = bgr + block_offset_bgr(wave_first + wu.owner[j], g, c.in_row_pitch, c.in_frame_pitch) +
                                 8u * wu.part[j] + (int64_t)(4 * half) * c.in_row_pitch;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u32x2 v; v.x = 0u; v.y = 0u;
                if (wu.live[j]) v = SVS_LD(reinterpret_cast<const u32x2 *>(src + r * c.in_row_pitch));
                raw[r][j] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u32x2 *buf = rowbuf + (r & 1) * 192;
#pragma unroll
            for (int j = 0; j < 3; ++j) buf[lane + 64 * j] = raw[r][j];
            wave_lds_fence();
            uint32_t gx, gy;
            bgr8_to_gray(buf[3 * lane], buf[3 * lane + 1], buf[3 * lane + 2], c, gx, gy);
            if (half == 0) { ax[r] = gx; ay[r] = gy; } else { ax[4 + r] = gx; ay[4 + r] = gy; }
        }
        wave_lds_fence();
    }
}

// Gray rows of a wave's 64 blocks -> interleaved BGR with B = G = R, through the wave-private tile `mine` (8 rows x 64
// lanes of 8 gray bytes): every store instruction covers 512 contiguous bytes.
__device__ __forceinline__ void wave_store_gray_as_bgr(u32x2 *mine, uint32_t lane, uint32_t gblock, bool live,
                                                       const uint32_t (&ax)[8], const uint32_t (&ay)[8],
                                                       uint8_t *bgr_out, const Geometry &g,
                                                       const ColourParams &c) {
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { u32x2 v; v.x = ax[r]; v.y = ay[r]; mine[r * 64 + lane] = v; }
    }
    wave_lds_fence();  // wave-private tile: LDS operations of one wave execute in order; this pins the compiler's order
    const uint32_t wave_first = gblock - lane;
    const WaveUnits wu = wave_units(lane, wave_first, g.total_blocks);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t owner = wu.owner[j], part = wu.part[j];
        if (!wu.live[j]) continue;
        // gray pixels feeding the unit's two dwords (v_perm_b32: selector bytes 0-3 pick from the low gray dword,
        // 4-7 from the high one): part 0 = p0 p0 p0 p1 | p1 p1 p2 p2, part 1 = p2 p3 p3 p3 | p4 p4 p4 p5,
        // part 2 = p5 p5 p6 p6 | p6 p7 p7 p7
        const uint32_t sel0 = part == 0 ? 0x01000000u : part == 1 ? 0x03030302u : 0x06060505u;
        const uint32_t sel1 = part == 0 ? 0x02020101u : part == 1 ? 0x05040404u : 0x07070706u;
        uint8_t *dst = bgr_out + block_offset_bgr(wave_first + owner, g, c.out_row_pitch, c.out_frame_pitch) + 8u * part;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const u32x2 v = mine[r * 64 + owner];
            u32x2 q;
            q.x = __builtin_amdgcn_perm(v.y, v.x, sel0);
            q.y = __builtin_amdgcn_perm(v.y, v.x, sel1);
            asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + r * c.out_row_pitch), "v"(q) : "memory");
        }
    }
}

// Stand-alone conversions (svs_bgr_to_gray_dev / svs_gray_to_bgr_dev), block-structured like the operator so that they
// share its coalesced row movers: 8x8 blocks, one lane per block.
__global__ __launch_bounds__(SVS_WG) void bgr_to_gray_kernel(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                             const Geometry g, const ColourParams c) {
    __shared__ __attribute__((aligned(16))) u32x2 rows[SVS_WG / 64][2 * 192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t gblock = tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x;
    const WaveUnits wu = wave_units(lane, gblock - lane, g.total_blocks);
    uint32_t ax[8], ay[8];
    wave_load_gray(bgr, g, c, wu, gblock - lane, lane, &rows[wave][0], ax, ay);
    if (gblock < g.total_blocks) {
        typename RowVec<1>::type v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].x = ax[r]; v[r].y = ay[r]; }
        store_rows<1>(gray + block_offset(gblock, g), g.row_pitch, v);
    }
}

__global__ __launch_bounds__(SVS_WG) void gray_to_bgr_kernel(const uint8_t *__restrict__ gray, uint8_t *__restrict__ bgr,
                                                             const Geometry g, const ColourParams c) {
    __shared__ __attribute__((aligned(16))) u32x2 tile[SVS_WG / 64][8][64];
    const uint32_t gblock = tile_id(g.xcd_chunk) * (uint32_t)SVS_WG + threadIdx.x;
    const bool live = gblock < g.total_blocks;
    uint32_t ax[8], ay[8];
    if (live) {
        typename RowVec<1>::type v[8];
        load_rows<1>(gray + block_offset(gblock, g), g.row_pitch, v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { ax[r] = v[r].x; ay[r] = v[r].y; }
    }
    wave_store_gray_as_bgr(&tile[threadIdx.x >> 6][0][0], threadIdx.x & 63u, gblock, live, ax, ay, bgr, g, c);
}

// Stego rows leave through a wave-private LDS tile (SVS_BGR_DIRECT_STORE disables it): each lane parks its 8 stego gray
// bytes per row, then the wave writes the BGR row as 192 consecutive 8-byte units - unit u = bytes [8*(u%3), +8) of the
// 24-byte row of the wave's block u/3 - so every store instruction covers 512 contiguous bytes instead of 8 bytes in
// every 24.
template <int U, int QM, bool EXACT>
__global__ __launch_bounds__(SVS_WG) void embed_bgr_kernel(const uint8_t *bgr_in,   // may alias bgr_out
                                                        uint8_t *bgr_out, uint8_t *__restrict__ gray_ref,
                                                        const Geometry g, const ColourParams c, const QimParams qp,
                                                        const uint32_t *__restrict__ bits, const uint64_t bit_offset,
                                                        const uint64_t n_bits, const uint32_t n_words) {
    // one wave-private 4 KB region per wave: row staging of the cooperative load, then (streaming arithmetic) the worklist and transposition
    // tile of the exact replay (svs::guard_phase2, 16 entries per round), then the stego tile of the cooperative store
    __shared__ __attribute__((aligned(16))) u32x2 lds_tile[SVS_WG / 64][8][64];
    static_assert(16 * sizeof(GuardEntry) + 8 * SVS_GUARD_TILE * sizeof(float) <= 8 * 64 * sizeof(u32x2), "wave region too small");
    const uint32_t tile = tile_id(g.xcd_chunk);
    const uint32_t gblock = tile * (uint32_t)SVS_WG + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool live = gblock < g.total_blocks;
    uint32_t ax[8], ay[8];
    // cooperative load, four rows at a time: +5..7 % over per-lane loads at a 24-byte stride with the streaming arithmetic (62 -> 78 VGPRs),
    // neutral in EXACT mode; all eight rows at once cost 100+ VGPRs and gained 1 % (profiles/history/r01_aux_kernel_rates.txt)
    wave_load_gray_halves(bgr_in, g, c, gblock - lane, lane, &lds_tile[wave][0][0], ax, ay);
    bool und = false;
    constexpr bool KEPT = !EXACT && U <= 2 && SVS_KEEP_WINDOW;   // see embed_kernel
    uint32_t hi_kept = 0;
    if (live) {
        if (gray_ref != nullptr) {  // the operator's first return value: the gray frame before embedding
            uint8_t *ref = gray_ref + block_offset(gblock, g);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u32x2 v; v.x = ax[r]; v.y = ay[r];
                SVS_ST(v, reinterpret_cast<u32x2 *>(ref + r * g.row_pitch));
            }
        }
        const uint32_t n = g.n_ac;
        const uint64_t first = (uint64_t)gblock * n;
        if (first < n_bits) {
            if constexpr (EXACT) {
                uint32_t hi, lo;
                payload_window(bits, n_words, bit_offset + first, hi, lo);
                embed_block_exact<8, QM>(ax, ay, n, block_budget(first, n_bits, n), hi, lo, qp);
            } else {
                und = guard_phase1<U, QM>(ax, ay, n, first, qp, bits, bit_offset, n_bits, n_words, KEPT ? &hi_kept : nullptr);
            }
        }
    }
    if constexpr (!EXACT) {   // undecided blocks: exact replay inside the wave (see embed_kernel)
        GuardEntry *entries = reinterpret_cast<GuardEntry *>(&lds_tile[wave][0][0]);
        float *t = reinterpret_cast<float *>(entries + 16);
        const GuardPayload pl{bits, bit_offset, n_bits, n_words};
        const uint64_t first = (uint64_t)gblock * g.n_ac;
        guard_phase2<QM, false, 16, KEPT>(entries, t, lane, g.n_ac, qp, pl, und, first, ax, ay, false, first, ax, ay, hi_kept, 0u);
    }
    wave_store_gray_as_bgr(&lds_tile[wave][0][0], lane, gblock, live, ax, ay, bgr_out, g, c);
}

// extract straight from interleaved BGR frames (gray computed on the fly).  One coefficient row: pocketfft-identical forward;
// two and more rows (round 4): the two-step FAST extraction of extract_kernel - FMA-factored forward, wave ballot of near-tie
// candidates, pocketfft coefficient 4 + per-block margin, 8-lane exact replay of tie blocks - instead of the pocketfft forward
// of every row for every block (0.88 ms per 200 x 4K BGR frames at n = 10 against an HBM floor of 0.75).
// FASTX = false keeps the pocketfft forward (delta below SVS_FAST_EXTRACT_DELTA_MIN).
template <int U, int QM, bool FASTX = (U >= 2)>
__global__ __launch_bounds__(SVS_WG) void extract_bgr_kernel(const uint8_t *__restrict__ bgr, const Geometry g,
                                                          const ColourParams c, const QimParams qp,
                                                          uint8_t *__restrict__ out, const uint64_t out_bytes) {
    __shared__ uint32_t flags[SVS_WG / 64][SVS_WAVE_BITS_DWORDS(1)];
    __shared__ __attribute__((aligned(16))) u32x2 rows[SVS_WG / 64][2 * 192];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = tile_id(g.xcd_chunk);
    const uint32_t gblock = tile * (uint32_t)SVS_WG + threadIdx.x;
    const uint32_t n = g.n_ac;
    uint32_t hi = 0, lo = 0;
    uint32_t ax[8], ay[8];
    const WaveUnits wu = wave_units(lane, gblock - lane, g.total_blocks);
    wave_load_gray(bgr, g, c, wu, gblock - lane, lane, &rows[wave][0], ax, ay);
    if constexpr (FASTX) {
        __shared__ GuardEntry entries[SVS_WG / 64][SVS_EXTRACT_CAP];
        __shared__ float tiles[SVS_WG / 64][8 * SVS_GUARD_TILE];
        bool tie = false;
        float off = 0.0f;
        if (gblock < g.total_blocks) tie = extract_block_cheap<U, QM>(ax, ay, n, qp, hi, lo, off);    // -> candidate
        if (__ballot(tie) != 0) {
            if (gblock < g.total_blocks) tie = extract_block_settle<QM>(ax, ay, n, qp, hi, off);
        }
        uint32_t hb = 0, lb = 0;
        extract_phase2<QM, false, SVS_EXTRACT_CAP>(&entries[wave][0], &tiles[wave][0], lane, n, qp, tie, ax, ay, hi, lo, false, ax, ay, hb, lb);
    } else {
        if (gblock < g.total_blocks) extract_block_exact<U, QM>(ax, ay, n, qp, hi, lo);
    }
    emit_wave_bits<U, 1>(&flags[wave][0], lane, (uint64_t)tile * (uint32_t)SVS_WG + wave * 64u, n, hi, lo, 0u, 0u, out,
                         out_bytes);
}

// ---------------------------------------------------------------------------------------
// SSIM evaluator: mean structural similarity of two gray frames as
// skimage.metrics.structural_similarity computes it with its defaults for 2-D uint8 input (what the
// reference's evaluation.calc_ssim calls, evaluation.py:21-26): 7x7 uniform window, K1 = 0.01, K2 = 0.03,
// sample covariance (NP/(NP-1)), float64 arithmetic, mean over the map cropped by 3 pixels per side.
// Window sums are exact integers here (skimage's running float sums differ from them by rounding only).
// One workgroup = 256 output columns x SSIM_BAND output rows; a thread walks down its column keeping the
// last 7 horizontal window sums of the five moments in registers.
// ---------------------------------------------------------------------------------------
#define SVS_SSIM_BAND 126  // output rows per workgroup (a multiple of 7 keeps the row groups aligned)
__global__ __launch_bounds__(256) void ssim_partial_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                           int32_t height, int32_t width, int64_t row_pitch,
                                                           int64_t frame_pitch, const double *__restrict__ data_range,
                                                           double *__restrict__ partial) {
    __shared__ __attribute__((aligned(8))) uint8_t ra[7][256 + 8], rb[7][256 + 8];  // 7 input rows at a time
    __shared__ double red[4];
    const int out_w = width - 6, out_h = height - 6;
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * SVS_SSIM_BAND, f = blockIdx.z;
    const int t = threadIdx.x;
    const uint8_t *pa = a + (int64_t)f * frame_pitch, *pb = b + (int64_t)f * frame_pitch;
    const double R = data_range[f];
    const double C1 = (0.01 * R) * (0.01 * R), C2 = (0.03 * R) * (0.03 * R);
    const double inv_np = 1.0 / 49.0, cov_norm = 49.0 / 48.0;
    // ring of the last 7 horizontal 7-sums of the five moments; slot = input row % 7, so with rows handled in groups
    // of 7 every slot index below is a compile-time constant
    uint32_t ha[7], hb[7], haa[7], hbb[7], hab[7];
    uint32_t va = 0, vb = 0, vaa = 0, vbb = 0, vab = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) ha[i] = hb[i] = haa[i] = hbb[i] = hab[i] = 0;
    double acc = 0.0;
    const int rows = min(SVS_SSIM_BAND, out_h - y0) + 6;  // input rows this band touches
    const bool col_ok = x0 + t < out_w;
    for (int r0 = 0; r0 < rows; r0 += 7) {
        __syncthreads();
        // stage 7 rows x 264 columns of both frames with 8-byte loads (x0 and the row pitch are multiples of 8, the
        // width is a multiple of 8, so a chunk is either wholly inside the frame or wholly outside)
        for (int i = t; i < 7 * 33; i += 256) {
            const int j = i / 33, c8 = i - j * 33;
            const int x = x0 + 8 * c8, y = y0 + r0 + j;
            u32x2 qa = {0u, 0u}, qb = {0u, 0u};
            if (x < width && y < height && r0 + j < rows) {
                qa = *reinterpret_cast<const u32x2 *>(pa + (int64_t)y * row_pitch + x);
                qb = *reinterpret_cast<const u32x2 *>(pb + (int64_t)y * row_pitch + x);
            }
            *reinterpret_cast<u32x2 *>(&ra[j][8 * c8]) = qa;
            *reinterpret_cast<u32x2 *>(&rb[j][8 * c8]) = qb;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            uint32_t sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
#pragma unroll
            for (int d = 0; d < 7; ++d) {
                const uint32_t u = ra[j][t + d], v = rb[j][t + d];
                sa += u; sb += v;
                saa += __umul24(u, u); sbb += __umul24(v, v); sab += __umul24(u, v);  // full-rate 24-bit multiplies
            }
            va += sa - ha[j]; vb += sb - hb[j]; vaa += saa - haa[j]; vbb += sbb - hbb[j]; vab += sab - hab[j];
            ha[j] = sa; hb[j] = sb; haa[j] = saa; hbb[j] = sbb; hab[j] = sab;
            const int r = r0 + j;
#if defined(SVS_SSIM_DIAG_NO_F64)
            if (r >= 6 && r < rows && col_ok) acc += (double)(va + vb + vaa + vbb + vab);
#else
            if (r >= 6 && r < rows && col_ok) {
                const double ux = va * inv_np, uy = vb * inv_np;
                const double vx = cov_norm * (vaa * inv_np - ux * ux), vy = cov_norm * (vbb * inv_np - uy * uy);
                const double vxy = cov_norm * (vab * inv_np - ux * uy);
                const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
                acc += (A1 * A2) / (B1 * B2);
            }
#endif
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0)
        partial[((int64_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// fixed-order sum of a frame's partials -> mean SSIM (deterministic: no atomics)
__global__ void ssim_finish_kernel(const double *__restrict__ partial, int32_t per_frame, double count,
                                   double *__restrict__ ssim) {
    const int f = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < per_frame; ++i) s += partial[(int64_t)f * per_frame + i];
        ssim[f] = s / count;
    }
}

// per-frame max - min of a plane (the data_range quirk of evaluation.calc_ssim, evaluation.py:26), in two steps:
// workgroups of SVS_RANGE_ROWS rows fold their bytes into lohi[frame] = {min, max} with integer atomics (exact and
// order-independent), then one thread per frame turns the pair into the double the SSIM kernel reads.
#define SVS_RANGE_ROWS 16
__global__ __launch_bounds__(256) void frame_minmax_kernel(const uint8_t *__restrict__ a, int32_t height, int32_t width,
                                                           int64_t row_pitch, int64_t frame_pitch,
                                                           uint32_t *__restrict__ lohi) {
    const int f = blockIdx.y, y0 = blockIdx.x * SVS_RANGE_ROWS;
    const int w8 = width / 8;  // width is a multiple of 8, rows are 8-byte aligned
    uint32_t lo = 0x00ff00ffu, hi = 0u;  // two 16-bit lanes each
    for (int y = y0; y < min(y0 + SVS_RANGE_ROWS, height); ++y) {
        const u32x2 *row = reinterpret_cast<const u32x2 *>(a + (int64_t)f * frame_pitch + (int64_t)y * row_pitch);
        for (int c = threadIdx.x; c < w8; c += 256) {
            const u32x2 v = SVS_LD(row + c);
            const uint32_t e0 = v.x & 0x00ff00ffu, o0 = (v.x >> 8) & 0x00ff00ffu;
            const uint32_t e1 = v.y & 0x00ff00ffu, o1 = (v.y >> 8) & 0x00ff00ffu;
            // packed 16-bit min / max (v_pk_min_u16 / v_pk_max_u16): four bytes per pair of instructions
            typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
            auto pk = [](uint32_t x) { return __builtin_bit_cast(u16x2, x); };
            u16x2 l = __builtin_elementwise_min(__builtin_elementwise_min(pk(e0), pk(o0)),
                                                __builtin_elementwise_min(pk(e1), pk(o1)));
            u16x2 h = __builtin_elementwise_max(__builtin_elementwise_max(pk(e0), pk(o0)),
                                                __builtin_elementwise_max(pk(e1), pk(o1)));
            lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(pk(lo), l));
            hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(pk(hi), h));
        }
    }
    uint32_t mn = min(lo & 0xffffu, lo >> 16), mx = max(hi & 0xffffu, hi >> 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_down(mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&lohi[2 * f], mn);
        atomicMax(&lohi[2 * f + 1], mx);
    }
}

__global__ void frame_range_finish_kernel(const uint32_t *__restrict__ lohi, int32_t n_frames, double *__restrict__ range) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames) range[f] = (double)lohi[2 * f + 1] - (double)lohi[2 * f];
}

#if defined(SVS_EXPERIMENTS)   // measurement kernels behind the experiments library's svs_ref_* / svs_probe_* hooks
// 8-byte-per-lane copy (what the one-block-per-lane kernels issue): non-temporal load, write-through or non-temporal store
template <int SC1>
__global__ __launch_bounds__(256) void copy8_kernel(const u32x2 *__restrict__ src, u32x2 *__restrict__ dst, uint64_t n8) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n8) {
        const u32x2 v = __builtin_nontemporal_load(src + i);
        if constexpr (SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
        else __builtin_nontemporal_store(v, dst + i);
    }
}

// mixed widths: which side of a copy is sensitive to the 8-byte access width?  SPLIT_LOAD: lane-strided 8-byte loads
// (lane i reads elements i and i + 64 of its wave's 1 KB) + one 16-byte store after a swap through LDS is not needed for
// the probe: the two halves are simply stored where they belong with the other width.
template <int WIDE_LOAD>
__global__ __launch_bounds__(256) void copy_mixed_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint64_t bytes) {
    // each wave moves 1 KB: either 64 x 16-byte loads + 2 x (64 x 8-byte) stores, or 2 x (64 x 8-byte) loads + 64 x 16-byte stores
    const uint64_t wave_base = ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)) * 16u;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave_base + 1024 > bytes) return;
    __shared__ __attribute__((aligned(16))) u32x4 buf[4][64];
    u32x4 *mine = buf[threadIdx.x >> 6];
    if constexpr (WIDE_LOAD) {
        mine[lane] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + wave_base) + lane);
        wave_lds_fence();
        const u32x2 *as8 = reinterpret_cast<const u32x2 *>(mine);
        const u32x2 a = as8[lane], b = as8[lane + 64];
        u32x2 *out = reinterpret_cast<u32x2 *>(dst + wave_base);
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(out + lane), "v"(a) : "memory");
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(out + lane + 64), "v"(b) : "memory");
    } else {
        const u32x2 *in = reinterpret_cast<const u32x2 *>(src + wave_base);
        u32x2 *as8 = reinterpret_cast<u32x2 *>(mine);
        as8[lane] = __builtin_nontemporal_load(in + lane);
        as8[lane + 64] = __builtin_nontemporal_load(in + lane + 64);
        wave_lds_fence();
        const u32x4 v = mine[lane];
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<u32x4 *>(dst + wave_base) + lane), "v"(v) : "memory");
    }
}

// ---- reference streams for tools/ab_bench.py: what plain copies / reads reach on the same box ----
// mode 0: one 16-byte element per thread, non-temporal;  mode 1: grid-stride, 4 x 16 B in flight per
// thread, non-temporal;  mode 2: as 1 with default cache policy
template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, uint64_t n16) {
    if constexpr (MODE >= 3) {
        // cache-policy experiments on the one-shot copy: 3 = nt load + sc1 (write-through) store, 4 = nt load + sc0 sc1 nt
        // store, 5 = sc1 load + sc1 store
        const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        if (i < n16) {
            u32x4 v;
            if constexpr (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + i) : "memory");
            else v = __builtin_nontemporal_load(src + i);
            if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst + i), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
        }
    } else if constexpr (MODE == 0) {
        const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    } else {
        const uint64_t stride = (uint64_t)gridDim.x * 256u;
        uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            u32x4 a, b, c, d;
            if constexpr (MODE == 1) {
                a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + stride);
                c = __builtin_nontemporal_load(src + i + 2 * stride); d = __builtin_nontemporal_load(src + i + 3 * stride);
                __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
                __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
            } else {
                a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
                dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
            }
        }
        for (; i < n16; i += stride) dst[i] = src[i];
    }
}

// read-only stream: xor-reduce, one dword written per workgroup (keeps the loads alive)
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ src, uint32_t *__restrict__ sink, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += stride)
        acc ^= __builtin_nontemporal_load(src + i);
    const uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (v == 0x12345678u) sink[blockIdx.x] = v;
}

// probe used by tests/test_gpu_primitives.py: what v_cvt_pk_u8_f32 does with a value
__global__ void probe_cvt_pk_u8_kernel(const float *__restrict__ in, uint32_t *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
#endif  // SVS_EXPERIMENTS

}  // namespace svs
