// svs_block.hpp - per-block arithmetic of the fused block-DCT / QIM frame operator.
//
// Reference behaviour being implemented: proses_frame_qim_dct, config_and_setup.py:106-174.
//
// Everything here is plain C++ on values held in registers, compiled for the gfx950 device by
// svs_device.hpp (the product) and - with g++ - by tests/hostemu/ (test infrastructure only: it
// lets the CPU-only test tier and the sanitizers exercise the very same arithmetic and bit
// bookkeeping; the shipped library has no CPU path).  Build both with -ffp-contract=off: fused
// operations are spelled fmaf() explicitly so that host and device round identically.
//
// Decomposition (see DESIGN.md "Kernels"):
//   * one LANE owns one 8x8 block; the block never leaves that lane's registers, so the
//     separable transform needs no cross-lane traffic at all (no LDS, no shuffles) and a
//     wavefront (64 lanes = 64 horizontally adjacent blocks) reads/writes 512 contiguous
//     bytes per row instruction.
//   * only the coefficient rows the payload touches are transformed: flat indices 1..n live in
//     rows u < U = n/8 + 1 of the coefficient matrix, so the vertical pass produces U outputs
//     per column and the horizontal pass runs on U rows (template parameter U).
//   * the streaming embed (n <= 15) uses linearity of the DCT as a PREDICTION: trunc(clip(x + IDCT(D' - D))), where D' - D
//     is non-zero only at the n requantised coefficients and x is the exact integer pixel, is the reference's byte wherever
//     a rigorous bound on the reference's own float32 round-trip noise (BETA, tools/guard_bound.py) separates the predicted
//     value from the integer grid; the other blocks are redone with pocketfft's exact operation sequence (namespace pf).
//   * everything else (n >= 16, delta outside the guard's range, the "exact" mode) replays pocketfft on all 64 coefficients.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SVS_HD __host__ __device__ __forceinline__
#else
#define SVS_HD inline
#endif

// compiler scheduling fence (device only): nothing is moved across it.  Used to keep the instruction scheduler from
// hoisting whole passes of independent work (and their live registers) ahead of where they are consumed.
#if defined(__HIP_DEVICE_COMPILE__)
#define SVS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SVS_SCHED_FENCE() ((void)0)
#endif

namespace svs {

// A block is held as two arrays of 8 little-endian dwords: rx[y] = pixels 0..3 of row y,
// ry[y] = pixels 4..7.  (Plain scalar arrays on purpose: an array of 8-byte structs is not fully
// scalarised by hipcc once 16-byte row loads are split into it, and ends up in LDS.)

// cos(k*pi/16)/2 and 1/sqrt(8): the orthonormal DCT-II basis (scipy norm='ortho',
// config_and_setup.py:135,168)
#define SVS_A0 0.35355339059327373f
#define SVS_C1 0.49039264020161522f
#define SVS_C2 0.46193976625564337f
#define SVS_C3 0.41573480615127262f
#define SVS_C4 0.35355339059327373f
#define SVS_C5 0.27778511650980114f
#define SVS_C6 0.19134171618254492f
#define SVS_C7 0.09754516100806417f

// Orthonormal 8-point DCT-II, first NOUT outputs only (even/odd split, FMA form: 36 ops for
// all eight outputs including the normalisation).
template <int NOUT>
SVS_HD void fdct8(const float (&x)[8], float (&X)[8]) {
    const float s0 = x[0] + x[7], s1 = x[1] + x[6], s2 = x[2] + x[5], s3 = x[3] + x[4];
    const float t0 = s0 + s3, t1 = s1 + s2;
    X[0] = (t0 + t1) * SVS_A0;
    if constexpr (NOUT > 1) {
        const float d0 = x[0] - x[7], d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
        const float t2 = s0 - s3, t3 = s1 - s2;
        X[1] = fmaf(d0, SVS_C1, fmaf(d1, SVS_C3, fmaf(d2, SVS_C5, d3 * SVS_C7)));
        if constexpr (NOUT > 2) X[2] = fmaf(t2, SVS_C2, t3 * SVS_C6);
        if constexpr (NOUT > 3) X[3] = fmaf(d0, SVS_C3, fmaf(d1, -SVS_C7, fmaf(d2, -SVS_C1, d3 * -SVS_C5)));
        if constexpr (NOUT > 4) X[4] = (t0 - t1) * SVS_C4;
        if constexpr (NOUT > 5) X[5] = fmaf(d0, SVS_C5, fmaf(d1, -SVS_C1, fmaf(d2, SVS_C7, d3 * SVS_C3)));
        if constexpr (NOUT > 6) X[6] = fmaf(t2, SVS_C6, t3 * -SVS_C2);
        if constexpr (NOUT > 7) X[7] = fmaf(d0, SVS_C7, fmaf(d1, -SVS_C5, fmaf(d2, SVS_C3, d3 * -SVS_C1)));
    }
}

// Orthonormal 8-point DCT-III (inverse of the above) of a vector whose entries NIN..7 are zero.
// SKIP0: entry 0 is known to be zero as well (row 0 of the modification matrix: DC is never
// touched, config_and_setup.py:140).
template <int NIN, bool SKIP0>
SVS_HD void idct8(const float (&X)[8], float (&x)[8]) {
    if constexpr (NIN == 2 && !SKIP0) {
        // two inputs: evaluate the 8 outputs directly (1 mul + 8 fma) instead of butterflies (13 ops)
        const float p = X[0] * SVS_A0;
        x[0] = fmaf(X[1], SVS_C1, p); x[7] = fmaf(X[1], -SVS_C1, p);
        x[1] = fmaf(X[1], SVS_C3, p); x[6] = fmaf(X[1], -SVS_C3, p);
        x[2] = fmaf(X[1], SVS_C5, p); x[5] = fmaf(X[1], -SVS_C5, p);
        x[3] = fmaf(X[1], SVS_C7, p); x[4] = fmaf(X[1], -SVS_C7, p);
        return;
    }
    float a, b;  // even part: DC and X4
    {
        const float p = SKIP0 ? 0.0f : X[0] * SVS_A0;
        if constexpr (NIN > 4) {
            const float r = X[4] * SVS_C4;
            a = SKIP0 ? r : p + r;
            b = SKIP0 ? -r : p - r;
        } else {
            a = p;
            b = p;
        }
    }
    float e0 = a, e1 = b, e2 = b, e3 = a;
    if constexpr (NIN > 2) {
        float g0 = X[2] * SVS_C2, g1 = X[2] * SVS_C6;
        if constexpr (NIN > 6) {
            g0 = fmaf(X[6], SVS_C6, g0);
            g1 = fmaf(X[6], -SVS_C2, g1);
        }
        if constexpr (SKIP0 && NIN <= 4) {
            e0 = g0; e3 = -g0; e1 = g1; e2 = -g1;
        } else {
            e0 = a + g0; e3 = a - g0; e1 = b + g1; e2 = b - g1;
        }
    }
    if constexpr (NIN > 1) {
        float o0 = X[1] * SVS_C1, o1 = X[1] * SVS_C3, o2 = X[1] * SVS_C5, o3 = X[1] * SVS_C7;
        if constexpr (NIN > 3) {
            o0 = fmaf(X[3], SVS_C3, o0); o1 = fmaf(X[3], -SVS_C7, o1);
            o2 = fmaf(X[3], -SVS_C1, o2); o3 = fmaf(X[3], -SVS_C5, o3);
        }
        if constexpr (NIN > 5) {
            o0 = fmaf(X[5], SVS_C5, o0); o1 = fmaf(X[5], -SVS_C1, o1);
            o2 = fmaf(X[5], SVS_C7, o2); o3 = fmaf(X[5], SVS_C3, o3);
        }
        if constexpr (NIN > 7) {
            o0 = fmaf(X[7], SVS_C7, o0); o1 = fmaf(X[7], -SVS_C5, o1);
            o2 = fmaf(X[7], SVS_C3, o2); o3 = fmaf(X[7], -SVS_C1, o3);
        }
        x[0] = e0 + o0; x[7] = e0 - o0;
        x[1] = e1 + o1; x[6] = e1 - o1;
        x[2] = e2 + o2; x[5] = e2 - o2;
        x[3] = e3 + o3; x[4] = e3 - o3;
    } else {
        x[0] = e0; x[7] = e0; x[1] = e1; x[6] = e1;
        x[2] = e2; x[5] = e2; x[3] = e3; x[4] = e3;
    }
}

// byte B (0..3) of a dword as float: v_cvt_f32_ubyte{0..3} on the device
template <int B>
SVS_HD float ubyte_to_float(uint32_t w) {
    return (float)((w >> (8 * B)) & 0xffu);
}

// Store pixel value v - an INTEGER-valued float (pixel + floor(change)) - clipped to [0,255] into
// byte B of `old`.  np.uint8(np.clip(x, 0, 255)) of the reference (config_and_setup.py:171) is
// clip-then-truncate; for x = pixel + change with an integer pixel, trunc(clip(x)) ==
// clip(pixel + floor(change)), which is what the callers pass in.
template <int B>
SVS_HD uint32_t put_pixel(float v, uint32_t old) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SVS_NO_CVT_PK_U8)
    // v_cvt_pk_u8_f32 saturates to [0,255] and rounds to nearest even (measured on gfx950:
    // profiles/history/r01_cvt_pk_u8_probe.json) - exact for the integer-valued input it gets here.
    return __builtin_amdgcn_cvt_pk_u8_f32(v, B, old);
#else
    const float c = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    const uint32_t u = (uint32_t)c;
    return (old & ~(0xffu << (8 * B))) | (u << (8 * B));
#endif
}

// Same store for a value that is NOT integer-valued: round to nearest even, then saturate (what v_cvt_pk_u8_f32
// does; rintf on the host)
template <int B>
SVS_HD uint32_t put_pixel_rne(float v, uint32_t old) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SVS_NO_CVT_PK_U8)
    return __builtin_amdgcn_cvt_pk_u8_f32(v, B, old);
#else
    const float r = rintf(v);
    const float c = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
    return (old & ~(0xffu << (8 * B))) | ((uint32_t)c << (8 * B));
#endif
}

// Eight float pixel values of one row -> their bytes in two dwords: np.uint8(np.clip(v, 0, 255)) of the reference
// (config_and_setup.py:171), i.e. clip, then truncate toward zero.  On the device this is v_cvt_pk_u8_f32 executed with the
// wave's FP32 rounding mode switched to round-toward-zero for exactly these eight instructions: the conversion saturates
// to [0, 255] and follows MODE.fp_round (measured on gfx950 with tools/probes/cvt_round_mode.hip: 126.99999 -> 126,
// 0.99999994 -> 0, -0.9 -> 0, 255.7 -> 255), which saves the v_floor_f32 per pixel of the floor-then-convert form.  One asm
// statement, so that no other floating-point instruction can be scheduled into the window.  It opens with s_nop 1: two of
// these statements can end up back to back, and s_setreg -> s_setreg on the same hardware register needs 2 wait states that
// the compiler's hazard recogniser cannot insert inside inline asm.  The kernels run with the default
// round-to-nearest-even mode, which the closing s_setreg restores.
SVS_HD void store_row_trunc(float p0, float p1, float p2, float p3, float p4, float p5, float p6, float p7, uint32_t &lo4,
                            uint32_t &hi4) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SVS_NO_CVT_PK_U8) && !defined(SVS_NO_RTZ_STORE)
    uint32_t a, b;
    asm volatile(
        "s_nop 1\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_u8_f32 %0, %2, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %1, %6, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %0, %3, 1, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %7, 1, %1\n\t"
        "v_cvt_pk_u8_f32 %0, %4, 2, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %8, 2, %1\n\t"
        "v_cvt_pk_u8_f32 %0, %5, 3, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %9, 3, %1\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
        : "=&v"(a), "=&v"(b)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7));
    lo4 = a;
    hi4 = b;
#else
    const float v[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
    uint32_t w[2] = {0, 0};
    for (int j = 0; j < 8; ++j) {
        const float c = v[j] < 0.0f ? 0.0f : (v[j] > 255.0f ? 255.0f : v[j]);   // NaN cannot occur (finite pixels)
        w[j >> 2] |= (uint32_t)c << (8 * (j & 3));
    }
    lo4 = w[0];
    hi4 = w[1];
#endif
}

// Forward transform of the coefficient rows u < U of one block.
// D[u][v] = sum_y sum_x a(u)a(v) p[y][x] cos((2y+1)u pi/16) cos((2x+1)v pi/16)
// (vertical axis first, as the reference does: axis=0 then axis=1, config_and_setup.py:135).
#ifndef SVS_PACKED_VERTICAL_U2
#define SVS_PACKED_VERTICAL_U2 1
#endif
#ifndef SVS_PACKED_VERTICAL_U1
#define SVS_PACKED_VERTICAL_U1 1
#endif
// Vertical pass for U = 2 on four columns held as the bytes of w[0..7]: the mirrored-row sums and differences the two
// outputs need are exact integers, so they are formed on 16-bit lanes (even bytes and odd bytes of the dwords: two
// columns per operation) and converted to float once - same values as the float path, fewer operations.
SVS_HD void vertical_u2_packed(const uint32_t (&w)[8], float (&v0)[4], float (&v1)[4]) {
    typedef int16_t i16x2 __attribute__((vector_size(4)));
    uint32_t lane[2][8];  // [0]: bytes 0 and 2 (columns 0, 2), [1]: bytes 1 and 3 (columns 1, 3), as 16-bit lanes
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        lane[0][r] = w[r] & 0x00ff00ffu;
        lane[1][r] = (w[r] >> 8) & 0x00ff00ffu;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t s[4];
        i16x2 d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = lane[h][k] + lane[h][7 - k];  // <= 510 per lane: no carry between lanes
            d[k] = __builtin_bit_cast(i16x2, lane[h][k]) - __builtin_bit_cast(i16x2, lane[h][7 - k]);
        }
        const uint32_t t = (s[0] + s[3]) + (s[1] + s[2]);  // <= 2040 per lane
#pragma unroll
        for (int c = 0; c < 2; ++c) {  // 16-bit lane c of half h is column 2c + h
            const float x0 = (float)(c ? (t >> 16) : (t & 0xffffu));
            v0[2 * c + h] = x0 * SVS_A0;
            const float d0 = (float)d[0][c], d1 = (float)d[1][c], d2 = (float)d[2][c], d3 = (float)d[3][c];
            v1[2 * c + h] = fmaf(d0, SVS_C1, fmaf(d1, SVS_C3, fmaf(d2, SVS_C5, d3 * SVS_C7)));
        }
    }
}

// U = 1: only the column sums are needed
SVS_HD void vertical_u1_packed(const uint32_t (&w)[8], float (&v0)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t t = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += h ? ((w[r] >> 8) & 0x00ff00ffu) : (w[r] & 0x00ff00ffu);  // <= 2040 per lane
        v0[h] = (float)(t & 0xffffu) * SVS_A0;
        v0[2 + h] = (float)(t >> 16) * SVS_A0;
    }
}

// Flat index 4 (coefficient (0, 4), basis +-1/8) exactly as pocketfft computes it, from row 0 of the vertical pass (which is
// pocketfft's, see forward_rows): for integer pixels this coefficient is a multiple of 1/8, so c / delta sits EXACTLY on a
// rounding tie in 1 of 8 delta blocks and only the reference's own float32 sequence says which way it falls.
// These are the operations of svs::pf::dct2_8 that feed output 4 (radb2 / radb4 sums, one twiddle product), nine in all.
SVS_HD float pf_row0_coefficient4(const float *v0) {
    const float c0 = 2.0f * v0[0], c7 = 2.0f * v0[7];
    const float c1 = v0[1] + v0[2], c3 = v0[3] + v0[4], c5 = v0[5] + v0[6];
    const float h0 = c0 + c7, h1 = c1 + c5;
    const float a0 = fmaf(2.0f, c3, h0);
    return fmaf(-2.0f, h1, a0) * 0x1.6a09e6p-3f;   // r[4] * (float(cos(pi/4)) / 4)
}

// `side` (optional): two by-products of row 0 of the vertical pass, V[0][x] = fl(colsum_x * a(0)) - the one part of this
// transform that is bit-identical to pocketfft's (its sums are exact integers and the product rounds once): the reference's
// own value of flat index 4 (pf_row0_coefficient4) and the sum of V[0] (= a(0) * the block's pixel sum, up to 8 roundings).
// Taken here so that the eight values need not stay in registers.
struct ForwardSide {
    float c4, v0_sum;
};
SVS_HD void forward_side(const float (&v0)[8], ForwardSide *side) {
    side->c4 = pf_row0_coefficient4(v0);
    side->v0_sum = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v0[4] + v0[5]) + (v0[6] + v0[7]));
}
template <int U>
SVS_HD void forward_rows(const uint32_t (&rx)[8], const uint32_t (&ry)[8], float (&D)[U][8], ForwardSide *side = nullptr) {
    float V[U][8];
    if constexpr (U == 1 && SVS_PACKED_VERTICAL_U1) {
        float a0[4], b0[4];
        vertical_u1_packed(rx, a0);
        vertical_u1_packed(ry, b0);
#pragma unroll
        for (int x = 0; x < 4; ++x) { V[0][x] = a0[x]; V[0][4 + x] = b0[x]; }
        if (side) forward_side(V[0], side);
        fdct8<8>(V[0], D[0]);
        return;
    }
    if constexpr (U == 2 && SVS_PACKED_VERTICAL_U2) {
        float a0[4], a1[4], b0[4], b1[4];
        vertical_u2_packed(rx, a0, a1);
        vertical_u2_packed(ry, b0, b1);
#pragma unroll
        for (int x = 0; x < 4; ++x) { V[0][x] = a0[x]; V[1][x] = a1[x]; V[0][4 + x] = b0[x]; V[1][4 + x] = b1[x]; }
        if (side) forward_side(V[0], side);
#pragma unroll
        for (int u = 0; u < U; ++u) fdct8<8>(V[u], D[u]);
        return;
    }
#define SVS_COL(X, W, B)                                                                \
    {                                                                                   \
        const float col[8] = {ubyte_to_float<B>(W[0]), ubyte_to_float<B>(W[1]), \
                              ubyte_to_float<B>(W[2]), ubyte_to_float<B>(W[3]), \
                              ubyte_to_float<B>(W[4]), ubyte_to_float<B>(W[5]), \
                              ubyte_to_float<B>(W[6]), ubyte_to_float<B>(W[7])}; \
        float out[8];                                                                   \
        fdct8<U>(col, out);                                                             \
        _Pragma("unroll") for (int u = 0; u < U; ++u) V[u][X] = out[u];                 \
    }
    SVS_COL(0, rx, 0) SVS_COL(1, rx, 1) SVS_COL(2, rx, 2) SVS_COL(3, rx, 3)
    SVS_COL(4, ry, 0) SVS_COL(5, ry, 1) SVS_COL(6, ry, 2) SVS_COL(7, ry, 3)
#undef SVS_COL
    if (side) forward_side(V[0], side);
#pragma unroll
    for (int u = 0; u < U; ++u) fdct8<8>(V[u], D[u]);
}

struct QimParams {
    float delta_f;      // (float)delta  - divisor
    float inv_delta_f;  // 1 / delta_f   - exact when delta is a power of two (QM_POW2)
    double delta_d;     // delta         - multiplier when (double)delta_f != delta (QM_DOUBLE)
    float tie_slope;    // FAST extraction: c00 * tie_slope bounds |c_fast/delta - c_pocketfft/delta| (see TIE_SLOPE)
    float tie2_sum, tie2_resid, tie2_c00;   // ... and its per-block refinement (SVS_TIE2_*), all per unit of 1/delta
    float tie2_max;                         // the largest value that margin takes over all blocks (any pixel sum)
    // GUARDED embed (embed_block_guarded): BETA = g_sum * (sum of pixels) + g_resid * sqrt(64 sum p^2 - (sum p)^2) + g_delta
    float g_sum, g_resid, g_delta;
};

// FAST extraction and rounding ties.  The FMA-factored forward transform (forward_rows) and pocketfft's are two float32
// evaluations of the same 64-term sums; for pixels in [0, 255] they differ by at most
//        |c_fast - c_pf| <= TIE_SLOPE * c00,     c00 = DC coefficient = (sum of the block's pixels) / 8,
// a forward error bound derived (and printed per coefficient row count) by tools/tie_bound.py: both algorithms are linear
// in the non-negative pixels, every path from a pixel to the output collects at most m factors (1 + d), |d| <= 2^-24, so
// the error is at most gamma_m * (sum of |path weights|) * (sum of pixels).  The largest slope over all coefficient
// indices and all U is 177 * 2^-24 = 1.055e-5; the constant below adds the roundings of the quantiser input itself
// (c * (1/delta): 3 more factors on |c| <= 2 c00) and 1 % of slack.  extract_block() reports a block in which some
// c/delta lies within that distance of a half-integer; the kernel then recomputes the block with the pocketfft-identical
// transform (extract_block_exact) - so FAST extraction returns the reference's bits for ANY input, ties included, while
// stego frames (coefficients sit near multiples of delta, far from ties) never take the second path.
#define SVS_TIE_SLOPE (1.055e-5 * 1.01 + 8.0 * 5.9604644775390625e-8)
// What the kernels test (round 3): the same difference bounded per block from its energy (tools/guard_bound.py --tie: first-
// order running error analysis of both operation sequences, as for the GUARDED embed),
//        |c_fast - c_pf| <= u' * (KDC * mean + KE * ||X - mean||_2),   u' = 2^-24 (1 + 2^-10),
// with ||X - mean||_2^2 = Q - S^2/64 <= S (16320 - S) / 64 (pixels <= 255), so that the block's pixel sum S is all it takes:
// 4x tighter than the global slope at mean 128 (3.5e-3 -> never-embedded noise: 0.7 % of the blocks at n = 10, delta = 8
// instead of 4 %).  Flat index 4 does not enter: its value is pocketfft's own (pf_row0_coefficient4).
// Constants of the largest row count (U = 8; U = 2: KE 24.3):
#define SVS_FAST_EXTRACT_DELTA_MIN 0x1p-10   // below, |c / delta| can reach 2^22 and extract_block_cheap's rounding constant no longer rounds: exact kernels
#define SVS_TIE2_KDC 64.0001
#define SVS_TIE2_KE 25.92

// How the quantiser is evaluated (all three give the reference's result, they differ in cost):
//   QM_F32    general delta: IEEE float32 division (about 10 instructions), float32 requantisation
//   QM_DOUBLE delta not representable in float32 (e.g. 0.1): divide by (float)delta, requantise with
//             a double multiply rounded once to float32 - what `float(q * delta)` does (:156)
//   QM_POW2   delta = 2^k: c / delta == c * (1/delta) exactly (both are the correctly rounded value of
//             the same real number), so the division is one multiply
enum QuantMode { QM_F32 = 0, QM_DOUBLE = 1, QM_POW2 = 2 };

// q = int(round(c / delta)) : float32 division, round half to even (config_and_setup.py:148,160).
// General delta: t = c * (1/delta) is within 3 * 2^-24 |t| of the correctly rounded quotient e = fl(c/delta),
// so rint(t) == rint(e) unless a half-integer lies within that distance of t; only then (about one
// coefficient in 10^5) is the 10-instruction IEEE division evaluated.  tests/test_block_arithmetic_cpu.py
// checks the shortcut against the division on 10^8 samples including every exact tie.
template <int QM>
SVS_HD int quant_index(float c, const QimParams &qp) {
#if defined(SVS_QUANT_ALWAYS_DIVIDE)  // A/B build: the plain division everywhere
    return (int)rintf(c / qp.delta_f);
#endif
    if constexpr (QM == QM_POW2) {
        return (int)rintf(c * qp.inv_delta_f);
    } else {
        const float t = c * qp.inv_delta_f;
        float r = rintf(t);
        const float miss = fabsf(fabsf(t - r) - 0.5f);       // distance of t from the nearest half-integer
        if (miss <= fabsf(t) * 4.76837158203125e-7f)          // 2^-21 |t|: comfortably above 3 * 2^-24 |t|
            r = rintf(c / qp.delta_f);
        return (int)r;
    }
}

// reference form of the above, used by the tests to validate the shortcut
// q with its parity forced to the payload bit by +1 (bit 1, q even) or -1 (bit 0, q odd) - config_and_setup.py:150-155:
// `q % 2` of a negative q is 0 or 1 in python, which is the low bit of the two's complement, so q + bit - (q & 1) is q with
// its low bit replaced by the bit: one v_bfi_b32 instead of and / subtract / add.
SVS_HD int force_parity(int q, int bit) { return (q & ~1) | bit; }
SVS_HD int quant_index_by_division(float c, float delta_f) { return (int)rintf(c / delta_f); }

// 64 stream bits starting at stream bit s of an MSB-first packed buffer viewed as dwords
// (touches at most dwords s/32 .. s/32+2, each only if below n_words)
SVS_HD void payload_window(const uint32_t *bits, uint32_t n_words, uint64_t s, uint32_t &hi, uint32_t &lo) {
    const uint32_t wi = (uint32_t)(s >> 5), sh = (uint32_t)(s & 31u);
    const uint32_t w0 = __builtin_bswap32(wi < n_words ? bits[wi] : 0u);
    const uint32_t w1 = __builtin_bswap32(wi + 1 < n_words ? bits[wi + 1] : 0u);
    const uint32_t w2 = __builtin_bswap32(wi + 2 < n_words ? bits[wi + 2] : 0u);
    const uint64_t a = ((uint64_t)w0 << 32) | w1, b = ((uint64_t)w1 << 32) | w2;
    hi = (uint32_t)((a << sh) >> 32);
    lo = (uint32_t)((b << sh) >> 32);
}

// n <= 15 (one and two coefficient rows): a block's window - and the windows of two adjacent blocks, 2 n <= 30 bits - lies
// inside the 64 stream bits that start at the dword holding stream bit s: two loads instead of three per block (six per lane
// with two blocks per lane).  payload_qword returns those 64 bits, MSB first; window32 the 32 stream
// bits starting `sh` bits in (sh < 32 for the block at s, < 47 for its right neighbour).
SVS_HD uint64_t payload_qword(const uint32_t *bits, uint32_t n_words, uint64_t s) {
    const uint32_t wi = (uint32_t)(s >> 5);
    const uint32_t w0 = __builtin_bswap32(wi < n_words ? bits[wi] : 0u);
    const uint32_t w1 = __builtin_bswap32(wi + 1 < n_words ? bits[wi + 1] : 0u);
    return ((uint64_t)w0 << 32) | w1;
}
SVS_HD uint32_t window32(uint64_t q, uint32_t sh) { return (uint32_t)((q << sh) >> 32); }

// bit i (0 = first) of the 64-bit MSB-first window hi:lo
SVS_HD uint32_t window_bit(uint32_t hi, uint32_t lo, int i) {
    return ((i < 32) ? (hi >> ((31 - i) & 31)) : (lo >> ((63 - i) & 31))) & 1u;
}

// bits a block takes from a budget of n_bits when its first stream bit is `first`
// (config_and_setup.py:130,132,141)
SVS_HD uint32_t block_budget(uint64_t first, uint64_t n_bits, uint32_t n) {
    if (first >= n_bits) return 0;
    const uint64_t left = n_bits - first;
    return left < n ? (uint32_t)left : n;
}

// frac(x) = x - floor(x) in [0, 1) (v_fract_f32) and min(a, |b|, |c|) (one v_min3_f32 with source modifiers)
SVS_HD float fract_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(x);
#else
    const float f = x - floorf(x);
    return f < 1.0f ? f : 0x1.fffffep-1f;   // v_fract_f32 clamps to the largest float below 1
#endif
}
SVS_HD float fmin3_abs(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    // spelled out: from the C form hipcc makes v_min_f32 |b|, |c| followed by a v_min3 that joins two such pairs - 56 instead
    // of 32 instructions of the 1.6-slot class per block (profiles/r04_valu_issue_rate.txt)
    float r;
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fminf(a, fminf(fabsf(b), fabsf(c)));
#endif
}

// (Rounds 1-3 had a separate FAST embed arithmetic here - FMA-factored forward transform on the payload rows, per-pixel grid
// test only for blocks that did not look "generic" - whose contract was "PSNR within 0.01 dB".  Each review found structured
// content on which the shortcut broke it (r02: exact cancellations, r03: coefficients requantised to 0 on smooth ramps,
// +0.68 dB), round 4's own probe a third family (two equal coefficients of a diagonal ramp cancelling on the anti-diagonal),
// and with the test on every pixel the arithmetic was no faster than the bit-identical kernels: n = 63 1.45 ms against the
// lane-per-block pocketfft kernel's 1.31 per 200 x 4K, n = 8..15 0.88 against the rigorous two-row kernel's 0.87
// (profiles/r04_*).  It is gone: every embed mode produces the reference's pixels.)

SVS_HD uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, acc, false);   // v_dot4_u32_u8
#else
    for (int i = 0; i < 4; ++i) acc += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return acc;
#endif
}

SVS_HD float guard_sqrt(float v) {   // any sqrt accurate to a few ulp will do: BETA's constants carry 2^-18 of slack
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);
#else
    return sqrtf(v);
#endif
}

// FAST extraction with two and more coefficient rows, in two steps (round 4).
//
// extract_block_cheap: parity bits of round(c_k/delta), k = 1..n, MSB-first into hi:lo (config_and_setup.py:160-161), from the
// FMA-factored forward transform.  t + 1.5 * 2^23 rounds t = c * (1/delta) to the nearest-even integer q and leaves q in the
// low mantissa bits of the sum (|t| < 2^22: the host routes delta < 2^-10 to the exact kernels), so the parity is bit 0 of the
// sum's pattern and one v_alignbit_b32 per coefficient - taken in DESCENDING order - shifts it into place: no rndne, no
// conversion, no shift / or (three 1.6-slot instructions per coefficient before, profiles/r04_valu_issue_rate.txt).
// Returns true when the block is a CANDIDATE: some t lies within qp.tie2_max - the largest value the per-block margin below
// can take for any block - of a rounding tie.  A block that is not a candidate has the reference's bits (the margin bounds
// |c_fast - c_pocketfft| / delta for EVERY coefficient, flat index 4 included: tools/guard_bound.py --tie), and needs nothing
// else; stego frames at delta >= 8 have no candidates at all (their coefficients sit within 4 / delta of the quantiser grid),
// so a wave of the kernel skips step two after one ballot.
// extract_block_settle (waves with a candidate; every lane of such a wave): what round 3 computed for every block - flat
// index 4 exactly as pocketfft has it (for integer pixels it is a multiple of 1/8: c / delta sits EXACTLY on a tie in 1 block of
// 8 delta, SURVEY N6, and only the reference's own float32 sequence says which way it falls; its bit is replaced), and the
// per-block margin  u' (KDC mean + KE ||X - mean||_2) / delta  from the pixel sum alone (see SVS_TIE2_*).  Returns true when
// a coefficient other than 4 is inside the margin of a tie: the caller then redoes the block with the pocketfft-identical
// transform (8 lanes per block on the device, extract_block_exact on the host emulation).  For a non-candidate block the
// result is false and the bits do not change - by the bound - so taking step two per wave (device) or per block (host
// emulation) gives the same stream.
template <int U, int QM, int NFIX = 0>
SVS_HD bool extract_block_cheap(const uint32_t (&rx)[8], const uint32_t (&ry)[8], uint32_t n_rt, const QimParams &qp,
                                uint32_t &hi, uint32_t &lo, float &off) {
    const uint32_t n = NFIX ? (uint32_t)NFIX : n_rt;
    float D[U][8];
    forward_rows<U>(rx, ry, D);
    hi = 0;
    lo = 0;
    off = 0.0f;        // largest |t - round(t)| over the used coefficients other than flat index 4 (0.5 = exactly on a tie)
    float d4 = 0.0f;   // t - round(t) of flat index 4
    const float kMagic = 12582912.0f;   // 1.5 * 2^23
#pragma unroll
    for (int k = 8 * U - 1; k >= 1; --k) {
        if ((uint32_t)k <= n) {  // wave-uniform
            const float t = D[k >> 3][k & 7] * qp.inv_delta_f;
            const float m = t + kMagic;
            const float d = t - (m - kMagic);
            if (k == 4) d4 = d;
            else off = fmaxf(off, fabsf(d));
            const uint32_t mb = __builtin_bit_cast(uint32_t, m);
            if (k - 1 < 32) hi = (mb << 31) | (hi >> 1);   // v_alignbit_b32: bit of coefficient i ends at 31 - i
            else lo = (mb << 31) | (lo >> 1);
        }
    }
    return !(fmaxf(off, fabsf(d4)) < 0.5f - qp.tie2_max);
}

template <int QM>
SVS_HD bool extract_block_settle(const uint32_t (&rx)[8], const uint32_t (&ry)[8], uint32_t n, const QimParams &qp, uint32_t &hi,
                                 float off) {
    float a0[4], b0[4], V0[8];
    vertical_u1_packed(rx, a0);      // row 0 of the vertical pass: fl(colsum * a(0)) - pocketfft's own values
    vertical_u1_packed(ry, b0);
#pragma unroll
    for (int x = 0; x < 4; ++x) { V0[x] = a0[x]; V0[4 + x] = b0[x]; }
    ForwardSide side;
    forward_side(V0, &side);
    if (n >= 4) {
        const uint32_t bit = (uint32_t)quant_index<QM>(side.c4, qp) & 1u;
        hi = (hi & ~(1u << 28)) | (bit << 28);   // i = 3 -> bit 31 - 3
    }
    // The per-block bound wants S = sum of the pixels and their energy Q = sum p^2 through 64 Q - S^2; pixels are at most
    // 255, so Q <= 255 S and 64 Q - S^2 <= S (16320 - S): no pass over the pixels (S = the vertical pass's DC row summed, over a(0)).
    const float S = side.v0_sum * (1.0000005f / SVS_A0);
    const float spread = guard_sqrt(fmaxf(S * (16320.0f - S), 0.0f)) * 1.000001f;
    const float c00 = side.v0_sum * (SVS_A0 * 1.000001f);     // the DC coefficient (a few roundings from fdct8's own value)
    const float margin = fmaf(qp.tie2_sum, S, fmaf(qp.tie2_resid, spread, fmaf(c00, qp.tie2_c00, 0x1p-20f)));
    return off >= 0.5f - margin;
}

// both steps for one block (host emulation; the kernels put a wave ballot between them)
template <int U, int QM, int NFIX = 0>
SVS_HD bool extract_block(const uint32_t (&rx)[8], const uint32_t (&ry)[8], uint32_t n_rt, const QimParams &qp,
                          uint32_t &hi, uint32_t &lo) {
    float off;
    if (!extract_block_cheap<U, QM, NFIX>(rx, ry, n_rt, qp, hi, lo, off)) return false;
    return extract_block_settle<QM>(rx, ry, NFIX ? (uint32_t)NFIX : n_rt, qp, hi, off);
}

// =====================================================================================================
// EXACT mode: bit-for-bit the float32 arithmetic of scipy.fftpack.dct/idct(norm='ortho') for length 8.
//
// scipy's DCT is pocketfft (C++, `T_dcst23<float>::exec`; scipy unpinned in the reference, vectors
// generated with scipy 1.15.3).  For N = 8 its published algorithm is: DCT-II = pre-butterfly ->
// BACKWARD real FFT of length 8 (radix passes radb2(ido=4), radb4(ido=1), factors [2,4]) scaled by
// fct = 1/sqrt(2N) = 0.25 -> twiddle post-pass with cos((k+1)pi/16) -> c[0] *= sqrt(2)/2;  DCT-III is
// the mirror image with the FORWARD real FFT (radf4(ido=1), radf2(ido=4)).  Every operation below is one
// IEEE float32 add/sub/mul in pocketfft's order (no FMA: build with -ffp-contract=off), so the results
// are bit-identical to scipy's - checked on random and integer vectors in tests/test_exact_mode_cpu.py and,
// through the whole operator, against every golden vector (stego PIXELS included).
// With these transforms the operator reproduces the reference's rounding-noise artefacts too
// (SURVEY N4/N6): exact .5 ties, 128 -> 127 on untouched flat blocks, delta <= 0 round trips.
// =====================================================================================================
namespace pf {
// float(cos((i+1) pi / 16)), i = 0..6: pocketfft's `twiddle[i]` for N = 8
#define SVS_PF_T0 0x1.f6297cp-1f
#define SVS_PF_T1 0x1.d906bcp-1f
#define SVS_PF_T2 0x1.a9b662p-1f
#define SVS_PF_T3 0x1.6a09e6p-1f
#define SVS_PF_T4 0x1.1c73b4p-1f
#define SVS_PF_T5 0x1.87de2ap-2f
#define SVS_PF_T6 0x1.8f8b84p-3f
#define SVS_PF_W 0x1.6a09e6p-1f      // cos(2pi/8) = sin(2pi/8) as float: the radix-2 pass twiddle
#define SVS_PF_SQRT2 0x1.6a09e6p+0f  // float(sqrt 2)

// Exactness-preserving rewrites used below (each keeps every result bit-identical to pocketfft's sequence):
//   * a multiplication by a power of two is exact, so it commutes with the roundings around it: the
//     `* fct` (0.25) after the FFT and the `0.5 *` of the DCT-II post-pass are folded into the twiddle
//     constants of the one multiplication every value passes through anyway (T/8, T/4 are exact floats);
//   * fl(a + fl(2*b)) == fmaf(2, b, a): pocketfft's `2*x` followed by an add/sub is one FMA.
// They remove 20 of 78 (DCT-II) and 8 of 66 (DCT-III) operations.  (Subnormal intermediates, which would
// break the first identity, cannot occur for pixel-derived data: non-zero values stay above 2^-40.)

// Every function below is written once for T = float (one line of 8 values) and T = f32x2 (two independent lines,
// component k of every value belonging to line k).  On gfx950 the two-line form compiles to packed-FP32 instructions
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE-754 single operations per issue slot), each component rounding
// exactly like the scalar operation - so the results stay bit-identical while the instruction count halves.
typedef float f32x2 __attribute__((vector_size(8)));

template <class T> SVS_HD T splat(float v);
template <> SVS_HD float splat<float>(float v) { return v; }
template <> SVS_HD f32x2 splat<f32x2>(float v) { const f32x2 r = {v, v}; return r; }

SVS_HD float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
SVS_HD f32x2 fma_t(f32x2 a, f32x2 b, f32x2 c) {
#if defined(__has_builtin)
#if __has_builtin(__builtin_elementwise_fma)
#define SVS_HAVE_ELEMENTWISE_FMA 1
#endif
#endif
#if defined(SVS_HAVE_ELEMENTWISE_FMA)
    return __builtin_elementwise_fma(a, b, c);
#else
    const f32x2 r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    return r;
#endif
}

// backward real FFT (halfcomplex -> real) of length 8, UNSCALED (the caller folds fct)
template <class T>
SVS_HD void rfft8_backward(const T (&c)[8], T (&o)[8]) {
    const T W = splat<T>(SVS_PF_W), two = splat<T>(2.0f), mtwo = splat<T>(-2.0f);
    // radb2, ido = 4, l1 = 1   (h3 = 2*c3 and h7 = -2*c4 are consumed as FMAs below)
    const T h0 = c[0] + c[7], h4 = c[0] - c[7];
    const T h1 = c[1] + c[5], tr2 = c[1] - c[5];
    const T ti2 = c[2] + c[6], h2 = c[2] - c[6];
    const T h6 = W * ti2 + W * tr2;
    const T h5 = W * tr2 - W * ti2;
    // radb4, ido = 1, l1 = 2:  a = h[4k] + h[4k+3], b = h[4k] - h[4k+3], out = a +- 2 h[4k+1], b +- 2 h[4k+2]
    const T a0 = fma_t(two, c[3], h0), b0 = fma_t(mtwo, c[3], h0);
    o[0] = fma_t(two, h1, a0);
    o[4] = fma_t(mtwo, h1, a0);
    o[6] = fma_t(two, h2, b0);
    o[2] = fma_t(mtwo, h2, b0);
    const T a1 = fma_t(mtwo, c[4], h4), b1 = fma_t(two, c[4], h4);
    o[1] = fma_t(two, h5, a1);
    o[5] = fma_t(mtwo, h5, a1);
    o[7] = fma_t(two, h6, b1);
    o[3] = fma_t(mtwo, h6, b1);
}

// forward real FFT (real -> halfcomplex) of length 8, UNSCALED (the caller folds fct)
template <class T>
SVS_HD void rfft8_forward(const T (&c)[8], T (&o)[8]) {
    const T W = splat<T>(SVS_PF_W);
    T y[8];
    // radf4, ido = 1, l1 = 2
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const T tr1 = c[k + 6] + c[k + 2];
        y[4 * k + 2] = c[k + 6] - c[k + 2];
        const T tr2 = c[k] + c[k + 4];
        y[4 * k + 1] = c[k] - c[k + 4];
        y[4 * k] = tr2 + tr1;
        y[4 * k + 3] = tr2 - tr1;
    }
    // radf2, ido = 4, l1 = 1
    const T tr2 = W * y[5] + W * y[6];
    const T ti2 = W * y[6] - W * y[5];
    o[0] = y[0] + y[4];
    o[7] = y[0] - y[4];
    o[4] = -y[7];
    o[3] = y[3];
    o[1] = y[1] + tr2;
    o[5] = y[1] - tr2;
    o[2] = ti2 + y[2];
    o[6] = ti2 - y[2];
}

// scipy.fftpack.dct(x, type=2, norm='ortho') for 8 float32 values
template <class T>
SVS_HD void dct2_8(const T (&x)[8], T (&X)[8]) {
    T c[8];
    c[0] = x[0] * splat<T>(2.0f);
    c[7] = x[7] * splat<T>(2.0f);
#pragma unroll
    for (int k = 1; k < 7; k += 2) {  // MPINPLACE(c[k+1], c[k])
        c[k + 1] = x[k + 1] - x[k];
        c[k] = x[k] + x[k + 1];
    }
    T r[8];
    rfft8_backward(c, r);
    // post-pass: pocketfft computes 0.5*(t1 +- t2) with t = T*(0.25 r) +- T*(0.25 r); here T/8 carries both scalings
    {
        const T t1 = splat<T>(SVS_PF_T0 * 0.125f) * r[7] + splat<T>(SVS_PF_T6 * 0.125f) * r[1];
        const T t2 = splat<T>(SVS_PF_T0 * 0.125f) * r[1] - splat<T>(SVS_PF_T6 * 0.125f) * r[7];
        X[1] = t1 + t2;
        X[7] = t1 - t2;
    }
    {
        const T t1 = splat<T>(SVS_PF_T1 * 0.125f) * r[6] + splat<T>(SVS_PF_T5 * 0.125f) * r[2];
        const T t2 = splat<T>(SVS_PF_T1 * 0.125f) * r[2] - splat<T>(SVS_PF_T5 * 0.125f) * r[6];
        X[2] = t1 + t2;
        X[6] = t1 - t2;
    }
    {
        const T t1 = splat<T>(SVS_PF_T2 * 0.125f) * r[5] + splat<T>(SVS_PF_T4 * 0.125f) * r[3];
        const T t2 = splat<T>(SVS_PF_T2 * 0.125f) * r[3] - splat<T>(SVS_PF_T4 * 0.125f) * r[5];
        X[3] = t1 + t2;
        X[5] = t1 - t2;
    }
    X[4] = r[4] * splat<T>(SVS_PF_T3 * 0.25f);
    X[0] = r[0] * splat<T>(SVS_PF_SQRT2 * 0.125f);
}

// scipy.fftpack.idct(X, type=2, norm='ortho') (= DCT-III) for 8 float32 values
template <class T>
SVS_HD void dct3_8(const T (&X)[8], T (&x)[8]) {
    T c[8];  // pre-pass with the FFT's fct = 0.25 folded into the constants
    c[0] = X[0] * splat<T>(SVS_PF_SQRT2 * 0.25f);
    {
        const T t1 = X[1] + X[7], t2 = X[1] - X[7];
        c[1] = splat<T>(SVS_PF_T0 * 0.25f) * t2 + splat<T>(SVS_PF_T6 * 0.25f) * t1;
        c[7] = splat<T>(SVS_PF_T0 * 0.25f) * t1 - splat<T>(SVS_PF_T6 * 0.25f) * t2;
    }
    {
        const T t1 = X[2] + X[6], t2 = X[2] - X[6];
        c[2] = splat<T>(SVS_PF_T1 * 0.25f) * t2 + splat<T>(SVS_PF_T5 * 0.25f) * t1;
        c[6] = splat<T>(SVS_PF_T1 * 0.25f) * t1 - splat<T>(SVS_PF_T5 * 0.25f) * t2;
    }
    {
        const T t1 = X[3] + X[5], t2 = X[3] - X[5];
        c[3] = splat<T>(SVS_PF_T2 * 0.25f) * t2 + splat<T>(SVS_PF_T4 * 0.25f) * t1;
        c[5] = splat<T>(SVS_PF_T2 * 0.25f) * t1 - splat<T>(SVS_PF_T4 * 0.25f) * t2;
    }
    c[4] = X[4] * splat<T>(SVS_PF_T3 * 0.5f);  // 2 * T3 * 0.25
    T r[8];
    rfft8_forward(c, r);
    x[0] = r[0];
    x[7] = r[7];
#pragma unroll
    for (int k = 1; k < 7; k += 2) {  // MPINPLACE(c[k], c[k+1])
        x[k] = r[k] - r[k + 1];
        x[k + 1] = r[k + 1] + r[k];
    }
}
}  // namespace pf

// all 64 coefficients of a block, pocketfft arithmetic: vertical transform first (axis 0), then
// horizontal (config_and_setup.py:135).  Coefficients the caller never reads are dead code.
SVS_HD void forward_exact(const uint32_t (&rx)[8], const uint32_t (&ry)[8], float (&D)[8][8]) {
    float V[8][8];
#define SVS_COL(X, W, B)                                                                \
    {                                                                                   \
        const float col[8] = {ubyte_to_float<B>(W[0]), ubyte_to_float<B>(W[1]),         \
                              ubyte_to_float<B>(W[2]), ubyte_to_float<B>(W[3]),         \
                              ubyte_to_float<B>(W[4]), ubyte_to_float<B>(W[5]),         \
                              ubyte_to_float<B>(W[6]), ubyte_to_float<B>(W[7])};        \
        float out[8];                                                                   \
        pf::dct2_8(col, out);                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) V[u][X] = out[u];                 \
    }
    SVS_COL(0, rx, 0) SVS_COL(1, rx, 1) SVS_COL(2, rx, 2) SVS_COL(3, rx, 3)
    SVS_COL(4, ry, 0) SVS_COL(5, ry, 1) SVS_COL(6, ry, 2) SVS_COL(7, ry, 3)
#undef SVS_COL
#pragma unroll
    for (int u = 0; u < 8; ++u) pf::dct2_8(V[u], D[u]);
}

// The same 64 coefficients, two lines per operation: the vertical pass transforms column pairs (2p, 2p+1), the
// horizontal pass row pairs (2q, 2q+1); D2[q][v] = (D[2q][v], D[2q+1][v]).  In between, the 2x2 sub-blocks are
// transposed in registers.
SVS_HD void forward_exact_paired(const uint32_t (&rx)[8], const uint32_t (&ry)[8], pf::f32x2 (&D2)[4][8]) {
    using pf::f32x2;
    f32x2 V2[4][8];  // V2[p][u] = (V[u][2p], V[u][2p+1])
#define SVS_COLPAIR(P, W, B0, B1)                                                       \
    {                                                                                   \
        f32x2 col[8];                                                                   \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                 \
            const f32x2 t = {ubyte_to_float<B0>(W[r]), ubyte_to_float<B1>(W[r])};       \
            col[r] = t;                                                                 \
        }                                                                               \
        pf::dct2_8(col, V2[P]);                                                         \
    }
    SVS_COLPAIR(0, rx, 0, 1) SVS_COLPAIR(1, rx, 2, 3) SVS_COLPAIR(2, ry, 0, 1) SVS_COLPAIR(3, ry, 2, 3)
#undef SVS_COLPAIR
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2 in[8];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
            const f32x2 a = V2[p2][2 * q], b = V2[p2][2 * q + 1];
            const f32x2 e = {a[0], b[0]}, o = {a[1], b[1]};
            in[2 * p2] = e;
            in[2 * p2 + 1] = o;
        }
        pf::dct2_8(in, D2[q]);
    }
}

// EXACT embed of one block: full DCT -> QIM on 1..n (first nb take payload) -> full IDCT (vertical
// first, :168) -> clip + truncate (:171).  A block that is entered is always round-tripped, even when
// nothing is changed (delta <= 0, n = 0): that is what produces the reference's x -> x-1 artefacts.
// The 64 coefficients of a CONSTANT block (all pixels = v), pocketfft arithmetic, without transforming 16 lines: every
// column is the same vector (v, .., v), so one dct2_8 gives the whole vertical pass; its AC outputs are exact zeros
// (differences of equal values, products and sums of zeros), so rows 1..7 stay zero through the horizontal pass, and row 0 is
// again a constant vector.  Two transforms instead of sixteen, same bits (the sign of a zero coefficient cannot reach a
// pixel: x + (-0) = x, and an exactly-zero pixel truncates to 0 either way).
SVS_HD void forward_exact_paired_constant(float v, pf::f32x2 (&D2)[4][8]) {
    using pf::f32x2;
    float col[8], V[8], row[8], D0[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) col[r] = v;
    pf::dct2_8(col, V);       // V[0] = the column's DC term, V[1..7] = 0
#pragma unroll
    for (int x = 0; x < 8; ++x) row[x] = V[0];
    pf::dct2_8(row, D0);      // D0[0] = DC, D0[1..7] = 0
    const f32x2 zero = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c) D2[q][c] = zero;
    D2[0][0][0] = D0[0];
}

// `constant_block`: the caller knows (wave-uniformly) that all 64 pixels are equal - the forward pass is then the two-line
// shortcut above (host emulation of flat content; the device replay transforms every block in full)
template <int U, int QM>
SVS_HD void embed_block_exact(uint32_t (&rx)[8], uint32_t (&ry)[8], uint32_t n, uint32_t nb, uint32_t hi, uint32_t lo,
                              const QimParams &qp, bool constant_block = false) {
    using pf::f32x2;
    f32x2 D2[4][8];
    if (constant_block) forward_exact_paired_constant(ubyte_to_float<0>(rx[0]), D2);
    else forward_exact_paired(rx, ry, D2);
#pragma unroll
    for (int k = 1; k < 8 * U; ++k) {
        if ((uint32_t)k <= n) {  // wave-uniform
            const int i = k - 1;
            const int bit = (int)window_bit(hi, lo, i);
            const float c = D2[k >> 4][k & 7][(k >> 3) & 1];
            int q = quant_index<QM>(c, qp);
            q = force_parity(q, bit);
            float cn;
            if constexpr (QM == QM_DOUBLE) cn = (float)((double)q * qp.delta_d);
            else cn = (float)q * qp.delta_f;
            D2[k >> 4][k & 7][(k >> 3) & 1] = ((uint32_t)i < nb) ? cn : c;
        }
    }
    f32x2 P2[4][8];  // vertical inverse of coefficient-column pairs: P2[p][y] = (P[y][2p], P[y][2p+1])
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
        f32x2 col[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 a = D2[q][2 * p2], b = D2[q][2 * p2 + 1];
            const f32x2 e = {a[0], b[0]}, o = {a[1], b[1]};
            col[2 * q] = e;
            col[2 * q + 1] = o;
        }
        pf::dct3_8(col, P2[p2]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // horizontal inverse of pixel rows 2q, 2q+1
        f32x2 in[8], px[8];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
            const f32x2 a = P2[p2][2 * q], b = P2[p2][2 * q + 1];
            const f32x2 e = {a[0], b[0]}, o = {a[1], b[1]};
            in[2 * p2] = e;
            in[2 * p2 + 1] = o;
        }
        pf::dct3_8(in, px);
        // np.uint8(np.clip(v, 0, 255)): clip, then truncate (every byte is overwritten: the input rows are dead)
#pragma unroll
        for (int k = 0; k < 2; ++k)
            store_row_trunc(px[0][k], px[1][k], px[2][k], px[3][k], px[4][k], px[5][k], px[6][k], px[7][k], rx[2 * q + k],
                            ry[2 * q + k]);
    }
}

// =====================================================================================================
// The streaming embed arithmetic (n <= 15; every mode): the reference's stego pixels, bit for bit, at the cost of a cheap
// sparse transform for almost every block.
//
// The reference's output for a block is trunc(clip(out)), out = pf_dct3(pf_dct3(D')) in float32 (config_and_setup.py:
// 166-171).  embed_block_guarded computes, with a handful of operations,
//   * the payload-carrying coefficients EXACTLY as pocketfft does (so every quantiser decision q_k is the reference's),
//   * pred = X + G(change): the pixel plus the sparse inverse of the n coefficient changes, and
//   * BETA, a rigorous bound on |out - pred| for this block (tools/guard_bound.py: first-order running error analysis of
//     every operation of svs::pf::dct2_8 / dct3_8, stage-wise Cauchy-Schwarz over the block's energy):
//         BETA = u' * (KDC * mean + KE * ||X - mean||_2 + KD * (1.5 delta + 0.01)) + 2^-20,   u' = 2^-24 (1 + 2^-10).
// If every pred is farther than BETA from the integers, floor(pred) IS floor(out) for all 64 pixels and the cheap result
// is the reference's (values outside [0, 255] clip to the same byte either way; flagging them too is merely conservative).
// Otherwise the function returns true and the caller redoes the block with the pocketfft-identical arithmetic
// (embed_block_exact on the host emulation; the 8-lanes-per-block replay inside the kernel, svs_device.hpp).
// With one coefficient row (n <= 7) the change is the same in all 8 rows of a column, so only 8 values are tested and a
// noise block (mean 128, sigma 65, delta 8: BETA = 1.1e-3) is flagged with probability 16 * BETA = 1.7 %; smooth content 0.5 %.
// Constants printed by tools/guard_bound.py (tests/test_guarded_mode_cpu.py re-derives them):
#define SVS_GUARD_KDC 17.0001      // per unit of the mean pixel value
#define SVS_GUARD_KE 31.05         // per unit of ||X - mean||_2 (the (2 -> 1) norm is at least 30.66: the bound is tight)
#define SVS_GUARD_KD_U1 19.61      // per unit of 1.5 delta + 0.01, at most 7 modified coefficients (incl. the kernel's own sparse inverse)
#define SVS_GUARD_KD_U2 54.78      // at most 15 modified coefficients
#define SVS_GUARD_UEFF (5.9604644775390625e-8 * (1.0 + 0.0009765625))
// delta range the guarded path is used for (outside it the caller takes the exact kernel): below, the changes are smaller
// than BETA and every block would be flagged; above, BETA itself exceeds 1/8
#define SVS_GUARD_DELTA_MIN 0.25
#define SVS_GUARD_DELTA_MAX 4096.0

// One coefficient through the quantiser, in the float domain: change = fl(q' delta) - c with q' = round-half-even(c / delta)
// forced to the parity of `bit` (config_and_setup.py:148-156).  t + 1.5 * 2^23 rounds t to the nearest-even integer q AND leaves
// q in the low mantissa bits of the sum (|t| < 2^22: delta >= 1/4 and |c| <= 2040 here), so the parity is forced on the bit
// pattern and q' comes back by subtracting the constant - no conversion to an integer and back (those are 1.6-slot
// instructions, profiles/r04_valu_issue_rate.txt).  Same results as quant_index / force_parity (tests: guarded mode vs the oracle).
template <int QM>
SVS_HD float qim_change(float c, uint32_t bit, const QimParams &qp) {
    if constexpr (QM == QM_DOUBLE) {
        const int q = force_parity(quant_index<QM>(c, qp), (int)bit);
        return (float)((double)q * qp.delta_d) - c;
    } else {
        const float kMagic = 12582912.0f;   // 1.5 * 2^23
        const float t = c * qp.inv_delta_f;
        float m = t + kMagic;
#if defined(SVS_QUANT_ALWAYS_DIVIDE)
        m = rintf(c / qp.delta_f) + kMagic;
#else
        if constexpr (QM != QM_POW2) {
            const float r = m - kMagic;
            const float miss = fabsf(fabsf(t - r) - 0.5f);       // distance of t from the nearest half-integer
            if (miss <= fabsf(t) * 4.76837158203125e-7f)          // see quant_index
                m = rintf(c / qp.delta_f) + kMagic;
        }
#endif
        const uint32_t mb = (__builtin_bit_cast(uint32_t, m) & ~1u) | bit;
        const float qf = __builtin_bit_cast(float, mb) - kMagic;
        return qf * qp.delta_f - c;
    }
}

// ---- one coefficient row (n <= 7) in the INTEGER domain (round 6; rounds 3-5 applied the same eight floors in the float domain:
// byte -> float, add, saturating float -> byte for each of the 64 pixels) ------------------------------------------------
// With n <= 7 the change is the same in all 8 rows of a pixel column, so the block's result is pixel + d[x] with eight
// integers d[x] = floor(change of column x).  When no pixel of the block can leave [0, 255] - min pixel + min d >= 0 and
// max pixel + max d <= 255 - the saturation of the store never acts, and because  sum_j (p_j + d_j) 256^j = P + D  with
// D = sum_j d_j 256^j (mod 2^32) and every p_j + d_j a byte, the four stego bytes of a row dword are ONE 32-bit add of the
// packed column deltas: 16 v_add_u32 per block instead of 64 x (byte -> float, add, float -> saturated byte).  The price is
// the block's min / max pixel - packed 16-bit min / max on the even / odd byte lanes the column sums split the rows into
// anyway.  The decision (BETA, `worst`) is guard_decide's, value for value; the quantiser step is qim_change (the float-
// domain form of the two-row kernel, same results as quant_index / force_parity for |c / delta| < 2^22: delta >= 1/4 here).
SVS_HD uint32_t pk_min_u16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
    return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16);
#endif
}
SVS_HD uint32_t pk_max_u16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
#else
    const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
    return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16);
#endif
}
SVS_HD int floor_to_int(float x) {   // (int)floor(x): v_cvt_flr_i32_f32
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return (int)floorf(x);
#endif
}
SVS_HD int imin3(int a, int b, int c) { const int m = a < b ? a : b; return m < c ? m : c; }
SVS_HD int imax3(int a, int b, int c) { const int m = a > b ? a : b; return m > c ? m : c; }

// pins three running values at this point of the instruction stream (device only): an opaque, ordered use-and-redefinition,
// so that the chains feeding them are complete here and the operands they consumed are dead - without it the instruction
// selector lets the min / max / sum-of-squares chains of guard_decide_int trail the transform, with their 32 lane splits alive
#if defined(__HIP_DEVICE_COMPILE__)
#define SVS_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#else
#define SVS_PIN3(a, b, c) ((void)0)
#endif
#define SVS_ROW1_UNDECIDED 1u   // guard_decide_int: BETA does not separate some column's change from the integer grid
#define SVS_ROW1_MAY_CLIP 2u    //                   min pixel + min d < 0 or max pixel + max d > 255: the plain add must not be used

// the eight column deltas of a block as two's complement 16-bit lanes: columns (0, 2), (1, 3), (4, 6), (5, 7) - the form the
// saturating store adds to the even / odd byte lanes of a row dword.  (|d| < 2^14 for every delta the guarded range admits:
// 1.5 * 4096 * sum of 7 basis values * a(0) < 5700.)
struct ColumnDeltas {
    uint32_t e_lo, o_lo, e_hi, o_hi;
};
// ... and as ONE 32-bit addend per row dword: sum_j d_j 256^j (mod 2^32) of columns 0..3 (e = e_lo, o = o_lo) / 4..7.  A lane
// pair (a, b) read as an integer is a + 65536 b + 65536 [a < 0]; without the last term it is the addend of its two columns.
SVS_HD uint32_t packed_addend(uint32_t e, uint32_t o) {
    const uint32_t ev = e - ((e & 0x8000u) << 1), ov = o - ((o & 0x8000u) << 1);
    return ev + (ov << 8);
}

SVS_HD uint32_t pk_add_i16(uint32_t a, uint32_t b) {   // lane-wise 16-bit add (wraps; the callers stay far inside the range)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b));
#else
    return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16);
#endif
}
SVS_HD uint32_t pk_clamp_u8_i16(uint32_t a) {   // lane-wise clamp of signed 16-bit lanes to [0, 255]
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
    const i16x2 zero = {0, 0}, top = {255, 255};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2, a), zero), top));
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; ++h) {
        int v = (int16_t)(a >> (16 * h));
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        r |= (uint32_t)v << (16 * h);
    }
    return r;
#endif
}

// -> SVS_ROW1_* flags and the column deltas (meaningful unless SVS_ROW1_UNDECIDED).
// (A compile-time coefficient count - n = 3, BASELINE configs[2] - removes a quarter of the instructions and changes nothing
// measurable: the kernel's arithmetic is hidden behind its memory traffic, 1.6144 vs 1.6056 ms per 600 x 4K in a same-process
// A/B.  Not instantiated.)
template <int QM>
SVS_HD uint32_t guard_decide_int(const uint32_t (&rx)[8], const uint32_t (&ry)[8], uint32_t n, uint32_t nb, uint32_t hi,
                                 const QimParams &qp, ColumnDeltas &cd) {
    float V[8];
    uint32_t S = 0, Q = 0, mn = 0x00ff00ffu, mx = 0u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t te = 0, to = 0;   // columns (0, 2) and (1, 3) of this half as 16-bit lanes, each <= 2040
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t w = half ? ry[r] : rx[r];
            const uint32_t e = w & 0x00ff00ffu, o = (w >> 8) & 0x00ff00ffu;
            te += e;
            to += o;
            Q = dot4_u8(w, w, Q);
            mn = pk_min_u16(pk_min_u16(mn, e), o);
            mx = pk_max_u16(pk_max_u16(mx, e), o);
            if (r & 1) SVS_PIN3(mn, mx, Q);
        }
        V[4 * half + 0] = (float)(te & 0xffffu) * SVS_A0;
        V[4 * half + 1] = (float)(to & 0xffffu) * SVS_A0;
        V[4 * half + 2] = (float)(te >> 16) * SVS_A0;
        V[4 * half + 3] = (float)(to >> 16) * SVS_A0;
        const uint32_t t = te + to;
        S += (t & 0xffffu) + (t >> 16);
        SVS_SCHED_FENCE();
    }
    float D[8];
    pf::dct2_8(V, D);   // row 0 of the coefficient matrix, bit-identical to scipy's
    SVS_SCHED_FENCE();
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        float change = 0.0f;
        if ((uint32_t)k <= n)  // wave-uniform; the budget is applied below, for the one block it concerns
            change = qim_change<QM>(D[k], (hi >> (32 - k)) & 1u, qp);
        D[k] = change;
    }
    if (nb < n) {   // the block the payload ends in: coefficients past the budget stay as they are (config_and_setup.py:141)
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if ((uint32_t)(k - 1) >= nb) D[k] = 0.0f;
    }
    D[0] = 0.0f;
    float P[8];
    idct8<8, true>(D, P);
    // BETA for this block: 64 Q - S^2 = 64 * (sum of squared deviations from the mean), an exact integer < 2^32
    const float spread = guard_sqrt((float)(64u * Q - S * S));
    const float beta = fmaf(qp.g_sum, (float)S, fmaf(qp.g_resid, spread, qp.g_delta));
    float worst = 0.0f;   // largest |frac(change) - 1/2| among the 8 columns
    int d[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const float ch = P[x] * SVS_A0;
        // v_fract_f32 is ch - floor(ch) (exact for these magnitudes) except that it stays below 1 where that difference would
        // round up to 1.0 (ch a hair below an integer): 1 - 2^-24 is flagged like 1.0 (BETA >= 2^-20)
        worst = fmaxf(worst, fabsf(fract_f32(ch) - 0.5f));
        d[x] = floor_to_int(ch);
    }
    uint32_t flags = (nb > 0 && !(worst < 0.5f - beta)) ? SVS_ROW1_UNDECIDED : 0u;   // nb == 0: the reference never enters the block (:130,:132)
    const int dmin = imin3(imin3(d[0], d[1], d[2]), imin3(d[3], d[4], d[5]), d[6] < d[7] ? d[6] : d[7]);
    const int dmax = imax3(imax3(d[0], d[1], d[2]), imax3(d[3], d[4], d[5]), d[6] > d[7] ? d[6] : d[7]);
    const uint32_t mnl = mn & 0xffffu, mnh = mn >> 16, mxl = mx & 0xffffu, mxh = mx >> 16;
    const int pmin = (int)(mnl < mnh ? mnl : mnh), pmax = (int)(mxl > mxh ? mxl : mxh);
    if (pmin + dmin < 0 || pmax + dmax > 255) flags |= SVS_ROW1_MAY_CLIP;
    cd.e_lo = ((uint32_t)d[0] & 0xffffu) | ((uint32_t)d[2] << 16);
    cd.o_lo = ((uint32_t)d[1] & 0xffffu) | ((uint32_t)d[3] << 16);
    cd.e_hi = ((uint32_t)d[4] & 0xffffu) | ((uint32_t)d[6] << 16);
    cd.o_hi = ((uint32_t)d[5] & 0xffffu) | ((uint32_t)d[7] << 16);
    return flags;
}

// the stores: pixel + d[column], clipped to [0, 255] (trunc(clip(x + c)) == clip(x + floor(c)) for integer x, :171)
SVS_HD void apply_deltas_plain(uint32_t (&rx)[8], uint32_t (&ry)[8], const ColumnDeltas &cd) {   // nothing can clip
    const uint32_t lo = packed_addend(cd.e_lo, cd.o_lo), hi = packed_addend(cd.e_hi, cd.o_hi);
#pragma unroll
    for (int r = 0; r < 8; ++r) { rx[r] += lo; ry[r] += hi; }
}
SVS_HD uint32_t add_clip_dword(uint32_t w, uint32_t de, uint32_t dodd) {
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque copy: without it the compiler keeps the 64 lane splits of guard_decide_int alive for this (rare) path - 50 VGPRs
    asm("" : "+v"(w));
#endif
    const uint32_t e = pk_clamp_u8_i16(pk_add_i16(w & 0x00ff00ffu, de));
    const uint32_t o = pk_clamp_u8_i16(pk_add_i16((w >> 8) & 0x00ff00ffu, dodd));
    return e | (o << 8);
}
SVS_HD void apply_deltas_clipped(uint32_t (&rx)[8], uint32_t (&ry)[8], const ColumnDeltas &cd) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        rx[r] = add_clip_dword(rx[r], cd.e_lo, cd.o_lo);
        ry[r] = add_clip_dword(ry[r], cd.e_hi, cd.o_hi);
    }
}

// decide, then apply: on return rx/ry hold the block's stego pixels - or, when the result is true (undecided), its original
// pixels, untouched (the host emulation, the one-block-per-lane and fused-colour kernels; embed_row1_kernel chooses the
// store form per WAVE instead of per lane)
template <int QM>
SVS_HD bool embed_block_guarded(uint32_t (&rx)[8], uint32_t (&ry)[8], uint32_t n, uint32_t nb, uint32_t hi, uint32_t lo,
                                const QimParams &qp) {
    (void)lo;   // one coefficient row: the window is its first word
    ColumnDeltas cd;
    const uint32_t flags = guard_decide_int<QM>(rx, ry, n, nb, hi, qp, cd);
    if (flags & SVS_ROW1_UNDECIDED) return true;
    if (flags & SVS_ROW1_MAY_CLIP) apply_deltas_clipped(rx, ry, cd);
    else apply_deltas_plain(rx, ry, cd);
    return false;
}

// Two coefficient rows (n = 8..15), rigorous: the same construction with 64 predictions instead of 8 (the change now varies
// down a column) - pocketfft-identical rows 0 and 1, QIM with the reference's decisions, sparse inverse, and every pixel's
// prediction tested against the grid with the bound of ITS position: the (2 -> 1) norms behind KE differ by pixel (24.7 in
// rows / columns 0, 3, 4, 7 crossed with each other, 31.0 in rows / columns 1, 2, 5, 6 crossed, 27.9 mixed), which takes the
// share of undecided noise blocks from 13 % to 12 %.  Every mode runs it (round 4: flags 0 and SVS_EXACT_GUARDED are the same launch).
#define SVS_GUARD_KE_CC 24.68
#define SVS_GUARD_KE_CE 27.93

// Outputs 0 and 1 of pocketfft's vertical transform (svs::pf::dct2_8 down a pixel column), for the four columns held as the
// bytes of w[0..7] - bit for bit, at a third of the operations (round 4; before, each column ran the float sequence of which
// the compiler kept the 41 operations behind X[0] and X[1]).  Pixels are integers, so every value of that sequence up to the
// first irrational twiddle is an exact small integer whichever way it is computed:
//     c0 = 2 x0, c7 = 2 x7, c1 = x1 + x2, c2 = x2 - x1, c3 = x3 + x4, c4 = x4 - x3, c5 = x5 + x6, c6 = x6 - x5      (pre-butterfly)
//     h4 = c0 - c7 = 2 d0,  tr2 = c1 - c5 = d1 + d2,  ti2 = c2 + c6 = d2 - d1,  with d_k = x_k - x_(7-k)              (radb2)
//     a1 = h4 - 2 c4 = 2 (d0 + d3),  b1 = h4 + 2 c4 = 2 (d0 - d3),  r[0] = 2 (sum of the column)                      (radb4)
// so they are formed on packed 16-bit lanes (two columns per operation) and converted to float once.  From there on the
// operations are pocketfft's own: h6 = W ti2 + W tr2, h5 = W tr2 - W ti2, r[1] = fl(2 h5 + a1), r[7] = fl(2 h6 + b1), the
// twiddle post-pass.  Two exact rescalings: fl(2 h + 2 a') = 2 fl(h + a') - r[1] and r[7] are carried at half their value -
// and that factor 2 is folded into the twiddle constants (T / 4 instead of T / 8; powers of two commute with rounding).
// S accumulates the sum of the 32 pixels (the guard's mean term) from the column sums that are there anyway.
// tests/test_guarded_mode_cpu.py holds this function against svs::pf::dct2_8 on random and extreme columns.
SVS_HD void vertical_pf01_packed(const uint32_t (&w)[8], float (&v0)[4], float (&v1)[4], uint32_t &S);

namespace pf {
SVS_HD void column_outputs01(float colsum, float tr2, float ti2, float a1h, float b1h, float &X0, float &X1) {
    X0 = colsum * (SVS_PF_SQRT2 * 0.25f);            // r[0] * (sqrt2 / 8), r[0] = 2 * colsum
    const float wtr = SVS_PF_W * tr2, wti = SVS_PF_W * ti2;
    const float h6 = wti + wtr, h5 = wtr - wti;
    const float o1 = h5 + a1h, o7 = h6 + b1h;        // r[1] / 2, r[7] / 2
    const float t1 = (SVS_PF_T0 * 0.25f) * o7 + (SVS_PF_T6 * 0.25f) * o1;
    const float t2 = (SVS_PF_T0 * 0.25f) * o1 - (SVS_PF_T6 * 0.25f) * o7;
    X1 = t1 + t2;
}
}  // namespace pf

SVS_HD void vertical_pf01_packed(const uint32_t (&w)[8], float (&v0)[4], float (&v1)[4], uint32_t &S) {
    typedef int16_t i16x2 __attribute__((vector_size(4)));
    uint32_t lane[2][8];  // [0]: bytes 0 and 2 (columns 0, 2), [1]: bytes 1 and 3 (columns 1, 3), as 16-bit lanes
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        lane[0][r] = w[r] & 0x00ff00ffu;
        lane[1][r] = (w[r] >> 8) & 0x00ff00ffu;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t s[4];
        i16x2 d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = lane[h][k] + lane[h][7 - k];  // <= 510 per lane: no carry between lanes
            d[k] = __builtin_bit_cast(i16x2, lane[h][k]) - __builtin_bit_cast(i16x2, lane[h][7 - k]);
        }
        const uint32_t t = (s[0] + s[3]) + (s[1] + s[2]);  // <= 2040 per lane
        S += (t & 0xffffu) + (t >> 16);
        const i16x2 tr2 = d[1] + d[2], ti2 = d[2] - d[1], a1h = d[0] + d[3], b1h = d[0] - d[3];   // |.| <= 510
#pragma unroll
        for (int c = 0; c < 2; ++c)  // 16-bit lane c of half h is column 2c + h
            pf::column_outputs01((float)(c ? (t >> 16) : (t & 0xffffu)), (float)tr2[c], (float)ti2[c], (float)a1h[c], (float)b1h[c],
                                 v0[2 * c + h], v1[2 * c + h]);
    }
}

// NFIX (8..15, or 0 = run-time n): the coefficient count at compile time - the quantiser loop loses its wave-uniform tests,
// outputs of the row-1 transform that nobody reads and inverse inputs that are known zeros disappear from the code.
// INPLACE: the stego bytes replace the pixels of rx / ry as they are computed (each column touches only its own byte of the
// row dwords) and an UNDECIDED block is left half-written - for a caller that has parked the original rows elsewhere (the
// two-row kernel parks them in LDS, where the exact replay wants them anyway: 16 registers and 16 moves less).
template <int QM, int NFIX = 0, bool INPLACE = false>
SVS_HD bool embed_block_guarded2(uint32_t (&rx)[8], uint32_t (&ry)[8], uint32_t n_rt, uint32_t nb, uint32_t hi, uint32_t lo,
                                 const QimParams &qp) {
    static_assert(NFIX == 0 || (NFIX >= 8 && NFIX <= 15), "two coefficient rows");
    const uint32_t n = NFIX ? (uint32_t)NFIX : n_rt;
    float V0[8], V1[8];
    uint32_t S = 0, Q = 0;
    {
        float a0[4], a1[4], b0[4], b1[4];
        vertical_pf01_packed(rx, a0, a1, S);
        vertical_pf01_packed(ry, b0, b1, S);
#pragma unroll
        for (int x = 0; x < 4; ++x) { V0[x] = a0[x]; V1[x] = a1[x]; V0[4 + x] = b0[x]; V1[4 + x] = b1[x]; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        Q = dot4_u8(rx[r], rx[r], Q);
        Q = dot4_u8(ry[r], ry[r], Q);
    }
    float D0[8], D1[8];
    pf::dct2_8(V0, D0);
    pf::dct2_8(V1, D1);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float change = 0.0f;
        if (k >= 1 && (uint32_t)k <= n)  // wave-uniform; the budget is applied below, for the one block it concerns
            change = qim_change<QM>(k < 8 ? D0[k] : D1[k - 8], window_bit(hi, lo, k - 1), qp);
        if (k < 8) D0[k] = change;
        else D1[k - 8] = change;
    }
    if (nb < n) {   // the block the payload ends in (see embed_block): coefficients past the budget stay as they are
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            if ((uint32_t)(k - 1) >= nb) {
                if (k < 8) D0[k] = 0.0f;
                else D1[k - 8] = 0.0f;
            }
        }
    }
    float P0[8], P1[8];
    idct8<8, true>(D0, P0);
    if constexpr (NFIX == 0) idct8<8, false>(D1, P1);
    else idct8<NFIX - 7, false>(D1, P1);      // row 1: entries 0..n-8 can be non-zero
    // BETA of the three position classes
    const float spread = guard_sqrt((float)(64u * Q - S * S));
    const float base = fmaf(qp.g_sum, (float)S, qp.g_delta);
    const float beta_ee = fmaf(qp.g_resid, spread, base);
    const float beta_ce = fmaf(qp.g_resid * (float)(SVS_GUARD_KE_CE / SVS_GUARD_KE), spread, base);
    const float beta_cc = fmaf(qp.g_resid * (float)(SVS_GUARD_KE_CC / SVS_GUARD_KE), spread, base);
    // Per pixel: ch = change(y, x) - (1/2 - 2^-16), small in magnitude, so its two roundings are part of KD's model of this
    // sparse inverse and the test below sees the prediction itself, not a copy rounded at the magnitude of the pixel (round
    // 3 tested pixel + change and paid 2^-14 of BETA for it).  The change lies ON the integer grid where fract(ch) = kMid.
    // The byte is v_cvt_pk_u8_f32(pixel + ch): round-to-nearest-even of a value whose fractional part is at least BETA - and
    // BETA >= 2^-14 (make_guard) - away from the rounding boundary, while the sum's own rounding is at most 2^-16: the floor.
    constexpr float kHalf = 0.5f - 0x1p-16f;
    constexpr float kMid = 0.5f + 0x1p-16f;
    float near_cc = 1.0f, near_ce = 1.0f, near_ee = 1.0f;   // smallest distance from the grid per position class
    uint32_t sx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sy[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // !INPLACE: the stego rows, committed only if decided
    uint32_t (&nx)[8] = INPLACE ? rx : sx;
    uint32_t (&ny)[8] = INPLACE ? ry : sy;
    const float ck[4] = {SVS_C1, SVS_C3, SVS_C5, SVS_C7};
    // CLS_C: class of this column's pixels in rows 0, 3, 4, 7;  CLS_E: in rows 1, 2, 5, 6
#define SVS_PREDCOL(X, W, NW, B, CLS_C, CLS_E)                                                  \
    {                                                                                           \
        const float e = fmaf(P0[X], SVS_A0, -kHalf), p1 = P1[X];                                \
        _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                         \
            const float ca = fmaf(p1, ck[y], e), cb = fmaf(p1, -ck[y], e);                      \
            if (y == 0 || y == 3) CLS_C = fmin3_abs(CLS_C, fract_f32(ca) - kMid, fract_f32(cb) - kMid); \
            else CLS_E = fmin3_abs(CLS_E, fract_f32(ca) - kMid, fract_f32(cb) - kMid);          \
            NW[y] = put_pixel_rne<B>(ca + ubyte_to_float<B>(W[y]), NW[y]);                      \
            NW[7 - y] = put_pixel_rne<B>(cb + ubyte_to_float<B>(W[7 - y]), NW[7 - y]);          \
        }                                                                                       \
    }
    SVS_PREDCOL(0, rx, nx, 0, near_cc, near_ce) SVS_PREDCOL(1, rx, nx, 1, near_ce, near_ee) SVS_PREDCOL(2, rx, nx, 2, near_ce, near_ee)
    SVS_PREDCOL(3, rx, nx, 3, near_cc, near_ce) SVS_PREDCOL(4, ry, ny, 0, near_cc, near_ce) SVS_PREDCOL(5, ry, ny, 1, near_ce, near_ee)
    SVS_PREDCOL(6, ry, ny, 2, near_ce, near_ee) SVS_PREDCOL(7, ry, ny, 3, near_cc, near_ce)
#undef SVS_PREDCOL
    const bool undecided = nb > 0 && !(near_cc >= beta_cc && near_ce >= beta_ce && near_ee >= beta_ee);
    if constexpr (!INPLACE) {
        if (undecided) return true;   // the block keeps its original pixels
#pragma unroll
        for (int r = 0; r < 8; ++r) { rx[r] = nx[r]; ry[r] = ny[r]; }
    }
    return undecided;
}

// BETA's coefficients for `rows` coefficient rows (1 or 2), rounded up; host side
inline void make_guard(double delta, int rows, QimParams *qp) {
    const double up = 1.0 + 0x1p-18;   // float evaluation of BETA in the kernel: conversions, sqrt, two FMAs
    const double kd = rows <= 1 ? SVS_GUARD_KD_U1 : SVS_GUARD_KD_U2;
    qp->g_sum = (float)(SVS_GUARD_UEFF * SVS_GUARD_KDC / 64.0 * up);
    qp->g_resid = (float)(SVS_GUARD_UEFF * SVS_GUARD_KE / 8.0 * up);
    double gd = SVS_GUARD_UEFF * kd * (1.5 * delta + 0.01) + 0x1p-20 + 0x1p-22;
    // two rows: the byte comes from v_cvt_pk_u8_f32's own rounding of pixel + (change - (1/2 - 2^-16)), which is the floor only
    // while the change's fractional part stays 2^-15 below 1 (embed_block_guarded2) - a FLOOR under BETA, not a term of it
    if (rows >= 2 && gd < 0x1p-14) gd = 0x1p-14;
    qp->g_delta = (float)(gd * up);
}

// EXACT embed of TWO horizontally adjacent blocks at once: every value is a pair (block A, block B) and every transform
// instruction a packed-FP32 one (v_pk_add / v_pk_mul / v_pk_fma_f32, each component rounding exactly like the scalar
// operation).  Unlike embed_block_exact - which pairs two LINES of one block and has to transpose 2x2 sub-blocks between
// the passes - the two blocks never exchange data, so the packed form costs no extra instructions: 2 x 928 scalar transform
// operations become 928 packed ones.  Same results as two embed_block_exact calls, bit for bit.
template <int U, int QM>
SVS_HD void embed_block_exact_pair(uint32_t (&ax)[8], uint32_t (&ay)[8], uint32_t (&bx)[8], uint32_t (&by)[8], uint32_t n,
                                   uint32_t nb_a, uint32_t nb_b, uint32_t hi_a, uint32_t lo_a, uint32_t hi_b, uint32_t lo_b,
                                   const QimParams &qp) {
    using pf::f32x2;
    f32x2 D[8][8];  // D[u][v] = (coefficient of A, coefficient of B)
    {
        f32x2 V[8][8];  // after the vertical pass: V[u][x]
#define SVS_COL2(X, WA, WB, B)                                                          \
    {                                                                                   \
        f32x2 col[8], out[8];                                                           \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                 \
            const f32x2 t = {ubyte_to_float<B>(WA[r]), ubyte_to_float<B>(WB[r])};       \
            col[r] = t;                                                                 \
        }                                                                               \
        pf::dct2_8(col, out);                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) V[u][X] = out[u];                 \
        SVS_SCHED_FENCE();                                                              \
    }
        SVS_COL2(0, ax, bx, 0) SVS_COL2(1, ax, bx, 1) SVS_COL2(2, ax, bx, 2) SVS_COL2(3, ax, bx, 3)
        SVS_COL2(4, ay, by, 0) SVS_COL2(5, ay, by, 1) SVS_COL2(6, ay, by, 2) SVS_COL2(7, ay, by, 3)
#undef SVS_COL2
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            pf::dct2_8(V[u], D[u]);
            SVS_SCHED_FENCE();
        }
    }
#pragma unroll
    for (int k = 1; k < 8 * U; ++k) {
        if ((uint32_t)k <= n) {  // wave-uniform
            const int i = k - 1;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int bit = (int)window_bit(which ? hi_b : hi_a, which ? lo_b : lo_a, i);
                const float c = D[k >> 3][k & 7][which];
                int q = quant_index<QM>(c, qp);
                q = force_parity(q, bit);
                float cn;
                if constexpr (QM == QM_DOUBLE) cn = (float)((double)q * qp.delta_d);
                else cn = (float)q * qp.delta_f;
                D[k >> 3][k & 7][which] = ((uint32_t)i < (which ? nb_b : nb_a)) ? cn : c;
            }
        }
    }
    f32x2 P[8][8];  // after the vertical inverse (axis 0 first, :168): P[y][v]
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        f32x2 col[8], out[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) col[u] = D[u][v];
        pf::dct3_8(col, out);
#pragma unroll
        for (int y = 0; y < 8; ++y) P[y][v] = out[y];
        SVS_SCHED_FENCE();
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        f32x2 px[8];
        pf::dct3_8(P[y], px);
        // np.uint8(np.clip(v, 0, 255)): clip, then truncate
        store_row_trunc(px[0][0], px[1][0], px[2][0], px[3][0], px[4][0], px[5][0], px[6][0], px[7][0], ax[y], ay[y]);
        store_row_trunc(px[0][1], px[1][1], px[2][1], px[3][1], px[4][1], px[5][1], px[6][1], px[7][1], bx[y], by[y]);
        SVS_SCHED_FENCE();
    }
}

// EXACT extract: parity bits of round(c_k/delta) with pocketfft-identical coefficients
template <int U, int QM>
SVS_HD void extract_block_exact(const uint32_t (&rx)[8], const uint32_t (&ry)[8], uint32_t n, const QimParams &qp,
                                uint32_t &hi, uint32_t &lo) {
    float D[8][8];
    forward_exact(rx, ry, D);
    hi = 0;
    lo = 0;
#pragma unroll
    for (int k = 1; k < 8 * U; ++k) {
        if ((uint32_t)k <= n) {  // wave-uniform
            const uint32_t bit = (uint32_t)quant_index<QM>(D[k >> 3][k & 7], qp) & 1u;
            const int i = k - 1;
            if (i < 32) hi |= bit << ((31 - i) & 31);
            else lo |= bit << ((63 - i) & 31);
        }
    }
}

inline int rows_for(int n) { return (n >> 3) + 1; }  // coefficient rows holding flat indices 1..n

// quantiser parameters and the cheapest exact evaluation mode for this delta (QuantMode); host side
inline int make_qim(double delta, QimParams *qp) {
    qp->delta_f = (float)delta;
    qp->inv_delta_f = 1.0f / qp->delta_f;
    qp->delta_d = delta;
    qp->tie_slope = (float)(SVS_TIE_SLOPE / (double)qp->delta_f) * 1.0000002f;  // rounded up
    {
        const double ueff = 5.9604644775390625e-8 * (1.0 + 0.0009765625), up = (1.0 + 0x1p-18) / (double)qp->delta_f;
        qp->tie2_sum = (float)(ueff * SVS_TIE2_KDC / 64.0 * up);
        qp->tie2_resid = (float)(ueff * SVS_TIE2_KE / 8.0 * up);
        qp->tie2_c00 = (float)(8.0 * 5.9604644775390625e-8 * up);   // the roundings of the quantiser input itself, |c| <= 2 c00
        // upper bound of extract_block_settle's margin over every pixel sum S in [0, 16320] (scan + slack for the scan step,
        // the float evaluation and the two 1.000001 factors)
        double worst = 0.0;
        for (int S = 0; S <= 16320; S += 8) {
            const double m = (double)qp->tie2_sum * S + (double)qp->tie2_resid * sqrt((double)S * (16320.0 - S)) +
                             (double)qp->tie2_c00 * S / 8.0 + 0x1p-20;
            if (m > worst) worst = m;
        }
        qp->tie2_max = (float)(worst * 1.001);
    }
    qp->g_sum = qp->g_resid = qp->g_delta = 0.0f;   // make_guard
    if ((double)qp->delta_f != delta) return QM_DOUBLE;
    int e = 0;
    const bool pow2 = frexp(delta, &e) == 0.5 && e > -100 && e < 100;
    return pow2 ? QM_POW2 : QM_F32;
}

}  // namespace svs
