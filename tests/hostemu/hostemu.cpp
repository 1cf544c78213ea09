// hostemu.cpp - TEST INFRASTRUCTURE ONLY (never loaded by the product package).
//
// Runs csrc/svs_block.hpp - the exact per-block arithmetic and bit bookkeeping the gfx950 kernels
// execute - on the CPU, block by block, so that the CPU-only test tier (and ASan/UBSan) can check
// it against the oracle without a GPU.  The lane/wave mapping, HBM access and LDS bit packing of
// the kernels are NOT modelled here; those are covered by the -m gpu tests.
// Build: g++ -O2 -ffp-contract=off -std=c++17 -shared -fPIC -I<csrc> hostemu.cpp -o libsvs_hostemu.so
#include <cmath>
#include <cstdint>
#include <cstring>

#include "svs_block.hpp"
#include "svs_stage.hpp"

namespace {

using svs::make_qim;   // the host-side parameter set-up the library itself uses

struct Blk {
    uint32_t x[8], y[8];
    void load(const uint8_t *p, size_t pitch) {
        for (int r = 0; r < 8; ++r) {
            std::memcpy(&x[r], p + r * pitch, 4);
            std::memcpy(&y[r], p + r * pitch + 4, 4);
        }
    }
    void store(uint8_t *p, size_t pitch) const {
        for (int r = 0; r < 8; ++r) {
            std::memcpy(p + r * pitch, &x[r], 4);
            std::memcpy(p + r * pitch + 4, &y[r], 4);
        }
    }
};

// two adjacent blocks through the packed pair form (what embed_exact_pair_kernel runs)
void embed_exact_pair_dispatch(Blk &a, Blk &b, uint32_t n, uint32_t nb_a, uint32_t nb_b, uint32_t hi_a, uint32_t lo_a,
                               uint32_t hi_b, uint32_t lo_b, const svs::QimParams &qp, int qm) {
    if (qm == svs::QM_DOUBLE) svs::embed_block_exact_pair<8, svs::QM_DOUBLE>(a.x, a.y, b.x, b.y, n, nb_a, nb_b, hi_a, lo_a, hi_b, lo_b, qp);
    else if (qm == svs::QM_POW2) svs::embed_block_exact_pair<8, svs::QM_POW2>(a.x, a.y, b.x, b.y, n, nb_a, nb_b, hi_a, lo_a, hi_b, lo_b, qp);
    else svs::embed_block_exact_pair<8, svs::QM_F32>(a.x, a.y, b.x, b.y, n, nb_a, nb_b, hi_a, lo_a, hi_b, lo_b, qp);
}

bool g_constant_shortcut = false;   // exact == 3: constant blocks take forward_exact_paired_constant, as the replay kernel does

bool g_generic_guarded2 = false;    // exact == 5: the two-row guard through its generic instantiation only

// Two coefficient rows: the instantiation the KERNEL launches (csrc/svs_device.hpp embed_kernel, svs_capi.hip launch_embed) -
// compile-time n for the GUI's default 10, and for every quantiser but the power-of-two one the in-place form (truncated
// inverse, stego bytes written over the row dwords, an undecided block left half-written: the kernel rebuilds it from the
// rows it parked in LDS, the caller here from the frame).  exact == 5 forces <QM, 0, false> so that the CPU tier can hold
// the two against each other.
template <int QM>
bool guarded2_as_launched(Blk &raw, uint32_t n, uint32_t nb, uint32_t hi, uint32_t lo, const svs::QimParams &qp) {
    if (g_generic_guarded2) return svs::embed_block_guarded2<QM, 0, false>(raw.x, raw.y, n, nb, hi, lo, qp);
    constexpr bool INPLACE = QM != svs::QM_POW2;
    if (n == 10) return svs::embed_block_guarded2<QM, 10, INPLACE>(raw.x, raw.y, n, nb, hi, lo, qp);
    return svs::embed_block_guarded2<QM, 0, INPLACE>(raw.x, raw.y, n, nb, hi, lo, qp);
}

// GUARDED (exact == 4 / 5; FAST too at n <= 15): -> true when the block has to be redone with the exact arithmetic
bool embed_guarded_dispatch(Blk &raw, uint32_t n, uint32_t nb, uint32_t hi, uint32_t lo, const svs::QimParams &qp, int qm) {
    if (svs::rows_for((int)n) == 2) {   // two coefficient rows: the per-pixel rigorous guard
        if (qm == svs::QM_DOUBLE) return guarded2_as_launched<svs::QM_DOUBLE>(raw, n, nb, hi, lo, qp);
        if (qm == svs::QM_POW2) return guarded2_as_launched<svs::QM_POW2>(raw, n, nb, hi, lo, qp);
        return guarded2_as_launched<svs::QM_F32>(raw, n, nb, hi, lo, qp);
    }
    if (qm == svs::QM_DOUBLE) return svs::embed_block_guarded<svs::QM_DOUBLE>(raw.x, raw.y, n, nb, hi, lo, qp);
    if (qm == svs::QM_POW2) return svs::embed_block_guarded<svs::QM_POW2>(raw.x, raw.y, n, nb, hi, lo, qp);
    return svs::embed_block_guarded<svs::QM_F32>(raw.x, raw.y, n, nb, hi, lo, qp);
}

void embed_exact_dispatch(Blk &raw, uint32_t n, uint32_t nb, uint32_t hi, uint32_t lo, const svs::QimParams &qp, int qm) {
    bool constant = g_constant_shortcut;
    for (int r = 0; r < 8 && constant; ++r)
        constant = raw.x[r] == (raw.x[0] & 0xffu) * 0x01010101u && raw.y[r] == (raw.x[0] & 0xffu) * 0x01010101u;
    if (qm == svs::QM_DOUBLE) svs::embed_block_exact<8, svs::QM_DOUBLE>(raw.x, raw.y, n, nb, hi, lo, qp, constant);
    else if (qm == svs::QM_POW2) svs::embed_block_exact<8, svs::QM_POW2>(raw.x, raw.y, n, nb, hi, lo, qp, constant);
    else svs::embed_block_exact<8, svs::QM_F32>(raw.x, raw.y, n, nb, hi, lo, qp, constant);
}

// -> true: some quantiser input is within the forward error bound of a tie (svs::extract_block's return value)
template <int QM>
bool extract_dispatch(int rows, const Blk &raw, uint32_t n, const svs::QimParams &d, uint32_t &hi, uint32_t &lo) {
    if (n == 10) return svs::extract_block<2, QM, 10>(raw.x, raw.y, n, d, hi, lo);   // as csrc/svs_capi.hip
    switch (rows) {
        case 1: return svs::extract_block<1, QM>(raw.x, raw.y, n, d, hi, lo);
        case 2: return svs::extract_block<2, QM>(raw.x, raw.y, n, d, hi, lo);
        case 3: return svs::extract_block<3, QM>(raw.x, raw.y, n, d, hi, lo);
        case 4: return svs::extract_block<4, QM>(raw.x, raw.y, n, d, hi, lo);
        case 5: return svs::extract_block<5, QM>(raw.x, raw.y, n, d, hi, lo);
        case 6: return svs::extract_block<6, QM>(raw.x, raw.y, n, d, hi, lo);
        case 7: return svs::extract_block<7, QM>(raw.x, raw.y, n, d, hi, lo);
        default: return svs::extract_block<8, QM>(raw.x, raw.y, n, d, hi, lo);
    }
}

template <int QM>
void extract_fast(int rows, const Blk &raw, uint32_t n, const svs::QimParams &d, uint32_t &hi, uint32_t &lo, uint64_t *redone) {
    if (extract_dispatch<QM>(rows, raw, n, d, hi, lo)) {   // near a tie: the kernels redo the block exactly
        svs::extract_block_exact<8, QM>(raw.x, raw.y, n, d, hi, lo);
        if (redone) ++*redone;
    }
}

}  // namespace

extern "C" {

// frames: contiguous [F][H][W]; bits: packed MSB-first, padded by the caller to a multiple of 4 bytes
uint64_t emu_embed(const uint8_t *gray, uint8_t *stego, int F, int H, int W, double delta, int n_ac,
                   const uint8_t *bits, uint64_t bits_bytes, uint64_t bit_offset, uint64_t n_bits, int exact,
                   uint64_t *n_replayed) {
    if (n_replayed) *n_replayed = 0;
    g_constant_shortcut = exact == 3;
    g_generic_guarded2 = exact == 5;
    if (exact == 5) exact = 4;
    const int n = n_ac < 0 ? 0 : (n_ac > 63 ? 63 : n_ac);
    const uint64_t bpf = (uint64_t)(H / 8) * (W / 8), total = bpf * F;
    std::memcpy(stego, gray, (size_t)F * H * W);
    uint64_t use = n_bits < total * n ? n_bits : total * n;
    if (!(delta > 0.0) || n == 0) use = 0;
    svs::QimParams qp;
    const int dbl = make_qim(use ? delta : 1.0, &qp);
    // exact == 0 (flags 0) and exact == 4 (SVS_EXACT_GUARDED): the cheap path wherever its rigorous error bound decides every
    // pixel, the exact arithmetic elsewhere (same routing as svs_embed_dev: one or two coefficient rows, delta inside the
    // guard's range; anything else is plain EXACT).  Every mode gives the reference's pixels.
    const bool in_range = delta >= SVS_GUARD_DELTA_MIN && delta <= SVS_GUARD_DELTA_MAX;
    const bool guarded = use > 0 && in_range && (exact == 4 || exact == 0) && svs::rows_for(n) <= 2;
    if (guarded) svs::make_guard(delta, svs::rows_for(n), &qp);
    if ((exact == 4 || exact == 0) && !guarded) exact = 1;
    if (use == 0) {
        if (n_bits > 0) {  // nothing consumed -> every block entered and round-tripped (either mode: svs_embed_dev)
            for (uint64_t gb = 0; gb < total; ++gb) {
                const uint64_t f = gb / bpf, b = gb % bpf;
                uint8_t *p = stego + f * (uint64_t)H * W + (b / (W / 8)) * 8 * W + (b % (W / 8)) * 8;
                Blk raw;
                raw.load(p, (size_t)W);
                svs::embed_block_exact<8, svs::QM_F32>(raw.x, raw.y, 0, 0, 0, 0, qp);
                raw.store(p, (size_t)W);
            }
        }
        return 0;
    }
    const uint32_t n_words = (uint32_t)(bits_bytes / 4);
    for (uint64_t gb = 0; gb < total; ++gb) {
        const uint64_t first = gb * n;
        const uint32_t nb = svs::block_budget(first, use, (uint32_t)n);
        if (nb == 0) break;
        const uint64_t f = gb / bpf, b = gb % bpf;
        const uint64_t by = b / (W / 8), bx = b % (W / 8);
        uint8_t *p = stego + f * (uint64_t)H * W + by * 8 * W + bx * 8;
        Blk raw;
        raw.load(p, (size_t)W);
        uint32_t hi, lo;
        svs::payload_window(reinterpret_cast<const uint32_t *>(bits), n_words, bit_offset + first, hi, lo);
        if (exact == 2 && (W / 8) % 2 == 0 && gb % 2 == 0 && svs::block_budget(first + n, use, (uint32_t)n) > 0) {
            // exact == 2: even/odd block pairs through the packed pair form, as embed_exact_pair_kernel does
            Blk other;
            other.load(p + 8, (size_t)W);
            uint32_t hi_b, lo_b;
            svs::payload_window(reinterpret_cast<const uint32_t *>(bits), n_words, bit_offset + first + n, hi_b, lo_b);
            embed_exact_pair_dispatch(raw, other, (uint32_t)n, nb, svs::block_budget(first + n, use, (uint32_t)n), hi, lo, hi_b,
                                      lo_b, qp, dbl);
            raw.store(p, (size_t)W);
            other.store(p + 8, (size_t)W);
            ++gb;
            continue;
        }
        if (guarded) {
            if (embed_guarded_dispatch(raw, (uint32_t)n, nb, hi, lo, qp, dbl)) {
                raw.load(p, (size_t)W);
                embed_exact_dispatch(raw, (uint32_t)n, nb, hi, lo, qp, dbl);
                if (n_replayed) ++*n_replayed;
            }
        } else {
            embed_exact_dispatch(raw, (uint32_t)n, nb, hi, lo, qp, dbl);
        }
        raw.store(p, (size_t)W);
    }
    return use;
}

// out_flags: one byte (0/1) per extracted bit, F*(H/8)*(W/8)*n entries
uint64_t emu_extract(const uint8_t *gray, int F, int H, int W, double delta, int n_ac, uint8_t *out_flags, int exact,
                     uint64_t *n_redone) {
    if (n_redone) *n_redone = 0;
    const int n = n_ac < 0 ? 0 : (n_ac > 63 ? 63 : n_ac);
    const uint64_t bpf = (uint64_t)(H / 8) * (W / 8), total = bpf * F;
    if (n == 0) return 0;
    if (!(delta > 0.0)) {
        std::memset(out_flags, 0, total * n);
        return total * n;
    }
    svs::QimParams qp;
    const int qm = make_qim(delta, &qp);
    for (uint64_t gb = 0; gb < total; ++gb) {
        const uint64_t f = gb / bpf, b = gb % bpf;
        const uint64_t by = b / (W / 8), bx = b % (W / 8);
        const uint8_t *p = gray + f * (uint64_t)H * W + by * 8 * W + bx * 8;
        Blk raw;
        raw.load(p, (size_t)W);
        uint32_t hi, lo;
        if (exact || svs::rows_for(n) == 1 || (double)qp.delta_f < SVS_FAST_EXTRACT_DELTA_MIN) {   // as svs_extract_dev routes: one row
            // and tiny steps use the pocketfft-identical forward in both modes
            if (qm == svs::QM_POW2) svs::extract_block_exact<8, svs::QM_POW2>(raw.x, raw.y, (uint32_t)n, qp, hi, lo);
            else svs::extract_block_exact<8, svs::QM_F32>(raw.x, raw.y, (uint32_t)n, qp, hi, lo);
        } else if (qm == svs::QM_POW2) extract_fast<svs::QM_POW2>(svs::rows_for(n), raw, (uint32_t)n, qp, hi, lo, n_redone);
        else extract_fast<svs::QM_F32>(svs::rows_for(n), raw, (uint32_t)n, qp, hi, lo, n_redone);
        for (int i = 0; i < n; ++i) out_flags[gb * n + i] = (uint8_t)svs::window_bit(hi, lo, i);
    }
    return total * n;
}

// forward coefficients of one block (for the DCT accuracy test): D[8][8]
void emu_forward_block(const uint8_t *block64, float *D64) {
    Blk raw;
    raw.load(block64, 8);
    float D[8][8];
    svs::forward_rows<8>(raw.x, raw.y, D);
    std::memcpy(D64, D, sizeof D);
}

// number of (c, delta) pairs on which the reciprocal-multiply quantiser differs from the IEEE division
uint64_t emu_quant_mismatches(const float *c, uint64_t n, double delta) {
    svs::QimParams qp;
    make_qim(delta, &qp);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i)
        bad += svs::quant_index<svs::QM_F32>(c[i], qp) != svs::quant_index_by_division(c[i], qp.delta_f);
    return bad;
}

// pocketfft-identical 8-point transforms (type 2 / type 3, norm='ortho')
void emu_pf_dct2(const float *x, float *X) {
    float a[8], b[8];
    std::memcpy(a, x, sizeof a);
    svs::pf::dct2_8(a, b);
    std::memcpy(X, b, sizeof b);
}

void emu_pf_dct3(const float *X, float *x) {
    float a[8], b[8];
    std::memcpy(a, X, sizeof a);
    svs::pf::dct3_8(a, b);
    std::memcpy(x, b, sizeof b);
}

// rows 0 and 1 of the vertical pass as the two-row guarded kernel computes them (packed integer first stages) -> V[2][8]
void emu_vertical_pf01(const uint8_t *block64, float *V16) {
    Blk raw;
    raw.load(block64, 8);
    float a0[4], a1[4], b0[4], b1[4];
    uint32_t S = 0;
    svs::vertical_pf01_packed(raw.x, a0, a1, S);
    svs::vertical_pf01_packed(raw.y, b0, b1, S);
    for (int x = 0; x < 4; ++x) { V16[x] = a0[x]; V16[4 + x] = b0[x]; V16[8 + x] = a1[x]; V16[12 + x] = b1[x]; }
    V16[16] = (float)S;
}

// number of (c, bit) pairs on which the float-domain quantiser step (qim_change) differs from the integer form
uint64_t emu_qim_change_mismatches(const float *c, const uint8_t *bit, uint64_t n, double delta) {
    svs::QimParams qp;
    const int qm = make_qim(delta, &qp);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        float got, want;
        if (qm == svs::QM_POW2) {
            got = svs::qim_change<svs::QM_POW2>(c[i], bit[i], qp);
            want = (float)svs::force_parity(svs::quant_index<svs::QM_POW2>(c[i], qp), bit[i]) * qp.delta_f - c[i];
        } else if (qm == svs::QM_DOUBLE) {
            got = svs::qim_change<svs::QM_DOUBLE>(c[i], bit[i], qp);
            want = (float)((double)svs::force_parity(svs::quant_index_by_division(c[i], qp.delta_f), bit[i]) * qp.delta_d) - c[i];
        } else {
            got = svs::qim_change<svs::QM_F32>(c[i], bit[i], qp);
            want = (float)svs::force_parity(svs::quant_index_by_division(c[i], qp.delta_f), bit[i]) * qp.delta_f - c[i];
        }
        bad += std::memcmp(&got, &want, 4) != 0;
    }
    return bad;
}

// the chunk plan of the host-pointer entry points (csrc/svs_stage.hpp): -> number of chunks; out[4 k ..] = f0, nf, r0, rows
// of chunk k (at most max_chunks are written); target_bytes = 0 takes the built-in rule for a batch of total_bytes
uint64_t emu_plan_chunks(int32_t n_frames, int32_t H, uint64_t row_bytes, uint64_t total_bytes, uint64_t target_bytes, int32_t *out,
                         uint64_t max_chunks) {
    uint64_t k = 0;
    svs::for_each_chunk(n_frames, H, (size_t)row_bytes, (size_t)(target_bytes ? target_bytes : svs::stage_chunk_rule(total_bytes)),
                        [&](const svs::Chunk &c) {
                            if (k < max_chunks) { out[4 * k] = c.f0; out[4 * k + 1] = c.nf; out[4 * k + 2] = c.r0; out[4 * k + 3] = c.rows; }
                            ++k;
                        });
    return k;
}

uint64_t emu_chunk_budget(uint64_t pass_bits, uint64_t use, uint64_t g0, uint32_t n) { return svs::chunk_budget(pass_bits, use, g0, n); }

void emu_idct8(const float *X, float *x) {
    float a[8], b[8];
    std::memcpy(a, X, sizeof a);
    svs::idct8<8, false>(a, b);
    std::memcpy(x, b, sizeof b);
}

}  // extern "C"
