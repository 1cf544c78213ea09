// valu_rate.hip - how many cycles one SIMD of gfx950 needs per wave64 vector instruction, for the instruction kinds the
// exact-mode kernels are made of, at 1..8 resident waves per SIMD.  Measurement probe (not part of the library).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/valu_rate tools/probes/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000, UNROLL = 16;   // 16 independent chains per wave: no dependent-issue stalls

template <int KIND>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cycles, float seed) {
    float a[UNROLL];
    f32x2 p[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i; p[i] = f32x2{seed + i, seed - i}; }
    const float c1 = seed * 0.5f, c2 = seed * 0.25f;
    const f32x2 q1 = {c1, c1}, q2 = {c2, c2};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
            else if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
            else if constexpr (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
            else if constexpr (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 6) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 7) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 8) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            // round 4: the instruction kinds of the guarded two-row embed kernel's phase 1
            else if constexpr (KIND == 10) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 11) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 13) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 15) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            else if constexpr (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 17) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 18) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 19) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 20) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 21) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 22) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 23) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 24) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c1) : "vcc");
            else if constexpr (KIND == 25) asm volatile("v_sub_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 26) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
            else if constexpr (KIND == 27) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i]));
            else if constexpr (KIND == 28) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 29) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 30) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 31) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 33) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 34) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 35) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 36) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 37) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 38) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 39) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 40) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 41) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 42) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 43) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 44) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 45) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 46) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 47) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 48) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            else if constexpr (KIND == 49) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c1) : "s20", "s21");
            else if constexpr (KIND == 50) asm volatile("v_sub_f32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 51) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 52) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(c1), "v"(c2));
            else if constexpr (KIND == 53) asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(a[i]));
            else if constexpr (KIND == 54) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
            else if constexpr (KIND == 55) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int cus) {
    printf("%-16s", name);
    for (int waves_per_simd : {2, 4, 8}) {
        const int blocks = cus * 4 * waves_per_simd;          // 64-thread workgroups: one wave each
        float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * 64 * blocks);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<KIND><<<blocks, 64>>>(out, cyc, 1.0f);           // warm-up
        hipEventRecord(e0);
        probe<KIND><<<blocks, 64>>>(out, cyc, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions issued per SIMD = waves_per_simd * ITER * UNROLL; kernel time -> ns per instruction per SIMD
        const double ns_per_instr = ms * 1e6 / ((double)waves_per_simd * ITER * UNROLL);
        printf("  %dw: %5.2f ns", waves_per_simd, ns_per_instr);
        hipFree(out); hipFree(cyc);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz; ns per wave64 instruction per SIMD (kernel time / instructions issued on one SIMD)\n",
           prop.gcnArchName, cus, prop.clockRate / 1000);
    run<0>("v_add_f32", cus); run<1>("v_fma_f32", cus); run<5>("v_mul_f32", cus); run<2>("v_pk_add_f32", cus);
    run<3>("v_pk_fma_f32", cus); run<4>("v_pk_mul_f32", cus); run<6>("v_cvt_f32_ubyte0", cus); run<7>("v_pk_add_u16", cus);
    run<8>("v_floor_f32", cus); run<9>("v_add_u32", cus);
    run<10>("v_fract_f32", cus); run<11>("v_rndne_f32", cus); run<12>("v_min_f32", cus); run<13>("v_min3_f32 |.|", cus);
    run<14>("v_and_b32", cus); run<15>("v_bfe_u32", cus); run<16>("v_perm_b32", cus); run<17>("v_cvt_pk_u8_f32", cus);
    run<18>("v_cvt_f32_i32", cus); run<19>("v_cvt_i32_f32", cus); run<20>("v_pk_sub_i16", cus); run<21>("v_dot4_u32_u8", cus);
    run<22>("v_bfi_b32", cus); run<23>("v_cndmask_b32", cus); run<24>("v_cmp_lt_f32", cus); run<25>("v_sub_f32 |.|", cus);
    run<26>("v_cvt_f32_i32 sdwa", cus); run<27>("v_lshrrev_b32", cus); run<28>("v_and_or_b32", cus); run<29>("v_max_f32 |.|", cus);
    run<30>("v_sub_u32", cus); run<31>("v_add3_u32", cus); run<32>("v_fmac_f32", cus); run<33>("v_mad_u32_u24", cus);
    run<34>("v_cvt_flr_i32_f32", cus); run<35>("v_pk_fma_f16", cus); run<36>("v_med3_f32", cus); run<37>("v_xor_b32", cus);
    run<38>("v_lshl_or_b32", cus); run<39>("v_cvt_f32_u32", cus); run<40>("v_pk_add_i16", cus); run<41>("v_sad_u8", cus);
    run<42>("v_mov_b32", cus);
    run<43>("v_sat_pk_u8_i16", cus); run<44>("v_alignbit_b32", cus); run<45>("v_pk_max_i16", cus); run<46>("v_max_u32", cus);
    run<47>("v_or_b32", cus); run<48>("v_lshlrev_b32", cus); run<49>("v_cndmask sgpr", cus); run<50>("v_sub_f32 clamp", cus);
    run<51>("v_mul_f32 const", cus); run<52>("v_fma_f32 clamp", cus); run<53>("v_cvt_f32_ubyte2", cus); run<54>("v_mul_u32_u24", cus);
    run<55>("v_ashrrev_i32", cus);
    return 0;
}
