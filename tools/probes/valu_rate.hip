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
    for (int waves_per_simd : {1, 2, 3, 4, 6, 8}) {
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
    return 0;
}
