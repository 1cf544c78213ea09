// cvt_round_mode.hip - does v_cvt_pk_u8_f32 follow the FP32 rounding mode of the MODE register on gfx950?
// (If it did, round-toward-minus-infinity would give floor + saturate + pack in one instruction.)  Measurement probe.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(const float *in, unsigned *out_rne, unsigned *out_down, unsigned *out_zero, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float v = in[i];
    unsigned a = 0, b = 0, c = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(a) : "v"(v));
    // MODE[1:0] = single-precision round mode: 0 nearest even, 1 +inf, 2 -inf, 3 toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=v"(b) : "v"(v));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=v"(c) : "v"(v));
    out_rne[i] = a; out_down[i] = b; out_zero[i] = c;
}

int main() {
    const float h[] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.999f, 3.0f, 126.99999f, 127.5f, 254.5f, 254.9f, 255.0f, 255.7f, 300.0f,
                       -0.3f, -0.5f, -0.9f, -1.5f, 1e-9f, 0.99999994f, 1.0000001f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *a, *b, *c;
    hipMalloc(&d, sizeof h); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&c, 4 * n);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, a, b, c, n);
    unsigned ha[64], hb[64], hc[64];
    hipMemcpy(ha, a, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 4 * n, hipMemcpyDeviceToHost);
    printf("%-14s %8s %8s %8s\n", "input", "nearest", "-inf", "zero");
    for (int i = 0; i < n; ++i) printf("%-14.8g %8u %8u %8u\n", h[i], ha[i] & 255, hb[i] & 255, hc[i] & 255);
    return 0;
}
