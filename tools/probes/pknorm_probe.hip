// pknorm_probe.hip - semantics of v_cvt_pknorm_i16_f32 and v_sat_pk_u8_i16 on gfx950 (candidate store path of the two-row
// FAST embed: two changes -> i16x2 in one instruction, packed add to the pixel pair, saturating pack).  Measurement probe.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float *in, int *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = in[i] * (1.0f / 32767.0f), b = (in[i] + 1.0f) * (1.0f / 32767.0f);
    unsigned r, s;
    asm volatile("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    const unsigned pix = 0x00fe0003u;   // pixel pair (3, 254) as u16 lanes
    unsigned sum;
    asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(sum) : "v"(r), "v"(pix));
    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(s) : "v"(sum));
    out[3 * i] = (int)(short)(r & 0xffff);
    out[3 * i + 1] = (int)(short)(r >> 16);
    out[3 * i + 2] = (int)s;
}
int main() {
    const float h[] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, -0.4f, -0.5f, -0.6f, -1.5f, -2.5f, 7.25f, -7.75f, 19.999f, -3.0f, 100.5f, -100.5f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; int *o; hipMalloc(&d, sizeof h); hipMalloc(&o, 12 * n);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o, n);
    int ho[64 * 3]; hipMemcpy(ho, o, 12 * n, hipMemcpyDeviceToHost);
    printf("%-10s %8s %8s   sat_pk_u8_i16(pixels (3,254) + pair)\n", "x", "norm(x)", "norm(x+1)");
    for (int i = 0; i < n; ++i) printf("%-10.4f %8d %8d   lo=%d hi=%d (raw 0x%08x)\n", h[i], ho[3 * i], ho[3 * i + 1], ho[3 * i + 2] & 255, (ho[3 * i + 2] >> 8) & 255, ho[3 * i + 2]);
    return 0;
}
