// pattern_copy.hip - why does the one-row embed launch run at 1.6 ms per 600 x 4K when a linear copy of the same bytes takes
// 1.49?  Copies with the embed kernel's ACCESS PATTERN (a lane owns two adjacent 8x8 blocks: eight rows of 16 bytes, rows one
// frame row apart; one contiguous eighth of the batch per XCD) and nothing else, at several resource footprints:
//   linear        16 bytes per lane, consecutive lanes consecutive addresses (the box's copy ceiling)
//   rows          the pattern, minimal registers, no LDS
//   rows+w5       the same, register-allocated for at most 5 waves per SIMD (embed_kernel: 98 VGPRs)
//   rows/1blk     one block per lane (8-byte accesses);  rows/4blk  four blocks per lane (32 bytes per row: 2 KB per wave and row)
//   rows pitch4096  the pattern on frames stored at a row pitch of 4096 bytes instead of 3840
//   rows identity / rows/4blk ident   the identity tile map instead of one eighth of the batch per XCD
//   lin32k eighth / identity   32 KB contiguous per workgroup (eight 16-byte accesses per lane, 4 KB apart), either tile map
// (second run: four blocks per lane as two 16-byte accesses 32 bytes apart is 3.2 ms - half-used lines per instruction)
// (first run, session r5j: 19 456 B of unused LDS per workgroup changes nothing, 1.685 vs 1.688; storing each row as it is
// loaded instead of after all eight loads is slower, 1.75)
// Measurement probe (not part of the library).  Build: hipcc --offload-arch=gfx950 -O3 -I<pkg>/csrc -I<repo>/include -o
// tools/probes/pattern_copy tools/probes/pattern_copy.hip ; run on the GPU box: tools/probes/pattern_copy [pairs]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "svs_device.hpp"

using svs::Geometry;
using svs::u32x2;
using svs::u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BPL, bool INTERLEAVED>
__device__ __forceinline__ void rows_copy(const uint8_t *src, uint8_t *dst, const Geometry &g) {
    const uint32_t gblock = (svs::tile_id(g.xcd_chunk) * 256u + threadIdx.x) * BPL;
    if (gblock >= g.total_blocks) return;
    const int64_t off = svs::block_offset(gblock, g);
    typename svs::RowVec<BPL>::type v[8];
    if constexpr (INTERLEAVED) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = __builtin_nontemporal_load(reinterpret_cast<const typename svs::RowVec<BPL>::type *>(src + off + r * g.row_pitch));
            if constexpr (BPL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + off + r * g.row_pitch), "v"(v[r]) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst + off + r * g.row_pitch), "v"(v[r]) : "memory");
        }
    } else {
        svs::load_rows<BPL>(src + off, g.row_pitch, v);
        svs::store_rows<BPL>(dst + off, g.row_pitch, v);
    }
}

template <int BPL, bool INTERLEAVED>
__global__ __launch_bounds__(256) void rows_kernel(const uint8_t *src, uint8_t *dst, const Geometry g) {
    extern __shared__ uint32_t unused_lds[];
    if (g.pad == 0x12345678u) unused_lds[threadIdx.x] = 1;      // never true: keeps the dynamic LDS allocation alive
    rows_copy<BPL, INTERLEAVED>(src, dst, g);
}

template <int BPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 5))) void rows_w5_kernel(const uint8_t *src, uint8_t *dst, const Geometry g) {
    extern __shared__ uint32_t unused_lds[];
    if (g.pad == 0x12345678u) unused_lds[threadIdx.x] = 1;
    rows_copy<BPL, false>(src, dst, g);
}

// four adjacent blocks per lane: 32 bytes per row (two 16-byte accesses), a wave's row segment is 2 KB
__global__ __launch_bounds__(256) void rows4_kernel(const uint8_t *src, uint8_t *dst, const Geometry g) {
    const uint32_t gblock = (svs::tile_id(g.xcd_chunk) * 256u + threadIdx.x) * 4u;
    if (gblock >= g.total_blocks) return;
    const int64_t off = svs::block_offset(gblock, g);
    u32x4 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off + r * g.row_pitch));
        b[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off + r * g.row_pitch + 16));
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + off + r * g.row_pitch), "v"(a[r]) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\ts_nop 1" ::"v"(dst + off + r * g.row_pitch), "v"(b[r]) : "memory");
    }
}

// the bytes of one embed tile (8 rows x 4 KB = 32 KB) as ONE contiguous stretch per workgroup: eight 16-byte loads per lane, 4 KB
// apart, then eight stores - what an LDS-staged kernel with linear global accesses would issue (workgroup = whole block rows)
__global__ __launch_bounds__(256) void linear32k_kernel(const uint8_t *src, uint8_t *dst, uint64_t bytes, uint32_t chunk) {
    const uint64_t base = (uint64_t)svs::tile_id(chunk) * 32768u + threadIdx.x * 16u;
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (base + i * 4096u + 16 <= bytes) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + base + i * 4096u));
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (base + i * 4096u + 16 <= bytes) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + base + i * 4096u), "v"(v[i]) : "memory");
}

__global__ __launch_bounds__(256) void linear_kernel(const u32x4 *src, u32x4 *dst, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + i), "v"(v) : "memory");
    }
}

static svs::FastDiv make_div(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    svs::FastDiv r;
    r.shift = 31 + l;
    r.mul = (uint32_t)(((1ull << r.shift) + d - 1) / d);
    r.div = d;
    r.pad = 0;
    return r;
}

int main(int argc, char **argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 3;
    const int F = 600, H = 2160, W = 3840;
    const uint64_t bytes = (uint64_t)F * H * W;
    Geometry g;
    const uint32_t wb = W / 8, bpf = wb * (H / 8);
    g.by_wb = make_div(wb);
    g.by_bpf = make_div(bpf);
    g.total_blocks = (uint32_t)((uint64_t)bpf * F);
    g.n_ac = 3;
    g.xcd_chunk = 0xFFFFFFFFu;
    g.pad = 0;
    g.row_pitch = W;
    g.frame_pitch = (int64_t)H * W;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t ev[11];
    for (auto &e : ev) CK(hipEventCreate(&e));
    Geometry gp = g;                 // the same frames at a row pitch of 4096 bytes: every row segment starts 4 KB aligned relative to its neighbours
    gp.row_pitch = 4096;
    gp.frame_pitch = (int64_t)H * 4096;
    const uint64_t bytes_p = (uint64_t)F * H * 4096;
    Geometry gi = g;                 // identity tile map
    gi.xcd_chunk = 0;
    const char *names[] = {"linear", "rows", "rows+w5", "rows/1blk", "lin32k eighth", "rows pitch4096", "rows identity", "lin32k identity"};
    printf("# 600 x 4K, 2 x 4.98 GB per launch: median ms over 9 launches in a burst (the first of 10 dropped), per (src, dst) allocation\n# pair");
    for (const char *n : names) printf(" %16s", n);
    printf("\n");
    std::vector<std::pair<uint8_t *, uint8_t *>> keep;
    for (int p = 0; p < pairs; ++p) {
        uint8_t *a, *b;
        CK(hipMalloc(&a, bytes_p));
        CK(hipMalloc(&b, bytes_p));
        CK(hipMemsetAsync(a, 0x5a, bytes_p, st));
        keep.push_back({a, b});
        printf("%6d", p);
        for (int k = 0; k < 8; ++k) {
            std::vector<float> ts;
            CK(hipEventRecord(ev[0], st));
            for (int rep = 0; rep < 10; ++rep) {      // a sustained burst: no synchronisation between the launches
                const uint32_t grid2 = (uint32_t)((g.total_blocks + 511) / 512), grid1 = (uint32_t)((g.total_blocks + 255) / 256);
                const uint32_t grid4 = (uint32_t)((g.total_blocks + 1023) / 1024);
                switch (k) {
                    case 0: hipLaunchKernelGGL(linear_kernel, dim3((uint32_t)((bytes / 16 + 255) / 256)), dim3(256), 0, st, (const u32x4 *)a, (u32x4 *)b, bytes / 16); break;
                    case 1: hipLaunchKernelGGL((rows_kernel<2, false>), dim3(grid2), dim3(256), 0, st, a, b, g); break;
                    case 2: hipLaunchKernelGGL((rows_w5_kernel<2>), dim3(grid2), dim3(256), 0, st, a, b, g); break;
                    case 3: hipLaunchKernelGGL((rows_kernel<1, false>), dim3(grid1), dim3(256), 0, st, a, b, g); break;
                    case 4: hipLaunchKernelGGL(linear32k_kernel, dim3((uint32_t)((bytes + 32767) / 32768)), dim3(256), 0, st, a, b, bytes, 0xFFFFFFFFu); break;
                    case 5: hipLaunchKernelGGL((rows_kernel<2, false>), dim3(grid2), dim3(256), 0, st, a, b, gp); break;
                    case 6: hipLaunchKernelGGL((rows_kernel<2, false>), dim3(grid2), dim3(256), 0, st, a, b, gi); break;
                    default: hipLaunchKernelGGL(linear32k_kernel, dim3((uint32_t)((bytes + 32767) / 32768)), dim3(256), 0, st, a, b, bytes, 0u); break;
                }
                CK(hipGetLastError());
                CK(hipEventRecord(ev[rep + 1], st));
            }
            CK(hipEventSynchronize(ev[10]));
            for (int rep = 1; rep < 10; ++rep) {
                float ms;
                CK(hipEventElapsedTime(&ms, ev[rep], ev[rep + 1]));
                ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf(" %16.4f", ts[ts.size() / 2]);
        }
        printf("\n");
    }
    uint8_t probe[64];
    CK(hipMemcpy(probe, keep[0].second, 64, hipMemcpyDeviceToHost));
    printf("# copied bytes ok: %s\n", probe[0] == 0x5a && probe[63] == 0x5a ? "yes" : "NO");
    return 0;
}
