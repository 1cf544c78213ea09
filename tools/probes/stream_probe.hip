// stream_probe.hip - round 6: which property of the one-row embed kernel's access pattern makes it sensitive to where its two
// buffers were placed, when a linear copy of the same bytes is not?  Pure copies of a 600 x 2160 x 3840 byte batch in which a
// lane moves R rows of 16 bytes, rows one frame row (3840 B) apart:
//     R = 8 is the embed kernel's pattern (two adjacent 8x8 blocks per lane), R = 1 the linear copy, R = 2, 4 in between;
//     tile map: identity, or one contiguous eighth of the grid per XCD (the embed kernel's);
//     order: all R loads, then all R stores (the kernel's) or row by row;
//     cap: workgroups per CU (= waves per SIMD) through unused dynamic LDS.
// Every configuration runs on the same P buffer pairs (hipMalloc, all resident): median ms per launch in bursts.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_probe tools/probes/stream_probe.hip ; run: tools/probes/stream_probe [pairs]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t PITCH = 3840, STRIPS = PITCH / 16;
constexpr uint64_t ROWS = 600ull * 2160ull, BYTES = ROWS * PITCH;

__device__ __forceinline__ uint32_t tile_of(uint32_t map) {
    const uint32_t i = blockIdx.x;
    if (map == 0) return i;
    const uint32_t n = gridDim.x, q = n / 8u, r = n % 8u, x = i % 8u;
    const uint32_t base = x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q, len = q + (x < r ? 1u : 0u);
    const uint32_t stagger = map >> 4;     // every XCD starts x * stagger tiles into its eighth (and wraps)
    return base + (i / 8u + x * stagger) % len;
}

template <int R, int ORDER, int WG = 256>
__global__ __launch_bounds__(WG) void rows_kernel(const uint8_t *src, uint8_t *dst, uint32_t map, uint32_t lanes_total, uint32_t never) {
    extern __shared__ uint32_t pad[];
    if (never == 0x12345678u) pad[threadIdx.x] = 1;
    const uint32_t L = tile_of(map) * (uint32_t)WG + threadIdx.x;
    if (L >= lanes_total) return;
    const uint32_t grp = L / STRIPS, s = L - grp * STRIPS;
    const int64_t off = (int64_t)grp * R * PITCH + s * 16;
    u32x4 v[R];
    if constexpr (ORDER == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off + r * (int64_t)PITCH));
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + off + r * (int64_t)PITCH), "v"(v[r]) : "memory");
    } else if constexpr (ORDER == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off + r * (int64_t)PITCH));
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + off + r * (int64_t)PITCH), "v"(v[r]) : "memory");
        }
    } else if constexpr (ORDER == 2) {   // read only: the rows are folded into one dword that is (never) stored
        u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < R; ++r) acc ^= __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off + r * (int64_t)PITCH));
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u && never == 1u) dst[off] = 1;
    } else {                             // write only
        const u32x4 c = {L, map, lanes_total, never};
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + off + r * (int64_t)PITCH), "v"(c) : "memory");
    }
}

// A 32 KB tile of 8 rows x 4096 B per workgroup of 256 (pitch 4096: one workgroup = one group of rows), moved two ways:
//   TRANSPOSED = 0  a lane moves 16 B of each of the 8 rows (the embed kernel's ownership: a wave's 8 accesses are 4096 B apart)
//   TRANSPOSED = 1  wave w moves rows 2w and 2w + 1 whole: its 8 accesses are 8 consecutive KB (what a kernel that passes its
//                   rows through an LDS transposition would issue) - the same bytes per workgroup, per wave and per instruction
template <int TRANSPOSED>
__global__ __launch_bounds__(256) void tile32k_kernel(const uint8_t *src, uint8_t *dst, uint32_t map, uint32_t tiles, uint32_t never) {
    extern __shared__ uint32_t pad[];
    if (never == 0x12345678u) pad[threadIdx.x] = 1;
    const uint32_t tile = tile_of(map);
    if (tile >= tiles) return;
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const int64_t base = (int64_t)tile * 32768;
    u32x4 v[8];
    int64_t off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        off[j] = TRANSPOSED ? base + (int64_t)(2 * w + j / 4) * 4096 + (64 * (j % 4) + l) * 16 : base + (int64_t)j * 4096 + threadIdx.x * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off[j]));
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + off[j]), "v"(v[j]) : "memory");
}

struct Cfg { const char *name; int R, order, map, cap, wg; };   // wg: threads per workgroup (0 = 256); cap counts workgroups per CU

template <int R, int ORDER, int WG = 256>
void launch(const uint8_t *s, uint8_t *d, int map, int cap) {
    const uint32_t lanes = (uint32_t)(BYTES / (16ull * R));
    const uint32_t lds = cap ? ((160u * 1024u / cap) & ~255u) : 0u;
    hipLaunchKernelGGL((rows_kernel<R, ORDER, WG>), dim3((lanes + WG - 1) / WG), dim3(WG), lds, 0, s, d, (uint32_t)map, lanes, 0u);
}
void go(const Cfg &c, const uint8_t *s, uint8_t *d) {
    if (c.order >= 10) {      // 32 KB tiles at pitch 4096: 10 = lane owns 8 rows, 11 = wave owns 2 whole rows
        const uint32_t tiles = (uint32_t)(BYTES / 32768);
        const uint32_t lds = c.cap ? ((160u * 1024u / c.cap) & ~255u) : 0u;
        if (c.order == 10) hipLaunchKernelGGL((tile32k_kernel<0>), dim3(tiles), dim3(256), lds, 0, s, d, (uint32_t)c.map, tiles, 0u);
        else hipLaunchKernelGGL((tile32k_kernel<1>), dim3(tiles), dim3(256), lds, 0, s, d, (uint32_t)c.map, tiles, 0u);
        return;
    }
    if (c.wg && c.R == 8 && c.order == 0) {
        switch (c.wg) {
            case 64: launch<8, 0, 64>(s, d, c.map, c.cap); return;
            case 128: launch<8, 0, 128>(s, d, c.map, c.cap); return;
            case 512: launch<8, 0, 512>(s, d, c.map, c.cap); return;
            case 1024: launch<8, 0, 1024>(s, d, c.map, c.cap); return;
        }
    }
#define GO(RR) if (c.R == RR) { if (c.order == 1) launch<RR, 1>(s, d, c.map, c.cap); else if (c.order == 2) launch<RR, 2>(s, d, c.map, c.cap); \
                      else if (c.order == 3) launch<RR, 3>(s, d, c.map, c.cap); else launch<RR, 0>(s, d, c.map, c.cap); }
    GO(1) GO(2) GO(4) GO(8) GO(16)
#undef GO
}

int main(int argc, char **argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 4, burst = 7;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;   // 0 hipMalloc, 1 fine-grained, 2 uncached, 3 physically contiguous
    const bool brief = argc > 3;
    std::vector<std::pair<uint8_t *, uint8_t *>> bufs;
    auto alloc = [&](uint8_t **p) {
        if (kind == 0) { CK(hipMalloc(p, BYTES)); return; }
        const unsigned flags = kind == 1 ? hipDeviceMallocFinegrained : kind == 2 ? hipDeviceMallocUncached : hipDeviceMallocContiguous;
        CK(hipExtMallocWithFlags((void **)p, BYTES, flags));
    };
    printf("# allocation kind %d (0 hipMalloc, 1 fine-grained, 2 uncached, 3 contiguous)\n", kind);
    for (int k = 0; k < pairs; ++k) {
        uint8_t *a, *b;
        alloc(&a); alloc(&b);
        CK(hipMemset(a, 17 + k, BYTES));
        bufs.push_back({a, b});
    }
    const Cfg cfgs[] = {
        {"R1 identity", 1, 0, 0, 0}, {"R1 eighth", 1, 0, 1, 0}, {"R2 eighth", 2, 0, 1, 0}, {"R4 eighth", 4, 0, 1, 0},
        {"R8 eighth", 8, 0, 1, 0}, {"R8 eighth cap4", 8, 0, 1, 4}, {"R8 eighth cap5", 8, 0, 1, 5}, {"R8 identity cap4", 8, 0, 0, 4},
        {"R8 eighth rowwise", 8, 1, 1, 0}, {"R8 eighth rowwise cap4", 8, 1, 1, 4}, {"R4 eighth cap4", 4, 0, 1, 4}, {"R4 eighth cap8", 4, 0, 1, 8},
        {"R2 eighth cap8", 2, 0, 1, 8}, {"R16 eighth cap4", 16, 0, 1, 4}, {"R16 eighth cap2", 16, 0, 1, 2}, {"R1 eighth cap4", 1, 0, 1, 4},
        {"R1 identity READ", 1, 2, 0, 0}, {"R8 eighth cap4 READ", 8, 2, 1, 4}, {"R8 eighth READ", 8, 2, 1, 0},
        {"R8 eighth wg64 cap16", 8, 0, 1, 16, 64}, {"R8 eighth wg128 cap8", 8, 0, 1, 8, 128}, {"R8 eighth wg512 cap2", 8, 0, 1, 2, 512},
        {"R8 eighth wg1024 cap1", 8, 0, 1, 1, 1024}, {"R8 identity wg64 cap16", 8, 0, 0, 16, 64}, {"R8 eighth wg64 cap32", 8, 0, 1, 32, 64},
        {"R8 e cap4 stagger 1", 8, 0, 1 | (1 << 4), 4}, {"R8 e cap4 stagger 8", 8, 0, 1 | (8 << 4), 4}, {"R8 e cap4 stagger 64", 8, 0, 1 | (64 << 4), 4},
        {"R8 e cap4 stagger 512", 8, 0, 1 | (512 << 4), 4}, {"R8 e cap4 stagger 2373", 8, 0, 1 | (2373 << 4), 4}, {"R8 e cap4 stagger 1187", 8, 0, 1 | (1187 << 4), 4},
        {"R8 e cap4 stagger 37", 8, 0, 1 | (37 << 4), 4}, {"R8 e cap4 stagger 4099", 8, 0, 1 | (4099 << 4), 4},
        {"T32K lane-rows cap4", 8, 10, 1, 4}, {"T32K wave-rows cap4", 8, 11, 1, 4}, {"T32K lane-rows", 8, 10, 1, 0}, {"T32K wave-rows", 8, 11, 1, 0},
        {"T32K wave-rows cap2", 8, 11, 1, 2}, {"T32K wave-rows ident c4", 8, 11, 0, 4},
        {"R1 identity WRITE", 1, 3, 0, 0}, {"R8 eighth cap4 WRITE", 8, 3, 1, 4}, {"R8 eighth WRITE", 8, 3, 1, 0}, {"R8 eighth cap2 WRITE", 8, 3, 1, 2},
    };
    hipEvent_t ev[16];
    for (auto &e : ev) CK(hipEventCreate(&e));
    printf("# median ms per copy of %.3f GB (read) + the same written; %d placements\n", BYTES / 1e9, pairs);
    for (const Cfg &c : cfgs) {
        if (brief && (c.wg != 0 || (c.order >= 2 && c.order < 10) || (c.map >> 4) != 0)) continue;
        if (brief && c.order < 2 && (c.map >> 4) == 0 && !(c.R == 8 && c.order == 0 && c.map == 1 && c.cap == 4) && !(c.R == 1 && c.cap == 0)) continue;
        printf("%-24s", c.name);
        for (auto &p : bufs) {
            CK(hipEventRecord(ev[0]));
            for (int i = 0; i < burst; ++i) { go(c, p.first, p.second); CK(hipEventRecord(ev[i + 1])); }
            CK(hipDeviceSynchronize());
            std::vector<float> t;
            for (int i = 1; i < burst; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); t.push_back(ms); }
            std::sort(t.begin(), t.end());
            printf("  %.4f", t[t.size() / 2]);
        }
        printf("\n");
    }
    return 0;
}
