// Developer probe (round 5): how fast do SCATTERED 16-byte chunks (the apply's staging pattern: ascending offsets, ~5.6 of every 8 chunks of a
// 128-byte line) travel HBM -> LDS (a) through registers (global_load_dwordx4 + ds_write_b128) and (b) by LDS-DMA (global_load_lds_dwordx4),
// and (c) LDS-DMA of CONTIGUOUS lines - each with `depth` wave-instructions in flight per wave.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void *lds_p;
__device__ __forceinline__ void dma16(uint32_t lds_addr, uint32_t voff, const uint8_t *base)
{
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(base) : "memory");
}
// every workgroup walks `per_wg` lists of 1024 offsets (4 wave-instructions per wave... 256 threads x 4), `rounds` times over different windows
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *g, const uint32_t *offs, int lists_per_wg, int nlists, uint32_t *sink, size_t window, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];      // 2 x 16 KiB
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_p)smem);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint8_t *gb = g + (size_t)((blockIdx.x * 7 + r * 13) % 61) * window;
        const uint8_t *gu = (const uint8_t *)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)gb)) |
                                              ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)gb >> 32)) << 32));
        for (int l = 0; l < lists_per_wg; ++l) {
            const uint32_t *lst = offs + (size_t)((blockIdx.x * lists_per_wg + l) % nlists) * 1024;
            const int buf = (l & 1) * 16384;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = lst[j * 256 + threadIdx.x];
            if (MODE == 0) {
                uint4 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { typedef uint32_t v4u __attribute__((ext_vector_type(4))); const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(gu + o[j])); q[j] = make_uint4(t.x, t.y, t.z, t.w); }
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4 *>(smem + buf + (j * 256 + threadIdx.x) * 16) = q[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16((uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + buf + (j * 256 + wave * 64) * 16)), o[j], gu);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            acc += *reinterpret_cast<const uint32_t *>(smem + buf + ((threadIdx.x * 37 + l) & 4095) * 4);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const size_t window = 28u << 20;           // one "globe": 28 MB; 61 of them = 1.7 GB (past the Infinity Cache)
    uint8_t *g; CK(hipMalloc(&g, window * 61)); CK(hipMemset(g, 1, window * 61));
    const int nlists = 4096;
    std::vector<uint32_t> h((size_t)nlists * 1024), hc((size_t)nlists * 1024);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int l = 0; l < nlists; ++l) {
        // scattered: ascending chunk numbers, each chunk of a line kept with probability 0.7, starting anywhere in the window
        uint32_t c = (uint32_t)(rnd() % ((window >> 4) - 4096));
        for (int i = 0; i < 1024; ++i) { do { ++c; } while (rnd() % 10 >= 7); h[(size_t)l * 1024 + i] = c * 16u; }
        uint32_t c2 = (uint32_t)(rnd() % ((window >> 4) - 4096)) & ~7u;
        for (int i = 0; i < 1024; ++i) hc[(size_t)l * 1024 + i] = (c2 + i) * 16u;      // contiguous: 128 whole lines
    }
    uint32_t *d, *dc, *sink; CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&dc, h.size() * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int occ : {8, 4}) {
        const int grid = 256 * occ, lists = 16, rounds = 8;
        const size_t shmem = occ == 8 ? 32768 / 2 * 2 : 32768;      // 32 KiB: at most 5 per CU; occ 8 is a request for the grid size only
        for (int mode = 0; mode < 3; ++mode) {
            const uint32_t *lst = mode == 2 ? dc : d;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), shmem, 0, g, lst, lists, nlists, sink, window, rounds);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), shmem, 0, g, lst, lists, nlists, sink, window, rounds);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)grid * lists * rounds * 1024 * 16;
                if (rep) printf("grid %5d (%d x 256 CUs) %-44s %7.1f GB/s (%.3f ms)\n", grid, occ, mode == 0 ? "scattered chunks, registers + ds_write_b128" : mode == 1 ? "scattered chunks, LDS-DMA" : "contiguous lines, LDS-DMA", bytes / ms / 1e6, ms);
            }
        }
    }
    return 0;
}
