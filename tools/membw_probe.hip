// Calibration probe (developer tool, not part of the library): streaming read / write / copy rates
// of this GPU with 16-byte accesses, and a known byte count for checking the FETCH_SIZE /
// WRITE_SIZE counter units under rocprofv3 --pmc.   hipcc --offload-arch=gfx950 -O3 membw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe_read(const uint4 *__restrict__ a, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345679u) *sink = acc;
}
__global__ __launch_bounds__(256) void probe_write(uint4 *__restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void probe_write_nt(uint4 *__restrict__ a, size_t n)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4u v = {(uint32_t)i, 1, 2, 3};
        __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(a) + i);
    }
}
// the apply's STORE pattern alone: a [frames][H][W] byte image written in tiles of TW bytes x TH rows, one tile per workgroup step, 16 bytes
// per lane, TW/16 lanes side by side (the staged apply: TW = 128 - 8 lanes write one 128-byte piece of a row, a wave 8 rows of a 3840-byte
// pitch, a workgroup 32); wider tiles write longer contiguous pieces.  Tiles in row-major order within a frame, grid-stride.
__global__ __launch_bounds__(256) void probe_write_tiles(uint8_t *__restrict__ img, int W, int H, int frames, int TW, int nt)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const int lpr = TW / 16, rows_per_wg = 256 / lpr;               // lanes per row piece, rows a workgroup covers
    const int tx_n = (W + TW - 1) / TW, ty_n = (H + rows_per_wg - 1) / rows_per_wg;
    const size_t tiles = (size_t)tx_n * ty_n * frames;
    const int lx = threadIdx.x % lpr, ly = threadIdx.x / lpr;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int f = (int)(t / ((size_t)tx_n * ty_n));
        const int r = (int)(t - (size_t)f * tx_n * ty_n), ty = r / tx_n, tx = r - ty * tx_n;
        const int y = ty * rows_per_wg + ly, x = tx * TW + lx * 16;
        if (y >= H || x >= W) continue;
        uint8_t *o = img + ((size_t)f * H + y) * W + x;
        v4u v = {(uint32_t)t, 1, 2, 3};
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(o));
        else *reinterpret_cast<v4u *>(o) = v;
    }
}
// the apply's mix, streaming: read 13 bytes for every 8 written (non-temporal stores)
__global__ __launch_bounds__(256) void probe_mix_nt(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n, uint32_t *sink)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
        acc ^= v.x ^ v.w;
        if ((i >> 6) % 13 < 8) { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(b) + i); }
    }
    if (acc == 0x12345679u) *sink = acc;
}
__global__ __launch_bounds__(256) void probe_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// the apply kernel's traffic mix: read 13 bytes, write 8 (per 21)
__global__ __launch_bounds__(256) void probe_mix(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
        if ((i & 7) < 5) b[i] = v; else if (v.x == 0x12345679u) b[0] = v;
    }
}

// whole 128-byte lines (8 lanes x 16 bytes) at pseudo-random places: runs of `run` consecutive lines, then a jump.
// What a gather that has nothing to re-use can expect from the memory system, against the streaming numbers above.
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void probe_gather(const uint4 *__restrict__ a, uint32_t nlines, uint32_t run, int iters, uint32_t *sink, uint32_t useful)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, group = gid >> 3, sub = gid & 7, groups = gridDim.x * 32;
    const uint32_t nruns = nlines / run;
    uint32_t acc = 0;
    for (int k = 0; k < iters; k += 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t seq = group + (uint32_t)(k + u) * groups;
            const uint32_t line = (mix32(seq / run) % nruns) * run + seq % run;
            v[u] = make_uint4(0, 0, 0, 0);
            if (((sub * 5u + line) & 7u) < useful) v[u] = a[(size_t)line * 8 + sub];      // `useful` of the line's 8 chunks, a different set per line
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345679u) *sink = acc;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20;
    const size_t n = bytes / 16;
    uint4 *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        for (int k = 0; k < 4; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(probe_read, dim3(g), dim3(256), 0, 0, a, n, sink);
                if (k == 1) hipLaunchKernelGGL(probe_write, dim3(g), dim3(256), 0, 0, b, n);
                if (k == 2) hipLaunchKernelGGL(probe_copy, dim3(g), dim3(256), 0, 0, a, b, n);
                if (k == 3) hipLaunchKernelGGL(probe_mix, dim3(g), dim3(256), 0, 0, a, b, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = k == 2 ? 2.0 * bytes : k == 3 ? bytes * (1.0 + 5.0 / 8.0) : (double)bytes;
            printf("%-5s grid %5d: %8.3f ms  %7.2f TB/s (bytes moved %.0f MiB)\n",
                   k == 0 ? "read" : k == 1 ? "write" : k == 2 ? "copy" : "mix", g, best, moved / best / 1e9, moved / 1048576.0);
        }
    }
    for (int g : {256 * 4, 256 * 8, 256 * 16}) {
        for (int k = 0; k < 2; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(probe_write_nt, dim3(g), dim3(256), 0, 0, b, n);
                if (k == 1) hipLaunchKernelGGL(probe_mix_nt, dim3(g), dim3(256), 0, 0, a, b, n, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = k == 0 ? (double)bytes : bytes * (1.0 + 8.0 / 13.0);
            printf("%-8s grid %5d: %8.3f ms  %7.2f TB/s (bytes moved %.0f MiB)\n", k == 0 ? "write-nt" : "mix-nt", g, best, moved / best / 1e9, moved / 1048576.0);
        }
    }
    // the apply's store pattern: 3840 x 2160 frames (as many as fit `bytes`), tiles of 128 .. 3840 bytes per row piece
    {
        const int W = 3840, H = 2160, frames = (int)(bytes / ((size_t)W * H));
        for (int g : {256 * 8, 256 * 16}) {
            for (int TW : {64, 128, 256, 512, 1024, 4096}) for (int nt = 0; nt < 2; ++nt) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(probe_write_tiles, dim3(g), dim3(256), 0, 0, reinterpret_cast<uint8_t *>(b), W, H, frames, TW, nt);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double moved = (double)W * H * frames;
                printf("%s tiles of %4d B x %3d rows, grid %5d: %8.3f ms  %7.2f TB/s (%.0f MiB)\n", nt ? "write-nt" : "write   ", TW, 256 / (TW / 16), g, best, moved / best / 1e9, moved / 1048576.0);
            }
        }
    }
    // random-line gather: 2 GiB of distinct lines (beyond the 256 MiB Infinity Cache), runs of 1..64 lines
    {
        const uint32_t nlines = (uint32_t)(bytes / 128);
        for (int g : {256 * 8, 256 * 16}) {
            for (uint32_t run : {1u, 2u, 4u, 8u, 16u, 64u, 1024u}) {
                const int iters = (int)((size_t)nlines / ((size_t)g * 32)) & ~3;
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(probe_gather, dim3(g), dim3(256), 0, 0, a, nlines, run, iters, sink, 8u);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double moved = (double)iters * g * 32 * 128;
                printf("gather grid %5d runs of %4u lines: %8.3f ms  %7.2f TB/s (%.0f MiB)\n", g, run, best, moved / best / 1e9, moved / 1048576.0);
            }
        }
    }
    // partial lines: only `useful` of a line's eight 16-byte chunks are requested (the line still travels whole)
    {
        const uint32_t nlines = (uint32_t)(bytes / 128);
        const int g = 256 * 8;
        for (uint32_t useful : {8u, 6u, 5u, 4u, 2u, 1u}) {
            const int iters = (int)((size_t)nlines / ((size_t)g * 32)) & ~3;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe_gather, dim3(g), dim3(256), 0, 0, a, nlines, 1u, iters, sink, useful);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = (double)iters * g * 32 * 128;
            printf("gather single lines, %u of 8 chunks requested: %8.3f ms  %7.2f TB/s of lines (%.2f TB/s of requested bytes)\n", useful, best, moved / best / 1e9, moved / best / 1e9 * useful / 8);
        }
    }
    return 0;
}
