// blinky-hip: a calibration kernel, not part of the warp.  bench.py prices the apply kernel's measured traffic against the
// nominal HBM peak (the contract) and, beside it, against what a plain streaming kernel with the SAME read : write ratio
// reaches on the box it runs on: no gather, no LDS, 16-byte accesses, non-temporal stores - the practical roofline of this
// memory system for that mix (tools/membw_probe.hip is the stand-alone version with more kernels).
#include "bk_internal.h"

namespace bk {

__global__ __launch_bounds__(256) void stream_mix_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n, uint32_t period,
                                                         uint32_t writes, uint32_t *sink)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = a[i];
        acc ^= v.x ^ v.w;
        if ((uint32_t)((i >> 6) % period) < writes) {       // whole 1 KiB wave-stores: `writes` of every `period` of them
            v4u w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(b) + i);
        }
    }
    if (acc == 0x12345679u) *sink = acc;
}

// mapped pixels of every owned row + W/32: the cost of a row in a multi-GPU stripe split (bk_comm_rebalance) for the
// direct-gather apply; the staged apply prices rows from its block map instead (coopmap_row_costs)
__global__ __launch_bounds__(256) void row_cost_kernel(const uint32_t *__restrict__ lmap, int W, uint32_t *__restrict__ cost)
{
    __shared__ uint32_t s_sum[4];
    const uint32_t *row = lmap + (size_t)blockIdx.x * W;
    uint32_t n = 0;
    for (int x = threadIdx.x; x < W; x += 256) n += row[x] != BK_NULL_OFFSET ? 1u : 0u;
    for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) cost[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3] + (uint32_t)max(1, W / 32);
}

}  // namespace bk

// cost_dev: device uint32 [H]; the owned rows get what they cost the context's apply variant, every other row 0 (on the
// context stream).  The two variants price in different units: the ranks of a job all run the same one.
int bk_row_costs_device(bk_ctx *ctx, uint32_t *cost_dev)
{
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "row costs: no lensmap (call bk_build first)");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    BK_HIP(ctx, hipMemsetAsync(cost_dev, 0, (size_t)ctx->H * sizeof(uint32_t), ctx->stream));
    if (ctx->apply_variant != 0) return bk::coopmap_row_costs(ctx, cost_dev + ctx->row0);
    if (ctx->rows() > 0) {
        hipLaunchKernelGGL(bk::row_cost_kernel, dim3((unsigned)ctx->rows()), dim3(256), 0, ctx->stream, ctx->d_offsets, ctx->W, cost_dev + ctx->row0);
        BK_HIP(ctx, hipGetLastError());
    }
    return BK_OK;
}

#if BK_DEBUG_API
// the row costs bk_comm_rebalance / bk_multi_rebalance would sum over the ranks, for this context's stripe (host uint32 [H])
extern "C" int bk_debug_row_costs(bk_ctx *ctx, uint32_t *host_out)
{
    if (!ctx || !host_out) return BK_E_INVALID;
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "bk_debug_row_costs: this context has no device");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    uint32_t *d = nullptr;
    BK_HIP(ctx, hipMalloc((void **)&d, (size_t)ctx->H * sizeof(uint32_t)));
    const int rc = bk_row_costs_device(ctx, d);
    hipError_t e = rc == BK_OK ? hipMemcpyAsync(host_out, d, (size_t)ctx->H * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (rc != BK_OK) return rc;
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "bk_debug_row_costs: %s", hipGetErrorString(e));
    return BK_OK;
}

// best of 5 passes over `bytes` read, writes / period of it written; *gbps = (bytes read + bytes written) / time
extern "C" int bk_debug_stream_mix(bk_ctx *ctx, size_t bytes, int period, int writes, double *gbps)
{
    if (!ctx || !gbps || bytes < (1u << 20) || period < 1 || writes < 0 || writes > period) return BK_E_INVALID;
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "bk_debug_stream_mix: this context has no device");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    uint4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&a, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&b, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&sink, 4);
    if (e == hipSuccess) e = hipMemsetAsync(a, 1, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 2, bytes, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const size_t n = bytes / 16;
    float best = 0;
    for (int rep = 0; rep < 5 && e == hipSuccess; ++rep) {
        e = hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(bk::stream_mix_kernel, dim3((unsigned)(ctx->num_cus * 8)), dim3(256), 0, ctx->stream, a, b, n, (uint32_t)period,
                           (uint32_t)writes, sink);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && (best == 0 || ms < best)) best = ms;
    }
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess || best <= 0) return ctx->fail(BK_E_HIP, "bk_debug_stream_mix: %s", hipGetErrorString(e));
    const double waves = (double)(n / 64), written = (waves / period) * writes * 1024.0;     // (to within one period)
    *gbps = ((double)bytes + written) / ((double)best * 1e-3) / 1e9;
    return BK_OK;
}

// FNV-1a-64 of a host buffer: the hash tests/golden/lensmaps.json records frames with, so that `bench.py --check` can hold the
// frames of its timed launch against the committed goldens without touching the CPU oracle
extern "C" int bk_debug_fnv1a64(const void *host, size_t bytes, uint64_t *out)
{
    if ((!host && bytes) || !out) return BK_E_INVALID;
    const unsigned char *p = (const unsigned char *)host;
    uint64_t h = 0xcbf29ce484222325ull;                 // the standard offset basis
    for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ull;
    *out = h;
    return BK_OK;
}
#endif
