// bk_apply.hip -- lensmap APPLY for gfx950: the per-frame 8-bit palettised gather.
//
// replaces render_lensmap (engine/NQ/fisheye.c:2406-2424):
//     if (*lmap) dst = rubix && tint != 255 ? palette[tint][**lmap] : **lmap
// Unmapped pixels (BK_NULL_OFFSET) leave dst untouched.
//
// Data in HBM:
//     lensmap  u32 [rows][W]     offset = plate*ps*ps + py*ps + px   (4 B/px, streamed once)
//     tints    u8  [rows][W]     only read when rubix is on
//     globe    u8  [F][6][ps][ps]
//     dst      u8  [F][..pitch..]
// Pure integer/byte work bound by HBM bandwidth: 6 algorithmic B/px (4 index + 1 texel + 1 store).
#include "bk_internal.h"

namespace bk {

// ---------------------------------------------------------------------------------------
// Variant 0: direct gather.  One thread = 4 consecutive pixels of one row: one 16-byte
// coalesced index load, 4 byte gathers per frame, one packed dword store per frame.
// A launch covers `nframes` frames; each thread re-uses its 4 indices for `fchunk` frames
// (the lensmap is frame-invariant), so index traffic is amortised across a batch.
// ---------------------------------------------------------------------------------------
template <bool RUBIX>
__global__ __launch_bounds__(256) void apply_direct4(
    const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ tints,
    const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames, int frame0,
    uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride,
    int W, int groups_per_row, int ngroups, int nframes, int fchunk,
    const uint8_t *__restrict__ pal)
{
    __shared__ uint8_t s_pal[RUBIX ? BK_MAX_PLATES * 256 : 4];
    if (RUBIX) {
        for (int i = threadIdx.x; i < BK_MAX_PLATES * 256; i += blockDim.x) s_pal[i] = pal[i];
        __syncthreads();
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int row = g / groups_per_row;
    const int x = (g - row * groups_per_row) * 4;
    const size_t px = (size_t)row * W + x;
    const uint4 idx = *reinterpret_cast<const uint4 *>(lmap + px);
    const uint32_t o[4] = {idx.x, idx.y, idx.z, idx.w};
    if ((o[0] & o[1] & o[2] & o[3]) == BK_NULL_OFFSET) return;   // all four unmapped
    const bool all = o[0] != BK_NULL_OFFSET && o[1] != BK_NULL_OFFSET &&
                     o[2] != BK_NULL_OFFSET && o[3] != BK_NULL_OFFSET;
    uint32_t t4 = 0xFFFFFFFFu;
    if (RUBIX) t4 = *reinterpret_cast<const uint32_t *>(tints + px);

    const int f_begin = blockIdx.y * fchunk;
    const int f_end = min(nframes, f_begin + fchunk);
    for (int f = f_begin; f < f_end; ++f) {
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        uint8_t *out = dst + (size_t)f * frame_stride + (size_t)row * dst_pitch + x;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = o[k] != BK_NULL_OFFSET ? gl[o[k]] : 0u;
        if (RUBIX) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t t = (t4 >> (8 * k)) & 0xFFu;
                if (t < (uint32_t)BK_MAX_PLATES) v[k] = s_pal[t * 256 + v[k]];      // (255 = none; a caller's out-of-range tint never indexes past the LUT)
            }
        }
        if (all && ((reinterpret_cast<uintptr_t>(out) & 3u) == 0)) {
            *reinterpret_cast<uint32_t *>(out) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (o[k] != BK_NULL_OFFSET) out[k] = (uint8_t)v[k];
        }
    }
}

// scalar form for widths that are not a multiple of 4
template <bool RUBIX>
__global__ __launch_bounds__(256) void apply_direct1(
    const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ tints,
    const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames, int frame0,
    uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride,
    int W, int npix, int nframes, int fchunk, const uint8_t *__restrict__ pal)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t o = lmap[i];
    if (o == BK_NULL_OFFSET) return;
    const int row = i / W, x = i - row * W;
    const uint32_t t = RUBIX ? tints[i] : 255u;
    const int f_begin = blockIdx.y * fchunk;
    const int f_end = min(nframes, f_begin + fchunk);
    for (int f = f_begin; f < f_end; ++f) {
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        uint32_t v = gl[o];
        if (RUBIX && t < (uint32_t)BK_MAX_PLATES) v = pal[t * 256 + v];
        dst[(size_t)f * frame_stride + (size_t)row * dst_pitch + x] = (uint8_t)v;
    }
}

// ---------------------------------------------------------------------------------------
// mapped-pixel bitmap: one wave ballot per 64 pixels (bit i = pixel i mapped).
// Used by the host to merge a warped frame into vid.buffer span by span.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_kernel(const uint32_t *__restrict__ lmap, size_t npix,
                                                   uint64_t *__restrict__ mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool mapped = i < npix && lmap[i] != BK_NULL_OFFSET;
    const uint64_t b = __ballot(mapped);
    if ((threadIdx.x & 63) == 0 && i < npix) mask[i >> 6] = b;
}

// ---------------------------------------------------------------------------------------
// synthetic plates: SURVEY.md 8(d)  s0 = seed, s <- s*1664525 + 1013904223, texel = s>>24.
// Each thread jumps ahead to its first element by composing the affine map in O(log i).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lcg_kernel(uint8_t *__restrict__ dst, int ps, int gp, int ph, uint32_t seed)
{
    // texel i = py*ps + px of the stream lands at dst[bk_texel_offset(px, py)]; each thread produces
    // up to 16 consecutive texels of one row = one 16-byte chunk of the tiled layout
    const int chunks_per_row = (ps + 15) / 16;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= chunks_per_row * ps) return;
    const int py = id / chunks_per_row, px0 = (id - py * chunks_per_row) * 16;
    const size_t i0 = (size_t)py * ps + px0;
    uint32_t A = 1u, C = 0u, a = 1664525u, c = 1013904223u;
    for (size_t e = i0; e; e >>= 1) {
        if (e & 1) { A = a * A; C = a * C + c; }
        c = a * c + c;
        a = a * a;
    }
    uint32_t s = A * seed + C;
    uint32_t w[4] = {0, 0, 0, 0};
    const int cnt = min(16, ps - px0);
    for (int k = 0; k < cnt; ++k) {
        s = s * 1664525u + 1013904223u;
        w[k >> 2] |= (s >> 24) << (8 * (k & 3));
    }
    uint8_t *out = dst + bk_texel_offset((uint32_t)gp, (uint32_t)ph, 0u, (uint32_t)px0, (uint32_t)py);   // px0 % 16 == 0: 16-byte aligned
    if (cnt == 16) *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (int k = 0; k < cnt; ++k) out[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
}

// ---------------------------------------------------------------------------------------
// lensmap entries cross the ABI in the reference layout (plate*ps*ps + py*ps + px, GLOBEPIXEL
// fisheye.c:349); on the device they address the tiled globe (bk_texel_offset, bk_build_params.h)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convert_offsets_kernel(uint32_t *__restrict__ buf, size_t n, uint32_t ps,
                                                              uint32_t gp, uint32_t ph, int to_device)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = buf[i];
    if (o == BK_NULL_OFFSET) return;
    uint32_t plate, px, py;
    if (to_device) {
        plate = o / (ps * ps);
        const uint32_t rem = o - plate * (ps * ps);
        py = rem / ps;
        px = rem - py * ps;
        buf[i] = plate < BK_MAX_PLATES ? bk_texel_offset(gp, ph, plate, px, py) : BK_NULL_OFFSET;   // never address outside the globe
    } else {
        bk_texel_coords(gp, ph, o, &plate, &px, &py);
        buf[i] = plate * (ps * ps) + py * ps + px;
    }
}

// one plate between the row-major staging buffer [ps][gp] and its tiled place in the globe: a thread
// moves one 16-byte chunk (16 texels of a row)
__global__ __launch_bounds__(256) void plate_retile_kernel(uint8_t *__restrict__ tiled, uint8_t *__restrict__ rowmajor,
                                                           int ps, int gp, int ph, int to_tiled)
{
    const int cpr = gp >> 4;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= cpr * ps) return;
    const int py = id / cpr, cx = id - py * cpr;
    uint4 *t = reinterpret_cast<uint4 *>(tiled + bk_texel_offset((uint32_t)gp, (uint32_t)ph, 0u, (uint32_t)cx * 16u, (uint32_t)py));
    uint4 *r = reinterpret_cast<uint4 *>(rowmajor + (size_t)py * gp + (size_t)cx * 16);
    if (to_tiled) *t = *r; else *r = *t;
}

// sparse patches of a device table (the handful of lensmap entries / texel corners the host re-derives on the
// platform libm after a build, bk_lens.cpp): dst[idx[i]] = val[i]
template <typename T>
__global__ __launch_bounds__(256) void scatter_kernel(T *__restrict__ dst, const uint32_t *__restrict__ idx,
                                                      const T *__restrict__ val, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}

// The reference stops building at the first malformed callback result (status -1, fisheye.c:2113-2115) and keeps what its scan -
// rows from the bottom up, pixels left to right - had set by then.  The GPU build evaluates every pixel; this pass takes away
// the entries the reference would not have reached (scan key <= the failing pixel's) and recounts the display flags.
__global__ __launch_bounds__(256) void truncate_scan_kernel(uint32_t *__restrict__ off, uint8_t *__restrict__ tints, int W, int row0, size_t n,
                                                            uint32_t bad_key, uint32_t plate_bytes, int *__restrict__ display)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lyl = (int)(i / (size_t)W), lx = (int)(i - (size_t)lyl * W), ly = row0 + lyl;
    const uint32_t key = (uint32_t)(ly * W + (W - 1 - lx)) + 1u;
    if (key <= bad_key) { off[i] = BK_NULL_OFFSET; tints[i] = 255; return; }
    const uint32_t o = off[i];
    if (o != BK_NULL_OFFSET) {
        const uint32_t plate = o / plate_bytes;
        if (plate < (uint32_t)BK_MAX_PLATES && __hip_atomic_load(&display[plate], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&display[plate], 1);
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
int launch_truncate_scan(bk_ctx *ctx, unsigned int bad_key, int display_out[BK_MAX_PLATES])
{
    const size_t n = (size_t)ctx->W * ctx->rows();
    BK_HIP(ctx, hipMemsetAsync(ctx->d_display, 0, BK_MAX_PLATES * sizeof(int), ctx->stream));
    if (n) hipLaunchKernelGGL(truncate_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_offsets, ctx->d_tints, ctx->W,
                              ctx->row0, n, bad_key, (uint32_t)ctx->plate_bytes(), ctx->d_display);
    BK_HIP(ctx, hipGetLastError());
    BK_HIP(ctx, hipMemcpyAsync(display_out, ctx->d_display, BK_MAX_PLATES * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// dst = address of pixel (0, row0) of frame 0, i.e. the first owned row
int launch_apply(bk_ctx *ctx, int frame0, int nframes, uint8_t *dst, int dst_pitch,
                 size_t frame_stride, int rubix_on)
{
    const int rows = ctx->rows();
    if (rows <= 0 || nframes <= 0) return BK_OK;
    if (ctx->apply_variant != 0) return launch_apply_coop(ctx, frame0, nframes, dst, dst_pitch, frame_stride, rubix_on);
    const size_t gstride = ctx->globe_stride();
    const int fchunk = nframes < 8 ? nframes : 8;
    const int fblocks = (nframes + fchunk - 1) / fchunk;
    if (ctx->W % 4 == 0) {
        const int gpr = ctx->W / 4;
        const int ngroups = gpr * rows;
        dim3 grid((ngroups + 255) / 256, fblocks);
        if (rubix_on)
            hipLaunchKernelGGL(apply_direct4<true>, grid, dim3(256), 0, ctx->stream, ctx->d_offsets,
                               ctx->d_tints, ctx->d_globe, gstride, ctx->nframes, frame0, dst,
                               dst_pitch, frame_stride, ctx->W, gpr, ngroups, nframes, fchunk, ctx->d_pal);
        else
            hipLaunchKernelGGL(apply_direct4<false>, grid, dim3(256), 0, ctx->stream, ctx->d_offsets,
                               ctx->d_tints, ctx->d_globe, gstride, ctx->nframes, frame0, dst,
                               dst_pitch, frame_stride, ctx->W, gpr, ngroups, nframes, fchunk, ctx->d_pal);
    } else {
        const int npix = ctx->W * rows;
        dim3 grid((npix + 255) / 256, fblocks);
        if (rubix_on)
            hipLaunchKernelGGL(apply_direct1<true>, grid, dim3(256), 0, ctx->stream, ctx->d_offsets,
                               ctx->d_tints, ctx->d_globe, gstride, ctx->nframes, frame0, dst,
                               dst_pitch, frame_stride, ctx->W, npix, nframes, fchunk, ctx->d_pal);
        else
            hipLaunchKernelGGL(apply_direct1<false>, grid, dim3(256), 0, ctx->stream, ctx->d_offsets,
                               ctx->d_tints, ctx->d_globe, gstride, ctx->nframes, frame0, dst,
                               dst_pitch, frame_stride, ctx->W, npix, nframes, fchunk, ctx->d_pal);
    }
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

int launch_mask(bk_ctx *ctx)
{
    const size_t npix = (size_t)ctx->W * ctx->rows();
    if (!npix) return BK_OK;
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_offsets, npix, ctx->d_mask);
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

int launch_fill_lcg(bk_ctx *ctx, uint8_t *plate_dst, uint32_t seed)
{
    const int threads = ((ctx->ps + 15) / 16) * ctx->ps;
    hipLaunchKernelGGL(lcg_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                       plate_dst, ctx->ps, ctx->gp, ctx->ph, seed);
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

int launch_convert_offsets(bk_ctx *ctx, uint32_t *buf, size_t n, int to_device)
{
    if (!n) return BK_OK;
    hipLaunchKernelGGL(convert_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       buf, n, (uint32_t)ctx->ps, (uint32_t)ctx->gp, (uint32_t)ctx->ph, to_device);
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

int launch_plate_retile(bk_ctx *ctx, uint8_t *plate_tiled, int to_tiled, uint8_t *rowmajor)
{
    const int threads = (ctx->gp >> 4) * ctx->ps;
    hipLaunchKernelGGL(plate_retile_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                       plate_tiled, rowmajor ? rowmajor : ctx->d_plate_stage, ctx->ps, ctx->gp, ctx->ph, to_tiled);
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

template <typename T>
static int scatter_impl(bk_ctx *ctx, T *dst, const uint32_t *h_idx, const T *h_val, size_t n)
{
    if (!n) return BK_OK;
    uint32_t *d_idx = nullptr;
    T *d_val = nullptr;
    BK_HIP(ctx, hipMalloc((void **)&d_idx, n * sizeof(uint32_t)));
    hipError_t e = hipMalloc((void **)&d_val, n * sizeof(T));
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, h_idx, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_val, h_val, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(scatter_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, d_idx, d_val, (uint32_t)n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the host arrays and the temporaries go away with the caller
    (void)hipFree(d_idx);
    (void)hipFree(d_val);
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "scatter: %s", hipGetErrorString(e));
    return BK_OK;
}
int launch_scatter32(bk_ctx *ctx, uint32_t *dst, const uint32_t *h_idx, const uint32_t *h_val, size_t n) { return scatter_impl<uint32_t>(ctx, dst, h_idx, h_val, n); }
int launch_scatter8(bk_ctx *ctx, uint8_t *dst, const uint32_t *h_idx, const uint8_t *h_val, size_t n) { return scatter_impl<uint8_t>(ctx, dst, h_idx, h_val, n); }

}  // namespace bk
