// bk_apply_tiled.hip -- lensmap APPLY, variant 1: tiled lensmap + LDS-staged plate regions.
//
// The lensmap is frame-invariant, so after every build it is "compiled" once into a form the
// per-frame gather can stream at full width:
//   * the screen is cut into tiles of 32 x (8*RG) pixels, one wavefront per tile (64 lanes, each
//     RG groups of 4 consecutive pixels; RG = 1, 2 or 4, default 2);
//   * for each tile a wave discovers, with ballots and shuffle reductions, which plates its
//     pixels read and the bounding box of the texels inside each plate (<= 3 regions: a cube
//     corner); the box is widened to 16-byte columns of the padded globe rows;
//   * each pixel's 32-bit globe offset is replaced by a 16-bit address inside the tile's LDS
//     staging area (0xFFFF = unmapped), stored tile-major so a wave reads its slab in one go.
// Per frame a wave then (1) copies its regions HBM -> LDS with coalesced 16-byte row loads,
// (2) gathers its texels from LDS (ds_read_u8: several times the rate of the texture
// addresser's byte gathers), (3) stores 4 packed pixels per lane and row group.  Tiles whose
// regions do not fit the LDS budget fall back to direct global gathers through the 32-bit table.
// Four horizontally adjacent tiles form a workgroup (a 128-pixel-wide strip: full 128-byte lines
// are written by one CU); the launch is persistent and workgroup b walks tiles of the horizontal
// screen band b % 8, so each XCD's L2 keeps its own slice of the index stream and shares plate rows.
//
// replaces render_lensmap (engine/NQ/fisheye.c:2406-2424); byte-exact.
#include "bk_internal.h"

namespace bk {

// tile = (32*XG) x (8*RG) pixels; a lane owns, in each of RG row groups, 4*XG consecutive pixels
constexpr int BK_TILE_LDS_CAP = 12288;                            // max bytes of LDS per wave-tile
constexpr uint32_t F_ALL = 1, F_SLOW = 2, F_EMPTY = 4;

struct TileHdr {              // 32 bytes
    uint32_t src[3];          // byte offset of region r inside one globe frame (16-byte aligned)
    uint16_t w16[3];          // region width in 16-byte chunks
    uint16_t rows[3];         // region height
    uint16_t nreg;
    uint16_t flags;
    uint16_t lds16;           // LDS bytes / 16 the regions need
    uint16_t pad;
};

struct TileMap {
    TileHdr *d_hdr = nullptr;
    uint16_t *d_idx = nullptr;      // [ntiles][RG][256]
    uint8_t *d_tint = nullptr;      // [ntiles][RG][256] tile-major tints (rubix)
    uint32_t *d_stats = nullptr;    // 64 replicas of: [0] max LDS bytes, [1] >3-region tiles, [2] empty tiles,
                                    // [3] 128-B lines staged, [4..36) LDS-need histogram (512 B bins)
    int blocks_x = 0, blocks_y = 0; // workgroup blocks of 128 x (8*RG) pixels
    int rg = 2, xg = 1;             // row groups per lane (tile height / 8), x groups per lane (tile width / 32)
    int lds_bytes = 0;              // LDS budget per wave of the apply launch
    uint32_t stats[40] = {0};
    int slow_tiles = 0;
    bool valid = false;
    size_t alloc_px = 0;
};

// developer ablation switch (tools/apply_probe.py): bit0 skip region loads, bit1 skip stores,
// bit2 skip the LDS gather, bit3 skip the LDS writes, bit4 header/index fetch only.  0 normally;
// results are wrong by design while it is non-zero.
__device__ int g_ablate = 0;

// tile t = block*4 + wave; wave w is the w-th column of tiles in the (128*XG)-pixel-wide block
__device__ __forceinline__ void tile_origin(int t, int rg, int xg, int blocks_x, int *ox, int *oy)
{
    const int blk = t >> 2, w = t & 3;
    const int by = blk / blocks_x, bx = blk - by * blocks_x;
    *ox = (bx * 4 + w) * 32 * xg;
    *oy = by * 8 * rg;
}

__device__ __forceinline__ int wave_min(int v)
{
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}

// ---------------------------------------------------------------------------------------------
// compile: one wave per tile
// ---------------------------------------------------------------------------------------------
template <int RG, int XG>
__global__ __launch_bounds__(256) void tile_compile_kernel(const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ tints,
                                                           int W, int rows, int ps, int gp, int blocks_x, int ntiles,
                                                           TileHdr *__restrict__ hdr, uint16_t *__restrict__ idx,
                                                           uint8_t *__restrict__ tint_t, uint32_t *__restrict__ stats)
{
    constexpr int NG = RG * XG, NP = 4 * NG;    // pixel groups / pixels per lane; group g = (row group g / XG, x half g % XG)
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    int ox, oy;
    tile_origin(t, RG, XG, blocks_x, &ox, &oy);
    const int ry = lane >> 3, cx = lane & 7;
    const int x0 = ox + cx * 4 * XG;

    uint32_t o[NP];
    uint8_t tn[NP];
    int plate[NP], px[NP], py[NP];
    const uint32_t plate_stride = (uint32_t)gp * (uint32_t)ps;
    bool all_l = true, any_l = false;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = oy + ((i >> 2) / XG) * 8 + ry, x = x0 + ((i >> 2) % XG) * 4 + (i & 3);
        const bool in = row < rows && x < W;
        o[i] = in ? lmap[(size_t)row * W + x] : BK_NULL_OFFSET;
        tn[i] = in ? tints[(size_t)row * W + x] : 255;
        plate[i] = -1; px[i] = 0; py[i] = 0;
        if (o[i] != BK_NULL_OFFSET) {
            const uint32_t p = o[i] / plate_stride, rem = o[i] - p * plate_stride;
            plate[i] = (int)p;
            py[i] = (int)(rem / (uint32_t)gp);
            px[i] = (int)(rem - (uint32_t)py[i] * (uint32_t)gp);
        }
        all_l = all_l && plate[i] >= 0;
        any_l = any_l || plate[i] >= 0;
    }
    const bool all = __all(all_l), any = __any(any_l);

    // regions: one per plate present in the tile (ballot), bounding box by shuffle reduction
    int nreg = 0, lds16 = 0, lines = 0;
    int r_plate[3] = {-1, -1, -1}, r_x0[3] = {0, 0, 0}, r_y0[3] = {0, 0, 0}, r_w16[3] = {0, 0, 0}, r_rows[3] = {0, 0, 0}, r_base16[3] = {0, 0, 0};
    bool slow = false;
    for (int p = 0; p < BK_MAX_PLATES; ++p) {
        bool mine = false;
        int mnx = 1 << 30, mxx = -1, mny = 1 << 30, mxy = -1;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (plate[i] == p) { mine = true; mnx = min(mnx, px[i]); mxx = max(mxx, px[i]); mny = min(mny, py[i]); mxy = max(mxy, py[i]); }
        if (__ballot(mine) == 0) continue;                       // wave-uniform
        mnx = wave_min(mnx); mxx = wave_max(mxx); mny = wave_min(mny); mxy = wave_max(mxy);
        if (nreg == 3) { slow = true; break; }
        const int xa = mnx & ~15, w16 = (mxx - xa) / 16 + 1, nrows = mxy - mny + 1, pitch16 = w16 | 1;
        r_plate[nreg] = p; r_x0[nreg] = xa; r_y0[nreg] = mny; r_w16[nreg] = w16; r_rows[nreg] = nrows; r_base16[nreg] = lds16;
        lds16 += nrows * pitch16;
        lines += nrows * (((xa + w16 * 16 + 127) >> 7) - (xa >> 7));      // 128-byte lines the staging touches
        ++nreg;
    }
    if (lds16 * 16 > BK_TILE_LDS_CAP || lds16 * 16 > 0xFFF0) slow = true;

#pragma unroll
    for (int r = 0; r < NG; ++r) {
        uint32_t a[4], tw = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = r * 4 + k;
            a[k] = 0xFFFFu;
            if (plate[i] >= 0) {
                a[k] = 0;
                if (!slow)
                    for (int q = 0; q < 3; ++q)
                        if (r_plate[q] == plate[i])
                            a[k] = (uint32_t)((r_base16[q] + (py[i] - r_y0[q]) * (r_w16[q] | 1)) * 16 + (px[i] - r_x0[q]));
            }
            tw |= (uint32_t)tn[i] << (8 * k);
        }
        // per row group: lane-major, 4*XG entries per lane (one 8- or 16-byte load in the apply kernel)
        const size_t slab = ((size_t)t * RG + r / XG) * (256 * XG) + (size_t)lane * (4 * XG) + (r % XG) * 4;
        *reinterpret_cast<uint2 *>(idx + slab) = make_uint2(a[0] | (a[1] << 16), a[2] | (a[3] << 16));
        *reinterpret_cast<uint32_t *>(tint_t + slab) = tw;
    }
    if (lane == 0) {
        TileHdr h;
        for (int r = 0; r < 3; ++r) {
            h.src[r] = r < nreg ? (uint32_t)r_plate[r] * plate_stride + (uint32_t)r_y0[r] * (uint32_t)gp + (uint32_t)r_x0[r] : 0u;
            h.w16[r] = (uint16_t)(r < nreg && !slow ? r_w16[r] : 0);
            h.rows[r] = (uint16_t)(r < nreg && !slow ? r_rows[r] : 0);
        }
        h.nreg = (uint16_t)(slow ? 0 : nreg);
        h.flags = (uint16_t)((all ? F_ALL : 0) | (slow ? F_SLOW : 0) | (any ? 0 : F_EMPTY));
        h.lds16 = (uint16_t)(slow ? 0 : lds16);
        h.pad = 0;
        hdr[t] = h;
        uint32_t *st = stats + (t & 63) * 40;       // 64 replicas: thousands of tiles must not serialise on one word
        if (!slow && any) {
            atomicMax(&st[0], (uint32_t)lds16 * 16u);
            atomicAdd(&st[4 + min(31, (lds16 * 16 + 511) / 512)], 1u);        // histogram of LDS need
            atomicAdd(&st[3], (uint32_t)lines);
        }
        if (slow) atomicAdd(&st[1], 1u);
        if (!any) atomicAdd(&st[2], 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------
template <int NG>
struct TileIdx {              // a lane's LDS addresses (two 16-bit per dword) and tints, per pixel group
    uint2 iw[NG];
    uint32_t t4[NG];
};

// What a wave fetches ahead for its NEXT tile while it works on the current one: the 32-byte
// header (as two vector loads, so that it is tracked by vmcnt like everything else - an s_load
// would share lgkmcnt with the LDS traffic and stall the gathers), its LDS indices and tints.
template <int NG>
struct TilePrefetch {
    uint4 h0, h1;
    TileIdx<NG> ix;
};

template <bool RUBIX, int RG, int XG>
__device__ __forceinline__ TilePrefetch<RG * XG> tile_fetch(const TileHdr *__restrict__ hdr, const uint16_t *__restrict__ idx,
                                                       const uint8_t *__restrict__ tint_t, int t, int lane)
{
    TilePrefetch<RG * XG> p;
    int tv;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tv) : "s"(t));     // make the address a VGPR: vector loads
    const uint4 *hp = reinterpret_cast<const uint4 *>(hdr) + 2 * (size_t)tv;
    p.h0 = hp[0];
    p.h1 = hp[1];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const size_t slab = ((size_t)t * RG + r) * (256 * XG) + (size_t)lane * (4 * XG);
        if (XG == 1) {
            p.ix.iw[r] = *reinterpret_cast<const uint2 *>(idx + slab);
            p.ix.t4[r] = RUBIX ? *reinterpret_cast<const uint32_t *>(tint_t + slab) : 0xFFFFFFFFu;
        } else {
            const uint4 v = *reinterpret_cast<const uint4 *>(idx + slab);
            p.ix.iw[r * XG] = make_uint2(v.x, v.y);
            p.ix.iw[r * XG + XG - 1] = make_uint2(v.z, v.w);
            const uint2 tt = RUBIX ? *reinterpret_cast<const uint2 *>(tint_t + slab) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            p.ix.t4[r * XG] = tt.x;
            p.ix.t4[r * XG + XG - 1] = tt.y;
        }
    }
    return p;
}

// The hot path: every pixel of the tile mapped, regions <= NQ KiB, aligned destination; the loop
// body is branch-free.  (Tried and measured without gain: a register prefetch of frame f+1's chunks,
// and issuing several frames' loads before their stores - with 24 waves per CU the memory system,
// not per-wave latency, is the limiter.)
template <int NQ, int RG, int XG>
__device__ __forceinline__ void hot_frames(const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames, int frame0,
                                           int f_begin, int f_end, uint8_t *__restrict__ out0, int dst_pitch, size_t frame_stride,
                                           uint8_t *lds, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3,
                                           uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3,
                                           bool k0, bool k1, bool k2, bool k3, const TileIdx<RG * XG> ix, bool wg_sync)
{
    // (explicit scalars, not arrays: the four chunks must stay in VGPRs)
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    const int abl = g_ablate;
    for (int f = f_begin; f < f_end; ++f) {
        if (wg_sync) __builtin_amdgcn_s_barrier();
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        if (!(abl & 1)) {
            q0 = *reinterpret_cast<const uint4 *>(gl + s0);
            if (NQ > 1) q1 = *reinterpret_cast<const uint4 *>(gl + s1);
            if (NQ > 2) q2 = *reinterpret_cast<const uint4 *>(gl + s2);
            if (NQ > 3) q3 = *reinterpret_cast<const uint4 *>(gl + s3);
        }
        if (!(abl & 8)) {
            if (k0) *reinterpret_cast<uint4 *>(lds + d0) = q0;
            if (NQ > 1 && k1) *reinterpret_cast<uint4 *>(lds + d1) = q1;
            if (NQ > 2 && k2) *reinterpret_cast<uint4 *>(lds + d2) = q2;
            if (NQ > 3 && k3) *reinterpret_cast<uint4 *>(lds + d3) = q3;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t w[RG * XG];
#pragma unroll
        for (int r = 0; r < RG * XG; ++r) {
            uint32_t v0 = ix.iw[r].x & 0xFFFFu, v1 = ix.iw[r].x >> 16, v2 = ix.iw[r].y & 0xFFFFu, v3 = ix.iw[r].y >> 16;
            if (!(abl & 4)) { v0 = lds[v0]; v1 = lds[v1]; v2 = lds[v2]; v3 = lds[v3]; }
            w[r] = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!(abl & 2)) {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                uint8_t *o = out0 + (size_t)f * frame_stride + (size_t)(r * 8) * dst_pitch;
                if (XG == 1) *reinterpret_cast<uint32_t *>(o) = w[r];
                else *reinterpret_cast<uint2 *>(o) = make_uint2(w[r * XG], w[r * XG + XG - 1]);
            }
        }
    }
}

template <bool RUBIX, int RG, int XG>
__device__ __forceinline__ void tile_process(
    const TilePrefetch<RG * XG> &pf, int t, int lane, uint8_t *lds, const uint8_t *pal_s,
    const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames,
    int frame0, uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride, int W, int rows, int gp,
    int blocks_x, int f_begin, int f_end, int lds_per_wave, bool wg_sync)
{
    // header words -> SGPRs (the values are wave-uniform)
    const uint32_t h_src[3] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h0.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h0.y),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h0.z)};
    const uint32_t w01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h0.w), w2r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h1.x);
    const uint32_t r12 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h1.y), nf = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h1.z);
    const uint32_t l16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pf.h1.w) & 0xFFFFu;
    const uint32_t h_w16[3] = {w01 & 0xFFFFu, w01 >> 16, w2r0 & 0xFFFFu};
    const uint32_t h_rows[3] = {w2r0 >> 16, r12 & 0xFFFFu, r12 >> 16};
    const uint32_t h_nreg = nf & 0xFFFFu, h_flags = nf >> 16;
    if (h_flags & F_EMPTY) {
        // (wg_sync: the four waves of a workgroup meet once per frame, whatever path their tile takes)
        if (wg_sync) for (int f = f_begin; f < f_end; ++f) __builtin_amdgcn_s_barrier();
        return;
    }
    if (g_ablate & 16) {                       // developer ablation: header + indices only
        if (pf.ix.iw[0].x == 0x12345678u && h_src[0] == 0xFFFFFFFFu) dst[0] = 1;
        return;
    }

    int ox, oy;
    tile_origin(t, RG, XG, blocks_x, &ox, &oy);
    const int ry = lane >> 3, cx = lane & 7;
    const int row0 = oy + ry, x = ox + cx * 4 * XG;        // group g covers row0 + 8 (g / XG), x + 4 (g % XG)
    // a tile whose regions exceed this launch's LDS budget takes the direct-gather path
    const bool slow = (h_flags & F_SLOW) != 0 || (int)l16 * 16 > lds_per_wave;
    const bool fast_store = (h_flags & F_ALL) && ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dst_pitch | (uintptr_t)frame_stride) & (uintptr_t)(4 * XG - 1)) == 0;

    // Frame-invariant staging plan: the 16-byte chunks of all regions are numbered row-major and
    // chunk c = lane + 64 j is lane's j-th: 16 bytes at globe offset q_src[j] -> LDS offset q_lds[j].
    // Tiles with <= MAXQ chunks per lane (regions <= 4 KiB) take the hot loop.  (A shift/mask "row
    // pass" plan without the division was tried: its idle lanes cost more load instructions than
    // the ALU it saved.)
    constexpr int MAXQ = 4;
    uint32_t q_src[MAXQ], q_lds[MAXQ];
    bool q_ok[MAXQ];
    uint32_t total_chunks = 0;
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) { q_ok[j] = false; q_src[j] = 0; q_lds[j] = 0; }
    if (!slow && h_nreg == 1) {
        // the common case (one plate under the tile): one division per chunk actually needed
        const uint32_t w16 = h_w16[0], pitch16 = w16 | 1u, total = w16 * h_rows[0];
        const uint32_t magic = w16 > 1 ? 0xFFFFFFFFu / w16 + 1u : 0u;
#pragma unroll
        for (int j = 0; j < MAXQ; ++j) {
            if (64u * j >= total) break;                              // wave-uniform
            const uint32_t c = (uint32_t)lane + 64u * j;
            const uint32_t yy = w16 > 1 ? __umulhi(c, magic) : c, xx = c - yy * w16;
            q_ok[j] = c < total;
            q_src[j] = q_ok[j] ? h_src[0] + yy * (uint32_t)gp + xx * 16u : 0u;
            q_lds[j] = (yy * pitch16 + xx) * 16u;
        }
        total_chunks = total;
    } else {
        uint32_t base16 = 0, cbase = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (slow || r >= (int)h_nreg) break;
            const uint32_t w16 = h_w16[r], nrows = h_rows[r], pitch16 = w16 | 1u, total = w16 * nrows;
            const uint32_t magic = w16 > 1 ? 0xFFFFFFFFu / w16 + 1u : 0u;
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const uint32_t c = (uint32_t)lane + 64u * j;          // chunk number across all regions
                if (c >= cbase && c < cbase + total) {
                    const uint32_t cc = c - cbase;
                    const uint32_t yy = w16 > 1 ? __umulhi(cc, magic) : cc, xx = cc - yy * w16;
                    q_ok[j] = true;
                    q_src[j] = h_src[r] + yy * (uint32_t)gp + xx * 16u;
                    q_lds[j] = (base16 + yy * pitch16 + xx) * 16u;
                }
            }
            base16 += nrows * pitch16;
            cbase += total;
        }
        total_chunks = cbase;
    }
    const bool piped = !slow && total_chunks <= 64u * MAXQ;

    if (!RUBIX && piped && fast_store) {
        uint8_t *out0 = dst + (size_t)row0 * dst_pitch + x;
        const uint32_t nq = (total_chunks + 63u) >> 6;       // wave-uniform
#define BK_HOT(N) hot_frames<N, RG, XG>(globe, globe_stride, globe_frames, frame0, f_begin, f_end, out0, dst_pitch, frame_stride, lds, \
                                    q_src[0], q_src[1], q_src[2], q_src[3], q_lds[0], q_lds[1], q_lds[2], q_lds[3],            \
                                    q_ok[0], q_ok[1], q_ok[2], q_ok[3], pf.ix, wg_sync)
        if (nq <= 1) BK_HOT(1);
        else if (nq == 2) BK_HOT(2);
        else if (nq == 3) BK_HOT(3);
        else BK_HOT(4);
#undef BK_HOT
        return;
    }

    // general path: partially mapped tiles, rubix, large regions, the direct-gather fallback
    if ((g_ablate & 32) && !slow) return;      // developer ablations: skip staged-general / direct-gather tiles
    if ((g_ablate & 64) && slow) return;
    constexpr int NG = RG * XG;
    uint32_t so[NG][4];
    if (slow) {
#pragma unroll
        for (int r = 0; r < NG; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = row0 + (r / XG) * 8, xx = x + (r % XG) * 4 + k;
                so[r][k] = (row < rows && xx < W) ? lmap[(size_t)row * W + xx] : BK_NULL_OFFSET;
            }
    }
    for (int f = f_begin; f < f_end; ++f) {
        if (wg_sync) __builtin_amdgcn_s_barrier();
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        if (!slow) {
            // stage the regions in rounds of 4 chunks per lane: the four 16-byte loads of a round are
            // issued back to back before their LDS stores (one exposed memory latency per round, not
            // per chunk; rows of the padded globe are 64-byte aligned)
            int base16 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r >= (int)h_nreg) break;
                const uint32_t w16 = h_w16[r], nrows = h_rows[r], pitch16 = w16 | 1u, total = w16 * nrows;
                const uint32_t magic = w16 > 1 ? 0xFFFFFFFFu / w16 + 1u : 0u;
                const uint8_t *src = gl + h_src[r];
                for (uint32_t c0 = 0; c0 < total; c0 += 256) {
                    uint4 qq[4];
                    uint32_t dd[4];
                    bool okk[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t c = c0 + (uint32_t)lane + 64u * j;
                        okk[j] = c < total;
                        const uint32_t cc = okk[j] ? c : 0u;
                        const uint32_t yy = w16 > 1 ? __umulhi(cc, magic) : cc, xx = cc - yy * w16;
                        qq[j] = *reinterpret_cast<const uint4 *>(src + (size_t)yy * gp + xx * 16);
                        dd[j] = (uint32_t)(base16 + yy * pitch16 + xx) * 16u;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (okk[j]) *reinterpret_cast<uint4 *>(lds + dd[j]) = qq[j];
                }
                base16 += nrows * pitch16;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int r = 0; r < NG; ++r) {
            const uint32_t a[4] = {pf.ix.iw[r].x & 0xFFFFu, pf.ix.iw[r].x >> 16, pf.ix.iw[r].y & 0xFFFFu, pf.ix.iw[r].y >> 16};
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (slow) v[k] = so[r][k] != BK_NULL_OFFSET ? gl[so[r][k]] : 0u;
                else v[k] = a[k] != 0xFFFFu ? lds[a[k]] : 0u;
            }
            if (RUBIX) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t tt = (pf.ix.t4[r] >> (8 * k)) & 0xFFu;
                    if (tt != 255u) v[k] = pal_s[tt * 256 + v[k]];
                }
            }
            uint8_t *out = dst + (size_t)f * frame_stride + (size_t)(row0 + (r / XG) * 8) * dst_pitch + x + (r % XG) * 4;
            if (fast_store) {
                *reinterpret_cast<uint32_t *>(out) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (a[k] != 0xFFFFu) out[k] = (uint8_t)v[k];
            }
        }
        if (!slow) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Persistent launch: the grid holds a bounded number of workgroups per CU; every wave walks a
// strided sequence of tiles inside its XCD's band and fetches the header / indices of its next tile
// before it starts on the current one.
template <bool RUBIX, int RG, int XG>
__global__ __launch_bounds__(256) void apply_tiled_kernel(
    const TileHdr *__restrict__ hdr, const uint16_t *__restrict__ idx, const uint8_t *__restrict__ tint_t,
    const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames,
    int frame0, uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride, int W, int rows, int gp,
    int blocks_x, int nblocks, int nframes, int fchunk, int lds_per_wave, const uint8_t *__restrict__ pal, int flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t *pal_s = smem + 4 * lds_per_wave;
    if (RUBIX) {
        for (int i = threadIdx.x; i < BK_MAX_PLATES * 256; i += 256) pal_s[i] = pal[i];
        __syncthreads();
    }
    uint8_t *lds = smem + wave * lds_per_wave;
    // XCD-banded mapping: workgroup b runs on XCD b % 8 (observed dispatch order); XCD k owns the
    // contiguous band [k*per, (k+1)*per) of workgroup blocks.  Correctness does not depend on it.
    const int per = (nblocks + 7) / 8;
    const int band = (int)(blockIdx.x & 7);
    const int wg_in_band = (int)(blockIdx.x >> 3), wgs_per_band = (int)(gridDim.x >> 3);
    const int l_end = min(nblocks, (band + 1) * per);
    int l = band * per + wg_in_band;
    if (l >= l_end) return;
    const int f_begin = blockIdx.y * fchunk, f_end = min(nframes, f_begin + fchunk);

    int t = __builtin_amdgcn_readfirstlane(l * 4 + wave);
    TilePrefetch<RG * XG> cur = tile_fetch<RUBIX, RG, XG>(hdr, idx, tint_t, t, lane);
    for (;;) {
        const int l_next = l + wgs_per_band;
        const bool has_next = l_next < l_end;
        const int t_next = __builtin_amdgcn_readfirstlane(l_next * 4 + wave);
        TilePrefetch<RG * XG> nxt = cur;
        if (has_next) nxt = tile_fetch<RUBIX, RG, XG>(hdr, idx, tint_t, t_next, lane);
        tile_process<RUBIX, RG, XG>(cur, t, lane, lds, pal_s, lmap, globe, globe_stride, globe_frames, frame0, dst, dst_pitch,
                                frame_stride, W, rows, gp, blocks_x, f_begin, f_end, lds_per_wave, (flags & 1) != 0);
        if (!has_next) break;
        l = l_next;
        t = t_next;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void tilemap_free(TileMap *tm)
{
    if (!tm) return;
    (void)hipFree(tm->d_hdr);
    (void)hipFree(tm->d_idx);
    (void)hipFree(tm->d_tint);
    (void)hipFree(tm->d_stats);
    delete tm;
}

void tilemap_invalidate(bk_ctx *ctx)
{
    if (ctx->tilemap) ctx->tilemap->valid = false;
    coopmap_invalidate(ctx);
}

// compile the tilemap for one tile height and read back its statistics
static int compile_shape(bk_ctx *ctx, TileMap *tm, int shape, double *cost_ps)
{
    const int rows = ctx->rows();
    const int rg = shape & 7, xg = shape >= 8 ? 2 : 1;       // shape = rg + 8 * (xg - 1)
    tm->rg = rg;
    tm->xg = xg;
    tm->blocks_x = (ctx->W + 128 * xg - 1) / (128 * xg);
    tm->blocks_y = (rows + 8 * rg - 1) / (8 * rg);
    const size_t ntiles = (size_t)tm->blocks_x * tm->blocks_y * 4;
    BK_HIP(ctx, hipMemsetAsync(tm->d_stats, 0, 64 * 40 * sizeof(uint32_t), ctx->stream));
    const dim3 grid((unsigned)((ntiles + 3) / 4)), block(256);
#define BK_COMPILE(N, X) hipLaunchKernelGGL((tile_compile_kernel<N, X>), grid, block, 0, ctx->stream, ctx->d_offsets, ctx->d_tints, ctx->W, rows, \
                                            ctx->ps, ctx->gp, tm->blocks_x, (int)ntiles, tm->d_hdr, tm->d_idx, tm->d_tint, tm->d_stats)
    if (xg == 1) { if (rg == 1) BK_COMPILE(1, 1); else if (rg == 2) BK_COMPILE(2, 1); else BK_COMPILE(4, 1); }
    else { if (rg == 1) BK_COMPILE(1, 2); else BK_COMPILE(2, 2); }
#undef BK_COMPILE
    BK_HIP(ctx, hipGetLastError());
    static thread_local uint32_t rep[64 * 40];
    BK_HIP(ctx, hipMemcpyAsync(rep, tm->d_stats, sizeof rep, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 40; ++k) tm->stats[k] = 0;
    for (int r = 0; r < 64; ++r) {
        tm->stats[0] = rep[r * 40] > tm->stats[0] ? rep[r * 40] : tm->stats[0];
        for (int k = 1; k < 40; ++k) tm->stats[k] += rep[r * 40 + k];
    }
    // LDS budget per wave, chosen by a small cost model over the histogram of per-tile needs: tiles
    // above the budget take the direct-gather path (~3x the time of a staged tile) while a larger
    // budget lowers occupancy for everyone (workgroups per CU = min(6, 160 KiB / (4 * budget));
    // 6 is what the kernel's SGPR count admits).
    uint64_t staged_all = 0;
    for (int b = 0; b < 32; ++b) staged_all += tm->stats[4 + b];
    int best_bin = 1;
    double best_c = -1;
    for (int bin = 1; bin * 512 <= BK_TILE_LDS_CAP; ++bin) {
        uint64_t fit = 0;
        for (int b = 0; b <= bin; ++b) fit += tm->stats[4 + b];
        const uint64_t over = staged_all - fit;
        int wgs = (160 * 1024) / (4 * bin * 512);
        if (wgs > 6) wgs = 6;
        if (wgs < 1) wgs = 1;
        const double c = ((double)fit + 3.0 * (double)over) * 6.0 / wgs;
        if (best_c < 0 || c < best_c) { best_c = c; best_bin = bin; }
    }
    const int cap = best_bin * 512;
    uint64_t over = 0;
    for (int b = best_bin + 1; b < 32; ++b) over += tm->stats[4 + b];
    tm->lds_bytes = cap;
    tm->slow_tiles = (int)(tm->stats[1] + over);
    // cost model (ps per frame) for comparing tile heights: a staged 128-byte line, per-tile fixed work,
    // a tile on the direct-gather path (per pixel group)
    const double nonempty = (double)(staged_all + tm->stats[1]);
    double c = 8.6 * tm->stats[3] + 120.0 * nonempty + 742.0 * rg * xg * tm->slow_tiles;
    if (cap > 4096) c *= 1.0 + (cap - 4096) / 8192.0;
    *cost_ps = c;
    return BK_OK;
}

static int ensure_tilemap(bk_ctx *ctx)
{
    if (!ctx->tilemap) ctx->tilemap = new TileMap();
    TileMap *tm = ctx->tilemap;
    if (tm->valid) return BK_OK;
    const int rows = ctx->rows();
    // buffers sized for the shortest tiles (most tiles); every height covers <= that many pixels + padding
    const size_t bx = (ctx->W + 127) / 128 + 1;
    const size_t max_px = bx * 4 * 256 * (size_t)((rows + 31) / 32 * 4 + 4);
    const size_t max_tiles = bx * 4 * (size_t)((rows + 7) / 8);
    if (max_px > tm->alloc_px) {
        (void)hipFree(tm->d_hdr); (void)hipFree(tm->d_idx); (void)hipFree(tm->d_tint);
        tm->d_hdr = nullptr; tm->d_idx = nullptr; tm->d_tint = nullptr;
        BK_HIP(ctx, hipMalloc((void **)&tm->d_hdr, max_tiles * sizeof(TileHdr)));
        BK_HIP(ctx, hipMalloc((void **)&tm->d_idx, max_px * sizeof(uint16_t)));
        BK_HIP(ctx, hipMalloc((void **)&tm->d_tint, max_px));
        tm->alloc_px = max_px;
    }
    if (!tm->d_stats) BK_HIP(ctx, hipMalloc((void **)&tm->d_stats, 64 * 40 * sizeof(uint32_t)));
    // ctx->tile_shape: 0 = default (32x16), shape = rg + 8*(xg-1): 1/2/4 = 32x8/16/32, 9/10 = 64x8/16;
    // -1 = compile every shape and keep the cheapest by the cost model
    static const int shapes[5] = {1, 2, 4, 9, 10};
    int best = ctx->tile_shape, compiled = -1;
    if (best == 0) best = 2;
    bool known = false;
    for (int sh : shapes) known = known || sh == best;
    if (!known) {
        double best_cost = 0;
        best = -1;
        for (int sh : shapes) {
            double c = 0;
            if (int r = compile_shape(ctx, tm, sh, &c)) return r;
            compiled = sh;
            if (best < 0 || c < best_cost) { best = sh; best_cost = c; }
        }
    }
    if (compiled != best) {
        double c = 0;
        if (int r = compile_shape(ctx, tm, best, &c)) return r;
    }
    tm->valid = true;
    return BK_OK;
}

int launch_apply_tiled(bk_ctx *ctx, int frame0, int nframes, uint8_t *dst, int dst_pitch, size_t frame_stride, int rubix_on)
{
    const int rows = ctx->rows();
    if (rows <= 0 || nframes <= 0) return BK_OK;
    if (int r = ensure_tilemap(ctx)) return r;
    TileMap *tm = ctx->tilemap;
    const int blocks_x = tm->blocks_x, nblocks = blocks_x * tm->blocks_y;
    const int fmax = ctx->apply_fchunk > 0 ? ctx->apply_fchunk : 8;
    const int fchunk = nframes < fmax ? nframes : fmax;
    const int fblocks = (nframes + fchunk - 1) / fchunk;
    const int per = (nblocks + 7) / 8;
    // persistent grid: enough workgroups to fill the chip a few times over; each walks its XCD band
    // with a stride, prefetching its next tile's header.
    int wgs_per_band = per;
    const int resident_per_band = ctx->num_cus * ctx->apply_wgs_per_cu / 8;
    if (fblocks * wgs_per_band > resident_per_band) wgs_per_band = (resident_per_band + fblocks - 1) / fblocks;
    if (wgs_per_band < 1) wgs_per_band = 1;
    if (wgs_per_band > per) wgs_per_band = per;
    dim3 grid((unsigned)(wgs_per_band * 8), (unsigned)fblocks);
    const size_t shmem = (size_t)4 * tm->lds_bytes + (rubix_on ? BK_MAX_PLATES * 256 : 0);
#define BK_APPLY(RBX, N, X) hipLaunchKernelGGL((apply_tiled_kernel<RBX, N, X>), grid, dim3(256), shmem, ctx->stream, tm->d_hdr, tm->d_idx, tm->d_tint,      \
                                            ctx->d_offsets, ctx->d_globe, ctx->globe_stride(), ctx->nframes, frame0, dst, dst_pitch, frame_stride, \
                                            ctx->W, rows, ctx->gp, blocks_x, nblocks, nframes, fchunk, tm->lds_bytes, ctx->d_pal, ctx->apply_flags)
    if (tm->xg == 1) {
        if (rubix_on) { if (tm->rg == 1) BK_APPLY(true, 1, 1); else if (tm->rg == 2) BK_APPLY(true, 2, 1); else BK_APPLY(true, 4, 1); }
        else { if (tm->rg == 1) BK_APPLY(false, 1, 1); else if (tm->rg == 2) BK_APPLY(false, 2, 1); else BK_APPLY(false, 4, 1); }
    } else {
        if (rubix_on) { if (tm->rg == 1) BK_APPLY(true, 1, 2); else BK_APPLY(true, 2, 2); }
        else { if (tm->rg == 1) BK_APPLY(false, 1, 2); else BK_APPLY(false, 2, 2); }
    }
#undef BK_APPLY
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

int set_ablation(bk_ctx *ctx, int bits)
{
    BK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &bits, sizeof(int)));
    return BK_OK;
}

int tilemap_stats(bk_ctx *ctx, int out[6])
{
    if (int r = ensure_tilemap(ctx)) return r;
    TileMap *tm = ctx->tilemap;
    out[0] = tm->blocks_x * tm->blocks_y * 4; out[1] = tm->slow_tiles; out[2] = (int)tm->stats[2]; out[3] = tm->lds_bytes;
    out[4] = 8 * tm->rg + 1000 * 32 * tm->xg; out[5] = (int)tm->stats[3];       // width*1000 + height
    return BK_OK;
}

}  // namespace bk
