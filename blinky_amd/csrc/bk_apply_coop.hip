// blinky-hip: workgroup-cooperative LDS-staged apply kernel (variant 2, the default) for gfx950.
//
// A predecessor that staged, per wave, the bounding boxes of the plate texels under a 32-pixel-wide tile
// (git history: bk_apply_tiled.hip; numbers in DESIGN.md 3.2) showed what limits this gather on MI355X:
// with row-major plates a 128-byte line spans ~3.5 such tiles and screen rows are slanted in plate space,
// so the same lines were requested over and over (TCP->TCC read requests = 5-6x the unique lines, L1
// hit rate ~23 %, TCP_PENDING_STALL ~2/3 of the kernel's cycles; kernel time tracked that request count)
// and, because waves drift apart in time, part of those requests went to HBM again.
//
// Here the unit of staging is the workgroup block of 128 x (8*RG) pixels and what is staged is the
// EXACT set of 16-byte globe chunks the block's pixels read - no bounding boxes.  `coop_compile_kernel`
// sorts the block's chunk numbers (bitonic sort in LDS), keeps the unique ones in ascending address
// order as the block's chunk list, and rewrites every pixel's lensmap entry as a 16-bit LDS address
// (slot * 16 + byte).  Per frame the 256 threads copy chunk list entry i to LDS slot i (neighbouring
// lanes fetch neighbouring chunks of a line: one request per line and block), meet at an s_barrier, gather
// their pixels from the shared copy (ds_read_u8) and store them.  A lane owns 4*RG consecutive pixels of one
// row and 32/RG lanes span the block's width, so every store instruction of a wave writes whole 128-byte
// lines (32-byte row segments per wave cost 1.3x in kernel time); the stores are non-temporal (the frame is
// never read back; L2 stays with the globe lines blocks share).  One staging buffer, two barriers per frame
// (chunks visible / gather done): measured against two alternating buffers with one barrier, the halved LDS
// footprint - more resident workgroups per CU - wins whenever LDS limits occupancy and costs nothing when
// it does not.  Lists larger than the buffer go through it in passes.
//
// Two kernel forms: persistent (a workgroup walks a strided list of blocks in its XCD's band and prefetches the
// next block's header, chunk list head and indices) and one-block-per-workgroup for launches whose whole grid
// is resident at once (fewer registers).  A batch launch re-uses a block's plan for up to 8 frames.
// The chunk list is layout-agnostic (byte offsets into a globe frame); with the globe stored as 16x8-texel
// lines (bk_build_params.h) a block's slanted footprint touches about half the lines it did row-major.
//
// replaces render_lensmap (engine/NQ/fisheye.c:2406-2424); byte-exact.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "bk_internal.h"

namespace bk {

constexpr int BK_PAL_BYTES = BK_MAX_PLATES * 256;         // the tint LUTs in LDS
constexpr int BK_COOP_MAX_FLIPS = 4;                   // recompiles per lensmap for a caller that alternates kinds of launch wanting different block heights
constexpr int BK_COOP_LDS_CAP = 65536;                 // max bytes of the staging buffer (a block has <= 4095 chunks)
constexpr uint32_t BK_COOP_MAX_CHUNKS = 4095;          // 16-bit LDS addresses: slot*16 + byte, 0xFFFF = unmapped
constexpr uint32_t CF_ALL = 0x1, CF_NONE = 0x10;       // << wave: that wave's rows of the block fully mapped / empty
constexpr uint32_t CF_SLOW = 0x100, CF_EMPTY = 0x200;  // direct-gather block / nothing mapped in the block
constexpr uint32_t BK_COOP_BINS = 65;                  // LDS-need histogram: 1 KiB bins, 0..64 KiB
constexpr int BK_COOP_STATS = 208;                     // words per stats replica
constexpr int BK_KF_WGMAP = 1 << 30;                   // kflags (set by the launcher): one-block form reads CoopMap::d_wgmap
constexpr int BK_COOP_BLOCK_COST = 4;                  // what a block costs beyond its lines and pixels, in lines (barriers, header)

struct CoopHdr {              // 8 bytes per block
    uint32_t nchunks;         // entries of the block's chunk list (0 for direct-gather / empty blocks)
    uint32_t flags;
};

struct CoopMap {
    CoopHdr *d_hdr = nullptr;
    uint32_t *d_list = nullptr;     // [nblocks][256*4*RG] byte offsets (16-byte aligned) into a globe frame, ascending; a TINTED map (below)
                                    // keeps the chunk's tint class in the low three bits: 0 = as it is, c = through palette row c - 1
    uint16_t *d_idx = nullptr;      // [nblocks][4 waves][RG][64 lanes][4] LDS addresses, 0xFFFF = unmapped
    // (r5) rubix: the tint is applied to the STAGED CHUNK, not to the gathered pixel.  A pixel's tint is a property of the texel it reads
    // (set_lensmap_grid, fisheye.c:1922-1960: the plate's number where the texel is on the grid), so a tinted map lists a chunk once per
    // tint class its pixels need - (chunk, class) is what is sorted and made unique - and the staging pass sends the 16 bytes of a
    // classed chunk through that palette row on their way to LDS.  The gather, its registers and its stores are the plain apply's:
    // no tint byte per pixel in registers (the tinted 128x32 kernel ran at 94 VGPRs = 5 workgroups per CU = two rounds of workgroups
    // for a 4K frame), no tint plane read per block visit.  At the default grid 10 % of the chunks are listed twice (a 16-texel row
    // that crosses a cell's edge).  Any tint plane works - bk_set_lensmap's too: classes are the tint VALUES.
    bool tinted = false;            // compiled for rubix launches; a plain launch gets a plain map (ensure_coopmap recompiles on a switch)
    // Work-balanced walk of the persistent apply: the live (non-empty) blocks in walk order, and where each XCD's band of
    // them starts - bands of equal COST (128-byte lines staged + a constant per block), not of equal block count.  A lens
    // that leaves part of the screen unmapped (hammer's ellipse, quincuncial under f_contain) otherwise gives the XCDs that
    // own the top and the bottom of the screen a fraction of the work of the ones in the middle (4K hammer: 42 K vs 362 K
    // chunks per band; the launch lasts as long as the fullest band).
    uint32_t *d_cost = nullptr;     // [nblocks] lines + BK_COOP_BLOCK_COST (0 = empty block)
    uint32_t *d_order = nullptr;    // [nblocks] block numbers of the live blocks, in walk order (bk_block_at)
    uint32_t *d_cum = nullptr;      // [nblocks] inclusive cost prefix over d_order (scratch of coop_order_kernel)
    uint32_t *d_bands = nullptr;    // [9] start of band k in d_order; [8] = live blocks
    uint32_t *d_wgmap = nullptr;    // [8 * ceil(nblocks / 8)] one-block-per-workgroup form: the block of workgroup b (XCD b % 8 takes
                                    // band b % 8), 0xFFFFFFFF = none; used when equal-count bands would be uneven (stats[7])
    uint32_t *d_stats = nullptr;    // 64 replicas of: [0] max chunks, [1] direct-gather blocks, [2] empty blocks,
                                    // [3] 128-B lines staged, [4] chunks staged, [8..73) blocks by LDS need (1 KiB bins),
                                    // [73..138) 128-B lines of those blocks, [138..203) chunks of those blocks
    int blocks_x = 0, blocks_y = 0;
    int rg = 4;
    int lds_bytes = 0;              // bytes of the staging buffer of the apply launch
    int tuned_frames = 0;           // frames per launch the block height was measured with (0 = cost model alone)
    int single_form = 0;            // single-frame launches: 0 = the launcher's rule, 1 = one block per workgroup, 2 = strided walk (measured)
    int fchunk = 0;                 // batch launches: frames a workgroup keeps a block for, 0 = the launcher's 8 (measured: 4 where the grid is small)
    bool lds_fixed = false;         // the staging buffer size was measured: the exact statistics do not re-choose it
    // what was measured, per KIND of launch (single frame / up to 16 / more): a caller that goes back and forth between kinds
    // measures each once; going back costs a recompile only where the kinds want different block heights, and that at most
    // BK_COOP_MAX_FLIPS times per lensmap (after that the map stays as it is)
    struct Tuned { int rg = 0, kb = 0, form = 0, fchunk = 0, frames = 0; } tuned[3];
    int flips = 0;
    int grown_rg = 0;               // != 0: this map's block height is one a resident session grew it to (res_launch): later sessions do not grow it again
    uint32_t stats[BK_COOP_STATS] = {0};
    uint32_t *h_stats = nullptr;    // pinned [64][BK_COOP_STATS]: the full compile's statistics land here asynchronously ...
    hipEvent_t stats_ready = nullptr;   // ... and are folded into `stats` when somebody asks (coopmap_stats / traffic model)
    bool stats_pending = false;
    int slow_blocks = 0;
    bool valid = false;
    size_t alloc_px = 0, alloc_blocks = 0;
};

// ---------------------------------------------------------------------------------------------
// compile: one workgroup per block
// ---------------------------------------------------------------------------------------------
template <int RG>
__global__ __launch_bounds__(256) void coop_compile_kernel(const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ tints,
                                                           int W, int rows, int blocks_x, int nblocks,
                                                           CoopHdr *__restrict__ hdr, uint32_t *__restrict__ list,
                                                           uint16_t *__restrict__ idx, int tinted,
                                                           uint32_t *__restrict__ stats, int row_stride, uint32_t *__restrict__ cost,
                                                           int block_cost)
{
    // row_stride > 1: a SURVEY pass for the cost model - only every row_stride-th row of blocks is looked at and nothing
    // but the statistics is written (ensure_coopmap scales them up); row_stride == 1: the real block map
    constexpr int NP = 4 * RG, N = 256 * NP;      // pixels per thread / per block
    __shared__ uint32_t key[N];                   // chunk numbers, sorted in place
    __shared__ uint32_t uniq[N];                  // unique chunk numbers, ascending
    __shared__ uint32_t s_wsum[4], s_wlines[4], s_wpx[4];
    __shared__ uint32_t s_flags;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool survey = row_stride > 1;
    const int sby = blockIdx.x / blocks_x, bx = blockIdx.x - sby * blocks_x;
    const int by = sby * row_stride, blk = by * blocks_x + bx;
    // pixel -> lane: a lane owns 4*RG consecutive pixels of ONE row (RG groups of 4), 32/RG lanes span the
    // block's 128-pixel width and a wave covers 2*RG consecutive rows: every store instruction of the apply
    // kernel writes whole 128-byte lines (one dword / dwordx2 / dwordx4 per lane)
    constexpr int LPR = 32 / RG;                  // lanes per row
    const int oy = by * 8 * RG;
    const int ry = wave * 2 * RG + lane / LPR;
    const int x0 = bx * 128 + (lane % LPR) * 4 * RG;
    if (threadIdx.x == 0) s_flags = 0;

    uint32_t o[NP];
    uint32_t kc[NP];                              // sort key: chunk number * 8 + tint class (0 on a plain map)
    bool all_l = true, any_l = false;
    uint32_t npx = 0;                             // mapped pixels of this lane
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = oy + ry, x = x0 + i;
        const bool in = row < rows && x < W;
        o[i] = in ? lmap[(size_t)row * W + x] : BK_NULL_OFFSET;
        const uint32_t tn = in && tinted ? tints[(size_t)row * W + x] : 255u;
        kc[i] = o[i] == BK_NULL_OFFSET ? 0xFFFFFFFFu : ((o[i] >> 4) << 3) | (tn < (uint32_t)BK_MAX_PLATES ? tn + 1u : 0u);
        key[threadIdx.x * NP + i] = kc[i];
        all_l = all_l && o[i] != BK_NULL_OFFSET;
        any_l = any_l || o[i] != BK_NULL_OFFSET;
        npx += o[i] != BK_NULL_OFFSET ? 1u : 0u;
    }
    const bool all = __all(all_l), any = __any(any_l);
    __syncthreads();
    if (lane == 0) atomicOr(&s_flags, (all ? CF_ALL << wave : 0u) | (any ? 0u : CF_NONE << wave));

    // bitonic sort of the N chunk numbers, ascending (unmapped = 0xFFFFFFFF sorts last)
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const uint32_t a = key[i], b = key[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
    }

    // unique: thread t owns sorted positions [t*NP, t*NP+NP)
    uint32_t cnt = 0, lcnt = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = threadIdx.x * NP + i;
        const uint32_t v = key[p], prev = p > 0 ? key[p - 1] : 0xFFFFFFFFu;
        const bool first = v != 0xFFFFFFFFu && (p == 0 || v != prev);
        cnt += first ? 1u : 0u;
        lcnt += (first && (p == 0 || (v >> 6) != (prev >> 6))) ? 1u : 0u;       // a new 128-byte line
    }
    uint32_t incl = cnt, lsum = lcnt;
    for (int m = 1; m < 64; m <<= 1) {
        const uint32_t u = __shfl_up(incl, m);
        if (lane >= m) incl += u;
    }
    for (int m = 32; m >= 1; m >>= 1) { lsum += __shfl_xor(lsum, m); npx += __shfl_xor(npx, m); }
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) { s_wlines[wave] = lsum; s_wpx[wave] = npx; }
    __syncthreads();
    uint32_t base = incl - cnt;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    const uint32_t nchunks = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    const uint32_t lines = s_wlines[0] + s_wlines[1] + s_wlines[2] + s_wlines[3];
    const uint32_t mapped = s_wpx[0] + s_wpx[1] + s_wpx[2] + s_wpx[3];
    const bool slow = nchunks > BK_COOP_MAX_CHUNKS;        // (only a 128x32 block whose 4096 pixels all read different chunks)
    {
        uint32_t k = base;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = threadIdx.x * NP + i;
            const uint32_t v = key[p];
            if (v != 0xFFFFFFFFu && (p == 0 || v != key[p - 1])) {
                uniq[k] = v;
                if (!slow && !survey) list[(size_t)blk * N + k] = ((v >> 3) << 4) | (v & 7u);
                ++k;
            }
        }
    }
    __syncthreads();

    // every pixel: slot of its chunk (binary search in the unique list) -> 16-bit LDS address
    if (!survey) {
    // (Tried: storing a u16 base per lane or per 4 pixels plus u8 offsets, 1.1-1.5 instead of 2 bytes per pixel.  Slots
    //  follow chunk numbers, so a row segment that crosses a tile-row boundary jumps by the whole staged tile row
    //  (> 1 KiB): 3 of 510 panini blocks at 1080p could use the compact form.  Removed again.
    //  Round 3: one byte per pixel of chained DIFFERENCES along the row, a 16-bit exception list for such jumps (4K panini 1.08,
    //  hammer 1.20 bytes per pixel), decoded once per block visit - parity-green, and slower everywhere: the ~130 VALU instructions
    //  of the decode run in every workgroup of a round at the same moment, exactly where the raw form has its addresses waiting in
    //  registers (4K panini single frame 8.5 -> 10.7 us, x16 +3-5 %; profiles/r03_barrier_and_pipelining_experiments.txt (5)).)
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        uint32_t a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = r * 4 + k;
            a[k] = 0xFFFFu;
            if (o[i] != BK_NULL_OFFSET) {
                a[k] = 0;
                if (!slow) {
                    const uint32_t c = kc[i];
                    uint32_t lo = 0, hi = nchunks;              // first slot with uniq[slot] >= c
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (uniq[mid] < c) lo = mid + 1; else hi = mid;
                    }
                    a[k] = lo * 16u + (o[i] & 15u);
                }
            }
        }
        const size_t slab = (((size_t)blk * 4 + wave) * RG + r) * 256 + (size_t)lane * 4;
        *reinterpret_cast<uint2 *>(idx + slab) = make_uint2(a[0] | (a[1] << 16), a[2] | (a[3] << 16));
    }
    }
    if (threadIdx.x == 0) {
        const uint32_t wflags = s_flags;
        const bool any_blk = nchunks != 0;
        CoopHdr h;
        h.nchunks = slow ? 0u : nchunks;
        h.flags = wflags | (slow ? CF_SLOW : 0u) | (any_blk ? 0u : CF_EMPTY);
        if (!survey) {
            hdr[blk] = h;
            // in units of one 128-byte line: the lines staged, the pixels stored (+ their share of the 2-byte addresses), a constant
            cost[blk] = !any_blk ? 0u : slow ? (uint32_t)N / 4u : lines + mapped * 5u / 512u + (uint32_t)block_cost;
        }
        uint32_t *st = stats + (blk & 63) * BK_COOP_STATS;
        if (!slow && any_blk) {
            atomicMax(&st[0], nchunks);
            const uint32_t bin = min(BK_COOP_BINS - 1u, (nchunks * 16u + 1023u) / 1024u);
            atomicAdd(&st[8 + bin], 1u);
            atomicAdd(&st[8 + BK_COOP_BINS + bin], lines);
            atomicAdd(&st[8 + 2 * BK_COOP_BINS + bin], nchunks);
            atomicAdd(&st[3], lines);
            atomicAdd(&st[4], nchunks);
        }
        if (slow) atomicAdd(&st[1], 1u);
        if (!any_blk) atomicAdd(&st[2], 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------
// The staging buffer's two barriers order LDS traffic only - chunks written / gather done.  __syncthreads() is a workgroup-scope FENCE as
// well: hipcc puts an s_waitcnt vmcnt(0) at it, which makes every wave wait there for the frame stores it has just issued and for the
// next frame's globe loads it has just put in flight - the very latency the pipelining is meant to hide.  No wave of this kernel reads global
// memory another wave wrote, so the raw barrier with the LDS counter alone is enough (ablation bit 2048 restores __syncthreads()).
#define BK_LDS_BARRIER()                                                           \
    do {                                                                           \
        if (kflags & 2048) __syncthreads();                                        \
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    } while (0)

template <int RG>
struct CoopIdx {              // a lane's LDS addresses (two 16-bit per dword), per row group
    uint2 iw[RG];
};
// (r5) a chunk of a TINTED block map on its way to LDS (CoopMap::tinted): list entry `e` carries the chunk's tint class in its low bits -
// class c > 0: the 16 texels go through palette row c - 1 here, once per chunk, and the gather behind it is the plain one
__device__ __forceinline__ uint4 bk_tint_chunk(uint4 q, uint32_t e, const uint8_t *pal_s)
{
    const uint32_t cls = e & 7u;
    if (cls) {
        const uint8_t *row = pal_s + (cls - 1u) * 256u;
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)row[w[i] & 0xFFu] | ((uint32_t)row[(w[i] >> 8) & 0xFFu] << 8) | ((uint32_t)row[(w[i] >> 16) & 0xFFu] << 16) |
                   ((uint32_t)row[w[i] >> 24] << 24);
        q = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return q;
}
#define BK_CHUNK_OFF(E_) (RUBIX ? (E_) & ~15u : (E_))            /* a list entry's byte offset into the globe frame */
#define BK_CHUNK_VAL(Q_, E_) (RUBIX && !(kflags & 65536) ? bk_tint_chunk((Q_), (E_), pal_s) : (Q_))   /* (developer bit 65536: timing without the tint) */
// What a thread fetches ahead for its workgroup's NEXT block: the header (a vector load, so that it
// is tracked by vmcnt like everything else) and its first four chunk-list entries.  The per-pixel LDS addresses are
// loaded by coop_block itself: they are not needed before the first gather, so that load rides behind the header and
// the globe loads instead of in front of them (8K hammer, 64 frames: 35.9 -> 34.6 us/frame), and the persistent form
// no longer carries a second set of them in registers.
template <int RG>
struct CoopPrefetch {
    uint2 h;
    uint32_t c[4];
};

// Launch order of the blocks: position l of the walk -> block number (row-major by * blocks_x + bx, which is how the
// block map is stored).  Patches of 8 block rows are walked column by column, so that the workgroups in flight at any
// moment - consecutive positions - cover a compact 2-D region of the screen: the globe lines that vertically
// neighbouring blocks share (1.3-1.4x of the distinct lines are staged, summed over blocks) are then requested close
// together in time and meet in L2 instead of going to HBM twice (8K hammer x 64 frames: 34.6 -> 33.9 us/frame; 4K
// hammer 9.0 -> 8.8, panini 3.83 -> 3.71; quincuncial 8.3 -> 8.5).  Ablation bit 16 restores the row-major walk.
__device__ __forceinline__ int bk_block_at(int l, int blocks_x, int nblocks, int kflags)
{
    if (kflags & 16) return l;
    constexpr int PH = 8;
    const int blocks_y = nblocks / blocks_x, per_patch = PH * blocks_x;
    const int srow = l / per_patch, rem = l - srow * per_patch;
    const int tall = min(PH, blocks_y - srow * PH);
    const int col = rem / tall, row = rem - col * tall;
    return (srow * PH + row) * blocks_x + col;
}

template <bool RUBIX, int RG>
__device__ __forceinline__ CoopPrefetch<RG> coop_fetch(const CoopHdr *__restrict__ hdr, const uint32_t *__restrict__ list, int blk)
{
    constexpr int N = 1024 * RG;
    CoopPrefetch<RG> p;
    int bv;
    asm volatile("v_mov_b32 %0, %1" : "=v"(bv) : "s"(blk));     // make the address a VGPR: vector load
    p.h = *(reinterpret_cast<const uint2 *>(hdr) + (size_t)bv);
    const uint32_t *lp = list + (size_t)blk * N + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) p.c[j] = 256 * j < N ? lp[256 * j] : 0u;
    return p;
}

// the block's per-pixel LDS addresses
template <bool RUBIX, int RG>
__device__ __forceinline__ CoopIdx<RG> coop_load_idx(const uint16_t *__restrict__ idx, const uint8_t *__restrict__, int blk,
                                                     int wave, int lane)
{
    CoopIdx<RG> ix;
    // (non-temporal loads of the list / indices were measured: slower, single frames by 25 % - between launches
    // they are served from L2 / Infinity Cache)
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const size_t slab = (((size_t)blk * 4 + wave) * RG + r) * 256 + (size_t)lane * 4;
        ix.iw[r] = *reinterpret_cast<const uint2 *>(idx + slab);
    }
    return ix;
}

// Stores of the warped frame.  WT = false: non-temporal (the frame is never read back here; a kernel boundary publishes it).
// WT = true (the resident apply, bk_apply_resident.inc): write-through `sc1` stores - the frame is published by a flag while the
// kernel goes on running, so its bytes have to be in memory, not in the XCD's L2, when the flag is raised; a 16-byte sc1 store
// costs what a plain one does and drops the line from L2, which is what a frame nobody reads back wants anyway.  They are inline
// assembly: the compiler does not count them in vmcnt - the publisher waits with an explicit s_waitcnt vmcnt(0).
// (WT: write-through BUFFER stores - the frame's base in a resource descriptor (SGPRs), a 32-bit byte offset per lane, aux 16 = sc1.
//  They are builtins, not inline assembly, on purpose: the compiler counts them in vmcnt.  Memory operations complete in order, and
//  a wait for the next block's chunk loads is "until at most N younger operations are outstanding" - with stores it cannot see, N
//  comes out too small and every such wait also sat out the acknowledgement of the previous block's write-through stores: a trip
//  to memory per block that nothing was waiting for.)
__device__ __forceinline__ uint8_t *bk_uniform_ptr(uint8_t *p)      // a pointer the caller knows to be wave-uniform, as SGPRs
{
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return reinterpret_cast<uint8_t *>((uintptr_t)lo | ((uintptr_t)hi << 32));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bk_frame_rsrc(uint8_t *dst) { return __builtin_amdgcn_make_buffer_rsrc(bk_uniform_ptr(dst), 0, 0x7FFFFFFF, 0x00020000); }

// one frame of one lane: its 4*RG texels out of the staged chunks, packed and stored
template <bool RUBIX, int RG, bool WT = false>
__device__ __forceinline__ void coop_gather_store(const uint8_t *buf, const CoopIdx<RG> &ix, bool fast_store, const uint8_t *pal_s,
                                                  uint8_t *__restrict__ dst, size_t frame_stride, int dst_pitch, int f, int row0, int x, int kflags)
{
    // (RUBIX: the chunks in `buf` are tinted already - bk_tint_chunk - and the gather is the plain one)
    if (fast_store) {
        uint32_t w[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            uint32_t i0 = ix.iw[r].x, i1 = ix.iw[r].y;
            if (WT) asm volatile("" : "+v"(i0), "+v"(i1));    // (the resident kernel: the unpacked addresses are not to be kept from frame to frame)
            const uint32_t v0 = buf[i0 & 0xFFFFu], v1 = buf[i0 >> 16], v2 = buf[i1 & 0xFFFFu], v3 = buf[i1 >> 16];
            w[r] = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
            // (the resident kernel gathers with the next block's chunks in flight in 16-24 registers: four texels at a time, not all
            //  sixteen byte reads and their addresses in registers at once - the compiler would otherwise spill the block's chunk
            //  offsets, and a scratch reload in front of every globe load makes the loads wait for one another)
            if (WT) asm volatile("" ::: "memory");
        }
        if (!(kflags & 4)) {
            // non-temporal stores: the frame is never read back here, and keeping it out of L2 leaves the cache to the
            // globe lines neighbouring blocks share (4K panini 3.8 -> 3.3 us/frame)
            typedef uint32_t v2u __attribute__((ext_vector_type(2)));
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            if (WT) {
                const uint32_t off = (uint32_t)row0 * (uint32_t)dst_pitch + (uint32_t)x;       // (f = 0, one frame per command)
                const __amdgpu_buffer_rsrc_t rd = bk_frame_rsrc(dst);
                if (RG == 1) __builtin_amdgcn_raw_buffer_store_b32(w[0], rd, (int)off, 0, 16);
                else if (RG == 2) { v2u v = {w[0], w[RG - 1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rd, (int)off, 0, 16); }
                else { v4u v = {w[0], w[1 % RG], w[2 % RG], w[3 % RG]}; __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off, 0, 16); }
            } else {
                uint8_t *o = dst + (size_t)f * frame_stride + (size_t)row0 * dst_pitch + x;
                if (RG == 1) __builtin_nontemporal_store(w[0], reinterpret_cast<uint32_t *>(o));
                else if (RG == 2) { v2u v = {w[0], w[RG - 1]}; __builtin_nontemporal_store(v, reinterpret_cast<v2u *>(o)); }
                else { v4u v = {w[0], w[1 % RG], w[2 % RG], w[3 % RG]}; __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(o)); }
            }
        }
    } else if (WT) {
        // the resident kernel in a tile that is only PARTLY mapped (the rim of hammer's ellipse, the block with stereographic's one NULL
        // pixel): write-through stores narrower than 16 bytes are a fabric write each (a byte 12x, a word 6x the time
        // per byte), so a lane stores as wide as its own pixels allow - all of them mapped and the frame aligned: the one wide store
        // of the fast path; else word by word, and bytes only in the words with a hole.  Without this the ONE workgroup with such a
        // tile ran at half the others' pace, and the slowest workgroup sets the frame rate.
        uint32_t w[RG], m = 0;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            uint32_t i0 = ix.iw[r].x, i1 = ix.iw[r].y;
            asm volatile("" : "+v"(i0), "+v"(i1));            // (likewise: no unpacked copies of the addresses kept from frame to frame)
            const uint32_t a[4] = {i0 & 0xFFFFu, i0 >> 16, i1 & 0xFFFFu, i1 >> 16};
            w[r] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t v = a[k] != 0xFFFFu ? buf[a[k]] : 0u;
                w[r] |= v << (8 * k);
                m |= a[k] != 0xFFFFu ? 1u << (4 * r + k) : 0u;
                asm volatile("" : "+v"(w[r]));               // (one texel at a time: sixteen conditional byte reads are not to be in flight - in registers - at once)
            }
        }
        if (kflags & 4) return;
        typedef uint32_t v2u __attribute__((ext_vector_type(2)));
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        uint32_t off = (uint32_t)row0 * (uint32_t)dst_pitch + (uint32_t)x;
        // (this path is rare; the compiler must not keep its sixteen per-pixel offsets in registers across the frame loop for it - it did,
        //  and paid with spills of the chunk offsets on the path that matters: the value is opaque from here on)
        asm volatile("" : "+v"(off));
        const __amdgpu_buffer_rsrc_t rd = bk_frame_rsrc(dst);
        const bool wide_ok = (kflags & 2048) != 0, word_ok = (kflags & 1024) != 0;      // the frame's alignment (set per command)
        if (m == (RG == 4 ? 0xFFFFu : RG == 2 ? 0xFFu : 0xFu) && wide_ok) {
            if (RG == 1) __builtin_amdgcn_raw_buffer_store_b32(w[0], rd, (int)off, 0, 16);
            else if (RG == 2) { v2u v = {w[0], w[RG - 1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rd, (int)off, 0, 16); }
            else { v4u v = {w[0], w[1 % RG], w[2 % RG], w[3 % RG]}; __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off, 0, 16); }
        } else {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const uint32_t mr = (m >> (4 * r)) & 0xFu;
                if (mr == 0xFu && word_ok) {
                    __builtin_amdgcn_raw_buffer_store_b32(w[r], rd, (int)(off + 4 * r), 0, 16);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((mr >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(w[r] >> (8 * k)), rd, (int)(off + 4 * r + k), 0, 16);
                }
            }
        }
    } else {
        uint32_t w[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const uint32_t a[4] = {ix.iw[r].x & 0xFFFFu, ix.iw[r].x >> 16, ix.iw[r].y & 0xFFFFu, ix.iw[r].y >> 16};
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = a[k] != 0xFFFFu ? buf[a[k]] : 0u;
            }
            w[r] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            if (!fast_store) {
                uint8_t *out = dst + (size_t)f * frame_stride + (size_t)row0 * dst_pitch + x + 4 * r;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (a[k] != 0xFFFFu) out[k] = (uint8_t)v[k];
            }
        }
    }
    (void)pal_s;
}

// Frames of one staged block.  A thread's first four chunks (16 KiB per block) are in registers;
// blocks with more read the rest of their list each frame (it stays in L2).
template <int NQ, bool RUBIX, int RG, bool DMA>
__device__ __forceinline__ void coop_frames(const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames, int frame0,
                                            int f_begin, int f_end, uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride,
                                            uint8_t *buf, const uint32_t *__restrict__ blist,
                                            uint32_t nchunks, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4, uint32_t s5,
                                            bool k0, bool k1, bool k2, bool k3, bool k4, bool k5, const CoopIdx<RG> ix, bool fast_store,
                                            bool tile_empty, const uint8_t *pal_s, int row0, int x, int kflags)
{
    // NQ = 5, 6 (strided walk only): a thread's fifth and sixth chunk ride in registers too - blocks of up to 24 KiB (the whole-globe
    // lenses have them: 9 % of 4K hammer's blocks are above 16 KiB) stay inside the frame pipeline instead of fetching the rest of
    // their list and chunks synchronously every frame, two dependent trips to memory that doubled those blocks' frames
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;     // (scalars, not an array: they must stay in VGPRs)
    constexpr uint32_t REG_CHUNKS = NQ > 4 ? 1536u : 1024u;           // chunks the register plan covers before the extra rounds
    const bool pipe = (kflags & 8) == 0;      // issue frame f+1's loads before frame f's gather (ablation bit 8 turns it off)
    typedef uint32_t bk_v4u __attribute__((ext_vector_type(4)));
    // kflags bit 128: the globe chunks are fetched with the non-temporal hint (streamed through L2, evicted first), so that
    // what L2 keeps from one launch to the next is the block map - headers, chunk lists, pixel addresses: the same bytes every
    // launch, and with the XCD bands fixed, the same XCD's L2 - instead of globe lines nobody will ask for again
    const bool nt_globe = (kflags & 128) != 0;
#define BK_COOP_LD(Q, PTR)                                                                                 \
    do {                                                                                                   \
        if (nt_globe) {                                                                                    \
            const bk_v4u t_ = __builtin_nontemporal_load(reinterpret_cast<const bk_v4u *>(PTR));           \
            Q = make_uint4(t_.x, t_.y, t_.z, t_.w);                                                        \
        } else {                                                                                           \
            Q = *reinterpret_cast<const uint4 *>(PTR);                                                     \
        }                                                                                                  \
    } while (0)
#define BK_COOP_LOADS(F)                                                                                   \
    do {                                                                                                   \
        const uint8_t *gl_ = globe + (size_t)((frame0 + (F)) % globe_frames) * globe_stride;              \
        BK_COOP_LD(q0, gl_ + BK_CHUNK_OFF(s0));                                                            \
        if (NQ > 1) BK_COOP_LD(q1, gl_ + BK_CHUNK_OFF(s1));                                                \
        if (NQ > 2) BK_COOP_LD(q2, gl_ + BK_CHUNK_OFF(s2));                                                \
        if (NQ > 3) BK_COOP_LD(q3, gl_ + BK_CHUNK_OFF(s3));                                                \
        if (NQ > 4) BK_COOP_LD(q4, gl_ + BK_CHUNK_OFF(s4));                                                \
        if (NQ > 5) BK_COOP_LD(q5, gl_ + BK_CHUNK_OFF(s5));                                                \
    } while (0)
    // DMA form (single-frame launches): the chunks go HBM -> LDS directly (global_load_lds_dwordx4: a wave-uniform LDS base
    // + lane * 16 - exactly "list entry i -> slot i"), no staging registers, no ds_write pass; the barrier's fence waits for them
    typedef __attribute__((address_space(3))) void *bk_lds_ptr;
    typedef const __attribute__((address_space(1))) void *bk_glb_ptr;
#define BK_COOP_DMA(SRC, SLOT0) __builtin_amdgcn_global_load_lds((bk_glb_ptr)(SRC), (bk_lds_ptr)(buf + ((SLOT0) + (threadIdx.x & ~63u)) * 16u), 16, 0, 0)
    if (!DMA && pipe && f_begin < f_end) BK_COOP_LOADS(f_begin);
    for (int f = f_begin; f < f_end; ++f) {
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        uint8_t *mine = buf + threadIdx.x * 16u;
        if (DMA) {
            if (k0) BK_COOP_DMA(gl + s0, 0u);
            if (NQ > 1 && k1) BK_COOP_DMA(gl + s1, 256u);
            if (NQ > 2 && k2) BK_COOP_DMA(gl + s2, 512u);
            if (NQ > 3 && k3) BK_COOP_DMA(gl + s3, 768u);
            if (NQ > 3) {
                for (uint32_t c0 = 1024; c0 < nchunks; c0 += 256) {
                    const uint32_t c = c0 + threadIdx.x;
                    if (c < nchunks) BK_COOP_DMA(gl + blist[c], c0);
                }
            }
        } else {
        if (!pipe && !(kflags & 2)) BK_COOP_LOADS(f);
        if (k0) *reinterpret_cast<uint4 *>(mine) = BK_CHUNK_VAL(q0, s0);
        if (NQ > 1 && k1) *reinterpret_cast<uint4 *>(mine + 4096) = BK_CHUNK_VAL(q1, s1);
        if (NQ > 2 && k2) *reinterpret_cast<uint4 *>(mine + 8192) = BK_CHUNK_VAL(q2, s2);
        if (NQ > 3 && k3) *reinterpret_cast<uint4 *>(mine + 12288) = BK_CHUNK_VAL(q3, s3);
        if (NQ > 4 && k4) *reinterpret_cast<uint4 *>(mine + 16384) = BK_CHUNK_VAL(q4, s4);
        if (NQ > 5 && k5) *reinterpret_cast<uint4 *>(mine + 20480) = BK_CHUNK_VAL(q5, s5);
        }
        if (!DMA && (NQ == 4 || NQ == 6)) {
            for (uint32_t c0 = REG_CHUNKS; c0 < nchunks; c0 += 1024) {      // blocks above what the registers hold: rounds of four loads
                const uint32_t c = c0 + threadIdx.x;
                const bool m0 = c < nchunks, m1 = c + 256u < nchunks, m2 = c + 512u < nchunks, m3 = c + 768u < nchunks;
                const uint32_t a0 = m0 ? blist[c] : 0u, a1 = m1 ? blist[c + 256u] : 0u, a2 = m2 ? blist[c + 512u] : 0u,
                               a3 = m3 ? blist[c + 768u] : 0u;
                BK_COOP_LD(q0, gl + BK_CHUNK_OFF(a0));
                BK_COOP_LD(q1, gl + BK_CHUNK_OFF(a1));
                BK_COOP_LD(q2, gl + BK_CHUNK_OFF(a2));
                BK_COOP_LD(q3, gl + BK_CHUNK_OFF(a3));
                uint8_t *md = mine + (size_t)c0 * 16u;
                if (m0) *reinterpret_cast<uint4 *>(md) = BK_CHUNK_VAL(q0, a0);
                if (m1) *reinterpret_cast<uint4 *>(md + 4096) = BK_CHUNK_VAL(q1, a1);
                if (m2) *reinterpret_cast<uint4 *>(md + 8192) = BK_CHUNK_VAL(q2, a2);
                if (m3) *reinterpret_cast<uint4 *>(md + 12288) = BK_CHUNK_VAL(q3, a3);
            }
        }
        if (DMA) __syncthreads();             // (the LDS-DMA form needs the fence's vmcnt(0): its loads ARE the LDS writes)
        else BK_LDS_BARRIER();                // the block's chunks are in `buf`
        if (!DMA && pipe && f + 1 < f_end) BK_COOP_LOADS(f + 1);
        if (tile_empty) { BK_LDS_BARRIER(); continue; }
        coop_gather_store<RUBIX, RG>(buf, ix, fast_store, pal_s, dst, frame_stride, dst_pitch, f, row0, x, kflags);
        BK_LDS_BARRIER();                     // every wave is done with `buf`
    }
}

#undef BK_COOP_LOADS
#undef BK_COOP_LD
#undef BK_COOP_DMA

// Frames of a block whose chunk list is larger than the launch's staging buffer: the list goes through
// LDS in passes of lds_buf/16 chunks; a pixel picks its texel up in the pass that holds its slot and the packed
// words are stored after the last pass.
template <bool RUBIX, int RG>
__device__ __forceinline__ void coop_frames_multipass(const uint8_t *__restrict__ globe, size_t globe_stride, int globe_frames, int frame0,
                                                   int f_begin, int f_end, uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride,
                                                   uint8_t *buf, uint32_t lds_buf, const uint32_t *__restrict__ blist,
                                                   uint32_t nchunks, const CoopIdx<RG> ix, bool fast_store, bool tile_empty,
                                                   const uint8_t *pal_s, int row0, int x)
{
    constexpr int kflags = 0;                                // (BK_CHUNK_VAL's developer bit: not here)
    const uint32_t cpb = lds_buf >> 4;                       // chunks per pass
    for (int f = f_begin; f < f_end; ++f) {
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
        uint32_t w[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) w[r] = 0;
        for (uint32_t base = 0; base < nchunks; base += cpb) {
            const uint32_t end = min(nchunks, base + cpb);
            for (uint32_t c0 = base; c0 < end; c0 += 1024) {
                const uint32_t c = c0 + threadIdx.x;
                const bool m0 = c < end, m1 = c + 256u < end, m2 = c + 512u < end, m3 = c + 768u < end;
                const uint32_t a0 = m0 ? blist[c] : 0u, a1 = m1 ? blist[c + 256u] : 0u, a2 = m2 ? blist[c + 512u] : 0u,
                               a3 = m3 ? blist[c + 768u] : 0u;
                const uint4 q0 = *reinterpret_cast<const uint4 *>(gl + BK_CHUNK_OFF(a0)), q1 = *reinterpret_cast<const uint4 *>(gl + BK_CHUNK_OFF(a1)),
                            q2 = *reinterpret_cast<const uint4 *>(gl + BK_CHUNK_OFF(a2)), q3 = *reinterpret_cast<const uint4 *>(gl + BK_CHUNK_OFF(a3));
                uint8_t *md = buf + (size_t)(c - base) * 16u;
                if (m0) *reinterpret_cast<uint4 *>(md) = BK_CHUNK_VAL(q0, a0);
                if (m1) *reinterpret_cast<uint4 *>(md + 4096) = BK_CHUNK_VAL(q1, a1);
                if (m2) *reinterpret_cast<uint4 *>(md + 8192) = BK_CHUNK_VAL(q2, a2);
                if (m3) *reinterpret_cast<uint4 *>(md + 12288) = BK_CHUNK_VAL(q3, a3);
            }
            __syncthreads();
            if (!tile_empty) {
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const uint32_t a[4] = {ix.iw[r].x & 0xFFFFu, ix.iw[r].x >> 16, ix.iw[r].y & 0xFFFFu, ix.iw[r].y >> 16};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t rel = a[k] - base * 16u;          // (0xFFFF = unmapped never lands inside: slots <= 4094)
                        if (a[k] != 0xFFFFu && rel < (end - base) * 16u) w[r] |= (uint32_t)buf[rel] << (8 * k);
                    }
                }
            }
            __syncthreads();
        }
        if (tile_empty) continue;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const uint32_t a[4] = {ix.iw[r].x & 0xFFFFu, ix.iw[r].x >> 16, ix.iw[r].y & 0xFFFFu, ix.iw[r].y >> 16};
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = (w[r] >> (8 * k)) & 0xFFu;
            }
            if (!fast_store) {
                uint8_t *out = dst + (size_t)f * frame_stride + (size_t)row0 * dst_pitch + x + 4 * r;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (a[k] != 0xFFFFu) out[k] = (uint8_t)v[k];
            }
        }
        if (fast_store) {                               // (one store of the lane's 4 * RG pixels, as everywhere else)
            typedef uint32_t v2u __attribute__((ext_vector_type(2)));
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            uint8_t *o = dst + (size_t)f * frame_stride + (size_t)row0 * dst_pitch + x;
            if (RG == 1) __builtin_nontemporal_store(w[0], reinterpret_cast<uint32_t *>(o));
            else if (RG == 2) { v2u v = {w[0], w[RG - 1]}; __builtin_nontemporal_store(v, reinterpret_cast<v2u *>(o)); }
            else { v4u v = {w[0], w[1 % RG], w[2 % RG], w[3 % RG]}; __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(o)); }
        }
    }
}

// direct-gather frames of a block that has no chunk list (more than 4095 unique chunks)
template <bool RUBIX, int RG>
__device__ __noinline__ void coop_slow_frames(const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ globe, size_t globe_stride,
                                              int globe_frames, int frame0, int f_begin, int f_end, uint8_t *__restrict__ dst,
                                              int dst_pitch, size_t frame_stride, int W, int rows, const uint8_t *__restrict__ tints,
                                              const uint8_t *pal_s, int row0, int x)
{
    // (a block without a chunk list: its pixels' tints straight from the lensmap's tint plane, [rows][W])
    uint32_t so[RG][4], st[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = row0, xx = x + 4 * r + k;
            so[r][k] = (row < rows && xx < W) ? lmap[(size_t)row * W + xx] : BK_NULL_OFFSET;
            if (k == 0) st[r] = 0xFFFFFFFFu;
            if (RUBIX && row < rows && xx < W) st[r] = (st[r] & ~(0xFFu << (8 * k))) | ((uint32_t)tints[(size_t)row * W + xx] << (8 * k));
        }
    for (int f = f_begin; f < f_end; ++f) {
        const uint8_t *gl = globe + (size_t)((frame0 + f) % globe_frames) * globe_stride;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            uint8_t *out = dst + (size_t)f * frame_stride + (size_t)row0 * dst_pitch + x + 4 * r;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (so[r][k] == BK_NULL_OFFSET) continue;
                uint32_t v = gl[so[r][k]];
                if (RUBIX) {
                    const uint32_t tt = (st[r] >> (8 * k)) & 0xFFu;
                    if (tt < (uint32_t)BK_MAX_PLATES) v = pal_s[tt * 256 + v];
                }
                out[k] = (uint8_t)v;
            }
        }
    }
}

__device__ __forceinline__ int lane_of() { return (int)(threadIdx.x & 63u); }

// everything a workgroup does for one block: `cur` holds the block's header and list head
template <bool RUBIX, int RG, bool DMA = false, int MAXQ = 4>
__device__ __forceinline__ void coop_block(const CoopPrefetch<RG> &cur, int l, const uint32_t *__restrict__ list,
                                           const uint16_t *__restrict__ idx, const uint8_t *__restrict__ tint_t,
                                           const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ globe, size_t globe_stride,
                                           int globe_frames, int frame0, int f_begin, int f_end, uint8_t *__restrict__ dst, int dst_pitch,
                                           size_t frame_stride, int W, int rows, int blocks_x, uint8_t *smem, int lds_buf,
                                           const uint8_t *pal_s, bool aligned, int ry, int cx, int wave, int kflags)
{
    constexpr int N = 1024 * RG;
    const uint32_t nchunks = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.h.x);      // wave-uniform values -> SGPRs
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.h.y);
    if (flags & CF_EMPTY) return;
    const int by = l / blocks_x, bx = l - by * blocks_x;
    const int row0 = by * 8 * RG + ry, x = bx * 128 + cx * 4 * RG;
    const bool tile_all = (flags >> wave) & 1u, tile_empty = (flags >> (4 + wave)) & 1u;
    // the pixel addresses: issued here, consumed at the first gather
    const CoopIdx<RG> ix = coop_load_idx<RUBIX, RG>(idx, tint_t, l, wave, lane_of());
    if (flags & CF_SLOW) {
        if (!tile_empty)
            coop_slow_frames<RUBIX, RG>(lmap, globe, globe_stride, globe_frames, frame0, f_begin, f_end, dst, dst_pitch,
                                        frame_stride, W, rows, tint_t, pal_s, row0, x);
    } else if ((int)(nchunks * 16u) > lds_buf) {
        // a chunk list larger than this launch's staging buffer goes through it in passes
        coop_frames_multipass<RUBIX, RG>(globe, globe_stride, globe_frames, frame0, f_begin, f_end, dst, dst_pitch, frame_stride,
                                         smem, (uint32_t)lds_buf, list + (size_t)l * N, nchunks, ix, tile_all && aligned,
                                         tile_empty, pal_s, row0, x);
    } else {
        const bool k0 = threadIdx.x < nchunks, k1 = threadIdx.x + 256u < nchunks, k2 = threadIdx.x + 512u < nchunks,
                   k3 = threadIdx.x + 768u < nchunks;
        const uint32_t s0 = k0 ? cur.c[0] : 0u, s1 = k1 ? cur.c[1] : 0u, s2 = k2 ? cur.c[2] : 0u, s3 = k3 ? cur.c[3] : 0u;
        // (MAXQ = 6, the strided walk: list entries five and six of the thread, once per block visit)
        const bool k4 = MAXQ > 4 && threadIdx.x + 1024u < nchunks, k5 = MAXQ > 4 && threadIdx.x + 1280u < nchunks;
        const uint32_t *bl = list + (size_t)l * N;
        const uint32_t s4 = k4 ? bl[threadIdx.x + 1024u] : 0u, s5 = k5 ? bl[threadIdx.x + 1280u] : 0u;
        const bool fast_store = tile_all && aligned;
        const uint32_t nq = (nchunks + 255u) >> 8;
#define BK_COOP(NQ_) coop_frames<NQ_, RUBIX, RG, DMA>(globe, globe_stride, globe_frames, frame0, f_begin, f_end, dst, dst_pitch, frame_stride, smem, \
                                                bl, nchunks, s0, s1, s2, s3, s4, s5, k0, k1, k2, k3, k4, k5,                       \
                                                ix, fast_store, tile_empty, pal_s, row0, x, kflags)
        if (nq <= 1) BK_COOP(1);
        else if (nq == 2) BK_COOP(2);
        else if (nq == 3) BK_COOP(3);
        else if (MAXQ <= 4 || nq == 4) BK_COOP(4);
        else if (nq == 5) BK_COOP(5);
        else BK_COOP(6);
#undef BK_COOP
    }
}

#define BK_COOP_KERNEL_ARGS                                                                                                        \
    const CoopHdr *__restrict__ hdr, const uint32_t *__restrict__ list, const uint16_t *__restrict__ idx,                          \
    const uint8_t *__restrict__ tint_t, const uint32_t *__restrict__ lmap, const uint8_t *__restrict__ globe, size_t globe_stride, \
    int globe_frames, int frame0, uint8_t *__restrict__ dst, int dst_pitch, size_t frame_stride, int W, int rows, int blocks_x,    \
    int nblocks, int nframes, int fchunk, int lds_buf, const uint8_t *__restrict__ pal, int kflags,                                \
    const uint32_t *__restrict__ order, const uint32_t *__restrict__ bands, const uint32_t *__restrict__ wgmap

// XCD-banded mapping: workgroup b runs on XCD b % 8 (observed dispatch order); XCD k owns the contiguous band
// [k*per, (k+1)*per) of blocks.  Correctness does not depend on it.
#define BK_COOP_PROLOGUE                                                                                                           \
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                                                               \
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                                                                   \
    uint8_t *pal_s = smem + lds_buf;                                                                                               \
    /* rubix: the palette's 384 dwords are REQUESTED here and put into LDS by BK_COOP_PAL_COMMIT, behind the block's header and    \
       list loads - a copy + barrier up here was a trip to memory of its own in front of everything a single-frame launch does */  \
    uint32_t pal_r0 = 0, pal_r1 = 0;                                                                                               \
    if (RUBIX) {                                                                                                                   \
        pal_r0 = reinterpret_cast<const uint32_t *>(pal)[threadIdx.x];                                                             \
        if (threadIdx.x < BK_MAX_PLATES * 64 - 256) pal_r1 = reinterpret_cast<const uint32_t *>(pal)[256 + threadIdx.x];           \
    }                                                                                                                              \
    const int per = (nblocks + 7) / 8;                                                                                             \
    const int band = (int)(blockIdx.x & 7);                                                                                        \
    const int wg_in_band = (int)(blockIdx.x >> 3), wgs_per_band = (int)(gridDim.x >> 3);                                          \
    const int l_end = min(nblocks, (band + 1) * per);                                                                              \
    int l = band * per + wg_in_band;                                                                                               \
    const int f_begin = blockIdx.y * fchunk, f_end = min(nframes, f_begin + fchunk);                                              \
    const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dst_pitch | (uintptr_t)frame_stride) & (uintptr_t)(4 * RG - 1)) == 0; \
    constexpr int LPR = 32 / RG;                  /* same pixel -> lane mapping as coop_compile_kernel */                        \
    const int ry = wave * 2 * RG + lane / LPR, cx = lane % LPR

#define BK_COOP_PAL_COMMIT                                                                                                         \
    if (RUBIX) {                                                                                                                   \
        reinterpret_cast<uint32_t *>(pal_s)[threadIdx.x] = pal_r0;                                                                 \
        if (threadIdx.x < BK_MAX_PLATES * 64 - 256) reinterpret_cast<uint32_t *>(pal_s)[256 + threadIdx.x] = pal_r1;               \
        __syncthreads();                                                                                                           \
    }

// persistent form: a workgroup walks a strided list of blocks and prefetches the next one.  By default the list is the
// XCD's band of the LIVE blocks in walk order, the bands cut at equal cost (CoopMap::d_order / d_bands; ablation bit 64:
// bands of equal block count over all blocks, empty ones included, as in rounds 1 and 2a).
template <bool RUBIX, int RG, int MAXQ = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void apply_coop_kernel(BK_COOP_KERNEL_ARGS)
{
    BK_COOP_PROLOGUE;
    (void)wgmap;
    const bool balanced = (kflags & (16 | 64)) == 0;
    int l_hi = l_end;
    if (balanced) {
        l = (int)bands[band] + wg_in_band;
        l_hi = (int)bands[band + 1];
    }
    if (l >= l_hi) return;
#define BK_BLOCK_OF(POS) (balanced ? (int)order[POS] : bk_block_at(POS, blocks_x, nblocks, kflags))
    int b_cur = BK_BLOCK_OF(l);
    int l_next = l + wgs_per_band;
    bool has_next = l_next < l_hi;
    int b_next = has_next ? BK_BLOCK_OF(l_next) : 0;
    CoopPrefetch<RG> cur = coop_fetch<RUBIX, RG>(hdr, list, b_cur);
    BK_COOP_PAL_COMMIT;
    for (;;) {
        CoopPrefetch<RG> nxt = cur;
        if (has_next) nxt = coop_fetch<RUBIX, RG>(hdr, list, b_next);
        const int l_nn = l_next + wgs_per_band;                    // two ahead: its block number is here before it is needed
        const bool has_nn = has_next && l_nn < l_hi;
        const int b_nn = has_nn ? BK_BLOCK_OF(l_nn) : 0;
        coop_block<RUBIX, RG, false, MAXQ>(cur, b_cur, list, idx, tint_t, lmap, globe, globe_stride, globe_frames, frame0, f_begin, f_end,
                                        dst, dst_pitch, frame_stride, W, rows, blocks_x, smem, lds_buf, pal_s, aligned, ry, cx, wave, kflags);
        if (!has_next) break;
        l_next = l_nn;
        b_cur = b_next;
        b_next = b_nn;
        has_next = has_nn;
        cur = nxt;
    }
#undef BK_BLOCK_OF
}


// one block per workgroup (launches short enough that the whole grid is resident at once, e.g. the engine's
// single frames): no next-block state, fewer registers, more workgroups per CU - every block starts at once
// (8 waves per SIMD = 8 workgroups per CU: the 128x32 form would take 67 VGPRs and 7; at 4K that is 1792 places for 2040
// blocks, and the 248 left over wait a whole block's latency for theirs - single frame 9.4 -> 8.4 us at <= 64 VGPRs)
template <bool RUBIX, int RG, bool DMA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void apply_coop_once_kernel(BK_COOP_KERNEL_ARGS)
{
    BK_COOP_PROLOGUE;
    (void)wgs_per_band; (void)order; (void)bands;
    int blk;
    if (kflags & BK_KF_WGMAP) {                       // bands of equal cost (only where equal counts would be uneven: one more
        const uint32_t m = wgmap[blockIdx.x];         // dependent load in front of the header)
        if (m == 0xFFFFFFFFu) return;
        blk = (int)m;
    } else {
        if (l >= l_end) return;
        blk = bk_block_at(l, blocks_x, nblocks, kflags);
    }
    const CoopPrefetch<RG> cur = coop_fetch<RUBIX, RG>(hdr, list, blk);
    BK_COOP_PAL_COMMIT;
    coop_block<RUBIX, RG, DMA>(cur, blk, list, idx, tint_t, lmap, globe, globe_stride, globe_frames, frame0, f_begin, f_end,
                               dst, dst_pitch, frame_stride, W, rows, blocks_x, smem, lds_buf, pal_s, aligned, ry, cx, wave, kflags);
}

// The live blocks in walk order, the eight band starts of equal cost, the workgroup -> block map of the one-block-per-workgroup
// grid (8 * per workgroups, per = ceil(nblocks / 8): the same bands, none longer than `per` blocks),
// and whether bands of equal block COUNT would be more than 10 % uneven (-> stats[7] of replica 0).  One workgroup.
__global__ __launch_bounds__(1024) void coop_order_kernel(const uint32_t *__restrict__ cost, int nblocks, int blocks_x, int per,
                                                          uint32_t *__restrict__ order, uint32_t *__restrict__ cum,
                                                          uint32_t *__restrict__ bands, uint32_t *__restrict__ wgmap,
                                                          uint32_t *__restrict__ stats)
{
    __shared__ uint32_t s_wn[16], s_wc[16], s_base_n, s_base_c, s_ucost[8], s_start[9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_base_n = 0; s_base_c = 0; }
    if (threadIdx.x < 8) s_ucost[threadIdx.x] = 0;
    __syncthreads();
    for (int l0 = 0; l0 < nblocks; l0 += 1024) {
        const int l = l0 + (int)threadIdx.x;
        uint32_t blk = 0, c = 0;
        if (l < nblocks) { blk = (uint32_t)bk_block_at(l, blocks_x, nblocks, 0); c = cost[blk]; }
        const uint32_t n = c ? 1u : 0u;
        if (c) atomicAdd(&s_ucost[min(7, l / per)], c);
        uint32_t in = n, ic = c;
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t u = __shfl_up(in, m), v = __shfl_up(ic, m);
            if (lane >= m) { in += u; ic += v; }
        }
        if (lane == 63) { s_wn[wave] = in; s_wc[wave] = ic; }
        __syncthreads();
        uint32_t bn = s_base_n, bc = s_base_c;
        for (int w = 0; w < wave; ++w) { bn += s_wn[w]; bc += s_wc[w]; }
        if (n) { order[bn + in - 1] = blk; cum[bn + in - 1] = bc + ic; }
        __syncthreads();
        if (threadIdx.x == 1023) { s_base_n = bn + in; s_base_c = bc + ic; }
        __syncthreads();
    }
    const uint32_t nlive = s_base_n, total = s_base_c;
    if (threadIdx.x <= 8) {
        const uint32_t k = threadIdx.x;
        uint32_t start = k == 8 ? nlive : 0u;
        if (k > 0 && k < 8) {                                   // first live block whose cost prefix passes k/8 of the total
            const uint32_t target = (uint32_t)((uint64_t)total * k / 8);
            uint32_t lo = 0, hi = nlive;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cum[mid] <= target) lo = mid + 1; else hi = mid;
            }
            start = lo;
        }
        s_start[k] = start;
    }
    __syncthreads();
    if (threadIdx.x <= 8) bands[threadIdx.x] = s_start[threadIdx.x];      // the strided walk: any band length will do
    __syncthreads();
    if (threadIdx.x == 0) {
        // the one-block grid has `per` workgroups per XCD: there no band may be longer than that, and the rest must still fit
        for (int k = 1; k < 8; ++k) {
            const int64_t need = (int64_t)nlive - (int64_t)(8 - k) * per;
            int64_t st = s_start[k];
            if (st < need) st = need;
            if (st < (int64_t)s_start[k - 1]) st = s_start[k - 1];
            if (st > (int64_t)s_start[k - 1] + per) st = (int64_t)s_start[k - 1] + per;
            s_start[k] = (uint32_t)st;
        }
        uint32_t umax = 0;
        for (int k = 0; k < 8; ++k) umax = max(umax, s_ucost[k]);
        stats[7] = (uint64_t)umax * 80u > (uint64_t)total * 11u ? 1u : 0u;
    }
    __syncthreads();
    for (int b = (int)threadIdx.x; b < 8 * per; b += 1024) {
        const uint32_t l = s_start[b & 7] + (uint32_t)(b >> 3);
        wgmap[b] = l < s_start[(b & 7) + 1] ? order[l] : 0xFFFFFFFFu;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void coopmap_free(CoopMap *cm)
{
    if (!cm) return;
    (void)hipFree(cm->d_hdr);
    (void)hipFree(cm->d_list);
    (void)hipFree(cm->d_idx);
    (void)hipFree(cm->d_cost); (void)hipFree(cm->d_order); (void)hipFree(cm->d_cum); (void)hipFree(cm->d_bands); (void)hipFree(cm->d_wgmap);
    (void)hipFree(cm->d_stats);
    if (cm->h_stats) (void)hipHostFree(cm->h_stats);
    if (cm->stats_ready) (void)hipEventDestroy(cm->stats_ready);
    delete cm;
}

void coopmap_invalidate(bk_ctx *ctx)
{
    resident_quiesce(ctx);          // (a resident apply kernel holds this block map in its registers)
    for (CoopMap *cm : {ctx->coopmap, ctx->coopmap_alt})            // (both flavours: the parked one describes the same lensmap)
        if (cm) {
            cm->valid = false;
            for (auto &t : cm->tuned) t = CoopMap::Tuned();
            cm->flips = 0;
            cm->grown_rg = 0;
        }
}

static void fold_stats(const uint32_t *rep, uint32_t *out, uint32_t scale)
{
    for (int k = 0; k < BK_COOP_STATS; ++k) out[k] = 0;
    for (int r = 0; r < 64; ++r) {
        out[0] = rep[r * BK_COOP_STATS] > out[0] ? rep[r * BK_COOP_STATS] : out[0];
        for (int k = 1; k < BK_COOP_STATS; ++k) out[k] += rep[r * BK_COOP_STATS + k] * scale;
    }
}

// one pass of coop_compile_kernel over the owned rows with block height 8*rg; row_stride > 1 = survey pass into statistics
// set `set` of d_stats (nothing else written)
static int coop_compile_launch(bk_ctx *ctx, CoopMap *cm, int rg, int row_stride, int set)
{
    const int rows = ctx->rows();
    const int bx = (ctx->W + 127) / 128, by = (rows + 8 * rg - 1) / (8 * rg);
    const int sampled_rows = (by + row_stride - 1) / row_stride;
    uint32_t *st = cm->d_stats + (size_t)set * 64 * BK_COOP_STATS;
    BK_HIP(ctx, hipMemsetAsync(st, 0, 64 * BK_COOP_STATS * sizeof(uint32_t), ctx->stream));
    const dim3 grid((unsigned)(bx * sampled_rows)), block(256);
#define BK_COMPILE(N) hipLaunchKernelGGL((coop_compile_kernel<N>), grid, block, 0, ctx->stream, ctx->d_offsets, ctx->d_tints, ctx->W, rows, \
                                         bx, bx * by, cm->d_hdr, cm->d_list, cm->d_idx, cm->tinted ? 1 : 0, st, row_stride, cm->d_cost, ctx->apply_block_cost >= 0 ? ctx->apply_block_cost : BK_COOP_BLOCK_COST)
    if (rg == 1) BK_COMPILE(1); else if (rg == 2) BK_COMPILE(2); else BK_COMPILE(4);
#undef BK_COMPILE
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

static int coop_choose_buffer(const CoopMap *cm, int rg, double npixels, int num_cus, double *cost_ns);
static int ensure_coopmap(bk_ctx *ctx, int launch_frames = 0, int want_tinted = -1);

// the statistics of the last full compile, once somebody needs them (the apply launch itself does not)
static int coop_stats_wait(bk_ctx *ctx, CoopMap *cm)
{
    if (!cm->stats_pending) return BK_OK;
    BK_HIP(ctx, hipEventSynchronize(cm->stats_ready));
    fold_stats(cm->h_stats, cm->stats, 1);
    // the buffer size was chosen on a survey (every 2nd / 8th row of blocks); now that the exact histogram is here, let the
    // model look again - the buffer is a launch parameter, nothing in the block map depends on it
    if (ctx->apply_lds_kb <= 0 && !cm->lds_fixed && (int)(cm->stats[0] * 16u) > cm->lds_bytes) {
        double c = 0;
        const int kb = coop_choose_buffer(cm, cm->rg, (double)ctx->W * ctx->rows(), ctx->num_cus, &c);
        if (kb * 1024 > cm->lds_bytes) cm->lds_bytes = kb * 1024;
    }
    uint64_t over = 0;                         // blocks that take more than one pass through the buffer
    for (int b = cm->lds_bytes / 1024 + 1; b < (int)BK_COOP_BINS; ++b) over += cm->stats[8 + b];
    cm->slow_blocks = (int)(cm->stats[1] + over);
    cm->stats_pending = false;
    return BK_OK;
}

// Cost model (ns per frame), fitted to MI355X measurements of five lenses x three block heights x five
// buffer sizes.  Throughput side: a staged 128-byte line ~13 ps, a staged block ~0.08 ns, a pixel ~0.5 ps.
// Latency side: a workgroup spends ~0.9 us per chunk-per-thread and frame on a block (load -> LDS -> barrier ->
// gather -> store) plus ~0.1 us per row group, and a CU overlaps only as many blocks as it holds workgroups
// (one staging buffer each; registers allow 8 in the one-block form) - and
// no more than the launch has: a small map does not fill the chip.  The two sides
// combine as a 3-norm; a block larger than the buffer takes ceil(need/buffer) passes; a block on the direct-gather
// path (no chunk list at all) adds ~16 ns per row group.  Returns the best buffer size in KiB.
static int coop_choose_buffer(const CoopMap *cm, int rg, double npixels, int num_cus, double *cost_ns)
{
    const int vg = 8;                     // workgroups per CU the registers allow (one-block form: <= 64 VGPRs)
    int best_bin = 1;
    double best_c = -1;
    // (from the largest buffer down, strictly better wins: among buffers of equal cost - all those that hold every block
    // and leave the CU its register-limited number of workgroups - the LARGEST is taken.  The statistics may come from a
    // survey of every 2nd or 8th block row, which underestimates the largest block; a block that does not fit is a
    // multi-pass straggler, and in a launch of a round or two of workgroups one straggler sets the end: 1080p
    // stereographic x16 with 10 such blocks of 510 ran 2.24 us/frame, without them 1.15)
    double live_all = 0;
    for (int b = 0; b < (int)BK_COOP_BINS; ++b) live_all += cm->stats[8 + b];
    for (int bin = BK_COOP_LDS_CAP / 1024; bin >= 1; --bin) {
        // blocks that need more than the buffer go through it in ceil(need / buffer) passes (slower per chunk:
        // no register plan, the list is re-read every frame)
        double lines_fit = 0, blocks_fit = 0, chunks_fit = 0, over_passes = 0, max_passes = 1, live = 0;
        for (int b = 0; b < (int)BK_COOP_BINS; ++b) {
            const double passes = b <= bin ? 1.0 : (double)((b + bin - 1) / bin);
            if (b > bin) over_passes += passes * cm->stats[8 + b];
            if (cm->stats[8 + b]) { live += cm->stats[8 + b]; if (passes > max_passes) max_passes = passes; }
            lines_fit += cm->stats[8 + BK_COOP_BINS + b];
            blocks_fit += passes * cm->stats[8 + b];
            chunks_fit += (b <= bin ? 1.0 : 1.5) * cm->stats[8 + 2 * BK_COOP_BINS + b];
        }
        int wgs = (160 * 1024) / (bin * 1024 + BK_PAL_BYTES);
        if (wgs > vg) wgs = vg;
        if (wgs < 1) wgs = 1;
        const double t_thr = 0.013 * lines_fit + 0.08 * blocks_fit + 0.00048 * npixels;
        // (... of which a batch launch - two frame groups - only fills 2 x live blocks: a small map does not fill the chip)
        const double slots = std::min((double)num_cus * wgs, std::max(1.0, 2.0 * live_all));
        const double t_lat = (900.0 * chunks_fit / 256.0 + 100.0 * rg * blocks_fit) / slots;
        // multi-pass blocks are long-running stragglers: ~12 ns per pass and row group in the throughput, and the slowest of
        // them sets the end of a launch that is only a few rounds of workgroups long (~1.2 us per extra pass, per round)
        const double rounds = 2.0 * live / ((double)num_cus * wgs);
        const double straggler = (max_passes - 1.0) * 1200.0 / (rounds > 1.0 ? rounds : 1.0);
        const double c = cbrt(t_thr * t_thr * t_thr + t_lat * t_lat * t_lat) + 12.0 * rg * over_passes + 16.0 * rg * (double)cm->stats[1] + straggler;
        if (best_c < 0 || c < best_c) { best_c = c; best_bin = bin; }
    }
    *cost_ns = best_c;
    return best_bin;
}

static int launch_compiled(bk_ctx *ctx, CoopMap *cm, int frame0, int nframes, uint8_t *dst, int dst_pitch, size_t frame_stride, int rubix_on);

// want_tinted: 1 / 0 = the launch that follows is a rubix / a plain one (CoopMap::tinted), -1 = whichever map there is
static int ensure_coopmap(bk_ctx *ctx, int launch_frames, int want_tinted)
{
    if (!ctx->coopmap) ctx->coopmap = new CoopMap();
    CoopMap *cm = ctx->coopmap;
    bool flavour_switch = false;
    if (want_tinted >= 0 && (want_tinted != 0) != cm->tinted) {
        // The other flavour of the map (f_rubix was switched; a tinted map lists some chunks twice).  (r6) BOTH flavours are kept: the
        // map this launch does not want is parked in ctx->coopmap_alt and comes back, compiled and tuned as it was, when f_rubix is
        // switched again - a toggle used to cost a compile + tuning (~1.5 ms, a dropped frame for an engine that toggles per frame),
        // now only the first launch of each flavour after a build does.  coopmap_invalidate invalidates both.
        // (a resident session of the current flavour holds the current map: it leaves first, while nothing has changed yet - ending it
        //  may finish pending frames, which come back here through res_launch asking for the flavour the map still has)
        if (cm->valid) resident_quiesce(ctx);
        std::swap(ctx->coopmap, ctx->coopmap_alt);
        if (!ctx->coopmap) ctx->coopmap = new CoopMap();
        cm = ctx->coopmap;
        if (cm->tinted != (want_tinted != 0)) {                  // (a fresh slot)
            cm->valid = false;
            for (auto &t : cm->tuned) t = CoopMap::Tuned();
            cm->flips = 0;
            cm->tinted = want_tinted != 0;
        }
        flavour_switch = !cm->valid;                              // compiled below: drain first (the buffers may be those of an older map of this flavour)
    }
    // The measured choice (below) holds for the KIND of launch it was measured with: single frames, batches of up to 16, long batches
    // (a 270-row stripe x 64 frames ran 40.8 us on the 128x8 blocks a 16-frame measurement had picked, 32.5 us on 128x16).  A caller
    // that changes kind gets a fresh measurement - ~1.5 ms once, against every launch after it (tuned_frames was written
    // and never read, throughput depended on which launch happened to come first after a build).
    auto kind_of = [](int frames) { return frames <= 1 ? 0 : frames <= 16 ? 1 : 2; };
    int recompile_rg = 0, recompile_kb = 0;               // != 0: this kind was measured before, on blocks of another height
    bool retune = false;
    if (cm->valid && ctx->blockmap_tuning && launch_frames > 0 && cm->tuned_frames > 0 && kind_of(launch_frames) != kind_of(cm->tuned_frames) &&
        !(ctx->tile_shape == 1 || ctx->tile_shape == 2 || ctx->tile_shape == 4)) {
        const CoopMap::Tuned &t = cm->tuned[kind_of(launch_frames)];
        if (t.rg == cm->rg) {                            // same blocks: only launch parameters differ
            cm->single_form = t.form; cm->fchunk = t.fchunk; cm->lds_bytes = t.kb * 1024; cm->lds_fixed = true; cm->tuned_frames = t.frames;
            return BK_OK;
        }
        if (t.rg != 0) {
            if (cm->flips >= BK_COOP_MAX_FLIPS) return BK_OK;
            ++cm->flips;
            recompile_rg = t.rg; recompile_kb = t.kb;
        }
        retune = true;
        cm->valid = false;
    }
    if (cm->valid) return BK_OK;
    // The block map is rewritten IN PLACE.  On the way here from a build nothing can be reading it (bk_build orders itself after the
    // context's stream, and a caller with two streams synchronizes after a build: INTEGRATION.md); a change of kind in the middle of
    // a caller's steady state can find launches of the old map in flight on another stream - found as a memory fault in bench.py's
    // two-stream job - so that case drains the device first, and every compile is complete before this function returns: the launch
    // that follows may be on one stream and the one after it on another.
    bk::Range range("blockmap compile + tuning");
    if (retune || flavour_switch) BK_HIP(ctx, hipDeviceSynchronize());
    struct Settle {                                       // (every way out of this function below)
        bk_ctx *c;
        ~Settle() { (void)hipStreamSynchronize(c->stream); }
    } settle{ctx};
    if (cm->valid) return BK_OK;
    const int rows = ctx->rows();
    const int forced = ctx->tile_shape == 1 || ctx->tile_shape == 2 || ctx->tile_shape == 4 ? ctx->tile_shape : 0;   // developer knob
    const size_t bx = (ctx->W + 127) / 128;
    const size_t max_blocks = bx * (size_t)((rows + 7) / 8);
    const size_t max_px = bx * 4 * 256 * (size_t)((rows + 31) / 32 * 4 + 4);
    if (max_px > cm->alloc_px || max_blocks > cm->alloc_blocks) {      // (the header count follows ceil(rows/8), the rest ceil(rows/32))
        (void)hipFree(cm->d_hdr); (void)hipFree(cm->d_list); (void)hipFree(cm->d_idx);
        (void)hipFree(cm->d_cost); (void)hipFree(cm->d_order); (void)hipFree(cm->d_cum); (void)hipFree(cm->d_bands); (void)hipFree(cm->d_wgmap);
        cm->d_hdr = nullptr; cm->d_list = nullptr; cm->d_idx = nullptr;
        cm->d_cost = nullptr; cm->d_order = nullptr; cm->d_cum = nullptr; cm->d_bands = nullptr; cm->d_wgmap = nullptr;
        cm->alloc_px = cm->alloc_blocks = 0;
        BK_HIP(ctx, hipMalloc((void **)&cm->d_hdr, max_blocks * sizeof(CoopHdr)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_list, max_px * sizeof(uint32_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_idx, max_px * sizeof(uint16_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_cost, max_blocks * sizeof(uint32_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_order, max_blocks * sizeof(uint32_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_cum, max_blocks * sizeof(uint32_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_bands, 16 * sizeof(uint32_t)));
        BK_HIP(ctx, hipMalloc((void **)&cm->d_wgmap, (max_blocks + 8) * sizeof(uint32_t)));
        cm->alloc_px = max_px;
        cm->alloc_blocks = max_blocks;
    }
    if (!cm->d_stats) {
        BK_HIP(ctx, hipMalloc((void **)&cm->d_stats, 4 * 64 * BK_COOP_STATS * sizeof(uint32_t)));
        BK_HIP(ctx, hipHostMalloc((void **)&cm->h_stats, 4 * 64 * BK_COOP_STATS * sizeof(uint32_t), hipHostMallocDefault));
        BK_HIP(ctx, hipEventCreateWithFlags(&cm->stats_ready, hipEventDisableTiming));
    }
    // Block height: the cheapest of 128x8 / 128x16 / 128x32 by the cost model, unless forced.  The model is fed by SURVEY
    // passes that look at every 8th row of blocks and write nothing but statistics (three of them cost 3/8 of one full
    // pass); only the winner is compiled in full, and the apply can be queued right behind it - its own statistics
    // (tile stats, traffic model) are fetched asynchronously.  Round 1 compiled all three heights in full and then the
    // winner again: 0.9 ms of wall time per lensmap at 4K, against 0.28 ms for the build itself.
    int cand[3] = {4, 2, 1}, ncand = 3;
    if (forced) { cand[0] = forced; ncand = 1; }
    int best_rg = cand[0], best_kb = 0;
    int c_rg[3] = {0, 0, 0}, c_kb[3] = {0, 0, 0}, nc = 0;          // the candidates with the staging buffer the model gives each,
    double c_cost[3] = {0, 0, 0};                                   // cheapest (by the model) first
    if (ncand > 1 || ctx->apply_lds_kb <= 0) {
        const int by_min = (rows + 31) / 32;
        // (every 8th row of blocks at 4K and above, every 2nd from 1080p up, everything below: the largest block decides the
        // buffer, and a sparse survey of a small map misses it - 1080p quincuncial, surveyed every 8th row: 40 of 306 blocks
        // did not fit the buffer chosen, 5.0 instead of 2.1 us/frame)
        const int stride = by_min >= 64 ? 8 : by_min >= 32 ? 2 : 1;
        for (int i = 0; i < ncand; ++i)
            if (int r = coop_compile_launch(ctx, cm, cand[i], stride, 1 + i)) return r;
        BK_HIP(ctx, hipMemcpyAsync(cm->h_stats + 64 * BK_COOP_STATS, cm->d_stats + 64 * BK_COOP_STATS, (size_t)ncand * 64 * BK_COOP_STATS * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < ncand; ++i) {
            const int by = (rows + 8 * cand[i] - 1) / (8 * cand[i]), sampled = (by + stride - 1) / stride;
            // scale the sampled rows up to all rows (integer factor on the counts; the cost model is smooth in them)
            const uint32_t scale = (uint32_t)((by + sampled - 1) / sampled);
            fold_stats(cm->h_stats + (size_t)(1 + i) * 64 * BK_COOP_STATS, cm->stats, scale);
            double c = 0;
            const int kb = coop_choose_buffer(cm, cand[i], (double)ctx->W * rows, ctx->num_cus, &c);
            if (g_debug.print_model) {                    // developer: the cost model's inputs, one line per candidate
                fprintf(stderr, "MODEL %dx%d rg %d kb %d cost %.1f maxchunks %u slow %u empty %u bins", ctx->W, rows, cand[i], kb, c, cm->stats[0],
                        cm->stats[1], cm->stats[2]);
                for (int b = 0; b < (int)BK_COOP_BINS; ++b)
                    if (cm->stats[8 + b]) fprintf(stderr, " %d:%u:%u:%u", b, cm->stats[8 + b], cm->stats[8 + BK_COOP_BINS + b], cm->stats[8 + 2 * BK_COOP_BINS + b]);
                fprintf(stderr, "\n");
            }
            int at = nc++;
            while (at > 0 && c_cost[at - 1] > c) { c_rg[at] = c_rg[at - 1]; c_kb[at] = c_kb[at - 1]; c_cost[at] = c_cost[at - 1]; --at; }
            c_rg[at] = cand[i]; c_kb[at] = kb; c_cost[at] = c;
        }
        best_rg = c_rg[0]; best_kb = c_kb[0];
    }
    auto clamp_kb = [&](int kb) {
        if (ctx->apply_lds_kb > 0) kb = ctx->apply_lds_kb > BK_COOP_LDS_CAP / 1024 ? BK_COOP_LDS_CAP / 1024 : ctx->apply_lds_kb;   // developer knob
        return kb < 1 ? 1 : kb;
    };
    auto compile_full = [&](int rg, int kb) -> int {
        cm->rg = rg;
        cm->blocks_x = (ctx->W + 127) / 128;
        cm->blocks_y = (rows + 8 * rg - 1) / (8 * rg);
        cm->lds_bytes = clamp_kb(kb) * 1024;
        if (int r = coop_compile_launch(ctx, cm, rg, 1, 0)) return r;
        const int nb = cm->blocks_x * cm->blocks_y;
        hipLaunchKernelGGL(coop_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, cm->d_cost, nb, cm->blocks_x, (nb + 7) / 8,
                           cm->d_order, cm->d_cum, cm->d_bands, cm->d_wgmap, cm->d_stats);
        BK_HIP(ctx, hipGetLastError());
        // its statistics travel back asynchronously; whoever needs them (launch configuration, traffic model) folds them in
        BK_HIP(ctx, hipMemcpyAsync(cm->h_stats, cm->d_stats, 64 * BK_COOP_STATS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        BK_HIP(ctx, hipEventRecord(cm->stats_ready, ctx->stream));
        cm->stats_pending = true;
        return BK_OK;
    };
    // MEASURED choice (bk_set_blockmap_tuning, on by default): the cost model ranks the block heights, but its picks miss by
    // 10-17 % where the launch is small (1080p hammer / quincuncial: 128x8 beats the 128x16 it picks; which side wins also moves
    // with how cold the globe ring is).  So the candidates the model puts within 20 % of its best are compiled in full and the very
    // launch the caller is about to make - same frame count, the context's own globe - is timed on each; the fastest stays.
    // A lensmap is built once per lens / zoom change and applied every frame: ~0.7 ms more here for up to 17 % per frame.
    int measured = 0;
    cm->single_form = 0;
    cm->fchunk = 0;
    cm->lds_fixed = false;
    cm->tuned_frames = 0;
    if (recompile_rg) {                                   // measured before: compile what won then
        const CoopMap::Tuned &t = cm->tuned[kind_of(launch_frames)];
        if (int r = compile_full(recompile_rg, recompile_kb)) return r;
        cm->single_form = t.form; cm->fchunk = t.fchunk; cm->lds_bytes = t.kb * 1024; cm->lds_fixed = true; cm->tuned_frames = t.frames;
        cm->valid = true;
        return BK_OK;
    }
    if (ctx->blockmap_tuning && !forced && nc > 0 && ctx->d_globe && !(ctx->apply_flags & (2 | 4 | 32))) {
        int keep = 1;
        while (keep < nc && c_cost[keep] <= 1.2 * c_cost[0]) ++keep;
        // (timed with the caller's own frame count, as far as a scratch frame buffer of 256 MB goes - in whole groups of 8 frames)
        int nf = launch_frames > 0 ? launch_frames : (ctx->nframes >= 16 ? 16 : ctx->nframes >= 8 ? 8 : 1);
        const int for_frames = nf;
        {
            const size_t frame_bytes = (size_t)rows * ctx->W;
            const int fit = (int)std::max<size_t>(16, ((size_t)256 << 20) / std::max<size_t>(1, frame_bytes) / 8 * 8);
            if (nf > 16 && nf > fit) nf = fit;
            if (nf > 64) nf = 64;
        }
        // Single-frame launches have two more things worth measuring per height.  (a) The form: one block per workgroup (dealt out by
        // the hardware as places free up) or the strided walk (prefetch, but a static split) - the launcher's rule of thumb is right
        // for most maps and 10-20 % off for some.  (b) The staging buffer: a launch is a whole number of ROUNDS of blocks - 4050 live
        // blocks on 7 x 256 places are 2.26 rounds and take 3 (4K mercator: 16.5 us where its bytes cost 12.8) - and a buffer of 20 KiB
        // instead of the 21-26 KiB that hold every block lets 8 workgroups share a CU: 1.98 rounds, with the few larger blocks going
        // through the buffer in two passes.  The buffer is a launch parameter, so (b) costs no compile.
        // Batch launches have one: the frames a workgroup keeps a block for.  Eight amortise the block's plan best, but a SMALL map -
        // 1080p: 1020 blocks x 2 frame groups on ~1800 places - then does not fill the chip for the 20 us the launch lasts
        // (C2 x16 moved 0.545 of the peak); four frames per visit double the workgroups.
        struct Variant { int form, kb, fchunk; };
        auto variants_of = [&](int kb, Variant *v) {
            int n = 0;
            v[n++] = {0, kb, 0};
            if (nf == 1 && ctx->apply_wgs_per_cu == 16) {
                v[n++] = {1, kb, 0};
                if (ctx->apply_lds_kb <= 0 && kb > 20 && kb <= 28) v[n++] = {1, 20, 0};
            }
            if (nf >= 8 && ctx->apply_fchunk <= 0) {
                const long long live = (long long)cm->blocks_x * cm->blocks_y - (cm->stats_pending ? 0 : (long long)cm->stats[2]);
                if (live * ((nf + 7) / 8) < 2ll * ctx->num_cus * 7) {
                    v[n++] = {0, kb, 4};
                    if (nf >= 16) { v[n++] = {0, kb, 2}; v[n++] = {0, kb, 16}; }     // (twice as many again / one workgroup per block for the whole batch)
                }
            }
            return n;
        };
        Variant vtmp[6];
        if (keep > 1 || variants_of(c_kb[0], vtmp) > 1) {
            uint8_t *scratch = nullptr;
            hipEvent_t t0, t1;
            bool pooled = true;                    // (stream-ordered allocation where the device has a memory pool, plain otherwise)
            if (hipMallocAsync((void **)&scratch, (size_t)nf * rows * ctx->W, ctx->stream) != hipSuccess) {
                (void)hipGetLastError();
                pooled = false;
                BK_HIP(ctx, hipMalloc((void **)&scratch, (size_t)nf * rows * ctx->W));
            }
            BK_HIP(ctx, hipEventCreate(&t0));
            BK_HIP(ctx, hipEventCreate(&t1));
            int rc = BK_OK, win = 0;
            Variant win_v = {0, c_kb[0], 0};
            float best_ms = -1;
            // every candidate: one warm-up launch, then a train of launches between two events - back to back, as a caller's
            // steady state issues them; the globe frames advance through the context's ring from launch to launch, candidate
            // after candidate, so that none of them is handed the cache state another one left behind
            // (short launches - stripes, small frames - are timed on longer trains: at 10 us a launch the fixed costs and their
            //  jitter are as large as the differences looked for)
            const double work = (double)nf * rows * ctx->W;
            // (r6: 5 / 3 instead of 2 for the large ones - a train of two 110 us launches picked 128x8 for 4K hammer x16 on one box and
            //  128x16 on the next, 0.67 against 0.70 of the peak; the tuning runs once per lensmap and kind of launch)
            const int train = work < 20e6 ? 12 : work < 40e6 ? 6 : work < 300e6 ? 5 : 3;
            const int span = ctx->nframes > nf ? ctx->nframes - nf + 1 : 1;
            int seq = 0;
            // (the cost model's pick is measured LAST: when it wins - the usual case - it is what is compiled at the end, and the map is not
            //  compiled a third time)
            float ms_pick = -1;
            Variant pick_v = {0, c_kb[0], 0};
            for (int i = keep - 1; i >= 0 && rc == BK_OK; --i) {
                rc = compile_full(c_rg[i], c_kb[i]);
                // (timed in the configuration the caller's steady state runs in: with the block map's statistics there - live
                //  blocks, uneven bands - the launch may take another form than in the first microseconds after a compile)
                if (rc == BK_OK) rc = coop_stats_wait(ctx, cm);
                Variant vs[6];
                const int nv = variants_of(cm->lds_bytes / 1024, vs);
                for (int k = 0; k < nv && rc == BK_OK; ++k) {
                    cm->single_form = vs[k].form;
                    cm->fchunk = vs[k].fchunk;
                    cm->lds_bytes = clamp_kb(vs[k].kb) * 1024;
                    rc = launch_compiled(ctx, cm, (seq++ * nf) % span, nf, scratch, ctx->W, (size_t)rows * ctx->W, cm->tinted ? 1 : 0);
                    if (rc == BK_OK && hipEventRecord(t0, ctx->stream) != hipSuccess) rc = ctx->fail(BK_E_HIP, "block map tuning: hipEventRecord failed");
                    for (int rep = 0; rep < train && rc == BK_OK; ++rep)
                        rc = launch_compiled(ctx, cm, (seq++ * nf) % span, nf, scratch, ctx->W, (size_t)rows * ctx->W, cm->tinted ? 1 : 0);
                    float ms = 0;
                    if (rc == BK_OK && (hipEventRecord(t1, ctx->stream) != hipSuccess || hipEventSynchronize(t1) != hipSuccess ||
                                        hipEventElapsedTime(&ms, t0, t1) != hipSuccess))
                        rc = ctx->fail(BK_E_HIP, "block map tuning: timing failed");
                    if (g_debug.print_model)
                        fprintf(stderr, "TUNE %dx%d x%d rg %d kb %d form %d frames per visit %d: %.2f us per launch\n", ctx->W, rows, nf, c_rg[i], vs[k].kb, vs[k].form, vs[k].fchunk ? vs[k].fchunk : 8, ms * 1e3 / train);
                    if (rc == BK_OK && (best_ms < 0 || ms < best_ms)) { best_ms = ms; win = i; win_v = vs[k]; }
                    if (rc == BK_OK && i == 0 && k == 0) { ms_pick = ms; pick_v = vs[k]; }
                }
                measured = i;
            }
            // the first variant of candidate 0 is the cost model's pick: another one replaces it only when it is measurably (3 %) faster
            if (rc == BK_OK && ms_pick >= 0 && !(best_ms < 0.97f * ms_pick)) { win = 0; win_v = pick_v; }
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
            if (pooled) (void)hipFreeAsync(scratch, ctx->stream);
            else { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(scratch); }
            if (rc != BK_OK) return rc;
            best_rg = c_rg[win]; best_kb = win_v.kb;
            cm->tuned_frames = for_frames;
            if (win == measured) measured = -1;           // the winner is what is compiled right now
            else measured = 0;
            if (measured != -1)
                if (int r = compile_full(best_rg, best_kb)) return r;
            measured = -1;
            cm->single_form = win_v.form;
            cm->fchunk = win_v.fchunk;
            cm->lds_bytes = clamp_kb(win_v.kb) * 1024;
            cm->lds_fixed = true;
            // (filed under the CALLER's kind of launch: an 8K x 64 launch is measured with the 16 frames a scratch buffer holds, and
            //  filed under 16 it would be measured again on every launch)
            CoopMap::Tuned &t = cm->tuned[kind_of(for_frames)];
            t.rg = best_rg; t.kb = cm->lds_bytes / 1024; t.form = win_v.form; t.fchunk = win_v.fchunk; t.frames = for_frames;
        } else {                                          // nothing to choose between for this kind: the model's pick, remembered like a measured one
            CoopMap::Tuned &t = cm->tuned[kind_of(for_frames)];
            t.rg = best_rg; t.kb = clamp_kb(best_kb); t.form = 0; t.fchunk = 0; t.frames = for_frames;
            cm->tuned_frames = for_frames;
        }
    }
    if (measured != -1)
        if (int r = compile_full(best_rg, best_kb)) return r;
    cm->valid = true;
    return BK_OK;
}

static int launch_compiled(bk_ctx *ctx, CoopMap *cm, int frame0, int nframes, uint8_t *dst, int dst_pitch, size_t frame_stride, int rubix_on);

int launch_apply_coop(bk_ctx *ctx, int frame0, int nframes, uint8_t *dst, int dst_pitch, size_t frame_stride, int rubix_on)
{
    const int rows = ctx->rows();
    if (rows <= 0 || nframes <= 0) return BK_OK;
    if (int r = ensure_coopmap(ctx, nframes, rubix_on ? 1 : 0)) return r;
    return launch_compiled(ctx, ctx->coopmap, frame0, nframes, dst, dst_pitch, frame_stride, rubix_on);
}

// the apply launch over an already compiled block map
static int launch_compiled(bk_ctx *ctx, CoopMap *cm, int frame0, int nframes, uint8_t *dst, int dst_pitch, size_t frame_stride, int rubix_on)
{
    const int rows = ctx->rows();
    const int blocks_x = cm->blocks_x, nblocks = blocks_x * cm->blocks_y;
    if ((rubix_on != 0) != cm->tinted) return ctx->fail(BK_E_STATE, "apply: the block map is not of this launch's flavour (internal)");
    // frames per block visit: 8, but a batch of 8..15 frames is split in two groups so that the grid has more
    // workgroups than one scheduling round holds (8 frames: 3.86 -> 3.70 us/frame)
    int fmax = ctx->apply_fchunk > 0 ? ctx->apply_fchunk : cm->fchunk > 0 ? cm->fchunk : 8;
    if (ctx->apply_fchunk <= 0 && cm->fchunk <= 0 && nframes >= 8 && nframes < 16) fmax = (nframes + 1) / 2;
    const int fchunk = nframes < fmax ? nframes : fmax;
    const int fblocks = (nframes + fchunk - 1) / fchunk;
    const int per = (nblocks + 7) / 8;
    // Fold the block map's statistics in FIRST if they have arrived: coop_stats_wait may enlarge cm->lds_bytes (the exact
    // histogram can show a block the strided survey missed), and the dynamic-LDS size of the launch, the staging-buffer
    // size the kernel is told and the palette's place behind it must all come from the same value.
    if (cm->stats_pending && hipEventQuery(cm->stats_ready) == hipSuccess) (void)coop_stats_wait(ctx, cm);
    const int lds_buf = cm->lds_bytes;
    const size_t shmem = (size_t)lds_buf + (rubix_on ? BK_PAL_BYTES : 0);
    // Grid.  One block per workgroup (the finest split, dealt to the CUs by the hardware as they free up) when that many
    // workgroups are about what the chip holds - `apply_wgs_per_cu` = 16 per CU, deliberately generous: 4K panini x16 runs
    // 3.93 us/frame that way against 4.08 for a strided walk by the 7 per CU that are truly resident - and otherwise a strided
    // walk in which every workgroup prefetches its next block's header and list behind the current block.  Single frames are
    // judged against the true residency (LDS, registers): the prefetch is worth more when a visit is one frame long (4K
    // hammer: 12.4 us strided against 13.4; at 1440p and below, where the grid fits, the one-block form wins by 5-20 %).
    int wgs_per_band = per;
    int per_cu = ctx->apply_wgs_per_cu;
    // (r5) batch launches: one block per workgroup up to 64 workgroups per CU's worth of them (16 384 on the chip), i.e. also for 64-frame
    // launches of a 4K map (2040 blocks x 8 frame groups), which used to fall to the strided walk at three blocks per workgroup: 4K panini
    // x64 242 -> 227 us per launch, trism/panini 204 -> 190, hammer 449 -> 439 (profiles/r05_c4_whole_vs_halves.txt) - the hardware deals
    // queued workgroups to CUs as they free up, a static walk cannot
    if (fchunk > 1 && per_cu == 16) per_cu = 64;
    if (fchunk == 1 && ctx->apply_wgs_per_cu == 16) {
        const int by_lds = (int)((160u * 1024u) / (shmem ? shmem : 1));
        const int by_regs = 8;                              // (one-block form: <= 64 VGPRs)
        per_cu = by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
        const int live = cm->stats_pending ? nblocks : nblocks - (int)cm->stats[2];
        // (up to 1.5 x what is resident the one-block form still wins: 4K panini, 2040 blocks on 1792 places, 9.0 against 9.3 us;
        //  the measured tuning may have settled the form for this block map: CoopMap::single_form)
        if (cm->single_form == 1 || (cm->single_form == 0 && 2 * live <= 3 * ctx->num_cus * per_cu)) per_cu = 1 << 20;         // one block each
    }
    {
        const long long resident_per_band = (long long)ctx->num_cus * per_cu / 8;
        if ((long long)fblocks * wgs_per_band > resident_per_band) wgs_per_band = (int)((resident_per_band + fblocks - 1) / fblocks);
    }
    // ... but no workgroup should walk more than about three blocks: the strided walk is a static split, and the longer a
    // workgroup lives the more the slowest one sets the end of the launch
    if (wgs_per_band < (per + 2) / 3) wgs_per_band = (per + 2) / 3;
    if (wgs_per_band < 1) wgs_per_band = 1;
    if (wgs_per_band > per) wgs_per_band = per;
    dim3 grid((unsigned)(wgs_per_band * 8), (unsigned)fblocks);
    const bool once = wgs_per_band == per && !(ctx->apply_flags & 32);     // every workgroup has exactly one block (ablation bit 32: persistent form anyway)
    // one-block form: take the cost-balanced workgroup -> block map if bands of equal block count are known to be uneven
    // (the block map's statistics arrive asynchronously: until they are here, the direct mapping)
    int kflags = ctx->apply_flags & ~BK_KF_WGMAP;
    if (once && !(kflags & (16 | 64)) && !cm->stats_pending && cm->stats[7]) kflags |= BK_KF_WGMAP;
    // Single-frame launches fetch the globe chunks non-temporally: between two of them it is the block map L2 should keep
    // (4K hammer 12.4 -> 11.6 us, quincuncial 12.8 -> 11.8, panini 8.4 -> 8.2); batch launches lose 3-10 % that way.
    // LDS-DMA staging (bit 256) measured neutral (panini 8.39 -> 8.26, hammer 12.42 -> 12.41): the launch is bound by what
    // crosses the fabric, not by the staging instructions - it stays a developer bit.  Bit 512 leaves both to the caller.
    if (fchunk == 1 && !(kflags & 512)) kflags |= 128;
    const bool dma = fchunk == 1 && (kflags & 256) != 0 && !rubix_on;        // (LDS-DMA cannot tint a chunk on its way)
    // the strided walk with six chunks per thread in registers (82 instead of 64-67 VGPRs: 6 instead of 7 workgroups per CU) only for
    // block maps that have blocks above 16 KiB - the whole-globe lenses, whose staging buffers allow 6 per CU or fewer anyway -
    // and for batch launches, where the frame pipeline is what it keeps those blocks in (ablation bit 4096: never)
    // (4K hammer x16 7.39 -> 7.07 us/frame; mercator, whose 21 KiB buffers let 7 workgroups share a CU, lost 7 % to the registers and keeps
    //  the narrow plan: only where the staging buffer already limits a CU to 6)
    const bool wideq = !once && !rubix_on && fchunk > 1 && !cm->stats_pending && cm->stats[0] > 1024u && shmem * 7 > 160u * 1024u && !(kflags & 4096);
    // (Tried in round 3 and removed: a strided walk in which a block's last frame issues the NEXT block's first globe loads and pixel
    //  addresses - apply_coop_pipe_kernel, git history.  Where it applied (block maps without blocks of more than 1024 chunks) it was
    //  slower - 4K panini at 128x16: 12.0 -> 18.3 us single frame, 4.5 -> 6.4 us/frame x16: 77-96 VGPRs against 67 - and the
    //  whole-globe lenses it was meant for have larger blocks.  What separates mercator's 16.5 us from hammer's 11.5 is not per-block
    //  latency but the quantisation of rounds: 4050 live blocks on 1792 resident places are 2.26 rounds and take 3.)
#define BK_APPLY_K(KERNEL, RBX, N) hipLaunchKernelGGL((KERNEL<RBX, N>), grid, dim3(256), shmem, ctx->stream, cm->d_hdr, cm->d_list, cm->d_idx, \
                                           ctx->d_tints, ctx->d_offsets, ctx->d_globe, ctx->globe_stride(), ctx->nframes, frame0, dst,    \
                                           dst_pitch, frame_stride, ctx->W, rows, blocks_x, nblocks, nframes, fchunk, lds_buf,           \
                                           ctx->d_pal, kflags, cm->d_order, cm->d_bands, cm->d_wgmap)
#define BK_APPLY_KD(N) hipLaunchKernelGGL((apply_coop_once_kernel<false, N, true>), grid, dim3(256), shmem, ctx->stream, cm->d_hdr, cm->d_list, cm->d_idx, \
                                           ctx->d_tints, ctx->d_offsets, ctx->d_globe, ctx->globe_stride(), ctx->nframes, frame0, dst,    \
                                           dst_pitch, frame_stride, ctx->W, rows, blocks_x, nblocks, nframes, fchunk, lds_buf,           \
                                           ctx->d_pal, kflags, cm->d_order, cm->d_bands, cm->d_wgmap)
#define BK_APPLY_KW(N) hipLaunchKernelGGL((apply_coop_kernel<false, N, 6>), grid, dim3(256), shmem, ctx->stream, cm->d_hdr, cm->d_list, cm->d_idx, \
                                           ctx->d_tints, ctx->d_offsets, ctx->d_globe, ctx->globe_stride(), ctx->nframes, frame0, dst,    \
                                           dst_pitch, frame_stride, ctx->W, rows, blocks_x, nblocks, nframes, fchunk, lds_buf,           \
                                           ctx->d_pal, kflags, cm->d_order, cm->d_bands, cm->d_wgmap)
#define BK_APPLY(N) do { if (once && dma) BK_APPLY_KD(N); else if (once) BK_APPLY_K(apply_coop_once_kernel, false, N);                 \
                         else if (wideq) BK_APPLY_KW(N); else BK_APPLY_K(apply_coop_kernel, false, N); } while (0)
#define BK_APPLY_R(N) do { if (once) BK_APPLY_K(apply_coop_once_kernel, true, N); else BK_APPLY_K(apply_coop_kernel, true, N); } while (0)
    if (rubix_on) { if (cm->rg == 1) BK_APPLY_R(1); else if (cm->rg == 2) BK_APPLY_R(2); else BK_APPLY_R(4); }
    else { if (cm->rg == 1) BK_APPLY(1); else if (cm->rg == 2) BK_APPLY(2); else BK_APPLY(4); }
#undef BK_APPLY_R
#undef BK_APPLY_K
#undef BK_APPLY_KD
#undef BK_APPLY_KW
#undef BK_APPLY
    BK_HIP(ctx, hipGetLastError());
    return BK_OK;
}

// ---------------------------------------------------------------------------------------------
// compulsory-traffic model of the staged apply (bench.py's roofline; bk_debug_traffic_model)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mark_lines_kernel(const uint32_t *__restrict__ lmap, size_t npix, uint32_t *__restrict__ bitmap,
                                                         unsigned long long *__restrict__ counts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o = i < npix ? lmap[i] : BK_NULL_OFFSET;
    const bool mapped = o != BK_NULL_OFFSET;
    if (mapped) {
        const uint32_t line = o >> 7;
        atomicOr(&bitmap[line >> 5], 1u << (line & 31u));
    }
    const uint64_t b = __ballot(mapped);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[0], (unsigned long long)__popcll(b));
}
__global__ __launch_bounds__(256) void count_bits_kernel(const uint32_t *__restrict__ bitmap, size_t nwords, unsigned long long *__restrict__ counts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = i < nwords ? (uint32_t)__popc(bitmap[i]) : 0u;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[1], (unsigned long long)c);
}

// out[0] distinct 128-byte globe lines the owned rows of the lensmap read (per frame: the compulsory globe traffic / 128)
// out[1] 128-byte lines staged per frame by all blocks (each block counts its own distinct lines: L2 / Infinity Cache
//        absorb what neighbouring blocks share)          out[2] 16-byte chunks staged per frame
// out[3] bytes of block map read per block visit, summed over blocks: headers + chunk lists + 16-bit pixel addresses
//        (a visit serves up to out[5] frames)            out[4] mapped pixels = bytes stored per frame
// out[5] frames per block visit                          out[6] blocks        out[7] block height in pixels
int coopmap_traffic_model(bk_ctx *ctx, uint64_t out[8])
{
    if (int r = ensure_coopmap(ctx)) return r;
    CoopMap *cm = ctx->coopmap;
    if (int r = coop_stats_wait(ctx, cm)) return r;
    const size_t npix = (size_t)ctx->W * ctx->rows();
    const size_t nlines = (ctx->globe_stride() + 127) / 128, nwords = (nlines + 31) / 32;
    uint32_t *bitmap = nullptr;
    unsigned long long *counts = nullptr, h[2] = {0, 0};
    BK_HIP(ctx, hipMalloc((void **)&bitmap, nwords * 4));
    hipError_t e = hipMalloc((void **)&counts, 16);
    if (e == hipSuccess) e = hipMemsetAsync(bitmap, 0, nwords * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(counts, 0, 16, ctx->stream);
    if (e == hipSuccess && npix) {
        hipLaunchKernelGGL(mark_lines_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_offsets, npix, bitmap, counts);
        hipLaunchKernelGGL(count_bits_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, ctx->stream, bitmap, nwords, counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, counts, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(bitmap);
    (void)hipFree(counts);
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "traffic model: %s", hipGetErrorString(e));
    const uint64_t nblocks = (uint64_t)cm->blocks_x * cm->blocks_y, live = nblocks - cm->stats[2];
    out[0] = h[1];
    out[1] = cm->stats[3];
    out[2] = cm->stats[4];
    out[3] = nblocks * sizeof(CoopHdr) + (uint64_t)cm->stats[4] * 4u + live * (uint64_t)(1024 * cm->rg) * 2u;
    out[4] = h[0];
    out[5] = 8;
    out[6] = nblocks;
    out[7] = (uint64_t)(8 * cm->rg);
    return BK_OK;
}

// The persistent grid gives XCD k the band k of the screen on the assumption that workgroup b of a launch runs on XCD
// b % 8 (round-robin dispatch).  It is only a locality assumption - results do not depend on it - but nothing in the
// programming model promises it, so a test looks: every workgroup reports the XCC it runs on (HW_REG_XCC_ID).
__global__ void xcd_probe_kernel(int *__restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xF);   // hwreg(HW_REG_XCC_ID, 0, 4)
}
int coopmap_xcd_probe(bk_ctx *ctx, int *out, int nwg)
{
    int *d = nullptr;
    BK_HIP(ctx, hipMalloc((void **)&d, (size_t)nwg * sizeof(int)));
    hipLaunchKernelGGL(xcd_probe_kernel, dim3((unsigned)nwg), dim3(256), 0, ctx->stream, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)nwg * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "xcd probe: %s", hipGetErrorString(e));
    return BK_OK;
}

int coopmap_band_balance(bk_ctx *ctx, uint32_t out[18])
{
    if (int r = ensure_coopmap(ctx)) return r;
    CoopMap *cm = ctx->coopmap;
    if (int r = coop_stats_wait(ctx, cm)) return r;
    const int nb = cm->blocks_x * cm->blocks_y;
    std::vector<uint32_t> cum((size_t)nb);
    BK_HIP(ctx, hipMemcpyAsync(out, cm->d_bands, 9 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipMemcpyAsync(cum.data(), cm->d_cum, (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out[9] = cm->stats[7] ? 1u : 0u;
    for (int k = 0; k < 8; ++k) {
        const uint32_t a = out[k], b = out[k + 1];
        out[10 + k] = b > a ? cum[b - 1] - (a ? cum[a - 1] : 0u) : 0u;
    }
    return BK_OK;
}

// What a row costs the staged apply, for a multi-GPU stripe split (bk_row_costs_device): every block's cost - the one the XCD
// bands are balanced with: lines staged + a share per mapped pixel + a constant - spread over the block's rows, x16 to keep
// integer resolution.  Mapped pixels alone miss by up to 30 % on whole-globe lenses: 8K hammer's rows at 35-55 degrees of
// latitude cross the cube's corners, where a pixel touches up to five times the globe lines it touches at a face's centre.
__global__ __launch_bounds__(256) void coop_row_cost_kernel(const uint32_t *__restrict__ cost, int blocks_x, int block_rows, int rows,
                                                            uint32_t *__restrict__ out)
{
    const int y = blockIdx.x * 256 + threadIdx.x;
    if (y >= rows) return;
    const int by = y / block_rows, here = min(block_rows, rows - by * block_rows);
    uint32_t s = 0;
    for (int bx = 0; bx < blocks_x; ++bx) s += cost[by * blocks_x + bx];
    out[y] = (s * 16u + (uint32_t)here / 2u) / (uint32_t)here + 16u;          // (+ a line's worth: empty rows are dealt out too)
}

// rows_out: device uint32 [ctx->rows()], on the context stream.  The block map is compiled by the cost model alone if there is
// none yet (no timed candidates: the stripe is about to change and with it the map).
int coopmap_row_costs(bk_ctx *ctx, uint32_t *rows_out)
{
    const int tuning = ctx->blockmap_tuning;
    const bool had_map = ctx->coopmap && ctx->coopmap->valid;
    ctx->blockmap_tuning = 0;
    const int r = ensure_coopmap(ctx);
    ctx->blockmap_tuning = tuning;
    if (r) return r;
    CoopMap *cm = ctx->coopmap;
    if (!had_map && tuning) cm->valid = false;      // (should the stripe stay as it is, the next apply compiles the measured map)
    if (ctx->rows() > 0) {
        hipLaunchKernelGGL(coop_row_cost_kernel, dim3((unsigned)((ctx->rows() + 255) / 256)), dim3(256), 0, ctx->stream,
                           cm->d_cost, cm->blocks_x, 8 * cm->rg, ctx->rows(), rows_out);
        BK_HIP(ctx, hipGetLastError());
    }
    return BK_OK;
}

int coopmap_stats(bk_ctx *ctx, int out[6])
{
    if (int r = ensure_coopmap(ctx)) return r;
    CoopMap *cm = ctx->coopmap;
    if (int r = coop_stats_wait(ctx, cm)) return r;
    out[0] = cm->blocks_x * cm->blocks_y; out[1] = cm->slow_blocks; out[2] = (int)cm->stats[2]; out[3] = cm->lds_bytes;
    out[4] = 8 * cm->rg + 1000 * 128; out[5] = (int)cm->stats[3];
    return BK_OK;
}

#include "bk_apply_resident.inc"

}  // namespace bk
