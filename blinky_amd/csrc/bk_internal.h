// bk_internal.h -- private state of libblinkyhip.so (see include/blinky_hip.h for the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/blinky_hip.h"
#ifndef BK_DEBUG_API
#define BK_DEBUG_API 0
#endif
#if BK_DEBUG_API
#include "../../include/blinky_hip_debug.h"
#endif
#include "bk_build_params.h"      // the device globe layout (bk_texel_offset)

namespace bk {

// process-wide developer / test switches (bk_debug_set_option, debug API builds only; always 0 otherwise)
struct DebugOptions { int no_memcache = 0, libm_rel_log2 = 0, print_model = 0, host_module = 0, no_direct_submit = 0, forward_careful = 0; };
extern DebugOptions g_debug;

// roctx ranges around the library's phases (bk_build, block-map compile, apply launches, plate uploads, the resident session):
// `rocprofv3 --marker-trace --kernel-trace` then groups the kernels by phase instead of by name.
// libroctx64 is resolved with dlopen on first use; without it (or without a profiler attached) a range costs two indirect calls.
struct Range {
    explicit Range(const char *name);
    ~Range();
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

// A run of mapped pixels in one output row (host-side, for merging a warped
// frame into the caller's vid.buffer without touching unmapped pixels).
struct Span { int row, x0, x1; };

struct Rubix { int numcells = 10; double cell = 4, pad = 1; };   // defaults: fisheye.c:672

struct LensProgram;   // bk_lens.cpp: parsed scripts + hiprtc modules
struct CoopMap;       // bk_apply_coop.hip: per-block staging plans of the workgroup-cooperative apply kernel
struct Resident;      // bk_apply_resident.inc: the resident single-frame apply (its kernel, command ring and stream)

}  // namespace bk

struct bk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // geometry (fisheye.c:704-708)
    int W = 0, H = 0, ps = 0;
    int gp = 0, ph = 0;              // padded plate width round_up(ps, 64) / height round_up(ps, 8) of the device globe
    int row0 = 0, row1 = 0;          // owned output rows
    int nframes = 1;

    // globe (fisheye.c:334-377)
    bk_plate plates[BK_MAX_PLATES] = {};
    int numplates = 0;
    bool globe_valid = false;
    int display[BK_MAX_PLATES] = {};

    // zoom / rubix (fisheye.c:453-474)
    int zoom_type = BK_ZOOM_NONE, zoom_fov = 0;
    double scale = -1;
    bk::Rubix rubix;

    // device memory
    uint8_t *d_globe = nullptr;      // [nframes][6][ph/8][gp/16] tiles of 16x8 texels (bk_texel_offset, bk_build_params.h)
    uint8_t *d_plate_stage = nullptr;  // [ps][gp] row-major staging of one plate for bk_upload_plate / bk_download_plate
    // bk_upload_plate_async: pinned host buffers the caller's rows are copied into, their device-side staging twins and
    // the event that says "this slot's DMA + retile are done" - three slots, so that the DMA of plate k overlaps the
    // engine's render of plate k+1
    static constexpr int kPlateSlots = 3;
    uint8_t *h_plate[kPlateSlots] = {};
    uint8_t *d_plate_slot[kPlateSlots] = {};
    hipEvent_t plate_ev[kPlateSlots] = {};
    size_t plate_slot_bytes = 0;
    int plate_next = 0;
    uint32_t *d_offsets = nullptr;   // [row1-row0][W]  offsets into the PADDED globe layout
    uint32_t *d_convert = nullptr;   // [row1-row0][W]  scratch for layout conversion at the ABI boundary
    uint8_t *d_tints = nullptr;      // [row1-row0][W]
    uint8_t *d_frame = nullptr;      // [row1-row0][W] staging for bk_apply (host dst)
    uint8_t *d_pal = nullptr;        // [6][256]
    uint64_t *d_mask = nullptr;      // mapped bits, 1 per pixel of the owned rows
    int *d_display = nullptr;        // [6] display flags + [1] error bits + [1] flagged-entry count + [1] first malformed result (scan key + 1) written by the build kernels
    unsigned int last_bad_key = 0;   // of the last bk_build: 1 + scan key of the first pixel whose callback returned a malformed result (0 = none)
    uint32_t *d_flag_list = nullptr; // entries a build flagged for re-evaluation on the platform libm (bk_device_rt.h)
    size_t flag_cap = 0;
    // scratch of the forward build, kept from one build to the next (four hipMalloc / hipFree of 320 MB at 4K cost more wall time than
    // its kernels): texel-corner screen coordinates, corner flags, the two key planes; released when an inverse map is built
    void *fwd_scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t fwd_scratch_bytes[4] = {0, 0, 0, 0};
    // fill_params' rubix grid bitmap for (platesize, rubix numcells / cell / pad): a division and an fmod per texel column, once - not per build
    unsigned int grid_cache[256] = {0};
    double grid_cache_key[4] = {-1, 0, 0, 0};
    hipStream_t build_aux = nullptr;     // a forward build clears its key planes here, beside the corner pass
    hipEvent_t build_ev[2] = {nullptr, nullptr};
    hipEvent_t build_time_ev[2] = {nullptr, nullptr};   // bk_build's timing pair, kept (creating and destroying them is four runtime calls a build)
    int *h_build_flags = nullptr;        // pinned: the counters of a forward build's two passes, read back without a stop in between
    bool fwd_tiles_used = false;         // the last build's quad pass went by bk_forward_tiles' flags (bk_debug_forward_tiles)
    void *fwd_tables = nullptr;          // BkBuildParams::fwd_quot + fwd_uv for platesize fwd_tables_ps (bk_lens.cpp)
    int fwd_tables_ps = -1;
    int last_flagged = 0, last_changed = 0;   // of the last bk_build: entries re-evaluated on the host / entries that changed
    uint8_t *h_frame = nullptr;      // pinned, [row1-row0][W]
    uint64_t *h_mask = nullptr;      // pinned
    size_t globe_bytes = 0;          // allocated size of d_globe
    size_t map_px = 0;               // allocated pixels for the map buffers

    bool lensmap_valid = false;
    std::vector<bk::Span> spans;     // mapped spans of the owned rows
    bool spans_valid = false;
    bool apply_in_flight = false;    // between bk_apply_begin and bk_apply_end
    bool fully_mapped = false;       // every pixel of the owned rows is mapped: bk_apply copies the frame whole

    int apply_variant = -1;          // -1 auto (= 2); 0 direct gather, 2 workgroup-cooperative LDS blocks
    int num_cus = 256;               // multiProcessorCount of the device
    int apply_flags = 0;             // developer ablations of the coop apply (bk_debug_set_ablation): 2 no globe loads, 4 no stores, 8 no load pipelining
    int apply_block_cost = -1;       // coop apply: constant term of a block's cost in the band balance (-1 = default; knob 600+n)
    int apply_lds_kb = 0;            // coop apply: force the staging buffer size in KiB (0 = cost model; knob 400+n)
    int apply_fchunk = 0;            // frames a workgroup keeps a block for (0 = default 8; knob 300+n)
    int apply_wgs_per_cu = 16;       // persistent apply grid: workgroups per CU (tunable, bk_debug_set_tile_shape)
    bool blockmap_tuning = true;     // bk_set_blockmap_tuning: block height of the staged apply chosen by timing the candidates
    int tile_shape = 0;              // coop apply: 0 = block height by cost model, 1/2/4 = force 128x8 / 128x16 / 128x32
    bk::CoopMap *coopmap = nullptr;       // owned; freed with bk::coopmap_free
    bk::CoopMap *coopmap_alt = nullptr;   // owned: the block map of the OTHER flavour (plain / tinted), parked while f_rubix is the other way
    bk::Resident *resident = nullptr;     // owned; freed with bk::resident_free (bk_apply_resident_begin .. _end)
    int resident_mode = 0;                // bk_set_resident_apply: bk_apply / bk_apply_begin..end / bk_upload_plate* go through the resident kernel
    int res_part = 0, res_parts = 1, res_reserve = 0;   // bk_set_resident_share: which CUs of every XCD the resident kernel may take
    uint64_t apply_ticket = 0;            // resident mode: the frame bk_apply_begin submitted
    uint8_t res_pal[BK_MAX_PLATES * 256] = {};   // resident mode: the palette the running session was begun with
    uint8_t pal_cache[BK_MAX_PLATES * 256] = {}; // what d_pal holds, uploaded on pal_stream (upload_pal)
    hipStream_t pal_stream = nullptr;
    bool pal_cached = false;
    bool res_rubix = false;
    bool apply_was_resident = false;      // latched by bk_apply_begin: the frame in flight went to the resident kernel
    bk::LensProgram *prog = nullptr;      // owned; freed with bk::lensprogram_free
    double last_build_ms = 0;
    double last_host_eval_ms = 0;    // of that: wall time of the host re-evaluation of the flagged entries
    double last_kernel_wall_ms = 0;  // inverse build: wall time of the kernel launch(es) + counter / flag-list read-back (sorted)
    int last_kernel_retries = 0;     // kernel re-runs because the flag list had to grow
    bool last_fixup_compiled = false;   // the flagged entries were re-derived by the compiled host module (not the interpreter)
    uint32_t *host_sink_off = nullptr; uint8_t *host_sink_tint = nullptr;   // bk_debug_host_build: a host-built table goes here instead of the device
    int last_build_path = 0;         // of the last bk_build: 0 = GPU kernels, 1 = host worker pool, 2 = one sequential host scan (bk_last_build_path)
    std::string last_build_why;
    int sequential_build = 1;        // bk_set_sequential_build: 0 never, 1 (default) when the callbacks carry state from pixel to pixel, 2 always
    bool async_compile = false;      // bk_set_async_compile: bk_build returns BK_PENDING instead of waiting for hiprtc

    int fail(int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)))
    {
        char buf[2048];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    int rows() const { return row1 - row0; }
    size_t plate_bytes() const { return (size_t)gp * ph; }
    size_t globe_stride() const { return (size_t)BK_MAX_PLATES * gp * ph; }
};

#define BK_HIP(ctx, expr)                                                               \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return (ctx)->fail(BK_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace bk {

// bk_apply.hip
int launch_apply(bk_ctx *ctx, int frame0, int nframes, uint8_t *dst_first_owned_row, int dst_pitch,
                 size_t frame_stride, int rubix_on);
int launch_mask(bk_ctx *ctx);                 // d_offsets -> d_mask
int launch_fill_lcg(bk_ctx *ctx, uint8_t *plate_dst, uint32_t seed);   // one plate, padded rows
int launch_convert_offsets(bk_ctx *ctx, uint32_t *buf, size_t n, int to_device);   // reference <-> device (tiled) layout
int launch_plate_retile(bk_ctx *ctx, uint8_t *plate_tiled, int to_tiled, uint8_t *rowmajor = nullptr);   // row-major staging (default d_plate_stage) <-> a plate of the globe
int launch_scatter32(bk_ctx *ctx, uint32_t *dst, const uint32_t *h_idx, const uint32_t *h_val, size_t n);   // dst[idx[i]] = val[i]; synchronous
int launch_scatter8(bk_ctx *ctx, uint8_t *dst, const uint32_t *h_idx, const uint8_t *h_val, size_t n);
// the owned rows of the lensmap as the reference leaves them when its scan stops at the pixel with scan key bad_key - 1 (everything it
// had not reached yet NULL) and the display flags of what is left; synchronous
int launch_truncate_scan(bk_ctx *ctx, unsigned int bad_key, int display_out[BK_MAX_PLATES]);
// bk_apply_coop.hip
void coopmap_invalidate(bk_ctx *ctx);
int launch_apply_coop(bk_ctx *ctx, int frame0, int nframes, uint8_t *dst_first_owned_row, int dst_pitch,
                      size_t frame_stride, int rubix_on);
int coopmap_stats(bk_ctx *ctx, int out[6]);   // blocks, direct-gather blocks, empty blocks, LDS bytes per buffer
int coopmap_traffic_model(bk_ctx *ctx, uint64_t out[8]);
int coopmap_xcd_probe(bk_ctx *ctx, int *out, int nwg);
int coopmap_band_balance(bk_ctx *ctx, uint32_t out[18]);
int coopmap_row_costs(bk_ctx *ctx, uint32_t *rows_out);      // device uint32 [rows()]: block costs spread over their rows, x16
}
int bk_row_costs_device(bk_ctx *ctx, uint32_t *cost_dev);     // bk_probe.hip: what every owned row costs the apply, into a device uint32 [H]
namespace bk {
void coopmap_free(CoopMap *);
// bk_apply_resident.inc (part of bk_apply_coop.hip): the resident single-frame apply
int resident_begin(bk_ctx *ctx, int rubix_on, double idle_ms);
int resident_submit(bk_ctx *ctx, int frame, uint8_t *dst_first_owned_row, int dst_pitch, uint64_t *ticket);
int resident_wait(bk_ctx *ctx, uint64_t ticket, double *gpu_us);
int resident_stop(bk_ctx *ctx);          // the kernel leaves (outstanding frames are finished first); begin again to resume
void resident_free(bk_ctx *ctx);
int resident_info(bk_ctx *ctx, int out[12]);
// every other device entry point of a context calls this first: a resident kernel fills the chip, whatever else the context
// launches would wait for it to leave
inline void resident_quiesce(bk_ctx *ctx) { if (ctx->resident) (void)resident_stop(ctx); }
bool resident_leaves_room(bk_ctx *ctx);
void hw_queue_note();      // one line on stderr, once, when GPU_MAX_HW_QUEUES could not be raised for the resident apply (bk_api.cpp)
/* job(i) for i in [0, parts) on the library's pool of host threads (bk_lens.cpp: the pool that re-derives flagged pixels), the caller
 * waiting; parts <= 1 or a pool of one thread: on the calling thread */
void host_parallel(size_t parts, const std::function<void(size_t)> &job);
bool resident_running(bk_ctx *ctx);      // a session's kernel is on the device right now

// bk_lens.cpp
void lensprogram_free(LensProgram *);
// BK_OK when the current lens + globe's kernels are loaded (or were found in a cache just now), BK_PENDING while hiprtc is
// still compiling them on another thread (asynchronous compilation only; nothing of the context's lensmap is touched)
int build_module_ready(bk_ctx *ctx);
// bk_api.cpp
void empty_context(bk_ctx *ctx);

}  // namespace bk
