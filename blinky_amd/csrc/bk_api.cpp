// bk_api.cpp -- context, memory and the non-script entry points of include/blinky_hip.h.
#include "bk_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include <dlfcn.h>

static thread_local std::string g_create_error;

// A resident apply kernel occupies the hardware queue its stream is mapped to for as long as it runs, and HIP maps all the streams of a
// process onto GPU_MAX_HW_QUEUES (default 4) of them, round robin: with a fifth stream in the process - several stripe contexts on one
// device, each with its own stream, exchange stream and resident stream - some stream shares a queue with a resident kernel and
// everything queued on it (a plate's DMA, the frame's copy back) waits until that kernel idles out (measured: 200 ms per call, the
// session's idle time, instead of 0.2 ms).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the
// process, so it is set here, at load time, unless the user has set it.
// (r6) It is a process-wide setting, so: BLINKY_HIP_KEEP_HW_QUEUES=1 leaves it alone; a value the user has set is never
// overwritten; and when it could not take effect - the HIP runtime was in the process before this library was loaded, with the variable
// unset - the first resident session says so once on stderr instead of leaving a 200 ms stall per call undiagnosed (bk::hw_queue_note).
static bool g_hwq_late = false;                 // libamdhip64 was already loaded when the variable was set here: it may have been read already
__attribute__((constructor)) static void bk_more_hardware_queues()
{
    const char *keep = getenv("BLINKY_HIP_KEEP_HW_QUEUES");
    if (keep && *keep && *keep != '0') return;
    if (getenv("GPU_MAX_HW_QUEUES")) return;
    if (void *h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD)) { g_hwq_late = true; (void)dlclose(h); }
    (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
}
void bk::hw_queue_note()
{
    static bool said = false;
    if (said) return;
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    const int n = v ? atoi(v) : 4;
    const char *keep = getenv("BLINKY_HIP_KEEP_HW_QUEUES");
    if ((g_hwq_late || n < 8) && !(keep && *keep && *keep != '0' && n >= 8)) {
        said = true;
        fprintf(stderr, "libblinkyhip: the resident apply keeps a kernel on the device; streams that share one of the runtime's %s hardware queues with it wait "
                        "until it idles out.  %s (set GPU_MAX_HW_QUEUES=16 in the environment before the process starts).\n",
                v && !g_hwq_late ? v : "(default 4)",
                g_hwq_late ? "The HIP runtime was loaded before this library could raise GPU_MAX_HW_QUEUES" : "GPU_MAX_HW_QUEUES is below 8");
    }
}

// ---- roctx ranges (bk::Range) --------------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        for (const char *name : {"libroctx64.so", "libroctx64.so.4", "librocprofiler-sdk-roctx.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
}  // namespace
bk::Range::Range(const char *name) { if (roctx().push) (void)roctx().push(name); }
bk::Range::~Range() { if (roctx().pop) (void)roctx().pop(); }

static int ensure_device(bk_ctx *ctx, bool keep_resident = false)
{
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "this context was created without a device (BK_DEVICE_NONE)");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    if (!keep_resident) bk::resident_quiesce(ctx);       // (a resident apply kernel leaves before anything else is launched)
    return BK_OK;
}

extern "C" const char *bk_version(void) { return "blinky-hip 0.5 (gfx950)"; }

extern "C" bk_ctx *bk_create(int device)
{
    int n = 0;
    hipError_t e = device == BK_DEVICE_NONE ? hipSuccess : hipGetDeviceCount(&n);
    if (device != BK_DEVICE_NONE && (e != hipSuccess || n <= 0)) {
        g_create_error = std::string("bk_create: no usable HIP device (") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                         "); libblinkyhip has no CPU fallback";
        return nullptr;
    }
    if (device == BK_DEVICE_NONE) {          // host-only context: scripts, zoom, code generation
        bk_ctx *ctx = new bk_ctx();
        ctx->device = BK_DEVICE_NONE;
        return ctx;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) {
        g_create_error = "bk_create: device index out of range";
        return nullptr;
    }
    bk_ctx *ctx = new bk_ctx();
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
    }
    if (hipSetDevice(device) != hipSuccess ||
        hipMalloc((void **)&ctx->d_pal, BK_MAX_PLATES * 256) != hipSuccess ||
        hipMalloc((void **)&ctx->d_display, 2 * (BK_MAX_PLATES + 3) * sizeof(int)) != hipSuccess) {
        g_create_error = "bk_create: hipSetDevice/hipMalloc failed";
        delete ctx;
        return nullptr;
    }
    return ctx;
}

static void free_plate_slots(bk_ctx *ctx)
{
    for (int i = 0; i < bk_ctx::kPlateSlots; ++i) {
        if (ctx->plate_ev[i]) { (void)hipEventSynchronize(ctx->plate_ev[i]); (void)hipEventDestroy(ctx->plate_ev[i]); ctx->plate_ev[i] = nullptr; }
        if (ctx->h_plate[i]) { (void)hipHostFree(ctx->h_plate[i]); ctx->h_plate[i] = nullptr; }
        if (ctx->d_plate_slot[i]) { (void)hipFree(ctx->d_plate_slot[i]); ctx->d_plate_slot[i] = nullptr; }
    }
    ctx->plate_slot_bytes = 0;
}

static void free_maps(bk_ctx *ctx)
{
    hipFree(ctx->d_offsets); ctx->d_offsets = nullptr;
    hipFree(ctx->d_tints); ctx->d_tints = nullptr;
    hipFree(ctx->d_convert); ctx->d_convert = nullptr;
    hipFree(ctx->d_frame); ctx->d_frame = nullptr;
    hipFree(ctx->d_mask); ctx->d_mask = nullptr;
    if (ctx->h_frame) hipHostFree(ctx->h_frame);
    if (ctx->h_mask) hipHostFree(ctx->h_mask);
    ctx->h_frame = nullptr; ctx->h_mask = nullptr;
    ctx->map_px = 0;
    ctx->lensmap_valid = false;
    ctx->spans_valid = false;
    bk::coopmap_invalidate(ctx);
}

extern "C" void bk_destroy(bk_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device < 0) { bk::lensprogram_free(ctx->prog); delete ctx; return; }
    hipSetDevice(ctx->device);
    bk::resident_free(ctx);
    hipStreamSynchronize(ctx->stream);
    free_maps(ctx);
    free_plate_slots(ctx);
    hipFree(ctx->d_globe);
    hipFree(ctx->d_plate_stage);
    hipFree(ctx->d_pal);
    hipFree(ctx->d_display);
    hipFree(ctx->d_flag_list);
    for (void *q : ctx->fwd_scratch) hipFree(q);
    hipFree(ctx->fwd_tables);
    if (ctx->h_build_flags) hipHostFree(ctx->h_build_flags);
    if (ctx->build_aux) hipStreamDestroy(ctx->build_aux);
    for (hipEvent_t e : ctx->build_ev) if (e) hipEventDestroy(e);
    for (hipEvent_t e : ctx->build_time_ev) if (e) hipEventDestroy(e);
    bk::coopmap_free(ctx->coopmap);
    bk::coopmap_free(ctx->coopmap_alt);
    bk::lensprogram_free(ctx->prog);
    delete ctx;
}

extern "C" const char *bk_last_error(const bk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int bk_set_stream(bk_ctx *ctx, void *hip_stream)
{
    if (!ctx) return BK_E_INVALID;
    ctx->stream = (hipStream_t)hip_stream;
    return BK_OK;
}

extern "C" int bk_synchronize(bk_ctx *ctx)
{
    if (!ctx) return BK_E_INVALID;
    if (int r = ensure_device(ctx, true)) return r;      // (while a resident session runs the context's stream carries DMAs only: every kernel launch ends the session first)
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// (re)allocate the row-dependent buffers: lens.pixels / lens.pixel_tints of fisheye.c:719-720
static int alloc_maps(bk_ctx *ctx)
{
    const size_t px = (size_t)ctx->W * ctx->rows();
    if (px == ctx->map_px && ctx->d_offsets) return BK_OK;
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_maps(ctx);
    if (!px) return BK_OK;
    const size_t words = (px + 63) / 64;
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_offsets, px * sizeof(uint32_t)));
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_tints, px));
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_convert, px * sizeof(uint32_t)));
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_frame, px));
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_mask, words * sizeof(uint64_t)));
    BK_HIP(ctx, hipHostMalloc((void **)&ctx->h_frame, px, hipHostMallocDefault));
    BK_HIP(ctx, hipHostMalloc((void **)&ctx->h_mask, words * sizeof(uint64_t), hipHostMallocDefault));
    ctx->map_px = px;
    return BK_OK;
}

static int alloc_globe(bk_ctx *ctx)
{
    const size_t need = (size_t)ctx->nframes * ctx->globe_stride();   // fisheye.c:718 per frame, plates padded to gp x ph
    if (need == ctx->globe_bytes && ctx->d_globe) return BK_OK;
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(ctx->d_globe);
    ctx->d_globe = nullptr;
    ctx->globe_bytes = 0;
    if (!need) return BK_OK;
    BK_HIP(ctx, hipMalloc((void **)&ctx->d_globe, need));
    BK_HIP(ctx, hipMemsetAsync(ctx->d_globe, 0, need, ctx->stream));
    ctx->globe_bytes = need;
    return BK_OK;
}

// leave a context EMPTY (W = H = 0, every size-dependent buffer freed; ctx->err is kept): what a failed bk_resize does,
// and what bk_multi_resize does to the contexts that had already taken the new size when a later one failed
void bk::empty_context(bk_ctx *ctx)
{
    const std::string msg = ctx->err;
    if (ctx->device >= 0) { (void)hipSetDevice(ctx->device); bk::resident_quiesce(ctx); (void)hipStreamSynchronize(ctx->stream); }
    free_maps(ctx);
    (void)hipFree(ctx->d_globe); ctx->d_globe = nullptr; ctx->globe_bytes = 0;
    (void)hipFree(ctx->d_plate_stage); ctx->d_plate_stage = nullptr;
    ctx->W = ctx->H = ctx->ps = ctx->gp = ctx->ph = 0;
    ctx->row0 = ctx->row1 = 0;
    ctx->lensmap_valid = false;
    ctx->spans_valid = false;
    ctx->err = msg;
}

extern "C" int bk_resize(bk_ctx *ctx, int width, int height)
{
    if (!ctx) return BK_E_INVALID;
    if (width <= 0 || height <= 0) return ctx->fail(BK_E_INVALID, "bk_resize: bad size %dx%d", width, height);
    if (ctx->device < 0) {                      // host-only: geometry for calc_zoom / code generation
        ctx->W = width; ctx->H = height; ctx->ps = std::min(width, height); ctx->gp = (ctx->ps + 63) & ~63;
        ctx->ph = (ctx->ps + 7) & ~7;
        ctx->row0 = 0; ctx->row1 = height;
        return BK_OK;
    }
    if (int r = ensure_device(ctx)) return r;
    if (width == ctx->W && height == ctx->H && ctx->d_offsets && ctx->d_globe && ctx->d_plate_stage) return BK_OK;
    // (6*ps*ps and W*H must fit the uint32 lensmap entries)
    const int ps = std::min(width, height);                                  // fisheye.c:707
    const int gp = (ps + 63) & ~63, ph = (ps + 7) & ~7;
    if ((uint64_t)BK_MAX_PLATES * gp * ph >= 0xFFFFFFFFull)
        return ctx->fail(BK_E_INVALID, "bk_resize: platesize %d overflows 32-bit offsets", ps);
    // failure-atomic: should an allocation fail the context is left EMPTY (W = H = 0, every buffer freed, every
    // device entry point answers BK_E_STATE) and a later bk_resize of the same size tries again; the reference
    // exits the process instead (fisheye.c:723-726)
    auto fail_empty = [&](int rc) {
        bk::empty_context(ctx);
        return rc;
    };
    ctx->W = width; ctx->H = height; ctx->ps = ps; ctx->gp = gp; ctx->ph = ph;
    ctx->row0 = 0; ctx->row1 = height;
    ctx->lensmap_valid = false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail_empty(ctx->fail(BK_E_HIP, "bk_resize: hipStreamSynchronize failed"));
    (void)hipFree(ctx->d_plate_stage);
    ctx->d_plate_stage = nullptr;
    free_plate_slots(ctx);                                  // (sized for the old platesize; re-created on the next async upload)
    if (hipMalloc((void **)&ctx->d_plate_stage, (size_t)gp * ps) != hipSuccess) {
        (void)hipGetLastError();
        return fail_empty(ctx->fail(BK_E_NOMEM, "bk_resize: out of device memory (plate staging, %dx%d)", width, height));
    }
    if (int r = alloc_maps(ctx)) return fail_empty(r);
    if (int r = alloc_globe(ctx)) return fail_empty(r);
    return BK_OK;
}

extern "C" int bk_set_rows(bk_ctx *ctx, int row0, int row1)
{
    if (!ctx) return BK_E_INVALID;
    if (row0 < 0 || row1 > ctx->H || row0 > row1)
        return ctx->fail(BK_E_INVALID, "bk_set_rows: [%d,%d) outside 0..%d", row0, row1, ctx->H);
    if (ctx->device < 0) {                      // a host-only context (BK_DEVICE_NONE) has no maps to allocate: the stripe is bookkeeping (as in bk_resize)
        ctx->row0 = row0; ctx->row1 = row1;
        return BK_OK;
    }
    if (int r = ensure_device(ctx)) return r;
    if (row0 == ctx->row0 && row1 == ctx->row1) return BK_OK;
    ctx->row0 = row0; ctx->row1 = row1;
    ctx->lensmap_valid = false;
    ctx->spans_valid = false;
    return alloc_maps(ctx);
}

extern "C" int bk_set_frames(bk_ctx *ctx, int nframes)
{
    if (!ctx || nframes < 1) return BK_E_INVALID;
    if (int r = ensure_device(ctx)) return r;
    ctx->nframes = nframes;
    if (!ctx->ps) return BK_OK;
    return alloc_globe(ctx);
}

extern "C" int bk_get_size(const bk_ctx *ctx, int *width, int *height, int *platesize, int *row0, int *row1)
{
    if (!ctx) return BK_E_INVALID;
    if (width) *width = ctx->W;
    if (height) *height = ctx->H;
    if (platesize) *platesize = ctx->ps;
    if (row0) *row0 = ctx->row0;
    if (row1) *row1 = ctx->row1;
    return BK_OK;
}

extern "C" int bk_set_rubixgrid(bk_ctx *ctx, int numcells, double cell_size, double pad_size)
{
    if (!ctx) return BK_E_INVALID;
    ctx->rubix.numcells = numcells; ctx->rubix.cell = cell_size; ctx->rubix.pad = pad_size;
    return BK_OK;
}

extern "C" int bk_set_zoom(bk_ctx *ctx, int zoom_type, int fov_degrees)
{
    if (!ctx || zoom_type < BK_ZOOM_NONE || zoom_type > BK_ZOOM_CONTAIN) return BK_E_INVALID;
    ctx->zoom_type = zoom_type;
    ctx->zoom_fov = (zoom_type == BK_ZOOM_FOV || zoom_type == BK_ZOOM_VFOV) ? fov_degrees : 0;   // clear_zoom :1273
    return BK_OK;
}

extern "C" int bk_set_apply_variant(bk_ctx *ctx, int variant)
{
    if (!ctx) return BK_E_INVALID;
    ctx->apply_variant = variant;
    return BK_OK;
}

extern "C" int bk_set_blockmap_tuning(bk_ctx *ctx, int measured)
{
    if (!ctx) return BK_E_INVALID;
    ctx->blockmap_tuning = measured != 0;
    bk::coopmap_invalidate(ctx);
    return BK_OK;
}

bk::DebugOptions bk::g_debug;
#if BK_DEBUG_API
extern "C" int bk_debug_set_option(const char *name, int value)
{
    if (!name) return BK_E_INVALID;
    if (!strcmp(name, "no_memcache")) bk::g_debug.no_memcache = value;
    else if (!strcmp(name, "libm_rel_log2")) bk::g_debug.libm_rel_log2 = value;
    else if (!strcmp(name, "print_model")) bk::g_debug.print_model = value;
    else if (!strcmp(name, "host_module")) bk::g_debug.host_module = value;
    else if (!strcmp(name, "no_direct_submit")) bk::g_debug.no_direct_submit = value;
    else if (!strcmp(name, "forward_careful")) bk::g_debug.forward_careful = value;          // forward builds pass by pass, every texel's ownership asked (tests)
    else return BK_E_INVALID;
    return BK_OK;
}
#endif
#if BK_DEBUG_API
extern "C" int bk_debug_set_ablation(bk_ctx *ctx, int bits)
{
    if (!ctx) return BK_E_INVALID;
    ctx->apply_flags = bits;
    return BK_OK;
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_set_tile_shape(bk_ctx *ctx, int lw)
{
    if (!ctx) return BK_E_INVALID;
    if (lw >= 600) { ctx->apply_block_cost = lw == 600 ? -1 : lw - 601; bk::coopmap_invalidate(ctx); return BK_OK; }   // 600 default, 601+n = n
    if (lw >= 400) { ctx->apply_lds_kb = lw - 400; bk::coopmap_invalidate(ctx); return BK_OK; }
    if (lw >= 300) { ctx->apply_fchunk = lw - 300; return BK_OK; }
    if (lw >= 100) { ctx->apply_wgs_per_cu = lw - 100; return BK_OK; }      // developer knob: 100+n = n workgroups per CU
    if (lw != 0 && lw != 1 && lw != 2 && lw != 4) return BK_E_INVALID;
    ctx->tile_shape = lw;
    bk::coopmap_invalidate(ctx);
    return BK_OK;
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_tile_stats(bk_ctx *ctx, int out[6])
{
    if (!ctx || !out) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "no lensmap");
    if (int r = ensure_device(ctx)) return r;
    return bk::coopmap_stats(ctx, out);
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_traffic_model(bk_ctx *ctx, uint64_t out[8])
{
    if (!ctx || !out) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "no lensmap");
    if (int r = ensure_device(ctx)) return r;
    return bk::coopmap_traffic_model(ctx, out);
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_band_balance(bk_ctx *ctx, uint32_t out[18])
{
    if (!ctx || !out) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "no lensmap");
    if (int r = ensure_device(ctx)) return r;
    return bk::coopmap_band_balance(ctx, out);
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_xcd_of_workgroups(bk_ctx *ctx, int *out, int nworkgroups)
{
    if (!ctx || !out || nworkgroups < 1) return BK_E_INVALID;
    if (int r = ensure_device(ctx)) return r;
    return bk::coopmap_xcd_probe(ctx, out, nworkgroups);
}
#endif

extern "C" double bk_last_build_ms(const bk_ctx *ctx) { return ctx ? ctx->last_build_ms : 0; }

// ---- lensmap table -------------------------------------------------------------------

extern "C" int bk_set_lensmap(bk_ctx *ctx, const uint32_t *offsets, const uint8_t *tints)
{
    if (!ctx || !offsets) return BK_E_INVALID;
    if (!ctx->d_offsets) return ctx->fail(BK_E_STATE, "bk_set_lensmap: call bk_resize first");
    if (int r = ensure_device(ctx)) return r;
    const size_t px = (size_t)ctx->W * ctx->rows();
    BK_HIP(ctx, hipMemcpyAsync(ctx->d_offsets, offsets, px * 4, hipMemcpyHostToDevice, ctx->stream));
    if (int r = bk::launch_convert_offsets(ctx, ctx->d_offsets, px, 1)) return r;      // reference -> padded layout
    if (tints) BK_HIP(ctx, hipMemcpyAsync(ctx->d_tints, tints, px, hipMemcpyHostToDevice, ctx->stream));
    else BK_HIP(ctx, hipMemsetAsync(ctx->d_tints, 255, px, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->lensmap_valid = true;
    ctx->spans_valid = false;
    bk::coopmap_invalidate(ctx);
    return BK_OK;
}

extern "C" int bk_read_lensmap(bk_ctx *ctx, uint32_t *offsets, uint8_t *tints)
{
    if (!ctx) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "bk_read_lensmap: no lensmap has been built");
    if (int r = ensure_device(ctx)) return r;
    const size_t px = (size_t)ctx->W * ctx->rows();
    if (offsets) {
        BK_HIP(ctx, hipMemcpyAsync(ctx->d_convert, ctx->d_offsets, px * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (int r = bk::launch_convert_offsets(ctx, ctx->d_convert, px, 0)) return r;  // padded -> reference layout
        BK_HIP(ctx, hipMemcpyAsync(offsets, ctx->d_convert, px * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (tints) BK_HIP(ctx, hipMemcpyAsync(tints, ctx->d_tints, px, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// ---- globe plates ---------------------------------------------------------------------

extern "C" void *bk_globe_device_ptr(bk_ctx *ctx, int frame)
{
    if (!ctx || !ctx->d_globe || frame < 0 || frame >= ctx->nframes) return nullptr;
    return ctx->d_globe + (size_t)frame * ctx->globe_stride();
}

extern "C" int bk_globe_pitch(const bk_ctx *ctx) { return ctx ? ctx->gp : 0; }
extern "C" int bk_globe_rows(const bk_ctx *ctx) { return ctx ? ctx->ph : 0; }
extern "C" uint32_t bk_globe_texel_offset(const bk_ctx *ctx, int plate, int px, int py)
{
    if (!ctx || plate < 0 || plate >= BK_MAX_PLATES || px < 0 || py < 0 || px >= ctx->ps || py >= ctx->ps) return BK_NULL_OFFSET;
    return bk_texel_offset((uint32_t)ctx->gp, (uint32_t)ctx->ph, (uint32_t)plate, (uint32_t)px, (uint32_t)py);
}

extern "C" int bk_upload_plate(bk_ctx *ctx, int frame, int plate, const uint8_t *src, int src_pitch)
{
    if (!ctx || !src) return BK_E_INVALID;
    if (!ctx->d_globe) return ctx->fail(BK_E_STATE, "bk_upload_plate: call bk_resize first");
    if (plate < 0 || plate >= BK_MAX_PLATES || frame < 0 || frame >= ctx->nframes || src_pitch < ctx->ps)
        return ctx->fail(BK_E_INVALID, "bk_upload_plate: bad frame/plate/pitch");
    if (ctx->resident_mode || bk::resident_running(ctx)) {        // re-tiled on the host, moved by one DMA: no kernel, the resident session stays
        if (int r = bk_upload_plate_async(ctx, frame, plate, src, src_pitch)) return r;
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return BK_OK;
    }
    if (int r = ensure_device(ctx)) return r;
    bk::Range range("bk_upload_plate");
    const size_t ps = ctx->ps;
    uint8_t *dst = ctx->d_globe + (size_t)frame * ctx->globe_stride() + (size_t)plate * ctx->plate_bytes();
    // the row memcpy loop of render_plate, fisheye.c:2441-2449: rows land in the staging buffer (gp bytes
    // apart), one kernel moves them into the plate's 16x8 tiles
    BK_HIP(ctx, hipMemcpy2DAsync(ctx->d_plate_stage, (size_t)ctx->gp, src, (size_t)src_pitch, ps, ps, hipMemcpyHostToDevice, ctx->stream));
    if (int r = bk::launch_plate_retile(ctx, dst, 1)) return r;
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// render_plate's memcpy (fisheye.c:2441-2449) lands in a pinned buffer and the DMA + retile are only ENQUEUED: the
// engine goes on to render the next plate while this one travels.  Three slots; the call blocks only if the slot it is
// about to reuse (three uploads ago) is still in flight.  Ordered with everything else on the context's stream.
extern "C" int bk_upload_plate_async(bk_ctx *ctx, int frame, int plate, const uint8_t *src, int src_pitch)
{
    if (!ctx || !src) return BK_E_INVALID;
    if (!ctx->d_globe) return ctx->fail(BK_E_STATE, "bk_upload_plate_async: call bk_resize first");
    if (plate < 0 || plate >= BK_MAX_PLATES || frame < 0 || frame >= ctx->nframes || src_pitch < ctx->ps)
        return ctx->fail(BK_E_INVALID, "bk_upload_plate_async: bad frame/plate/pitch");
    // beside a resident kernel (the session is not ended): a pure DMA needs none of the CUs it holds - the plate is re-tiled on the host;
    // with a place per CU reserved for this context's kernels (bk_set_resident_share) the usual path runs beside it: rows by DMA, re-tiled
    // by a kernel - a 4K plate costs the host 0.11 ms that way and 0.35 ms re-tiled here
    const bool beside = ctx->resident_mode != 0 || bk::resident_running(ctx);
    const bool dma_only = beside && !bk::resident_leaves_room(ctx);
    if (int r = ensure_device(ctx, beside)) return r;
    // (the slots hold a whole plate image: rows gp apart for the re-tiling kernel, or - resident mode - the plate's tiles themselves)
    const size_t ps = ctx->ps, gp = ctx->gp, bytes = ctx->plate_bytes();
    if (ctx->plate_slot_bytes != bytes) {
        free_plate_slots(ctx);
        for (int i = 0; i < bk_ctx::kPlateSlots; ++i) {
            BK_HIP(ctx, hipHostMalloc((void **)&ctx->h_plate[i], bytes, hipHostMallocDefault));
            BK_HIP(ctx, hipMalloc((void **)&ctx->d_plate_slot[i], bytes));
            BK_HIP(ctx, hipEventCreateWithFlags(&ctx->plate_ev[i], hipEventDisableTiming));
        }
        ctx->plate_slot_bytes = bytes;
        ctx->plate_next = 0;
    }
    const int slot = ctx->plate_next;
    ctx->plate_next = (slot + 1) % bk_ctx::kPlateSlots;
    BK_HIP(ctx, hipEventSynchronize(ctx->plate_ev[slot]));               // (a never-recorded event is complete)
    uint8_t *h = ctx->h_plate[slot];
    uint8_t *dst = ctx->d_globe + (size_t)frame * ctx->globe_stride() + (size_t)plate * ctx->plate_bytes();
    if (dma_only) {
        // render_plate's row memcpy (fisheye.c:2441-2449) writing the device layout directly: 16 texels of a row are one 16-byte piece
        // of a 16x8 tile (bk_texel_offset) - then ONE linear DMA into the globe.  No kernel: the resident apply holds the CUs, the
        // SDMA engines do not need one (profiles/r05_resident_apply.txt (2)).  Padding texels (ps..gp, ps..ph) are never read.
        // (tile row by tile row: eight source rows read side by side, the destination written front to back, whole 128-byte tiles at a
        //  time - 2.1 ms for the six plates of a 4K globe on one host thread, what a plain row memcpy of them takes; row by row: 2.9)
        // (r6) ... and on several: one host thread re-tiled at 13 GB/s, 28 MB of plates per 4K frame = 2.1 ms of the drop-in's 1.8 ms-per-frame
        // budget in this mode; the tile rows are independent - eight pool threads take them in runs of 16
        const size_t tpr = gp >> 4, full = ps >> 4, rest = ps & 15, tile_rows = (ps + 7) / 8;
        const size_t run = 16, parts = (tile_rows + run - 1) / run;
        bk::host_parallel(ps >= 512 ? std::min<size_t>(parts, 8) : 1, [&](size_t part) {
            const size_t nparts = ps >= 512 ? std::min<size_t>(parts, 8) : 1;
            for (size_t ty = part * run; ty < tile_rows; ty += nparts * run)
                for (size_t t = ty; t < std::min(tile_rows, ty + run); ++t) {
                    const size_t nr = ps - t * 8 < 8 ? ps - t * 8 : 8;
                    uint8_t *trow = h + t * tpr * 128;
                    const uint8_t *s0 = src + t * 8 * (size_t)src_pitch;
                    for (size_t cx = 0; cx < full; ++cx)
                        for (size_t r = 0; r < nr; ++r) memcpy(trow + cx * 128 + r * 16, s0 + r * (size_t)src_pitch + cx * 16, 16);
                    if (rest)
                        for (size_t r = 0; r < nr; ++r) memcpy(trow + full * 128 + r * 16, s0 + r * (size_t)src_pitch + full * 16, rest);
                }
        });
        BK_HIP(ctx, hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        BK_HIP(ctx, hipEventRecord(ctx->plate_ev[slot], ctx->stream));
        return BK_OK;
    }
    for (size_t y = 0; y < ps; ++y) memcpy(h + y * gp, src + y * (size_t)src_pitch, ps);      // rows gp apart, as the staging twin expects
    BK_HIP(ctx, hipMemcpyAsync(ctx->d_plate_slot[slot], h, gp * ps, hipMemcpyHostToDevice, ctx->stream));
    if (int r = bk::launch_plate_retile(ctx, dst, 1, ctx->d_plate_slot[slot])) return r;
    BK_HIP(ctx, hipEventRecord(ctx->plate_ev[slot], ctx->stream));
    return BK_OK;
}

extern "C" int bk_download_plate(bk_ctx *ctx, int frame, int plate, uint8_t *dst_host, int dst_pitch)
{
    if (!ctx || !dst_host) return BK_E_INVALID;
    if (!ctx->d_globe) return ctx->fail(BK_E_STATE, "bk_download_plate: call bk_resize first");
    if (plate < 0 || plate >= BK_MAX_PLATES || frame < 0 || frame >= ctx->nframes || dst_pitch < ctx->ps)
        return ctx->fail(BK_E_INVALID, "bk_download_plate: bad frame/plate/pitch");
    if (int r = ensure_device(ctx)) return r;
    const size_t ps = ctx->ps;
    uint8_t *src = ctx->d_globe + (size_t)frame * ctx->globe_stride() + (size_t)plate * ctx->plate_bytes();
    if (int r = bk::launch_plate_retile(ctx, src, 0)) return r;
    BK_HIP(ctx, hipMemcpy2DAsync(dst_host, (size_t)dst_pitch, ctx->d_plate_stage, (size_t)ctx->gp, ps, ps, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

extern "C" int bk_fill_plate_lcg(bk_ctx *ctx, int frame, int plate, uint32_t seed_frame)
{
    if (!ctx) return BK_E_INVALID;
    if (!ctx->d_globe) return ctx->fail(BK_E_STATE, "bk_fill_plate_lcg: call bk_resize first");
    if (plate < 0 || plate >= BK_MAX_PLATES || frame < 0 || frame >= ctx->nframes)
        return ctx->fail(BK_E_INVALID, "bk_fill_plate_lcg: bad frame/plate");
    if (int r = ensure_device(ctx)) return r;
    const uint32_t seed = 0x9E3779B9u * (uint32_t)(plate + 1 + 6 * (int)seed_frame);
    return bk::launch_fill_lcg(ctx, ctx->d_globe + (size_t)frame * ctx->globe_stride() + (size_t)plate * ctx->plate_bytes(), seed);
}

// ---- plain device buffers for hosts that do not link HIP themselves (stripe / frame buffers of the multi-GPU exchange)
extern "C" void *bk_dev_alloc(bk_ctx *ctx, size_t bytes)
{
    if (!ctx || !bytes || ensure_device(ctx) != BK_OK) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); ctx->fail(BK_E_NOMEM, "bk_dev_alloc: out of device memory (%zu bytes)", bytes); return nullptr; }
    if (hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) { (void)hipFree(p); return nullptr; }
    return p;
}
extern "C" void bk_dev_free(bk_ctx *ctx, void *p)
{
    if (!ctx || !p || ensure_device(ctx) != BK_OK) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(p);
}
extern "C" int bk_dev_read(bk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (!ctx || !dst_host || !src_dev) return BK_E_INVALID;
    if (int r = ensure_device(ctx)) return r;
    BK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// ---- apply ------------------------------------------------------------------------------

static int upload_pal(bk_ctx *ctx, int rubix_on, const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!rubix_on) return BK_OK;
    if (!pal) return ctx->fail(BK_E_INVALID, "rubix_on needs the palette LUTs");
    // (r5) the LUTs a caller passes are the same from frame to frame (f_rubix's palettes change with the game's palette, not with the
    // view): an upload per call was a copy between every two single-frame launches - 2.6 us of a 10 us launch.  The device copy is
    // kept while the bytes AND the stream are the same (an upload is ordered with the launches of the stream it was issued on only).
    if (ctx->pal_cached && ctx->pal_stream == ctx->stream && memcmp(ctx->pal_cache, pal, sizeof ctx->pal_cache) == 0) return BK_OK;
    ctx->pal_cached = false;
    BK_HIP(ctx, hipMemcpyAsync(ctx->d_pal, pal, BK_MAX_PLATES * 256, hipMemcpyHostToDevice, ctx->stream));
    memcpy(ctx->pal_cache, pal, sizeof ctx->pal_cache);
    ctx->pal_stream = ctx->stream;
    ctx->pal_cached = true;
    return BK_OK;
}

extern "C" int bk_apply_device(bk_ctx *ctx, int frame0, int nframes, void *dst_dev, int dst_pitch,
                               size_t frame_stride, int x0, int y0, int rubix_on,
                               const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!ctx || !dst_dev) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "bk_apply: no lensmap (bk_build / bk_set_lensmap first)");
    if (dst_pitch < ctx->W + x0 || x0 < 0 || y0 < 0 || frame0 < 0 || nframes < 1)
        return ctx->fail(BK_E_INVALID, "bk_apply_device: bad pitch/origin/frames");
    if (int r = ensure_device(ctx)) return r;
    bk::Range range("bk_apply_device");
    if (int r = upload_pal(ctx, rubix_on, pal)) return r;
    uint8_t *first = (uint8_t *)dst_dev + (size_t)(y0 + ctx->row0) * dst_pitch + x0;
    return bk::launch_apply(ctx, frame0, nframes, first, dst_pitch, frame_stride, rubix_on);
}

// ---- resident single-frame apply (bk_apply_resident.inc) ------------------------------------------------------------
extern "C" int bk_set_resident_share(bk_ctx *ctx, int part, int parts, int reserve_cus_per_xcd /* = workgroup slots per CU */)
{
    if (!ctx) return BK_E_INVALID;
    if (parts < 1 || parts > 8 || part < 0 || part >= parts || reserve_cus_per_xcd < 0 || reserve_cus_per_xcd > 7)
        return ctx->fail(BK_E_INVALID, "bk_set_resident_share: part %d of %d, %d workgroup slots reserved per CU", part, parts, reserve_cus_per_xcd);
    if (ctx->device >= 0 && (part != ctx->res_part || parts != ctx->res_parts || reserve_cus_per_xcd != ctx->res_reserve)) {
        BK_HIP(ctx, hipSetDevice(ctx->device));
        bk::resident_quiesce(ctx);                  // (the running kernel sits on the old share)
    }
    ctx->res_part = part; ctx->res_parts = parts; ctx->res_reserve = reserve_cus_per_xcd;
    return BK_OK;
}

extern "C" int bk_set_resident_apply(bk_ctx *ctx, int on)
{
    if (!ctx) return BK_E_INVALID;
    if (!on && ctx->resident_mode && ctx->device >= 0) { BK_HIP(ctx, hipSetDevice(ctx->device)); bk::resident_quiesce(ctx); }
    ctx->resident_mode = on ? 1 : 0;
    return BK_OK;
}

extern "C" int bk_apply_resident_begin(bk_ctx *ctx, int rubix_on, const uint8_t pal[BK_MAX_PLATES][256], double idle_ms)
{
    if (!ctx) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "bk_apply_resident_begin: no lensmap (bk_build / bk_set_lensmap first)");
    if (ctx->apply_variant == 0) return ctx->fail(BK_E_STATE, "bk_apply_resident_begin: the resident apply is the staged variant (bk_set_apply_variant 2 / -1)");
    if (ctx->rows() <= 0) return ctx->fail(BK_E_STATE, "bk_apply_resident_begin: this context owns no rows");
    if (int r = ensure_device(ctx)) return r;
    bk::Range range("bk_apply_resident_begin");
    if (int r = upload_pal(ctx, rubix_on, pal)) return r;
    if (int r = bk::resident_begin(ctx, rubix_on, idle_ms)) return r;
    // what this session tints with: bk_apply_begin compares its own call's rubix flag and palette against these, whoever began the session
    ctx->res_rubix = rubix_on != 0;
    if (rubix_on) memcpy(ctx->res_pal, pal, sizeof ctx->res_pal);
    return BK_OK;
}

extern "C" int bk_apply_resident_submit(bk_ctx *ctx, int frame, void *dst_dev, int dst_pitch, int x0, int y0, uint64_t *ticket)
{
    if (!ctx || !dst_dev) return BK_E_INVALID;
    if (!ctx->resident) return ctx->fail(BK_E_STATE, "bk_apply_resident_submit without bk_apply_resident_begin");
    if (dst_pitch < ctx->W + x0 || x0 < 0 || y0 < 0 || frame < 0) return ctx->fail(BK_E_INVALID, "bk_apply_resident_submit: bad pitch/origin/frame");
    if (int r = ensure_device(ctx, true)) return r;
    uint8_t *first = (uint8_t *)dst_dev + (size_t)(y0 + ctx->row0) * dst_pitch + x0;
    return bk::resident_submit(ctx, frame, first, dst_pitch, ticket);
}

// a batch through the resident kernel: frame f of the batch is globe (frame0 + f) % resident globes -> dst_dev + f * frame_stride
// (bk_apply_device's addressing); one command per frame, the call blocks only while the command ring is full
extern "C" int bk_apply_resident_submit_batch(bk_ctx *ctx, int frame0, int nframes, void *dst_dev, int dst_pitch, size_t frame_stride,
                                              int x0, int y0, uint64_t *last_ticket)
{
    if (!ctx || !dst_dev) return BK_E_INVALID;
    if (!ctx->resident) return ctx->fail(BK_E_STATE, "bk_apply_resident_submit_batch without bk_apply_resident_begin");
    if (dst_pitch < ctx->W + x0 || x0 < 0 || y0 < 0 || frame0 < 0 || nframes < 1) return ctx->fail(BK_E_INVALID, "bk_apply_resident_submit_batch: bad pitch/origin/frames");
    if (int r = ensure_device(ctx, true)) return r;
    uint8_t *first = (uint8_t *)dst_dev + (size_t)(y0 + ctx->row0) * dst_pitch + x0;
    for (int f = 0; f < nframes; ++f)
        if (int r = bk::resident_submit(ctx, (frame0 + f) % ctx->nframes, first + (size_t)f * frame_stride, dst_pitch, last_ticket)) return r;
    return BK_OK;
}

extern "C" int bk_apply_resident_wait(bk_ctx *ctx, uint64_t ticket, double *gpu_us)
{
    if (!ctx) return BK_E_INVALID;
    if (int r = ensure_device(ctx, true)) return r;
    return bk::resident_wait(ctx, ticket, gpu_us);
}

#if BK_DEBUG_API
extern "C" int bk_debug_resident_latency(bk_ctx *ctx, int frames, void *dst_dev, int dst_pitch, int globes, double *host_us, double *device_us)
{
    if (!ctx || !dst_dev || frames < 1 || globes < 1) return BK_E_INVALID;
    std::vector<double> wall((size_t)frames), dev((size_t)frames);
    for (int i = 0; i < frames; ++i) {
        struct timespec t0, t1;
        uint64_t ticket = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if (int r = bk_apply_resident_submit(ctx, (7 * i) % globes, dst_dev, dst_pitch, 0, 0, &ticket)) return r;
        if (int r = bk_apply_resident_wait(ctx, ticket, &dev[(size_t)i])) return r;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        wall[(size_t)i] = (double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3;
    }
    std::sort(wall.begin(), wall.end());
    std::sort(dev.begin(), dev.end());
    if (host_us) *host_us = wall[(size_t)frames / 2];
    if (device_us) *device_us = dev[(size_t)frames / 2];
    return BK_OK;
}
#endif

extern "C" int bk_apply_resident_end(bk_ctx *ctx)
{
    if (!ctx) return BK_E_INVALID;
    if (ctx->device < 0) return BK_OK;
    if (int r = ensure_device(ctx, true)) return r;
    return bk::resident_stop(ctx);
}

extern "C" int bk_apply_resident_info(bk_ctx *ctx, int out[12])
{
    if (!ctx || !out) return BK_E_INVALID;
    return bk::resident_info(ctx, out);
}

// mapped spans of the owned rows from the device bitmap (once per lensmap)
static int ensure_spans(bk_ctx *ctx)
{
    if (ctx->spans_valid) return BK_OK;
    const size_t px = (size_t)ctx->W * ctx->rows();
    const size_t words = (px + 63) / 64;
    if (int r = bk::launch_mask(ctx)) return r;
    BK_HIP(ctx, hipMemcpyAsync(ctx->h_mask, ctx->d_mask, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->spans.clear();
    for (int r = 0; r < ctx->rows(); ++r) {
        int start = -1;
        for (int x = 0; x <= ctx->W; ++x) {
            bool m = false;
            if (x < ctx->W) {
                const size_t i = (size_t)r * ctx->W + x;
                m = (ctx->h_mask[i >> 6] >> (i & 63)) & 1u;
            }
            if (m && start < 0) start = x;
            if (!m && start >= 0) { ctx->spans.push_back({r, start, x}); start = -1; }
        }
    }
    ctx->spans_valid = true;
    // every pixel of the owned rows mapped (panini, stereographic, ...): no merge needed, the frame can be copied whole
    ctx->fully_mapped = ctx->spans.size() == (size_t)ctx->rows();
    for (const bk::Span &s : ctx->spans) if (s.x0 != 0 || s.x1 != ctx->W) { ctx->fully_mapped = false; break; }
    return BK_OK;
}

// bk_apply in two halves, so that a host (or bk_multi_apply, over several devices) can do something else while the GPU
// warps and the frame travels: begin = enqueue the warp of the owned rows into the staging frame and, unless every
// pixel is mapped, its copy into pinned host memory; end = wait and deliver into the caller's buffer.
extern "C" int bk_apply_begin(bk_ctx *ctx, int frame, int rubix_on, const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!ctx) return BK_E_INVALID;
    if (!ctx->lensmap_valid) return ctx->fail(BK_E_STATE, "bk_apply: no lensmap (bk_build / bk_set_lensmap first)");
    if (frame < 0 || frame >= ctx->nframes) return ctx->fail(BK_E_INVALID, "bk_apply: bad frame %d", frame);
    if (ctx->resident_mode && ctx->apply_variant != 0) {
        // the frame as a command to the resident kernel (begun here if it is not running, or was begun with another palette)
        if (int r = ensure_device(ctx, true)) return r;
        bk::Range range("bk_apply_begin (resident)");
        if (rubix_on && !pal) return ctx->fail(BK_E_INVALID, "rubix_on needs the palette LUTs");
        bk::Resident *R = ctx->resident;
        const bool same = R && ctx->res_rubix == (rubix_on != 0) && (!rubix_on || memcmp(ctx->res_pal, pal, sizeof ctx->res_pal) == 0);
        if (!ctx->spans_valid || !R || !same) {
            if (int r = ensure_device(ctx)) return r;                 // (ends a session begun with other settings; the mask kernel below needs the CUs)
            if (int r = ensure_spans(ctx)) return r;
            if (int r = upload_pal(ctx, rubix_on, pal)) return r;
            BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (int r = bk::resident_begin(ctx, rubix_on, 0.0)) return r;
            ctx->res_rubix = rubix_on != 0;
            if (rubix_on) memcpy(ctx->res_pal, pal, sizeof ctx->res_pal);
        }
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));               // the plates of this frame have arrived (bk_upload_plate_async)
        uint8_t *first = ctx->d_frame;                                // a tight staging frame of the owned rows: its pixel (0, row0) is d_frame
        if (int r = bk::resident_submit(ctx, frame, first, ctx->W, &ctx->apply_ticket)) return r;
        ctx->apply_in_flight = true;
        ctx->apply_was_resident = true;                               // bk_apply_end follows THIS call's path, whatever bk_set_resident_apply says by then
        return BK_OK;
    }
    ctx->apply_was_resident = false;
    ctx->apply_ticket = 0;
    if (int r = ensure_device(ctx)) return r;
    bk::Range range("bk_apply_begin");
    if (int r = ensure_spans(ctx)) return r;
    if (int r = upload_pal(ctx, rubix_on, pal)) return r;
    // warp the owned rows into a tight staging frame
    if (int r = bk::launch_apply(ctx, frame, 1, ctx->d_frame, ctx->W, 0, rubix_on)) return r;
    if (!ctx->fully_mapped)
        BK_HIP(ctx, hipMemcpyAsync(ctx->h_frame, ctx->d_frame, (size_t)ctx->W * ctx->rows(), hipMemcpyDeviceToHost, ctx->stream));
    ctx->apply_in_flight = true;
    return BK_OK;
}

extern "C" int bk_apply_end(bk_ctx *ctx, uint8_t *dst, int dst_pitch, int x0, int y0)
{
    if (!ctx || !dst) return BK_E_INVALID;
    if (!ctx->apply_in_flight) return ctx->fail(BK_E_STATE, "bk_apply_end without bk_apply_begin");
    const bool resident = ctx->apply_was_resident && ctx->apply_ticket != 0;
    if (int r = ensure_device(ctx, resident)) return r;
    ctx->apply_in_flight = false;
    const int rows = ctx->rows();
    bool pinned = false;
    if (resident) {
        const uint64_t t = ctx->apply_ticket;
        ctx->apply_ticket = 0;
        if (int r = bk::resident_wait(ctx, t, nullptr)) return r;     // the frame is in memory
        // Through the pinned frame (one DMA), then host rows - unless a place per CU is reserved for this context's kernels: a 2-D copy
        // into the caller's pageable, pitched buffer is a shader copy in this runtime for small frames, and a shader needs a place on a
        // CU - with every place taken by the resident kernel (a 6-per-CU form filling the chip: rubix at 640x480) it waited for the
        // kernel's idle exit, 200 ms per frame, and the session restarted
        pinned = !bk::resident_leaves_room(ctx);
        if (pinned || !ctx->fully_mapped)
            BK_HIP(ctx, hipMemcpyAsync(ctx->h_frame, ctx->d_frame, (size_t)ctx->W * rows, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (ctx->fully_mapped && !pinned) {
        // nothing to preserve between the mapped pixels: one 2-D copy straight into the caller's buffer, no host merge
        BK_HIP(ctx, hipMemcpy2DAsync(dst + (size_t)(y0 + ctx->row0) * dst_pitch + x0, (size_t)dst_pitch, ctx->d_frame, (size_t)ctx->W,
                                     (size_t)ctx->W, (size_t)rows, hipMemcpyDeviceToHost, ctx->stream));
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return BK_OK;
    }
    // merge only the mapped spans into the caller's buffer (VBUFFER(x+scr_vrect.x, y+scr_vrect.y), fisheye.c:2414-2421)
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // (r6: a 4K frame is 8.3 MB of row copies - on a few pool threads when it is that large)
    const size_t nspans = ctx->spans.size(), nparts = (size_t)ctx->W * rows >= ((size_t)1 << 21) ? std::min<size_t>(8, (nspans + 63) / 64) : 1;
    bk::host_parallel(nparts, [&](size_t part) {
        const size_t lo = nspans * part / nparts, hi = nspans * (part + 1) / nparts;
        for (size_t i = lo; i < hi; ++i) {
            const bk::Span &s = ctx->spans[i];
            memcpy(dst + (size_t)(y0 + ctx->row0 + s.row) * dst_pitch + x0 + s.x0, ctx->h_frame + (size_t)s.row * ctx->W + s.x0, (size_t)(s.x1 - s.x0));
        }
    });
    return BK_OK;
}

extern "C" int bk_apply(bk_ctx *ctx, int frame, uint8_t *dst, int dst_pitch, int x0, int y0,
                        int rubix_on, const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!ctx || !dst) return BK_E_INVALID;
    if (int r = bk_apply_begin(ctx, frame, rubix_on, pal)) return r;
    return bk_apply_end(ctx, dst, dst_pitch, x0, y0);
}

// ---- rubix palettes (fisheye.c:835-908), integer host precompute -----------------------------

static int closest_pal_index(const uint8_t *basepal, int r, int g, int b)
{
    int best = 256 * 256 * 256, besti = 0;
    for (int i = 0; i < 256; ++i) {
        const int dr = basepal[3 * i] - r, dg = basepal[3 * i + 1] - g, db = basepal[3 * i + 2] - b;
        const int d = dr * dr + dg * dg + db * db;
        if (d < best) { best = d; besti = i; }   // first minimum wins, fisheye.c:847
    }
    return besti;
}

extern "C" void bk_create_palmap(const uint8_t *basepal, uint8_t pal_out[BK_MAX_PLATES][256])
{
    static const int tint[BK_MAX_PLATES][3] = {{255, 255, 255}, {0, 0, 255}, {255, 0, 0},
                                               {255, 255, 0}, {255, 0, 255}, {0, 255, 255}};   // fisheye.c:866-886
    const int percent = 256 / 6;
    for (int j = 0; j < BK_MAX_PLATES; ++j)
        for (int i = 0; i < 256; ++i) {
            int c[3];
            for (int k = 0; k < 3; ++k) {
                int v = basepal[3 * i + k];
                v += percent * (tint[j][k] - v) >> 8;                                         // fisheye.c:895
                c[k] = v < 0 ? 0 : v > 255 ? 255 : v;
            }
            pal_out[j][i] = (uint8_t)closest_pal_index(basepal, c[0], c[1], c[2]);
        }
}
