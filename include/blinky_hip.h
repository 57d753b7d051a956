/*
 * blinky_hip.h -- C ABI of libblinkyhip.so: the MI355X (gfx950) implementation of
 * Blinky's globe->screen warp (lensmap build + per-frame lensmap apply).
 *
 * The reference has no FFI for this path: engine/NQ/fisheye.c is one C
 * translation unit of static functions (engine/include/fisheye.h:4-9 is its
 * whole public face).  This header is the seam a maintainer binds instead of
 * those statics; every entry point names the reference code it replaces
 * (paths relative to /root/reference/engine).  Plain C types only; the host
 * side (blinky_amd/host/fisheye_hip.c) stays C, exactly like the reference.
 *
 * Conventions
 *   - every int-returning function returns BK_OK (0) or a negative BK_E_* code;
 *     bk_last_error() gives the message (the reference's Con_Printf text where
 *     one exists).  Nothing ever exit()s (the reference does on OOM, fisheye.c:723-726).
 *   - a lensmap entry is offset = ptr - globe.pixels = plate*ps*ps + py*ps + px
 *     (GLOBEPIXEL, fisheye.c:349) as uint32, BK_NULL_OFFSET for a NULL pointer;
 *     tints are the bytes of lens.pixel_tints (255 = none, fisheye.c:432-449).
 *   - all device work is enqueued on the context's HIP stream (bk_set_stream);
 *     host-pointer entry points are synchronous with respect to their buffers.
 *   - there is no CPU fallback: without a usable GPU bk_create fails.
 */
#ifndef BLINKY_HIP_H
#define BLINKY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BK_MAX_PLATES   6             /* MAX_PLATES, fisheye.c:352 */
#define BK_NULL_OFFSET  0xFFFFFFFFu
#define BK_DEVICE_NONE  (-2)          /* bk_create: host-only context (scripts, zoom, code generation) */

enum {
    BK_PENDING = 1,        /* bk_build with bk_set_async_compile: the lens is still compiling, call again later */
    BK_OK = 0,
    BK_E_INVALID = -1,     /* bad argument / call order */
    BK_E_HIP = -2,         /* HIP runtime or hiprtc failure (message has the detail) */
    BK_E_SCRIPT = -3,      /* lens / globe script failed to load, run or compile */
    BK_E_ZOOM = -4,        /* calc_zoom() failure (fisheye.c:1293-1386) */
    BK_E_NOMEM = -5,
    BK_E_STATE = -6        /* no valid lens/globe/lensmap for the requested operation */
};

enum { BK_MAP_NONE = 0, BK_MAP_INVERSE = 1, BK_MAP_FORWARD = 2 };                 /* fisheye.c:391 */
enum { BK_ZOOM_NONE = 0, BK_ZOOM_FOV, BK_ZOOM_VFOV, BK_ZOOM_COVER, BK_ZOOM_CONTAIN }; /* fisheye.c:457 */

typedef struct bk_ctx bk_ctx;

/* One globe plate as LUA_load_globe leaves it (fisheye.c:353-361, 1796-1869). */
typedef struct {
    float forward[3], right[3], up[3];
    float fov;      /* radians */
    float dist;     /* 0.5 / tan(fov/2) */
} bk_plate;

/* What LUA_load_lens reads back from the script's globals (fisheye.c:1684-1747). */
typedef struct {
    int    map_type;             /* BK_MAP_* after the `map` override */
    int    has_inverse, has_forward;
    int    max_fov, max_vfov;    /* 0 when absent */
    double lens_width, lens_height;   /* 0 when absent */
    char   onload[128];          /* "" when absent or not a string (fisheye.c:1087-1102) */
} bk_lens_info;

/* ---- lifecycle -------------------------------------------------------------------
 * replaces: the three malloc blocks and static state of fisheye.c (306-528, 712-727),
 * init_lua / lua_close (fisheye.c:1222-1265, 678-681). */
bk_ctx     *bk_create(int device);            /* -1: current HIP device; BK_DEVICE_NONE: no device at all */
void        bk_destroy(bk_ctx *ctx);
const char *bk_last_error(const bk_ctx *ctx); /* ctx may be NULL: error of the last failed bk_create */
int         bk_set_stream(bk_ctx *ctx, void *hip_stream);   /* hipStream_t; NULL = default stream */
int         bk_synchronize(bk_ctx *ctx);

/* ---- scripts (the Lua callback surface) -----------------------------------------
 * replaces: LUA_load_globe (fisheye.c:1752-1875) and LUA_load_lens (1659-1750);
 * `src` is the text of <basedir>/lua-scripts/{globes,lenses}/<name>.lua.
 * One interpreter state lives in the context, so script globals leak between
 * loads exactly as in the reference (only the names of fisheye.c:1880-1903 are cleared).
 * The language: what runs while a script LOADS (the chunk, what it calls) is Lua 5.2 (closures, varargs, metatables, goto, coroutines,
 * the string library with patterns, table.*, math.*, bit32.*, os.*, io.*, pcall / xpcall / error, load / dofile / require with
 * package.path / package.preload relative to the working directory; r6: coroutines, the rest of os / io, _G as a proxy of the globals;
 * not provided: string.dump, os.exit - an error the script can see -, and of the debug library what needs a register stack
 * (getlocal / setlocal answer nil, hooks are accepted and never fire; traceback, getinfo, get/setmetatable, get/setupvalue,
 * upvalueid / upvaluejoin and getregistry work); a coroutine created
 * while loading is nil inside the per-pixel callbacks: the build evaluates them on copies of the script state, and a coroutine is a
 * native stack of the original).  What the per-pixel CALLBACKS (lens_inverse, lens_forward,
 * globe_plate and everything they call) may use is narrower - they become GPU code at bk_build: numbers, booleans, nil, string
 * constants, local tables (arrays, records {x = ..}, matrices {{..}, {..}}), constant tables of the chunk, every control structure but goto, the math library, functions
 * defined inside a callback, functions and constant tables passed as arguments, methods of constant objects, varargs.  A script whose
 * callbacks go beyond that (recursion, tables made at run time, functions as values, strings) is NOT refused: bk_build evaluates
 * them with the library's own interpreter on the host instead - seconds instead of milliseconds at 4K - and bk_last_build_path
 * names the construct that forced it (fisheye.c:1551, 1597, 1640 lua_call whatever the script defines). */
int bk_load_globe(bk_ctx *ctx, const char *src, size_t len, const char *chunkname);
int bk_load_lens(bk_ctx *ctx, const char *src, size_t len, const char *chunkname);
int bk_clear_lens(bk_ctx *ctx);    /* lens.valid = false ("not a valid lens", fisheye.c:1080-1083) */
int bk_clear_globe(bk_ctx *ctx);   /* globe.valid = false (fisheye.c:1157-1160) */
int bk_get_lens_info(const bk_ctx *ctx, bk_lens_info *out);
int bk_get_globe(const bk_ctx *ctx, bk_plate plates[BK_MAX_PLATES], int *numplates);
/* bypass the script for the globe (plates already in LUA_load_globe's float form) */
int bk_set_globe_plates(bk_ctx *ctx, const bk_plate *plates, int numplates);

/* ---- geometry / parameters ------------------------------------------------------
 * bk_resize  replaces F_RenderView's size-change block (fisheye.c:704-727): ps = min(W,H),
 *            (re)allocates globe[6*ps*ps], lensmap[W*H], tints[W*H] in HBM.
 * bk_set_rows restricts this context to output rows [row0,row1) (multi-GPU stripes);
 *            the lensmap, tints and frames it holds cover only those rows.
 * bk_set_frames sizes the resident globe ring: nframes globes of 6*ps*ps bytes. */
int bk_resize(bk_ctx *ctx, int width, int height);
int bk_set_rows(bk_ctx *ctx, int row0, int row1);
int bk_set_frames(bk_ctx *ctx, int nframes);
int bk_set_zoom(bk_ctx *ctx, int zoom_type, int fov_degrees);                       /* cmd_fov/vfov/cover/contain, fisheye.c:955-965,1032-1058 */
int bk_set_rubixgrid(bk_ctx *ctx, int numcells, double cell_size, double pad_size); /* cmd_rubixgrid, fisheye.c:939-953 */

/* ---- lensmap build ----------------------------------------------------------------
 * replaces: create_lensmap (fisheye.c:2367-2397) = calc_zoom + resume_lensmap_inverse
 * (2084-2124) or resume_lensmap_forward (2126-2217), run to completion on the GPU,
 * including the NULL/255 clears of F_RenderView (731-732).
 * display_out (nullable) receives globe.plates[i].display; scale_out lens.scale.
 * A lens_inverse that returns a MALFORMED RESULT (status -1, fisheye.c:1565-1584) ends the reference's scan at that pixel and keeps
 * what it had set by then (2113-2115; rows from the bottom up, pixels left to right): bk_build returns BK_E_SCRIPT with exactly
 * that table in place - the entries the reference's scan had not reached are NULL - and the display flags of what is left.
 * (Stripes: a context knows only its rows; bk_multi_build hands every stripe the group's first failing pixel, a bk_comm host
 * does the same with bk_last_build_bad_key (max over the ranks) and bk_truncate_build.)  Any other run-time failure (arithmetic
 * on nil, a runaway loop - Lua errors the reference's unprotected lua_call does not survive; a malformed lens_forward result)
 * returns BK_E_SCRIPT and leaves an EMPTY lensmap.
 * Script globals (and locals of the script's chunk) that lens_inverse / lens_forward / globe_plate ASSIGN are per-pixel state
 * on the GPU, initialised from their value after the chunk ran: the reference's result for scratch variables and for caches
 * keyed by one of the callback's parameters (`if y ~= lasty then maxx = ..; lasty = y end`: eckert4.lua) - every bundled script.
 * A script that really carries state from one pixel to the next (a counter, a running sum) would not behave as in the reference's
 * sequential scan (fisheye.c:2084-2124), so by DEFAULT (bk_set_sequential_build mode 1) an inverse-map lens whose callbacks read
 * a script global before assigning it outside such a keyed cache, or assign a chunk local at all (bk_lens_carries_state), is built
 * as ONE sequential scan on the host, in the reference's order, by the compiled host module (else the script interpreter):
 * seconds instead of milliseconds at 4K, the reference's result for any script.  (r6) A FORWARD-map lens that carries state is
 * scanned on the host too, by the interpreter, in the reference's own call order (fisheye.c:2126-2217: per plate the last row's
 * lower corners, then row by row from the last up the upper corners left to right and the texels' globe_plate calls).
 * mode 2 = every lens; 0 = never (the GPU build whatever the script does, unless the emitter declines it: bk_last_build_path).
 * Memory: a FORWARD-map build (lenses with lens_forward only) keeps its scratch - screen coordinates of every texel corner and two
 * key planes, about 10 bytes per plate texel + 8 per pixel: 320 MB at 3840x2160 on a cube globe - in the context from one build to
 * the next (allocating it cost more than the build's kernels); it is released when the context builds an inverse map or is destroyed. */
int bk_build(bk_ctx *ctx, int display_out[BK_MAX_PLATES], double *scale_out);
/* how the last bk_build evaluated the lens callbacks: 0 = the GPU kernels; 1 = the interpreter on the host's worker pool (the
 * emitter declined a construct and the callbacks carry no state: every worker has its own copy of the script state); 2 = ONE
 * sequential scan on the host in the reference's order (state carried from call to call, or bk_set_sequential_build 2) - inverse
 * maps: fisheye.c:2084-2124; forward maps: 2126-2217, the corners' and texels' calls in the reference's own sequence.  `why`
 * (nullable) receives the construct the emitter declined and the state finding. */
int bk_last_build_path(const bk_ctx *ctx, char *why /* nullable */, size_t cap);
int bk_set_sequential_build(bk_ctx *ctx, int mode);
int bk_lens_carries_state(bk_ctx *ctx, char *global_name /* nullable */, size_t cap);
/* after a bk_build that returned BK_E_SCRIPT for a malformed result: 1 + scan key (ly * W + (W - 1 - lx)) of the first failing
 * pixel in the reference's scan order, 0 if none; bk_truncate_build NULLs everything the reference's scan would not have reached
 * before the pixel with that key (a no-op for key 0) and recounts display_out (nullable) */
unsigned int bk_last_build_bad_key(const bk_ctx *ctx);
int bk_truncate_build(bk_ctx *ctx, unsigned int bad_key, int display_out[BK_MAX_PLATES]);
int bk_calc_zoom(bk_ctx *ctx, double *scale_out);                                   /* calc_zoom only */
/* Lens modules are compiled with hiprtc on first use (0.2-1.1 s per lens) and kept (a) in the process and (b) on disk:
 * bk_set_cache_dir(dir) / $BLINKY_HIP_CACHE / $XDG_CACHE_HOME/blinky_hip / $HOME/.cache/blinky_hip, in that order
 * ("" or "off" disables the disk cache); a code object is keyed by generated source, embedded headers, GPU arch and
 * library version.  With bk_set_async_compile(ctx, 1) a bk_build that would have to wait for hiprtc starts the
 * compilation on another thread, returns BK_PENDING and leaves the previous lensmap (and display flags) in place,
 * so a render loop keeps drawing - the reference's time-sliced builder never stalls the game either
 * (fisheye.c:2084-2217) - and calls bk_build again on a later frame. */
int bk_set_cache_dir(const char *dir);        /* process-wide; NULL returns to the environment / default location */
int bk_set_async_compile(bk_ctx *ctx, int on);
/* Exactness bookkeeping of the last bk_build.  The kernels evaluate the scripts' transcendentals with a portable
 * libm; the reference's Lua VM calls the platform's.  Every value on the device carries a bound on that
 * discrepancy, and each pixel / texel corner whose DISCRETE outcome (a float narrowing, a comparison, a floor)
 * could depend on it is flagged, re-derived by the host interpreter on the platform libm - the evaluator
 * bk_calc_zoom and bk_load_globe already use - and patched before bk_build returns.
 * flagged = entries re-derived on the host, changed = entries whose value differed from the device's. */
int bk_last_build_fixups(const bk_ctx *ctx, int *flagged, int *changed);
/* direct access to the table (lens.pixels / lens.pixel_tints of the owned rows) */
int bk_set_lensmap(bk_ctx *ctx, const uint32_t *offsets, const uint8_t *tints);
int bk_read_lensmap(bk_ctx *ctx, uint32_t *offsets, uint8_t *tints);

/* ---- globe plates -------------------------------------------------------------------
 * bk_upload_plate replaces render_plate's row memcpy loop (fisheye.c:2441-2449):
 * ps rows of ps bytes from src (pitch src_pitch) into plate `plate` of globe `frame`. */
int   bk_upload_plate(bk_ctx *ctx, int frame, int plate, const uint8_t *src, int src_pitch);
/* the same, pipelined: the rows are copied into a pinned staging buffer (render_plate's memcpy) and the DMA and the
 * re-tiling are only enqueued on the context's stream, so that the caller renders the next plate meanwhile; src may be
 * reused as soon as the call returns.  Three staging slots: the call waits only for the upload three calls ago. */
int   bk_upload_plate_async(bk_ctx *ctx, int frame, int plate, const uint8_t *src, int src_pitch);
/* the way back (the plate copy cmd_saveglobe reads, fisheye.c:1396-1465): ps rows of ps bytes to dst_host */
int   bk_download_plate(bk_ctx *ctx, int frame, int plate, uint8_t *dst_host, int dst_pitch);
/* the plate image f_saveglobe encodes (WritePCXplate's pixel loop, fisheye.c:1438-1456): texel, or 0xFE where
 * the ray through the texel belongs to another plate (ray_to_plate_index / the globe's globe_plate) unless
 * with_margins.  Computed on the device; PCX packing is the caller's (blinky_amd/host/fisheye_hip.c). */
int   bk_save_plate(bk_ctx *ctx, int frame, int plate, int with_margins, uint8_t *dst_host, int dst_pitch);
/* Device layout of a globe frame: 6 plates of bk_globe_pitch() = round_up(ps,64) by bk_globe_rows() =
 * round_up(ps,8) texels, each stored as 16x8-texel tiles of 128 bytes (tiles row-major, rows of a tile
 * 16 bytes apart): a 128-byte line covers a compact patch, so the slanted footprints of the warp touch
 * about half as many lines as with row-major plates.  bk_globe_texel_offset gives the byte offset of a
 * texel inside a frame (0xFFFFFFFF if out of range) for callers that fill plates on the device themselves.
 * Lensmap entries cross this ABI in the reference layout (plate*ps*ps + py*ps + px); the library converts. */
void    *bk_globe_device_ptr(bk_ctx *ctx, int frame);   /* device address of globe `frame` (6*pitch*rows bytes) */
int      bk_globe_pitch(const bk_ctx *ctx);
int      bk_globe_rows(const bk_ctx *ctx);
uint32_t bk_globe_texel_offset(const bk_ctx *ctx, int plate, int px, int py);
/* synthetic plates: the SURVEY.md 8(d) LCG stream generated on the device */
int   bk_fill_plate_lcg(bk_ctx *ctx, int frame, int plate, uint32_t seed_frame);

/* ---- lensmap apply ------------------------------------------------------------------
 * replaces: render_lensmap (fisheye.c:2406-2424).  Unmapped pixels leave dst untouched.
 * pal (nullable unless rubix_on): the 6 tint LUTs of create_palmap (fisheye.c:857-908).
 * bk_apply        dst is host memory (vid.buffer), pitch vid.rowbytes, origin scr_vrect.{x,y};
 *                 it is synchronous.  Only the owned rows are touched.
 * bk_apply_device dst is device memory holding nframes frames of frame_stride bytes each;
 *                 frame f is warped from globe (frame0+f) % nframes_resident.  Asynchronous on the stream.
 *                 dst addresses pixel (0,0) of frame 0 of the WHOLE view; only the owned rows
 *                 [row0,row1) (bk_set_rows) are written, at dst + f*frame_stride + y*pitch + x.  A caller that
 *                 keeps just its stripe (frame_stride = rows*pitch) passes stripe_base - row0*pitch. */
int bk_apply(bk_ctx *ctx, int frame, uint8_t *dst, int dst_pitch, int x0, int y0,
             int rubix_on, const uint8_t pal[BK_MAX_PLATES][256]);
/* bk_apply in two halves: begin enqueues the warp (and the frame's way to pinned host memory) and returns; end waits and
 * delivers into dst exactly as bk_apply does.  Lets a host overlap the GPU's part with its own work, and is how
 * bk_multi_apply keeps several devices busy from one thread. */
int bk_apply_begin(bk_ctx *ctx, int frame, int rubix_on, const uint8_t pal[BK_MAX_PLATES][256]);
int bk_apply_end(bk_ctx *ctx, uint8_t *dst, int dst_pitch, int x0, int y0);
int bk_apply_device(bk_ctx *ctx, int frame0, int nframes, void *dst_dev, int dst_pitch,
                    size_t frame_stride, int x0, int y0, int rubix_on,
                    const uint8_t pal[BK_MAX_PLATES][256]);

/* ---- resident single-frame apply ------------------------------------------------------------------------------------
 * replaces: the per-frame call of render_lensmap (fisheye.c:803 -> 2406-2424) for hosts whose globes stay in device memory.
 * One bk_apply_device launch per frame re-reads the whole block map of the staged apply (2 bytes per pixel and more: half of a
 * 4K frame's traffic) because nothing stays in the caches across a kernel boundary.  bk_apply_resident_begin launches a kernel
 * that STAYS on the device with the block map in its registers; bk_apply_resident_submit hands it one frame - warp globe `frame`
 * into dst_dev (same addressing as bk_apply_device: pixel (0,0) of the whole view, only the owned rows are written, unmapped
 * pixels untouched) - by writing a command into pinned host memory, and returns at once with a ticket; submissions pipeline
 * (up to 32 in flight).  bk_apply_resident_wait(ticket) returns when that frame - and every earlier one - is complete IN
 * MEMORY (any stream, DMA or peer may read it); *gpu_us (nullable) = device time from the kernel seeing the command to the
 * frame's completion.  bk_apply_resident_end makes the kernel leave.
 * Rules: the globe frame and dst of a submission must not be written by anyone else between submit and wait, and whatever
 * produced that globe frame must have COMPLETED (host-synchronised) before the submit - the kernel is not ordered with any
 * stream.  rubix_on / pal are fixed for the session.  The kernel fills the GPU: every other device entry point of this context
 * ends the session first (the next submit starts it again transparently), kernels of other contexts / libraries queue behind
 * it, and a device-wide synchronise blocks until it leaves - which it does by itself after idle_ms (<= 0: 200 ms) without a
 * submission; a later submit relaunches it.  The staged apply variant only. */
int bk_apply_resident_begin(bk_ctx *ctx, int rubix_on, const uint8_t pal[BK_MAX_PLATES][256], double idle_ms);
int bk_apply_resident_submit(bk_ctx *ctx, int frame, void *dst_dev, int dst_pitch, int x0, int y0, uint64_t *ticket);
/* a batch: frame f is warped from globe (frame0 + f) % resident globes into dst_dev + f * frame_stride (bk_apply_device's
 * addressing), one command per frame; blocks only while the command ring is full; *last_ticket = the last frame's */
int bk_apply_resident_submit_batch(bk_ctx *ctx, int frame0, int nframes, void *dst_dev, int dst_pitch, size_t frame_stride,
                                   int x0, int y0, uint64_t *last_ticket);
int bk_apply_resident_wait(bk_ctx *ctx, uint64_t ticket, double *gpu_us);
int bk_apply_resident_end(bk_ctx *ctx);
/* out = {kernel on the device, worker workgroups, blocks per workgroup held in registers, chunk offsets per thread and block,
 * block height, workgroups per CU, kernel launches so far, submissions not yet known complete; of the last session that was ended:
 * min / median / max over its workgroups of the frames a workgroup ran on into the next one without draining; 0} */
int bk_apply_resident_info(bk_ctx *ctx, int out[12]);
/* (r5) The resident kernel beside other work, and under the drop-in calls.
 * bk_set_resident_share: how much of every CU the resident kernel may occupy.  A CU holds up to eight of its workgroups (fewer for the
 * forms that need more registers or LDS); `reserve_slots_per_cu` of those places stay free on every CU - kernels of this context, of
 * other contexts and of other libraries are scheduled there while the kernel is resident, provided they fit what a place leaves
 * (a quarter to an eighth of a CU's registers and LDS) - and the rest is split evenly between `parts` contexts of the same device, of
 * which this one is number `part`: stripe contexts on a one-GPU box each keep their own resident kernel.  A form that fits a CU no more
 * often than the reserve asks to leave free (64 KiB of staging for a scrambled table: twice) keeps one place per context - the reserve
 * yields, such a workgroup leaves most of the CU to others anyway.  Default 0 / 1 / 0: the whole chip.  Takes effect at the next bk_apply_resident_begin / relaunch.  (Measured and not used: a CU-masked stream -
 * hipExtStreamCreateWithCUMask streams are blocking streams, so every null-stream operation of the process, PyTorch's default
 * stream included, then waits for the resident kernel to leave.)
 * bk_set_resident_apply(ctx, 1): the per-frame calls of the drop-in go through the resident kernel - bk_apply / bk_apply_begin .. _end
 * submit the frame as a command, and none of bk_apply* / bk_upload_plate* ends the session.  How plates and frames travel depends on the
 * share: with reserve_slots_per_cu >= 1 (recommended for a drop-in: fisheye_hip.c's default) this context's small kernels run beside
 * the resident kernel, so plates are re-tiled on the device and a fully mapped frame is copied straight into the caller's buffer, as
 * without a resident kernel; with 0 the kernel may hold every place of the chip, and bk_upload_plate / bk_upload_plate_async re-tile
 * the plate on the HOST (render_plate's row memcpy, fisheye.c:2441-2449, writes tiles instead of rows) and move it with one DMA, the
 * frame comes back through a pinned copy + host rows - about 1 ms more host time per 3840x2160 frame, no kernel launch at all per frame
 * (a session is begun by the first bk_apply after a build, with that call's
 * rubix flag and palette, and begun again when they change).  Everything else (bk_build, bk_resize, bk_apply_device ...) still ends it.
 * A process-wide requirement of the resident kernel: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues
 * (default 4), and a stream that shares a queue with the resident kernel waits until it idles out.  Loading this library sets the
 * variable to 16 if it is unset (the runtime reads it at its first call); BLINKY_HIP_KEEP_HW_QUEUES=1 prevents that, a value already set
 * is kept, and when the runtime was loaded first the first session prints one line on stderr saying so. */
int bk_set_resident_share(bk_ctx *ctx, int part, int parts, int reserve_slots_per_cu);
int bk_set_resident_apply(bk_ctx *ctx, int on);

/* ---- multi-GPU: row stripes + RCCL over xGMI ---------------------------------------------------------------
 * No counterpart in the reference (fisheye.c is single-threaded CPU code).  Every output pixel is independent, so
 * GPU r of N owns rows [H*r/N, H*(r+1)/N): it builds and keeps only that stripe of the lensmap (no exchange, ever),
 * holds a full replica of the globe, warps its stripe; ONE exchange step reassembles frames.  Stripe buffers are
 * tight: [nframes][rows_r][W] bytes; frame buffers are tight [slots][H][W] with `frame_stride` bytes between slots.
 * An exchange is asynchronous: it starts once the work queued on the context's stream so far (the warp of the stripe)
 * has finished and runs on the communicator's own stream, so that the NEXT batch's warp overlaps the stripes' travel;
 * `slot` (0..BK_COMM_SLOTS-1) names it for bk_comm_wait.  librccl is loaded on first use.
 *
 * bk_comm: ONE rank.  Ranks may be processes (one per GPU, e.g. under torchrun / mpirun: rank 0 calls
 * bk_comm_unique_id and ships the 128 bytes to the others by any means) or several contexts of one process. */
#define BK_COMM_ID_BYTES 128
#define BK_COMM_SLOTS 4     /* exchanges in flight that a caller can tell apart (double / quadruple buffering) */
typedef struct bk_comm bk_comm;
int         bk_comm_unique_id(uint8_t id[BK_COMM_ID_BYTES]);                      /* ncclGetUniqueId */
/* joins the communicator (ncclCommInitRank on the context's device; waits until all nranks joined, at most BLINKY_HIP_COMM_TIMEOUT
 * seconds - default 60 - after which it fails and bk_last_error(NULL) says that a rank never arrived) and restricts the
 * context to this rank's stripe (bk_set_rows); needs bk_resize first.  nranks == 1 needs no id. */
bk_comm    *bk_comm_create(bk_ctx *ctx, int nranks, int rank, const uint8_t id[BK_COMM_ID_BYTES]);
void        bk_comm_destroy(bk_comm *c);
const char *bk_comm_last_error(const bk_comm *c);                                  /* c may be NULL: last failed create */
int         bk_comm_stripe(const bk_comm *c, int rank, int *row0, int *row1);      /* any rank's rows */
int         bk_comm_restripe(bk_comm *c);                                          /* after a later bk_resize */
/* stripes of equal WORK instead of equal height (collective: every rank calls it after a bk_build): what every row costs
 * the apply - for the default (staged) apply the costs of the row's blocks in this rank's block map: globe lines staged,
 * mapped pixels, a constant per block; for the direct-gather variant its mapped pixels - is summed over the ranks
 * (ncclAllReduce), the rows are cut into shares of equal cost (multiples of 8 rows) and this rank's context gets its new
 * stripe (bk_set_rows).  Build again afterwards; bk_comm_stripe tells the new stripes.  For lenses that leave part of the
 * screen unmapped or sample the globe unevenly (hammer's ellipse: with equal heights the ranks that own the top and the
 * bottom of the screen have a fraction of the middle ranks' work).  Every rank must run the same apply variant. */
int         bk_comm_rebalance(bk_comm *c);
/* display[] |= every other rank's (which plates the WHOLE frame reads, fisheye.c:1976): ncclAllReduce(MAX); synchronous */
int         bk_comm_or_display(bk_comm *c, int display[BK_MAX_PLATES]);
/* every frame of the batch onto `root` (what a single display needs): grouped ncclSend / ncclRecv; frames_dev is
 * used on the root only and receives frame f at slot f */
int         bk_comm_gather(bk_comm *c, const void *stripe_dev, int nframes, int root, void *frames_dev, size_t frame_stride, int slot);
/* batches: frame f is reassembled on rank f % N at slot f / N (a gather whose root rotates): all N*(N-1) xGMI links
 * carry stripes at once and every GPU ends up with 1/N of the batch as whole frames */
int         bk_comm_exchange_rotating(bk_comm *c, const void *stripe_dev, int nframes, void *frames_dev, size_t frame_stride, int slot);
/* the context's stream waits - on the device, the host does not block - for the exchange last posted with `slot`:
 * call before warping into the stripe buffer, or reading the frames, that exchange used */
int         bk_comm_wait(bk_comm *c, int slot);
int         bk_comm_synchronize(bk_comm *c);         /* host waits for the context's stream and every posted exchange */

/* bk_multi: N stripe contexts + their communicators driven by ONE host thread - what a single-process C host
 * (blinky_amd/host/fisheye_hip.c) uses to spread F_RenderView's warp over the node's GPUs.  ncclCommInitAll; every
 * exchange is posted for all ranks inside one ncclGroup.  A device named twice (one-GPU box; RCCL refuses duplicates)
 * or BLINKY_HIP_COMM=copy selects device-to-device copies over the same schedule instead. */
typedef struct bk_multi bk_multi;
bk_multi   *bk_create_multi(int ndev, const int *devices);
void        bk_destroy_multi(bk_multi *m);
const char *bk_multi_last_error(const bk_multi *m);
int         bk_multi_size(const bk_multi *m);
bk_ctx     *bk_multi_ctx(bk_multi *m, int i);        /* stripe context i (introspection, per-context calls) */
bk_comm    *bk_multi_comm(bk_multi *m, int i);       /* after bk_multi_resize */
int         bk_multi_uses_rccl(const bk_multi *m);
int         bk_multi_lensmap_valid(const bk_multi *m);   /* 1 when every stripe context holds a lensmap built for its current rows */
/* the per-context calls applied to every stripe context (same meaning as their bk_* namesakes) */
int bk_multi_load_globe(bk_multi *m, const char *src, size_t len, const char *chunkname);
int bk_multi_load_lens(bk_multi *m, const char *src, size_t len, const char *chunkname);
int bk_multi_clear_lens(bk_multi *m);
int bk_multi_clear_globe(bk_multi *m);
int bk_multi_resize(bk_multi *m, int width, int height);          /* + stripes: context i owns rows [H*i/N, H*(i+1)/N) */
int bk_multi_rebalance(bk_multi *m, int *bounds_out /* [N+1], nullable */);   /* bk_comm_rebalance for the group; then bk_multi_build again */
int bk_multi_set_frames(bk_multi *m, int nframes);
int bk_multi_set_zoom(bk_multi *m, int zoom_type, int fov_degrees);
int bk_multi_set_rubixgrid(bk_multi *m, int numcells, double cell_size, double pad_size);
int bk_multi_upload_plate(bk_multi *m, int frame, int plate, const uint8_t *src, int src_pitch);   /* replica on every GPU */
int bk_multi_fill_plate_lcg(bk_multi *m, int frame, int plate, uint32_t seed_frame);
int bk_multi_synchronize(bk_multi *m);       /* host waits for every stream of every rank */
int bk_multi_wait(bk_multi *m, int slot);    /* bk_comm_wait on every rank */
/* all stripes built concurrently (one host thread per device); display_out = OR over the stripes */
int bk_multi_build(bk_multi *m, int display_out[BK_MAX_PLATES], double *scale_out);
/* host frame: every device warps its stripe and copies it straight into dst (N PCIe links side by side) */
/* bk_set_resident_apply on every stripe context; contexts that share a device (a device named twice) split the places of its CUs
 * between them (bk_set_resident_share) so that each keeps its own resident kernel */
int bk_multi_set_resident_apply(bk_multi *m, int on);
int bk_multi_apply(bk_multi *m, int frame, uint8_t *dst, int dst_pitch, int x0, int y0, int rubix_on,
                   const uint8_t pal[BK_MAX_PLATES][256]);
/* device frames: warp into per-device stripe buffers, then gather / rotating exchange as bk_comm_* */
int bk_multi_apply_stripes(bk_multi *m, int frame0, int nframes, void *const *stripes_dev, int rubix_on,
                           const uint8_t pal[BK_MAX_PLATES][256]);
int bk_multi_gather(bk_multi *m, void *const *stripes_dev, int nframes, int root, void *frames_dev, size_t frame_stride, int slot);
int bk_multi_exchange_rotating(bk_multi *m, void *const *stripes_dev, int nframes, void *const *frames_dev, size_t frame_stride, int slot);

/* plain device buffers for hosts that do not link HIP themselves (zero-filled; ordered on the context's stream) */
void *bk_dev_alloc(bk_ctx *ctx, size_t bytes);
void  bk_dev_free(bk_ctx *ctx, void *dev_ptr);
int   bk_dev_read(bk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* synchronous */

/* rubix palettes: create_palmap / find_closest_pal_index (fisheye.c:835-908); basepal = 768 bytes */
void bk_create_palmap(const uint8_t *basepal, uint8_t pal_out[BK_MAX_PLATES][256]);

/* ---- introspection ---------------------------------------------------------------------
 * (developer knobs, ablations, statistics and test hooks - the bk_debug_* entry points - are declared in
 *  include/blinky_hip_debug.h and exist only when the library is built with BK_DEBUG_API, the Makefile's default) */
int         bk_get_size(const bk_ctx *ctx, int *width, int *height, int *platesize, int *row0, int *row1);
const char *bk_version(void);
/* selects the apply kernel: 0 = direct gather, 2 = workgroup-cooperative LDS staging (default) */
int         bk_set_apply_variant(bk_ctx *ctx, int variant);
/* How the staged apply picks its block height (128x8 / 128x16 / 128x32 pixels) after a build: 1 (default) = the candidates
 * its cost model ranks within 20 % of the best are compiled and the caller's own first launch is TIMED on each (about
 * 0.7 ms more per lensmap at 3840x2160, for up to 17 % per frame where the model misses); 0 = the cost model alone. */
int         bk_set_blockmap_tuning(bk_ctx *ctx, int measured);
/* milliseconds of the last bk_build's device work (HIP events on the context stream) */
double      bk_last_build_ms(const bk_ctx *ctx);
/* The entries a build flags (results that hinge on the platform libm's last bits) are re-derived on the host: by the script
 * interpreter, or - when the machine has a C++ compiler - by the generated lens code itself compiled for the host and loaded
 * with dlopen (a fraction of a microsecond per entry instead of 4-10 us).  The compiler ($BLINKY_HIP_HOSTCXX, else c++ / g++ /
 * clang++ on $PATH, else ROCm's clang; "off" = none) runs on another thread, its output is cached beside the device code
 * objects (bk_set_cache_dir); until it is there the interpreter answers - same results either way.
 * bk_set_host_compile(0) switches the mechanism off for the process.  bk_host_module_ready: 1 when the current lens + globe's
 * module is loaded; with wait != 0 it is compiled now if need be. */
int         bk_set_host_compile(int on);
int         bk_host_module_ready(bk_ctx *ctx, int wait);
/* host-side script arithmetic (chunk execution, calc_zoom, globe loading, re-derivation of flagged pixels): 0 = the
 * platform libm, which is what the reference's Lua VM calls (default: scale, lens_width, plates bit-identical to the
 * reference on the same machine); 1 = the portable bkm.h functions the GPU kernels use.  (Values >= 2 select the
 * stand-in libms of the test suite and exist only in builds with the debug API: include/blinky_hip_debug.h.) */
int         bk_set_host_math(bk_ctx *ctx, int portable);
/* text the scripts print()ed since the context was created (the reference sends it to stdout) */
const char *bk_script_console(bk_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
