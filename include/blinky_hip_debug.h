/* blinky_hip_debug.h -- developer knobs, ablations, statistics and test hooks of libblinkyhip.so.
 *
 * NOT part of the drop-in boundary (include/blinky_hip.h).  These entry points exist only when the library is built
 * with -DBK_DEBUG_API=1 (the default of blinky_amd/csrc/Makefile; `make DEBUG_API=0` leaves them - and the test-only
 * stand-in libms of bk_set_host_math - out).  tests/, tools/ and bench.py's traffic model use them; an engine never does.
 * The library reads no test switches from the environment: what used to be BLINKY_HIP_NO_MEMCACHE /
 * BLINKY_HIP_TEST_LIBM_REL_LOG2 / BLINKY_HIP_DEBUG_MODEL is bk_debug_set_option below. */
#ifndef BLINKY_HIP_DEBUG_H
#define BLINKY_HIP_DEBUG_H

#include "blinky_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* process-wide test / developer switches (0 = off, the default):
 *   "no_memcache"        != 0: the in-process code-object cache is bypassed, so that the disk cache can be observed
 *   "libm_rel_log2"      n in 8..52: generate the build kernels with the assumed libm discrepancy 2^-n instead of 2^-50
 *                        (paired with bk_set_host_math(ctx, n): the flag -> host fix-up path on thousands of pixels)
 *   "print_model"        != 0: the block-shape cost model prints its inputs to stderr, one line per candidate
 *   "host_module"        1: flagged entries are re-derived by the compiled host module only (it is waited for; an error if
 *                        there is none), 2: by the script interpreter only; 0: whichever is there (the default)
 *   "forward_careful"    != 0: a forward-map build goes pass by pass (a stop and a look at the flag lists after each) and asks every
 *                        texel whether its ray selects its own plate - no tile is taken on bk_forward_tiles' word
 * returns BK_E_INVALID for an unknown name */
int         bk_debug_set_option(const char *name, int value);
/* bk_debug_module_from_cache: 1 if the current module was loaded from the disk cache (test hook; BLINKY_HIP_NO_MEMCACHE
 * in the environment bypasses the in-process cache so that the disk path can be observed). */
int         bk_debug_module_from_cache(const bk_ctx *ctx);
/* bk_debug_forward_tiles: after a forward-map build on the device, how many of its 16 x 16 texel tiles the quad pass took as wholly
 * inside their plate's own region (bk_forward_tiles) and how many tiles there are; taken = -1 when the last build did not use the
 * shortcut (an inverse map, a globe_plate script, "forward_careful", a build that went pass by pass). */
int         bk_debug_forward_tiles(bk_ctx *ctx, int *taken, int *total);
/* developer only: timing ablations of the staged apply (2 no globe loads, 4 no stores, 8 no load
 * pipelining; 16 row-major block walk, 32 persistent form always, 64 XCD bands of equal block count instead of equal
 * cost, 128 non-temporal globe loads, 256 LDS-DMA staging (global_load_lds) in single-frame launches of the one-block form,
 * 512 no automatic choice of 128 / 256 for single-frame launches, 2048 __syncthreads() instead of the raw LDS barriers, 4096 never
 * the six-chunks-per-thread register plan of the strided walk - results stay exact for 8..4096); results are wrong while
 * bits 2/4 are set.  0 restores normal operation. */
int         bk_debug_set_ablation(bk_ctx *ctx, int bits);
/* staged apply statistics of the current lensmap: out = {blocks, blocks on the direct-gather fallback,
 * empty blocks, bytes of one LDS staging buffer, 128000 + block height in pixels, 128-byte lines staged per frame} */
int         bk_debug_tile_stats(bk_ctx *ctx, int out[6]);
/* What the staged apply has to move for the current lensmap (bench.py's compulsory-traffic roofline):
 * out = {distinct 128-byte globe lines the owned rows read per frame, lines staged per frame summed over blocks,
 * 16-byte chunks staged per frame, bytes of block map read per block visit summed over blocks, mapped pixels
 * (= bytes stored per frame), frames served per block visit, blocks, block height in pixels} */
int         bk_debug_traffic_model(bk_ctx *ctx, uint64_t out[8]);
/* How the staged apply splits the current lensmap's blocks over the 8 XCDs: out[0..8] = where each XCD's band starts in
 * the list of live (non-empty) blocks in walk order - bands of equal cost, not of equal block count - with out[8] = live
 * blocks; out[9] = 1 if bands of equal block count would be more than 10 % uneven (single-frame launches then take the
 * balanced workgroup -> block map too); out[10 + k] = cost of band k (128-byte lines staged + pixel and block terms) */
int         bk_debug_band_balance(bk_ctx *ctx, uint32_t out[18]);
/* calibration (bench.py): GB/s of a plain streaming kernel that reads `bytes` with 16-byte loads and writes `writes` of
 * every `period` KiB of it back with non-temporal stores - what this memory system gives a kernel with that read : write
 * ratio and nothing else to do (best of 5 passes; allocates and frees 2 x bytes) */
int         bk_debug_stream_mix(bk_ctx *ctx, size_t bytes, int period, int writes, double *gbps);
/* the HOST build paths (scripts the GPU emitter declines, state-carrying scripts; blinky_hip.h at bk_last_build_path) on any context, a
 * device-less one included: mode 1 = worker pool, 2 = one sequential scan in the reference's order, 0 = what bk_build would choose.
 * offsets [rows * W] come back in the reference's layout (plate * ps * ps + py * ps + px, 0xFFFFFFFF = NULL); returns what bk_build returns */
int         bk_debug_host_build(bk_ctx *ctx, int mode, uint32_t *offsets, uint8_t *tints, int display_out[BK_MAX_PLATES], double *scale_out);
/* FNV-1a-64 of a HOST buffer - the hash tests/golden/lensmaps.json pins frames with (bench.py --check) */
int         bk_debug_fnv1a64(const void *host, size_t bytes, uint64_t *out);
/* the resident apply one frame at a time, on the C host's clock (what fisheye_hip.c pays, without a binding in between): `frames`
 * times bk_apply_resident_submit + bk_apply_resident_wait of globe (7 i) % globes into dst_dev; medians of the host wall clock and of
 * the device's own figure (command seen -> frame complete), microseconds.  Needs a session (bk_apply_resident_begin). */
int         bk_debug_resident_latency(bk_ctx *ctx, int frames, void *dst_dev, int dst_pitch, int globes, double *host_us, double *device_us);
/* which XCD (HW_REG_XCC_ID) each workgroup of a 1-D launch of `nworkgroups` runs on: the apply kernel's screen bands
 * assume workgroup b -> XCD b % 8 (locality only; a test checks the assumption on the box it runs on) */
int         bk_debug_xcd_of_workgroups(bk_ctx *ctx, int *out, int nworkgroups);
/* developer knobs: 0 = block height by the cost model, 1 / 2 / 4 = force 128x8 / 128x16 / 128x32 pixel blocks;
 * 100+n = n workgroups per CU in the persistent grid; 300+n = frames per block visit; 400+n = staging buffer KiB;
 * 600 / 601+n = default / n as the constant term of a block's cost in the band balance */
int         bk_debug_set_tile_shape(bk_ctx *ctx, int lw);
/* the HIP translation unit generated for the current lens + globe scripts (needed = strlen+1);
 * compile != 0 also runs it through hiprtc (works on a BK_DEVICE_NONE context) */
int         bk_debug_kernel_source(bk_ctx *ctx, char *buf, size_t cap, size_t *needed, int compile);
/* test hook: the kernel-argument block (BkBuildParams, blinky_amd/csrc/bk_build_params.h) bk_build would launch the
 * current lens + globe with; works on a BK_DEVICE_NONE context (tests/hostemu runs the generated code on the host) */
int         bk_debug_build_params(bk_ctx *ctx, void *out, size_t cap, size_t *needed);
/* test hook: the host re-evaluation bk_build applies to the pixels it flags (bk_last_build_fixups), over any pixel
 * indices (row-major inside the owned rows); offsets in the reference layout.  Works without a device. */
int         bk_debug_host_entries(bk_ctx *ctx, const uint32_t *ids, size_t n, uint32_t *offsets, uint8_t *tints);
/* the same for the texel corners of the forward build (corner number plate * (ps+1)^2 + j * (ps+1) + i): screen x, y, and
 * whether lens_forward gave a position */
int         bk_debug_host_corners(bk_ctx *ctx, const uint32_t *ids, size_t n, int32_t *sx, int32_t *sy, uint8_t *ok);
/* evaluate a callback with the HOST interpreter, for diagnosing a script: which 0 = lens_inverse(x,y),
 * 1 = lens_forward(x,y,z), 2 = globe_plate(x,y,z); *nout = number of results, -1 for a single nil */
int         bk_debug_eval(bk_ctx *ctx, int which, const double *args, int nargs, double out[8], int *nout);
/* the same on the DEVICE (the generated code), over n argument tuples of nargs doubles: out gets 8
 * doubles per tuple, nout the result count (-1 = a single nil, <= -100 = runtime error bits) */
int         bk_debug_eval_device(bk_ctx *ctx, int which, const double *args, int nargs, int n, double *out, int *nout);
/* where the last bk_build's time went: out = {bk_last_build_ms (HIP events around the whole build: kernels + host fix-up),
 * wall ms of the host re-evaluation of the flagged entries inside it, flagged entries, worker threads of the fix-up pool,
 * wall ms of the inverse kernel launch(es) + read-back and sorting of the flag list, kernel re-runs because the list grew
 * (+ 1000 when the compiled host module did the re-evaluation)} */
int         bk_debug_build_breakdown(const bk_ctx *ctx, double out[6]);
/* bk_set_host_math(ctx, n >= 2), test mode: the host interpreter's libm becomes bkm.h with every inexact result moved
 * pseudo-randomly by up to 2^-n relative, standing in for "another libm" when the tests check the exactness flags
 * (tests/test_exactness_cpu.py); + 64 moves every result up by that amount instead, + 128 down. */
/* test hook, no device needed: the N+1 stripe bounds bk_comm_rebalance / bk_multi_rebalance derive from per-row costs.
 * W >= 1: row_cost holds the mapped pixels of each of H rows and is priced as the direct-gather apply's rows are
 * (+ W/32 per row); W == 0: row_cost is taken as it is (bk_debug_row_costs' output). */
int         bk_debug_stripe_bounds(const uint32_t *row_cost, int H, int W, int nranks, int *bounds_out);
/* what bk_comm_rebalance / bk_multi_rebalance sum over the ranks: the cost of every row of this context's stripe to its apply
 * variant (host uint32 [H], rows of other stripes 0); needs a built lensmap */
int         bk_debug_row_costs(bk_ctx *ctx, uint32_t *host_out);

#ifdef __cplusplus
}
#endif
#endif
