#!/usr/bin/env python3
"""bench.py -- Blinky warp hot path on MI355X: lensmap APPLY throughput (+ lensmap BUILD ms).

Workload (BASELINE.json metric): 3840x2160 output, cube globe (6 faces of 2160x2160), panini
lens, f_fov 180.  Synthetic data: LCG globe faces generated on the device;
the lensmap is built on the device from the bundled Lua scripts before the timed region.

A *step* = one pass of the hot path over one batch: `frames` frames warped by ONE bk_apply_device launch from
`frames` consecutive globes of a resident ring of `ring` distinct globes (default 64 = 1.8 GB), into one of four
rotating output buffers.  The ring advances every step, so a globe is re-read only after the whole ring - several
times the 256 MiB Infinity Cache - has gone by: the timed region runs against HBM, not against the last-level
cache (the figure for a ring that fits the cache is reported beside it as `roofline.warm_ring`).
With N > 1 ranks each rank owns a stripe of output rows (it builds and keeps only that stripe of the lensmap, holds
a full globe replica) and the step ends with the frames' reassembly: one grouped RCCL send/recv in which frame f's
stripes travel to rank f % N (blinky_amd.multigpu.exchange_rotating), overlapped with the next batch's warp.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N`; a world size that differs from --gpus is an error.

The timed region of K steps is repeated (--repeats; by default as many regions as it takes to keep the GPU busy for
--min-gpu-seconds, at least 31; each bracketed by barrier + synchronize, max over ranks); `value` is the MEDIAN region,
min / max are printed beside it.  `value_one_stream` is the same measurement with every launch on one stream - the figure
`roofline.kernel_ms_per_launch` (the kernel alone, HIP events) must be consistent with.
At N = 1 the line also carries `configs_extra` (BASELINE.json configs[1], 1920x1080 cube/stereographic, timed the same
way), `predicted_stripe_complete` (rank r's stripe for N = 2 / 4 / 8 timed on this GPU, max over ranks) and `build_ms` - the
metric's other half: bk_build at the headline's size as a call, for its own lens, quincuncial and two forward-map lenses.
Prints ONE JSON line (rank 0).  `value` = whole-job Mpixels/s = W*H*frames*steps / time.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 3840, 2160
GLOBE, LENS, ZOOM = "cube", "panini", "f_fov 180"
ALGO_BYTES_PER_PX = 6          # 4 B lensmap index + 1 B texel + 1 B store
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
DEFAULT_FRAMES = 64            # frames per step at every --gpus N (one workload for the whole scaling curve)
EXTRA_FRAMES = 16              # configs_extra / value_at_16_frames: the 16-frame launches rounds 1-4 reported
KERNEL_SOURCES = ("blinky_amd/csrc/bk_apply_coop.hip", "blinky_amd/csrc/bk_apply.hip", "blinky_amd/csrc/bk_build_params.h")


def kernel_source_hash():
    """identifies the apply kernels a PMC traffic record was measured on (profiles/apply_traffic.json)"""
    h = hashlib.sha1()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(frames_budget_s=6.0):
    """The reference CPU path timed on this host's cores (1 thread: the reference is single-threaded).
    Uses oracle/_ref (unmodified fisheye.c) when the prebuilt library travelled with the repo,
    otherwise the oracle's C restatement.  Bounded sample: the 4K lensmap build once plus
    ~frames_budget_s of render_lensmap() calls."""
    import ctypes as C
    import numpy as np
    import oracle_ffi as O
    if O.have_ref():
        t0 = time.time()
        lm, _ = O.ref_run(GLOBE, LENS, None, W, H, want_frame=False)        # build (create_lensmap) + one apply
        build_s = time.time() - t0
        ref = C.CDLL(O.REF_SO)
        ref.ref_time_apply.restype = C.c_double
        best = C.c_double()
        ref.ref_time_apply(2, C.byref(best))
        reps = max(3, int(frames_budget_s / (best.value * 1e-3)))
        total = ref.ref_time_apply(reps, C.byref(best))
        kind = "reference"
    else:
        t0 = time.time()
        lm = O.lensmap(GLOBE, LENS, None, W, H)
        build_s = time.time() - t0
        globe = O.lcg_globe(lm.ps, 6, 0)
        dst = np.zeros((H, W), np.uint8)
        t0 = time.time(); O.apply(lm.offsets, lm.tints, W, H, globe, dst); one = time.time() - t0
        reps = max(3, int(frames_budget_s / one))
        t0 = time.time()
        for _ in range(reps):
            O.apply(lm.offsets, lm.tints, W, H, globe, dst)
        total = time.time() - t0
        best = C.c_double(total / reps * 1e3)
        kind = "port"
    # next to the faithful single thread, the same gather restated row-parallel on every host core
    ncores = len(os.sched_getaffinity(0))
    mt_best, _ = O.time_apply_mt(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, 0), 12, ncores)
    return {"value": round(W * H * reps / total / 1e6, 1), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "allcores": {"value": round(W * H / mt_best / 1e6, 1), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
                         "sample": f"best of 12 x row-parallel ({ncores} pinned threads) restatement of render_lensmap, same lensmap and plates"},
            "sample": f"{reps} x render_lensmap at {W}x{H} {GLOBE}/{LENS} on LCG plates (best {best.value:.2f} ms/frame); "
                      f"lensmap build {build_s * 1e3:.0f} ms wall incl. setup",
            "build_ms": round(build_s * 1e3, 1)}


def trace(msg):
    """progress notes on stderr (BENCH_TRACE=1): which phase a crash or a hang belongs to"""
    if os.environ.get("BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def synthetic_basepal():
    """pal[i] = (i * 37) mod 256, i < 768 (the real gfx/palette.lmp is not in the tree)"""
    import numpy as np
    return ((np.arange(768) * 37) % 256).astype(np.uint8)


def compulsory_bytes(model, F):
    """what the staged apply has to move per F-frame launch: every mapped pixel stored once, every distinct globe line the
    lensmap touches read once per frame, the block map (headers + chunk lists + 16-bit pixel addresses) once per block visit"""
    visits = -(-F // int(model["frames_per_visit"])) if F >= 8 else 1
    if 8 <= F < 16:
        visits = 2
    return F * (model["mapped_pixels"] + 128 * model["unique_globe_lines"]) + visits * model["blockmap_bytes_per_visit"]


def resident_measure(torch, ctx, dst, W, rows, R, rubix=False, pal=None, frames=600, repeats=5, singles=100, orig_first_row=0):
    """The engine's real call is ONE frame per F_RenderView (fisheye.c:803): the resident apply (bk_apply_resident_*) - one kernel
    that stays on the device with the block map in registers, a frame is a command - over the same cold ring.  -> dict:
    pipelined = `frames` submissions back to back (bk_apply_resident_submit_batch), host wall clock / frames, median of
    `repeats`; one_at_a_time = submit, wait, submit ...: host wall clock and the device's own figure per frame."""
    # eight rotating output buffers: consecutive frames are in flight together (a stripe's resident kernel runs up to eight frames side
    # by side), and frames that share a destination would make their write-through stores collide on the same lines
    NB = 8
    obuf = torch.zeros((NB, rows, W), dtype=torch.uint8, device="cuda")
    obase = obuf.data_ptr() - (orig_first_row * W)
    torch.cuda.synchronize()
    ctx.resident_begin(rubix, pal, idle_ms=200)
    info = ctx.resident_info()
    ctx.resident_wait(ctx.resident_submit(dst, W, frame=0))
    frames = max(NB, frames // NB * NB)
    pipe = []
    for rep in range(repeats):
        t0 = time.perf_counter()
        for b in range(frames // NB):
            last = ctx.resident_submit_batch(obase, W, rows * W, frame0=(rep * frames + b * NB) % R, nframes=NB)
        ctx.resident_wait(last)
        pipe.append((time.perf_counter() - t0) / frames * 1e6)
    wall, dev = [], []
    for i in range(singles):
        t0 = time.perf_counter()
        t = ctx.resident_submit(dst, W, frame=(7 * i) % R)
        dev.append(ctx.resident_wait(t))
        wall.append((time.perf_counter() - t0) * 1e6)
    c_host, c_dev = ctx.resident_latency(dst, W, frames=max(50, singles), globes=R)      # the same on the C host's own clock (no ctypes call in between)
    ctx.resident_end()
    px = W * rows
    med = statistics.median(pipe)
    bpp = ALGO_BYTES_PER_PX + (1 if rubix else 0)
    return {"us": round(med, 3), "us_min": round(min(pipe), 3),
            "algorithmic_frac": round(bpp * px / (med * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "mpx_s": round(px / med, 1), "frames_per_measurement": frames,
            "one_at_a_time_host_us": round(c_host, 2), "one_at_a_time_device_us": round(c_dev, 2),
            "one_at_a_time_host_us_python": round(statistics.median(wall), 2),
            "one_at_a_time_is": "submit + wait per frame; host_us = the C host's clock around bk_apply_resident_submit + _wait (bk_debug_resident_latency), "
                                "device_us = command seen on the device -> frame complete in memory; _python = the same through the ctypes binding",
            "workgroups": info["workgroups"], "blocks_in_registers": info["blocks_in_registers"], "block_h": info["block_h"],
            "what": "bk_apply_resident_*: one frame per command to a kernel that stays on the device (block map in registers); "
                    "pipelined submissions, host wall clock per frame, cold ring"}


class OneGpuWorkload:
    """One lensmap + a resident ring of LCG globes on one GPU, timed the way the headline is: K-step regions between
    synchronizes (wall clock, steps alternating between two streams) and the kernel alone under HIP events."""

    def __init__(self, torch, blinky_amd, S, device_index, globe, lens, zoom, W, H, F, rows=None, ring_bytes=1.8e9, ring_max=64, rubix=False):
        self.torch, self.W, self.H, self.F = torch, W, H, F
        self.rubix = rubix
        self.pal = blinky_amd.ffi.create_palmap(synthetic_basepal()) if rubix else None
        self.ctx = ctx = blinky_amd.Context(device_index)
        self.stream = torch.cuda.current_stream()
        ctx.set_stream(self.stream.cuda_stream)
        S.configure(ctx, globe, lens, zoom, (W, H))
        ps = min(W, H)
        globe_bytes = 6 * ((ps + 63) // 64 * 64) * ((ps + 7) // 8 * 8)
        self.R = R = max(F, min(ring_max * 4, max(ring_max, int(ring_bytes // globe_bytes))))     # well past the 256 MiB Infinity Cache
        ctx.set_frames(R)
        ctx.resize(W, H)
        self.r0, self.r1 = rows if rows else (0, H)
        ctx.set_rows(self.r0, self.r1)
        self.rows = self.r1 - self.r0
        self.display, self.scale = ctx.build()
        self.build_ms = ctx.last_build_ms()
        self.tile_stats = ctx.tile_stats()
        for f in range(R):
            for p in range(6):
                ctx.fill_plate_lcg(f, p, f)
        self.out = [torch.zeros((F, self.rows, W), dtype=torch.uint8, device=torch.device("cuda", device_index)) for _ in range(4)]
        self.streams = [self.stream, torch.cuda.Stream(device=torch.device("cuda", device_index))]
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()

    def origin(self, t):
        return t.data_ptr() - self.r0 * self.W

    def launch(self, i, nframes=None, ring=None):
        nf = nframes or self.F
        self.ctx.apply_device(self.origin(self.out[i % 4]), self.W, self.rows * self.W, frame0=(i * nf) % (ring or self.R), nframes=nf,
                              rubix_on=self.rubix, pal=self.pal)

    def kernel_ms(self, nframes=None, launches=50, repeats=9, ring=None):
        """median / min / max over `repeats` of HIP events around `launches` back-to-back launches on one stream"""
        torch = self.torch
        self.ctx.set_stream(self.stream.cuda_stream)
        out = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            self.e0.record(self.stream)
            for i in range(launches):
                self.launch(i, nframes, ring)
            self.e1.record(self.stream)
            torch.cuda.synchronize()
            out.append(self.e0.elapsed_time(self.e1) / launches)
        return statistics.median(out), min(out), max(out)

    def job_seconds_per_step(self, steps=50, repeats=31, nstreams=2):
        """median wall-clock seconds per step over `repeats` regions of `steps` steps (synchronize on both sides)"""
        torch = self.torch
        out = []
        for k in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            handles = [st.cuda_stream for st in self.streams]
            for i in range(steps):
                self.ctx.set_stream(handles[i % nstreams])
                self.launch(k * steps + i)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps)
        self.ctx.set_stream(self.stream.cuda_stream)
        return statistics.median(out)

    def resident_us(self, frames=600):
        return resident_measure(self.torch, self.ctx, self.origin(self.out[0]), self.W, self.rows, self.R, self.rubix, self.pal, frames=frames, orig_first_row=self.r0)

    def close(self):
        self.torch.cuda.synchronize()
        self.ctx.close()
        self.out = None


def build_times(torch, blinky_amd, S, device_index, W, H, lenses=(("panini", "f_fov 180"), ("quincuncial", None), ("winkel2", None), ("polyconic", None))):
    """the metric's other half - lensmap BUILD ms at WxH on the cube globe - for the headline's lens, C3's quincuncial and two forward-map
    lenses (winkel2, polyconic: fisheye.c:2126-2338): bk_build as a call on the host's clock (best and median of 9 after the compile) and
    its device part under HIP events (bk_last_build_ms)."""
    import statistics
    rec = {}
    for lens, zoom in lenses:
        ctx = blinky_amd.Context(device_index)
        try:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            S.configure(ctx, "cube", lens, zoom, (W, H))
            ctx.build()                                   # hiprtc / code cache, scratch, host module
            ctx.build()
            wall, dev = [], []
            for _ in range(9):
                t0 = time.perf_counter()
                ctx.build()
                wall.append((time.perf_counter() - t0) * 1e3)
                dev.append(ctx.last_build_ms())
            flagged, changed = ctx.last_build_fixups()
            rec[lens] = {"call_ms_best": round(min(wall), 3), "call_ms_median": round(statistics.median(wall), 3), "device_ms": round(min(dev), 3),
                         "entries_rederived_on_host": flagged, "map": "forward" if ctx.lens_info().map_type == 2 else "inverse"}
        except Exception as e:      # noqa: BLE001
            rec[lens] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            ctx.close()
    return rec


def extra_config(torch, blinky_amd, S, device_index, name, globe, lens, zoom, W, H, F, steps, args=None, rubix=False, ring_max=64):
    """one more BASELINE.json configuration, driver-timed beside the headline (N = 1 only); with `args`, its HBM traffic is measured
    the way the headline's is (measure_traffic: two rocprofv3 --pmc child runs of the same launch).  rubix: the tint LUT path
    (fisheye.c:2416-2419), 7 algorithmic bytes per pixel."""
    trace(f"extra {name}: workload")
    wl = OneGpuWorkload(torch, blinky_amd, S, device_index, globe, lens, zoom, W, H, F, ring_max=ring_max, rubix=rubix)
    for i in range(5):
        wl.launch(i)
    trace("  kernel")
    k_med, k_min, k_max = wl.kernel_ms(launches=steps)
    trace("  single frames")
    s_med, _, _ = wl.kernel_ms(nframes=1, launches=max(steps, 30))
    trace("  job, two streams")
    job2 = wl.job_seconds_per_step(steps=steps, nstreams=2)
    trace("  job, one stream")
    job1 = wl.job_seconds_per_step(steps=steps, repeats=11, nstreams=1)
    trace("  resident")
    model = wl.ctx.traffic_model()
    # (rubix, r5: the tint plane is not read by the apply any more - a tinted block map carries the tint classes in its chunk list,
    #  `blockmap_bytes_per_visit` counts those entries - so the compulsory model is the plain one over the tinted map; the
    #  ALGORITHMIC count stays the reference's 7 B/px: fisheye.c:2416-2419 reads a tint byte per pixel and frame)
    comp = compulsory_bytes(model, F)
    bpp = ALGO_BYTES_PER_PX + (1 if rubix else 0)
    algo = bpp * W * H * F
    rec = {"name": name, "workload": f"{W}x{H} {globe}/{lens} {zoom or 'onload zoom'}{' rubix on' if rubix else ''}, {F} frames/step from a ring of {wl.R} distinct globes",
           "value": round(W * H * F / job2 / 1e6, 1), "value_one_stream": round(W * H * F / job1 / 1e6, 1), "unit": "Mpixels/s",
           "ms_per_step": round(job2 * 1e3, 5), "kernel_us_per_launch": round(k_med * 1e3, 3), "kernel_us_min": round(k_min * 1e3, 3),
           "kernel_us_max": round(k_max * 1e3, 3), "us_per_frame": round(k_med * 1e3 / F, 4),
           "algorithmic_bytes_per_px": bpp,
           "algorithmic_frac": round(algo / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "frac_compulsory": round(comp / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "compulsory_bytes_per_launch": int(comp),
           "single_frame": {"us": round(s_med * 1e3, 3),
                            "algorithmic_frac": round(bpp * W * H / (s_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "what": "one bk_apply_device launch per frame, HIP events around a train of them, cold ring"},
           "lensmap_build_ms": round(wl.build_ms, 3), "tile_stats": wl.tile_stats, "lens_scale": wl.scale}
    try:
        rec["single_frame"]["resident"] = wl.resident_us(frames=max(100, min(600, int(3000 / max(1.0, s_med * 1e3)))))
    except Exception as e:      # noqa: BLE001
        rec["single_frame"]["resident"] = {"error": f"{type(e).__name__}: {e}"}
    ring, block_h = wl.R, int(wl.tile_stats["tile_h"]) % 1000
    wl.close()
    trace("  traffic")
    if args is not None and not args.no_live_traffic:
        traffic, src = measure_traffic(args, F, ring, block_h, config=f"{W}x{H}:{globe}:{lens}:{zoom or ''}:{1 if rubix else 0}")
        rec["traffic"] = traffic
        rec["traffic_source"] = src
        rec["frac_traffic"] = round(traffic / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None
    return rec


def predicted_stripes(torch, blinky_amd, S, device_index, globe, lens, zoom, W, H, F, steps, t1_ms=None, resident_us_1=None):
    """What row-striping would give on N GPUs, measured on ONE: for N = 2 / 4 / 8 every rank's stripe is built and its F-frame launch
    timed here; a step of the N-GPU job lasts as long as its slowest rank's launch (stripe-complete throughput: nothing is exchanged).
    The stripes are the ones `bench.py --gpus N` uses: cut by the block map's row costs (bk_comm_rebalance), multiples of 8 rows.
    t1_ms None: the one-GPU launch of the same F frames is timed here too."""
    from blinky_amd import ffi
    trace(f"predicted stripes {globe}/{lens} x{F}")
    if t1_ms is None:
        wl = OneGpuWorkload(torch, blinky_amd, S, device_index, globe, lens, zoom, W, H, F, ring_max=32)
        for i in range(3):
            wl.launch(i)
        t1_ms = wl.kernel_ms(launches=max(10, steps // 2), repeats=5)[0]
        wl.close()
    full = blinky_amd.Context(device_index)
    S.configure(full, globe, lens, zoom, (W, H))
    full.build()
    cost = full.row_costs()
    full.close()
    out = {"frames_per_launch": F, "one_gpu_us_per_launch": round(t1_ms * 1e3, 2)}
    for n in (2, 4, 8):
        bounds = ffi.stripe_bounds_from_costs(cost, 0, n)
        per_rank, res_rank, res_err = [], [], None
        for r in range(n):
            wl = OneGpuWorkload(torch, blinky_amd, S, device_index, globe, lens, zoom, W, H, F, rows=(bounds[r], bounds[r + 1]), ring_max=32)
            for i in range(3):
                wl.launch(i)
            per_rank.append(wl.kernel_ms(launches=max(10, steps // 2), repeats=5)[0])
            if resident_us_1:
                # the per-frame display pipeline: the stripe through ITS resident kernel, frames submitted one by one (pipelined); a
                # stripe has fewer blocks than the chip has places, so the kernel runs several frames side by side (frame stride)
                try:
                    res_rank.append(wl.resident_us(frames=300)["us"])
                except Exception as e:      # noqa: BLE001
                    res_err = f"{type(e).__name__}: {e}"
            wl.close()
        worst = max(per_rank)
        out[str(n)] = {"slowest_rank_us_per_launch": round(worst * 1e3, 2), "fastest_rank_us_per_launch": round(min(per_rank) * 1e3, 2),
                       "stripe_rows": [bounds[r + 1] - bounds[r] for r in range(n)],
                       "stripe_complete_mpx_s": round(W * H * F / (worst * 1e-3) / 1e6, 1), "speedup_vs_1": round(t1_ms / worst, 3)}
        if resident_us_1:
            out.setdefault("frames1_resident", {"what": "one frame per command through each stripe's resident kernel (bk_apply_resident_*), pipelined "
                                                        "submissions; a step = one frame, as long as the slowest rank's",
                                                "one_gpu_us_per_frame": round(resident_us_1, 3)})
            out["frames1_resident"][str(n)] = ({"error": res_err} if res_err or len(res_rank) != n else
                                               {"slowest_rank_us_per_frame": round(max(res_rank), 3), "fastest_rank_us_per_frame": round(min(res_rank), 3),
                                                "stripe_complete_mpx_s": round(W * H / max(res_rank), 1), "speedup_vs_1": round(resident_us_1 / max(res_rank), 3)})
    return out


def traffic_child(args):
    """`bench.py --traffic-child`: what bench.py runs under `rocprofv3 --pmc <counter> --kernel-trace` to MEASURE the bytes the
    dominant kernel moves (measure_traffic): the same context, lensmap, ring and batch launch as the timed run, 3 + 20 launches
    on one stream, no timing, no JSON."""
    import torch
    import blinky_amd
    import scripts as S
    F, R = args.frames, max(args.ring, args.frames)
    W, H, globe, lens, zoom = globals()["W"], globals()["H"], GLOBE, LENS, ZOOM
    rubix, pal = False, None
    if args.child_config:                                   # "WxH:globe:lens:zoom[:rubix]" - configs_extra
        parts = args.child_config.split(":")
        size, globe, lens, zoom = parts[:4]
        W, H = [int(v) for v in size.split("x")]
        zoom = zoom or None
        rubix = len(parts) > 4 and parts[4] == "1"
        pal = blinky_amd.ffi.create_palmap(synthetic_basepal()) if rubix else None
    ctx = blinky_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_frames(R)
    S.configure(ctx, globe, lens, zoom, (W, H))
    if args.variant >= 0:
        ctx.set_apply_variant(args.variant)
    if args.shape:
        ctx.set_tile_shape(args.shape)                   # the block height the parent's (measured) tuning chose
    ctx.build()
    for f in range(R):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, f)
    out = [torch.zeros((F, H, W), dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(23 if W * H * F < 5e8 else 8):
        ctx.apply_device(out[i % 4].data_ptr(), W, H * W, frame0=(i * F) % R, nframes=F, rubix_on=rubix, pal=pal)
    torch.cuda.synchronize()


def measure_traffic(args, F, R, block_h, config=None):
    """HBM bytes per launch of the dominant kernel, measured NOW: two child runs of this script under rocprofv3, one --pmc
    counter each (FETCH_SIZE, WRITE_SIZE; counters only ever together with --kernel-trace), per MI355X_MICROARCH.md's HBM
    section: values in KiB; on gfx950 FETCH_SIZE tallies the 128-byte TCC->EA read requests at 64 bytes, so reads are doubled
    (cross-checked against TCC_EA0_RDREQ_* in profiles/).  Returns (bytes or None, provenance dict)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    vals, t0 = {}, time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bk_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--traffic-child",
               "--frames", str(F), "--ring", str(R), "--variant", str(args.variant), "--shape", str(block_h // 8)]
        if config:
            cmd += ["--child-config", config]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, {"error": f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode})", "stderr_tail": r.stderr[-300:]}
            con = sqlite3.connect(dbs[0])
            rows = list(con.execute("select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name = ? and "
                                    "kernel_name like '%apply_%' group by kernel_name, grid_size order by count(*) * grid_size desc", (counter,)))
            con.close()
            if not rows:
                return None, {"error": f"no apply kernel in the {counter} pass"}
            vals[counter] = rows[0]
        except Exception as e:      # noqa: BLE001
            return None, {"error": f"{type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    return int(round((2 * f[3] + w[3]) * 1024)), {
        "measured": "live, in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) around "
                    "`bench.py --traffic-child` (the same launch, 20 of them); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies its 128-byte read requests at 64 bytes)",
        "kernel": f[0][:80], "grid_work_items": f[1], "launches_profiled": f[2], "FETCH_SIZE_KiB_per_launch": round(f[3], 2),
        "WRITE_SIZE_KiB_per_launch": round(w[3], 2), "seconds": round(time.time() - t0, 1)}


COMPACT_LIMIT = 3072          # bytes: the driver keeps an ~8 KB tail of stdout and parses its LAST line (r5's 21 KB line was cut: parsed = null)


def compact_line(out):
    """The ONE JSON line the driver parses: the contract's keys and nothing that grows with the number of configurations.  The full
    record (configs_extra, predicted stripes, timed regions, the traffic model, every prose note) goes to bench_detail.json."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    rf, cfg = out["roofline"], out["config"]
    sf = rf.get("single_frame") or {}
    res = sf.get("resident") if isinstance(sf.get("resident"), dict) else {}
    line = pick(out, ("metric", "value", "value_one_stream", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_one_stream",
                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = pick(cfg, ("workload", "frames_per_step", "ring_globes", "parallelism", "streams"))
    line["roofline"] = pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "frac_compulsory", "kernel_ms_per_launch",
                                 "kernel_ms_min", "kernel_ms_max"))
    line["roofline"]["single_frame_us"] = sf.get("us")
    line["roofline"]["resident_us"] = res.get("us")
    line["roofline"]["resident_submit_wait_host_us"] = res.get("one_at_a_time_host_us")
    line["roofline"]["frac_note"] = ("frac = the contract's 6 B/px / kernel time / peak; the kernel reads a 2-byte address once per 8 frames instead of "
                                     "a 4-byte index per pixel and frame, so frac > 1 is not utilisation: frac_traffic (PMC bytes) is")
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "build_ms"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        if isinstance(cb.get("allcores"), dict):
            line["cpu_baseline"]["allcores"] = pick(cb["allcores"], ("value", "cores", "kind"))
    line.update(pick(out, ("lensmap_build_ms", "value_at_16_frames", "stripe_complete_mpx_s")))
    if isinstance(out.get("build_ms"), dict):         # lens -> bk_build as a call, best of 9 (ms); winkel2 / polyconic are forward maps
        line["build_ms"] = {k: v.get("call_ms_best", "error") for k, v in out["build_ms"].items() if isinstance(v, dict)}
    if out.get("n_gpus", 1) > 1:
        line.update(pick(out, ("assembled_on_rank0_mpx_s", "scaling_reference_mpx_s", "speedup_vs_scaling_reference")))
        line["exchange"] = pick(out.get("exchange") or {}, ("bound_mpx_s", "bound_all_on_rank0_mpx_s"))
        line["stripes"] = out.get("stripes")
        line["first_step_check_ok"] = (out.get("first_step_check") or {}).get("ok")
    extras = out.get("configs_extra")
    if extras:
        # one short entry per extra configuration: [kernel us per launch, frac_traffic (PMC), resident us per frame]
        line["extras"] = {str(c.get("name", "?")).split(" (")[0].rstrip(","): ([c.get("kernel_us_per_launch"), c.get("frac_traffic"),
                                                                               ((c.get("single_frame") or {}).get("resident") or {}).get("us")]
                                                                              if "error" not in c else "error") for c in extras}
        line["extras_are"] = "[kernel us/launch, frac_traffic, resident us/frame]"
    line["detail"] = out.get("detail_file")
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:                     # never let an optional key cost the headline
        for k in ("extras", "extras_are", "stripes", "exchange"):
            line.pop(k, None)
        line["roofline"].pop("frac_note", None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out, detail_path):
    """full record -> bench_detail.json (and gpurun_out/ when that scratch directory exists); compact line -> the LAST line of stdout"""
    paths = [detail_path or os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not detail_path:
        paths.append(os.path.join(ROOT, "gpurun_out", f"bench_detail_n{out.get('n_gpus', 1)}.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
            written = written or os.path.relpath(p, ROOT)
        except OSError as e:
            print(f"[bench] could not write {p}: {e}", file=sys.stderr, flush=True)
    out["detail_file"] = written
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` outside torchrun: one process per GPU, this node, RCCL"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a torchrun environment: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per step (batch warped by one launch); default 64 for EVERY --gpus N, so that the driver's 1/2/4/8 curve divides "
                         "like by like (rounds 1-4 ran 16 at N = 1: `value_at_16_frames` keeps that figure); a 270-row stripe of 16 frames is a "
                         "~10 us launch against a ~5.5 us back-to-back launch floor (launch-bound, not bandwidth-bound)")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep stripes of equal height instead of stripes of equal block-map cost")
    ap.add_argument("--ring", type=int, default=64, help="distinct resident globes the steps cycle through")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step timed region is measured (median reported); 0 = as many as keep the GPU busy "
                         "for --min-gpu-seconds, at least 31")
    ap.add_argument("--min-gpu-seconds", type=float, default=3.0,
                    help="with --repeats 0: total GPU-timed seconds the regions should add up to (a 50-step region is ~3 ms)")
    ap.add_argument("--no-extra", action="store_true", help="skip configs_extra (1080p C2) and predicted_stripe_complete")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two child runs under rocprofv3 --pmc); use the stamped record in profiles/ if it matches")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-config", default="", help=argparse.SUPPRESS)
    ap.add_argument("--shape", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps alternate between: the tail of a batch launch (its last, partly filled round of "
                         "workgroups) then overlaps the ramp of the next; 1 = every launch on one stream")
    ap.add_argument("--variant", type=int, default=-1, help="apply kernel variant (-1 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", default="", help="where the full record goes (default: bench_detail.json beside bench.py); stdout's last line is the compact one")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region, compare the FNV-1a-64 of every frame this rank holds with the reference's golden frames "
                         "(tests/golden/lensmaps.json, recorded from the unmodified fisheye.c)")
    ap.add_argument("--no-first-step-check", action="store_true",
                    help="N > 1: skip the check of the first step's reassembled frames against a full-frame warp (on by default: the first "
                         "time the exchange runs on a node is the time to find out)")
    args = ap.parse_args()
    if args.traffic_child:
        if args.frames <= 0:
            args.frames = DEFAULT_FRAMES
        return traffic_child(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a "
                 f"{world}-GPU number as a {args.gpus}-GPU one")

    import gc
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    # A timing harness in CPython switches the cyclic collector off (as timeit does): with torch loaded a full collection takes
    # 35-40 ms and a young one 1 ms, inside whatever launch train happens to be running - that was the "millisecond launch" of the
    # rubix line in rounds 4 and 5 (profiles/r06_rubix_outlier.txt).  Reference counting still frees everything that is not a cycle.
    gc.disable()

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BLINKY_BENCH_BACKEND=gloo + BLINKY_BENCH_ONE_GPU=1: developer smoke of the N > 1 control flow on a
        # single-GPU box (all ranks on cuda:0, stripes exchanged through host memory); never a measurement
        dist.init_process_group(os.environ.get("BLINKY_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"[bench] process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    host_exchange = world > 1 and dist.get_backend() == "gloo"
    if os.environ.get("BLINKY_BENCH_ONE_GPU") == "1":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"[bench] rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import blinky_amd
    import scripts as S
    from blinky_amd import multigpu

    if args.frames <= 0:
        args.frames = DEFAULT_FRAMES                # the same workload at every N
    F, R = args.frames, max(args.ring, args.frames)
    ctx = blinky_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_frames(R)
    S.configure(ctx, GLOBE, LENS, ZOOM, (W, H))
    bounds = multigpu.stripe_bounds(H, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    ctx.set_rows(r0, r1)
    # the stripe exchange behind the C ABI: bk_comm = librccl called from libblinkyhip (grouped ncclSend / ncclRecv on the
    # communicator's own stream); torch.distributed only ships the unique id.  The gloo developer smoke keeps host tensors.
    comm, comm_fallback = None, None
    if world > 1 and not host_exchange:
        # Should libblinkyhip's own communicator not come up on this node, say so and exchange through the process group's RCCL instead
        # of losing the measurement; every rank takes the same decision.  (bk_comm_create waits for the other ranks at most
        # BLINKY_HIP_COMM_TIMEOUT seconds - default 60 - and then fails with a message that says which rank was waiting.)
        try:
            comm = multigpu.rccl_comm(ctx, world, rank, dev)
            ok = 1 if comm.stripe(rank) == (r0, r1) else 0
            why = "" if ok else "stripe bounds differ"
        except Exception as e:      # noqa: BLE001
            comm, ok, why = None, 0, f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: bk_comm could not be created ({why}); using torch.distributed for the exchange", file=sys.stderr, flush=True)
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            reasons = [None] * world
            dist.all_gather_object(reasons, why)
            comm_fallback = "; ".join(f"rank {r}: {w}" for r, w in enumerate(reasons) if w)[:300] or "a rank could not create bk_comm"
    if args.variant >= 0:
        ctx.set_apply_variant(args.variant)

    # ---- lensmap build (each rank: its own stripe; no exchange) ----------------------------------
    t0 = time.time()
    ctx.build()                                   # includes hiprtc compilation of the lens (or the module cache)
    build_first_wall_ms = (time.time() - t0) * 1e3
    # stripes of equal WORK instead of equal height (bk_comm_rebalance: the block map's costs row by row, summed over the ranks by one
    # all-reduce; multiples of 8 rows): a lens that leaves part of the screen unmapped or samples the globe unevenly would otherwise
    # give the ranks that own the top and the bottom of the screen a fraction of the middle ranks' work
    rebalanced = False
    if world > 1 and not args.no_rebalance:
        try:
            if comm:
                comm.rebalance()
                bounds = [comm.stripe(r)[0] for r in range(world)] + [H]
            else:
                cost = torch.from_numpy(ctx.row_costs().astype("int64")).to(dev if not host_exchange else "cpu")
                dist.all_reduce(cost)
                bounds = blinky_amd.ffi.stripe_bounds_from_costs(cost.cpu().numpy().astype("uint32"), 0, world)
                ctx.set_rows(bounds[rank], bounds[rank + 1])
            r0, r1 = bounds[rank], bounds[rank + 1]
            ctx.build()
            rebalanced = True
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: stripes not rebalanced ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
    t0 = time.time()
    display, scale = ctx.build()                  # module cached: emit + launch (+ host fix-up of flagged pixels) only
    display = comm.or_display(display) if comm else multigpu.or_display(display, world, dev)   # which plates the whole frame reads
    build_wall_ms = (time.time() - t0) * 1e3
    build_kernel_ms = ctx.last_build_ms()
    fix_flagged, fix_changed = ctx.last_build_fixups()
    t0 = time.time()
    tile_stats = ctx.tile_stats()                 # compiles the block map (chunk lists) of the lensmap for the apply kernel
    tilemap_first_wall_ms = (time.time() - t0) * 1e3      # includes the one-off buffer allocations
    ctx.build()
    ctx.synchronize()
    t0 = time.time()
    tile_stats = ctx.tile_stats()
    tilemap_wall_ms = (time.time() - t0) * 1e3            # steady state: what a zoom / lens change costs on top of the build
    model = ctx.traffic_model()
    for f in range(R):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, f)
    torch.cuda.synchronize()

    rows = r1 - r0
    # Draw_TileClear stand-in: 0.  N = 1: four rotating output buffers.  N > 1: two stripe buffers, so that batch i+1 is
    # warped while batch i's stripes travel.
    NB = 4 if world == 1 else 2
    stripes = [torch.zeros((F, rows, W), dtype=torch.uint8, device=dev) for _ in range(NB)]
    stripe = stripes[0]
    nown = (F + world - 1) // world
    xdev = torch.device("cpu") if host_exchange else dev
    frames_out = [torch.zeros((nown, H, W), dtype=torch.uint8, device=xdev) for _ in range(2)] if world > 1 else None
    pending = [[], []]
    root_frames = torch.zeros((F, H, W), dtype=torch.uint8, device=dev) if (comm and rank == 0) else None

    def origin(t):
        # bk_apply_device takes the address of the frame's pixel (0,0) and writes the owned rows [r0, r1) only;
        # a stripe buffer holds just those rows, so its frame origin lies r0 rows before it
        return t.data_ptr() - r0 * W

    def first_globe(i):
        return (i * F) % R                       # the ring advances by one batch every step

    # Consecutive steps alternate between `--streams` HIP streams (bk_set_stream): a batch launch ends with a partly filled
    # round of workgroups, and with the next batch queued on another stream its workgroups fill the CUs as they free up
    # (4K panini x16: 3.89 -> 3.52 us/frame).  Output / stripe buffers rotate with the same period or a multiple of it, so a
    # buffer is always written from the same stream.
    nstreams = max(1, min(args.streams, NB))
    while NB % nstreams:
        nstreams -= 1
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]

    stream_handles = [st.cuda_stream for st in streams]

    def step(i):
        if world > 1 and comm is None:             # (exchange through torch.distributed - the gloo smoke, the RCCL fallback: torch ops follow torch's current stream)
            with torch.cuda.stream(streams[i % nstreams]):
                ctx.set_stream(stream_handles[i % nstreams])
                step_on_current_stream(i)
        else:                                      # the library launches on the stream it is handed: no torch stream switch per step (~3 us of host time)
            ctx.set_stream(stream_handles[i % nstreams])
            step_on_current_stream(i)

    def step_on_current_stream(i):
        # one batch: warp this rank's stripe of F frames, then (N > 1) reassemble frame f on rank f % world (one grouped
        # RCCL send/recv per batch; buffers alternate, the exchange of batch i overlaps the warp of batch i+1)
        b = i % NB
        if comm:
            comm.wait(b)                          # (device-side) the exchange that last used buffer pair b has finished
        elif world > 1:
            for w in pending[b]:
                w.wait()
            pending[b] = []
        ctx.apply_device(origin(stripes[b]), W, rows * W, frame0=first_globe(i), nframes=F)
        if comm:
            comm.exchange_rotating(stripes[b].data_ptr(), F, frames_out[b].data_ptr(), H * W, slot=b)
        elif world > 1:
            pending[b] = multigpu.exchange_rotating(stripes[b].cpu() if host_exchange else stripes[b], bounds, rank, world, frames_out[b], wait=False)

    def drain():
        if comm:
            comm.synchronize()
        for b in range(2):
            for w in pending[b]:
                w.wait()
            pending[b] = []

    def step_root(i):
        # the single-display variant: every frame of the batch gathered onto rank 0
        ctx.set_stream(stream.cuda_stream)
        if comm:
            comm.wait(0)
            ctx.apply_device(origin(stripe), W, rows * W, frame0=first_globe(i), nframes=F)
            comm.gather(stripe.data_ptr(), F, 0, root_frames.data_ptr() if rank == 0 else None, H * W, slot=0)
            return
        ctx.apply_device(origin(stripe), W, rows * W, frame0=first_globe(i), nframes=F)
        multigpu.gather_stripes(stripe.cpu() if host_exchange else stripe, bounds, rank, world, 0, None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # Should the grouped send/recv not be usable on this node, fall back to the plain gather onto rank 0 (and say so)
    # rather than lose the measurement; every rank takes the same decision.
    exchange_mode = "rotating"

    def exchange_works():
        ok = 1
        try:
            step(0)
            drain()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: rotating exchange failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        pending[0], pending[1] = [], []
        return int(t.item()) == 1

    if world > 1 and not exchange_works():
        if comm is not None:
            # libblinkyhip's communicator came up but its exchange did not run: try the process group's RCCL before
            # giving up on rotating roots (the JSON line says which transport was timed)
            print(f"[bench] rank {rank}: falling back from bk_comm to torch.distributed", file=sys.stderr, flush=True)
            comm, comm_fallback = None, "the communicator came up but its first exchange failed"
            if not exchange_works():
                exchange_mode = "root"
        else:
            exchange_mode = "root"
    run_step = step if exchange_mode == "rotating" else step_root

    def timed_region(first):
        """EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        barrier()
        t0 = time.perf_counter()
        for i in range(first, first + args.steps):
            run_step(i)
        drain()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    first_step_check = None
    if world > 1 and not args.no_first_step_check and exchange_mode == "rotating":
        # step 0 for real, then every frame this rank ends up holding against the same frame warped whole here; every rank's
        # verdict (and its error text) travels to rank 0 and into the JSON line
        msg = "ok"
        try:
            run_step(0)
            drain()
            torch.cuda.synchronize()
            chk = blinky_amd.Context(local_rank)
            chk.set_stream(stream.cuda_stream)
            chk.set_frames(F)
            S.configure(chk, GLOBE, LENS, ZOOM, (W, H))
            chk.build()
            g0 = first_globe(0)
            mine = multigpu.owned_frames(F, rank, world)
            for f in mine:
                for p in range(6):
                    chk.fill_plate_lcg(f, p, (g0 + f) % R)
            whole = torch.zeros((H, W), dtype=torch.uint8, device=dev)
            bad = []
            for f in mine:
                chk.apply_device(whole.data_ptr(), W, H * W, frame0=f, nframes=1)
                torch.cuda.synchronize()
                if not torch.equal(frames_out[0][f // world].to(dev), whole):
                    bad.append(f)
            chk.close()
            if bad:
                msg = f"MISMATCH in frames {bad}"
        except Exception as e:      # noqa: BLE001
            msg = f"{type(e).__name__}: {e}"
        verdicts = [None] * world
        dist.all_gather_object(verdicts, msg)
        first_step_check = {"ranks": verdicts, "ok": all(v == "ok" for v in verdicts)}
        if not first_step_check["ok"]:
            if rank == 0:
                print(f"[bench] first-step check FAILED: {verdicts}", file=sys.stderr, flush=True)
                print(json.dumps({"metric": "warped Mpixels/s (lensmap apply)", "value": None, "n_gpus": world, "error": "first-step check failed",
                                  "first_step_check": first_step_check}))
            sys.exit(3)
    for i in range(args.warmup):
        run_step(i)
    drain()
    nrep = args.repeats
    if nrep <= 0:
        # as many regions as keep the GPU busy for --min-gpu-seconds (a 50-step 4K region is ~3 ms: 31 of them would be
        # 0.1 s of GPU work inside a run of several seconds); every rank derives the same count from the max over ranks
        probe = timed_region(args.warmup)
        nrep = int(min(20000, max(31, args.min_gpu_seconds / max(probe, 1e-6))))
    regions = [timed_region(args.warmup + k * args.steps) for k in range(max(1, nrep))]
    elapsed = statistics.median(regions)
    # the same job with every launch on ONE stream (what roofline.kernel_ms_per_launch is to be compared with)
    one_stream_elapsed = None
    if nstreams > 1:
        nstreams_saved, nstreams = nstreams, 1
        for i in range(args.warmup):
            run_step(i)
        drain()
        one_stream_elapsed = statistics.median([timed_region(args.warmup + k * args.steps) for k in range(max(1, nrep // 4))])
        nstreams = nstreams_saved
    # N = 1: the same job in 16-frame steps (the headline of rounds 1-4), for continuity - not `value`
    elapsed16 = None
    if world == 1 and F != EXTRA_FRAMES and F > EXTRA_FRAMES:
        def region16(first):
            barrier()
            t0 = time.perf_counter()
            for i in range(first, first + args.steps):
                ctx.set_stream(stream_handles[i % nstreams])
                ctx.apply_device(origin(stripes[i % NB]), W, rows * W, frame0=(i * EXTRA_FRAMES) % R, nframes=EXTRA_FRAMES)
            barrier()
            return time.perf_counter() - t0
        region16(0)
        elapsed16 = statistics.median([region16(k * args.steps) for k in range(max(5, min(61, nrep // 8)))])
        ctx.set_stream(stream.cuda_stream)
    # N > 1: rank 0 alone also runs the ONE-GPU job on the same workload (whole frames, same F, same streams) while the others wait:
    # the denominator of the scaling curve measured in the same run, on the same box (`scaling_reference_mpx_s`)
    scaling_reference = None
    if world > 1 and not os.environ.get("BLINKY_BENCH_NO_REFERENCE"):
        ref_elapsed = 0.0
        if rank == 0:
            try:
                ref = OneGpuWorkload(torch, blinky_amd, S, local_rank, GLOBE, LENS, ZOOM, W, H, F, ring_max=R, ring_bytes=0)
                for i in range(3):
                    ref.launch(i)
                ref_elapsed = ref.job_seconds_per_step(steps=args.steps, repeats=max(5, min(31, nrep // 8)), nstreams=nstreams)
                ref.close()
            except Exception as e:      # noqa: BLE001
                print(f"[bench] rank 0: one-GPU scaling reference failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        barrier()
        if ref_elapsed > 0:
            scaling_reference = W * H * F / ref_elapsed / 1e6
    root_elapsed = None
    if world > 1:
        # extra (not `value`): all frames assembled on rank 0 - bounded by one GPU's xGMI ingest
        nroot = max(3, args.steps // 5)
        step_root(0)
        barrier()
        t0 = time.perf_counter()
        for i in range(nroot):
            step_root(i)
        barrier()
        root_elapsed = max_over_ranks((time.perf_counter() - t0) / nroot)

    torch.cuda.synchronize()
    ctx.set_stream(stream.cuda_stream)

    # ---- the dominant kernel alone, HIP events on the launch stream (roofline) ------------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def kernel_ms(ring, nframes=F, launches=args.steps, repeats=15):
        """median over `repeats` of: HIP events around `launches` back-to-back launches cycling a ring of `ring` globes"""
        out = []
        for k in range(repeats):
            barrier()
            e0.record(stream)
            for i in range(launches):
                ctx.apply_device(origin(stripes[i % NB]), W, rows * W, frame0=(i * nframes) % ring, nframes=nframes)
            e1.record(stream)
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / launches)
        return statistics.median(out), min(out), max(out)

    k_med, k_min, k_max = kernel_ms(R)
    kw_med, _, _ = kernel_ms(F)                           # the ring that fits the Infinity Cache: the same F globes every launch
    stripe_mpx = W * rows * F / (k_med * 1e-3) / 1e6
    if world > 1:
        t = torch.tensor([stripe_mpx], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        stripe_complete_mpx = float(t.item())
    else:
        stripe_complete_mpx = stripe_mpx
    single_ms, _, _ = kernel_ms(R, nframes=1)            # single-frame launches (what the engine drop-in issues), cold ring
    single_resident = None
    if world == 1:
        try:
            single_resident = resident_measure(torch, ctx, origin(stripes[0]), W, rows, R, orig_first_row=r0)
        except Exception as e:      # noqa: BLE001
            single_resident = {"error": f"{type(e).__name__}: {e}"}

    if args.check:
        # Every frame this rank ends up holding after the LAST step of the timed region, against the reference: FNV-1a-64 per frame
        # compared with tests/golden/lensmaps.json (frame 0 recorded from the unmodified fisheye.c, the rest from the oracle's
        # render_lensmap over the reference's lensmap: tests/golden/make_golden.py).  Frame f of that step is LCG globe (g0 + f) % R,
        # which the goldens cover while R <= 64.  Nothing of the CPU oracle is touched here: the hash is bk_debug_fnv1a64.
        from concurrent.futures import ThreadPoolExecutor
        gold = None
        for rec in json.load(open(os.path.join(ROOT, "tests", "golden", "lensmaps.json")))["lensmaps"]:
            if (rec["globe"], rec["lens"], rec["W"], rec["H"]) == (GLOBE, LENS, W, H) and rec["zoom"] in (None, ZOOM) and "fnv_frames" in rec:
                gold, gold_rec = rec["fnv_frames"], rec
        if gold is None or R > len(gold):
            sys.exit(f"[check] no golden frames for {W}x{H} {GLOBE}/{LENS} over a ring of {R} globes (tests/golden/lensmaps.json)")
        last_i = args.warmup + max(1, nrep // 4 if one_stream_elapsed is not None else nrep) * args.steps - 1   # the last step of the last timed region
        g0 = first_globe(last_i)
        if world > 1:
            mine = multigpu.owned_frames(F, rank, world) if exchange_mode == "rotating" else []
            held = {f: frames_out[last_i % NB][f // world] for f in mine}
        else:
            ctx.apply_device(origin(stripe), W, rows * W, frame0=g0, nframes=F)      # the timed launch once more, into a buffer read back here
            torch.cuda.synchronize()
            held = {f: stripe[f] for f in range(F)}
            off, tin = ctx.read_lensmap()                  # ... and the table the timed launches gathered through IS the reference's
            mine_is = (blinky_amd.ffi.fnv1a64(off), blinky_amd.ffi.fnv1a64(tin), repr(scale))
            if mine_is != (gold_rec["fnv_offsets"], gold_rec["fnv_tints"], gold_rec["scale"]):
                sys.exit(f"[check] the lensmap this run built differs from the reference's golden: offsets / tints / scale {mine_is} against "
                         f"{(gold_rec['fnv_offsets'], gold_rec['fnv_tints'], gold_rec['scale'])}")
            del off, tin
        with ThreadPoolExecutor(8) as pool:
            got = dict(zip(held, pool.map(lambda t: blinky_amd.ffi.fnv1a64(t.cpu().numpy()), held.values())))
        bad = [f for f in held if got[f] != gold[(g0 + f) % R]]
        print(f"[check] rank {rank}: {'OK' if not bad else 'MISMATCH in frames ' + str(bad)} ({len(held)} frames against tests/golden/lensmaps.json)",
              file=sys.stderr, flush=True)
        if bad:
            sys.exit(3)

    if rank == 0:
        px_per_step = W * H * F
        value = px_per_step * args.steps / elapsed / 1e6
        algo_bytes = ALGO_BYTES_PER_PX * W * rows * F                      # per launch on this rank
        achieved = algo_bytes / (k_med * 1e-3) / 1e9
        # what the kernel has to move per launch: every mapped pixel stored once, every distinct globe line the lensmap
        # touches read once per frame, the block map (headers + chunk lists + 16-bit pixel addresses) once per block visit
        compulsory = compulsory_bytes(model, F)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "apply_traffic.json")
        if world == 1 and not args.no_live_traffic:
            trace("traffic children")
            traffic, traffic_src = measure_traffic(args, F, R, int(model["block_height"]))
        if traffic is None and os.path.exists(tpath) and world == 1:
            live_error = traffic_src
            try:
                rec = json.load(open(tpath))
                want = f"{W}x{H} {GLOBE}/{LENS} x{F} ring{R}"
                if rec.get("workload") == want and rec.get("kernel_source_sha1_16") == kernel_source_hash():
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_src = {k: rec.get(k) for k in ("profile", "commit", "kernel_source_sha1_16", "FETCH_SIZE_KiB_per_launch",
                                                           "WRITE_SIZE_KiB_per_launch")}
                else:
                    traffic_src = {"stale": True, "why": f"profiles/apply_traffic.json was measured on workload {rec.get('workload')!r}, "
                                   f"kernel sources {rec.get('kernel_source_sha1_16')}; this run is {want!r}, {kernel_source_hash()} - "
                                   "rerun tools/profile_bench.sh"}
            except Exception as e:      # noqa: BLE001
                traffic_src = {"stale": True, "why": f"unreadable record: {e}"}
            if live_error and traffic_src is not None:
                traffic_src["live_measurement"] = live_error
        t_launch = k_med * 1e-3
        # calibration: a plain streaming kernel with the apply's read : write ratio (compulsory model), on this box, now
        stream_mix = None
        try:
            wr = F * model["mapped_pixels"]
            rd = max(1, compulsory - wr)
            period = 64
            writes = max(0, min(period, int(round(period * wr / rd))))
            trace("stream mix")
            gbps = ctx.stream_mix(1 << 30, period, writes)
            used = traffic if traffic else compulsory
            stream_mix = {"GB/s": round(gbps, 1), "read_KiB_per_written_KiB": round(period / max(1, writes), 3),
                          "what": "plain streaming kernel, 16-byte loads, non-temporal stores, the apply's read : write ratio, "
                                  "1 GiB read (bk_debug_stream_mix): the practical roofline of this memory system for that mix",
                          "apply_traffic_over_it": round(used / t_launch / 1e9 / gbps, 4),
                          "apply_traffic_is": "traffic (PMC)" if traffic else "compulsory model"}
        except Exception as e:      # noqa: BLE001
            stream_mix = {"error": f"{type(e).__name__}: {e}"}
        out = {
            "metric": "warped Mpixels/s (lensmap apply)", "value": round(value, 1), "unit": "Mpixels/s",
            "value_one_stream": round(px_per_step * args.steps / one_stream_elapsed / 1e6, 1) if one_stream_elapsed else round(value, 1),
            "ms_per_step_one_stream": round(one_stream_elapsed / args.steps * 1e3, 4) if one_stream_elapsed else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            # (rounds 1-4 reported the N = 1 job in 16-frame steps; the same job, same run)
            "value_at_16_frames": round(W * H * EXTRA_FRAMES * args.steps / elapsed16 / 1e6, 1) if elapsed16 else None,
            # N > 1: the one-GPU job on the same workload (whole frames, same frames per step), timed by rank 0 in this run
            "scaling_reference_mpx_s": round(scaling_reference, 1) if scaling_reference else None,
            "speedup_vs_scaling_reference": round(value / scaling_reference, 3) if scaling_reference else None,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} {GLOBE}/{LENS} {ZOOM}, {F} frames/step from a resident ring of {R} distinct globes "
                                   f"({R * 6 * 2160 * 2176 / 1e9:.2f} GB, advanced every step), one lensmap; value / ms_per_step = the job with steps "
                                   f"alternating between {nstreams} HIP stream(s), value_one_stream = the same on one stream, roofline.kernel_ms_per_launch "
                                   "= the kernel alone under HIP events; 64 frames/step since r5 (r1-r4 ran 16: compare those with value_at_16_frames)",
                       "frames_per_step": F, "ring_globes": R,
                       "parallelism": f"row-stripes x{world}" + ("" if world == 1 else (" + bk_comm (librccl grouped ncclSend/ncclRecv behind the C ABI)" if comm else " + gloo host exchange (developer smoke)" if host_exchange else f" + torch.distributed RCCL send/recv (bk_comm unavailable: {comm_fallback})") +
                                                                     (": frame f reassembled on rank f%N" if exchange_mode == "rotating" else ": every frame gathered onto rank 0")),
                       "apply_variant": args.variant, "streams": nstreams,
                       "streams_note": "consecutive steps alternate between this many HIP streams, so the tail of one batch launch "
                                       "overlaps the ramp of the next; roofline.* is the kernel alone on one stream"},
            "timed_regions": {"count": len(regions), "steps_each": args.steps, "value_is": "median", "gpu_timed_seconds": round(sum(regions), 3),
                              "mpx_s_median": round(value, 1),
                              "mpx_s_min": round(px_per_step * args.steps / max(regions) / 1e6, 1),
                              "mpx_s_max": round(px_per_step * args.steps / min(regions) / 1e6, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the engine's real call: ONE frame per launch, nothing amortised - the one contract-formula figure <= 1
                         "single_frame": {"us": round(single_ms * 1e3, 3),
                                          "algorithmic_frac": round(ALGO_BYTES_PER_PX * W * rows / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "target_us_at_0.70": round(ALGO_BYTES_PER_PX * W * rows / (0.70 * HBM_PEAK_GBS * 1e9) * 1e6, 3),
                                          "what": "one bk_apply_device launch per frame (HIP events around a train of them, cold ring); `resident`: "
                                                  "the same frames as commands to the resident kernel",
                                          "resident": single_resident},
                         "frac_is": "ALGORITHMIC bytes (6 B/px, SURVEY.md 8(d)) / kernel time / peak, as the bench contract defines it; the "
                                    "kernel moves fewer bytes than that (2-byte LDS addresses read once per 8 frames instead of a 4-byte "
                                    "index per pixel and frame), so this ratio can exceed 1 and is NOT a bandwidth utilisation - "
                                    "frac_traffic and frac_compulsory are",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "frac_traffic": round(traffic / t_launch / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         "compulsory_bytes_per_launch": int(compulsory),
                         "frac_compulsory": round(compulsory / t_launch / 1e9 / HBM_PEAK_GBS, 4),
                         "stream_mix": stream_mix,
                         # the timed job, not the kernel alone: launches overlap when the steps alternate streams
                         "job": {"launches_per_s": round(args.steps / elapsed, 1),
                                 "traffic_GBps": round((traffic if traffic else compulsory) * args.steps / elapsed / 1e9, 1),
                                 "frac_traffic": round((traffic if traffic else compulsory) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                                 "bytes_are": "traffic (PMC)" if traffic else "compulsory model"} if world == 1 else None,
                         "kernel_ms_per_launch": round(k_med, 5), "kernel_ms_min": round(k_min, 5), "kernel_ms_max": round(k_max, 5),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "ring_globes": R, "model": {k: int(v) for k, v in model.items()},
                         "warm_ring": {"ring_globes": F, "kernel_ms_per_launch": round(kw_med, 5),
                                       "achieved": round(algo_bytes / (kw_med * 1e-3) / 1e9, 1),
                                       "frac_compulsory": round(compulsory / (kw_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "the same F globes every launch: touched set fits the 256 MiB Infinity Cache"}},
            "lensmap_build_ms": round(build_kernel_ms, 3),
            "lensmap_build_wall_ms": round(build_wall_ms, 2),
            "lensmap_build_first_wall_ms_incl_hiprtc": round(build_first_wall_ms, 1),
            "lensmap_build_host_fixups": {"flagged": fix_flagged, "changed": fix_changed},
            "lensmap_blockmap_compile_wall_ms": round(tilemap_wall_ms, 3),
            "lensmap_blockmap_first_wall_ms_incl_alloc": round(tilemap_first_wall_ms, 3), "tile_stats": tile_stats,
            "stripe_complete_mpx_s": round(stripe_complete_mpx, 1),
            "assembled_on_rank0_mpx_s": round(W * H * F / root_elapsed / 1e6, 1) if root_elapsed else None,
            # what xGMI allows (one link per GPU pair, ~64 GB/s per direction): a rank receives (N-1)/N of every frame it
            # owns, each peer's share over that peer's own link -> whole-job bound = N * N * link rate (1 B per pixel)
            "exchange": None if world == 1 else {
                "bytes_received_per_rank_per_step": int(len(multigpu.owned_frames(F, 0, world)) * (H - (bounds[1] - bounds[0])) * W),
                "xgmi_link_GBps_assumed": 64, "bound_mpx_s": round(world * world * 64e9 / 1e6, 1),
                "bound_all_on_rank0_mpx_s": round(64e9 * world / 1e6, 1)},
            "stripes": {"rows_per_rank": [bounds[r + 1] - bounds[r] for r in range(world)], "rebalanced": rebalanced},
            "first_step_check": first_step_check,
            "single_frame_launch_us": round(single_ms * 1e3, 2),
            "single_frame_mpx_s": round(W * rows / (single_ms * 1e-3) / 1e6, 1),
            "lens_scale": scale,
        }
        if world == 1 and not args.no_extra:
            # the other configurations BASELINE.json names that fit one GPU, timed by this same run: C2 (1080p stereographic), C3 (4K
            # quincuncial, the inverse-only full-sphere lens), C5 (8K hammer, 64 frames in ONE launch over a ring of 64 distinct 8K
            # globes = 7.2 GB), the headline with the rubix tint LUTs on (7 B/px), and 4K hammer as the whole-globe single-frame case
            FX = EXTRA_FRAMES
            extras = [("C2 (BASELINE.json configs[1])", "cube", "stereographic", None, 1920, 1080, FX, args.steps, False, 64),
                      ("C2x64 (the same map, 64 frames per launch: as many bytes per launch as the headline's)", "cube", "stereographic", None, 1920, 1080, 64,
                       max(6, args.steps // 3), False, 64),
                      ("C3 (BASELINE.json configs[2])", "cube", "quincuncial", None, 3840, 2160, FX, args.steps, False, 64),
                      ("C5 (BASELINE.json configs[4], on one GPU)", "cube", "hammer", None, 7680, 4320, 64, max(6, args.steps // 5), False, 64),
                      ("headline, rubix on (fisheye.c:2416-2419)", GLOBE, LENS, ZOOM, W, H, FX, args.steps, True, 64),
                      ("4K cube/hammer (whole-globe lens)", "cube", "hammer", None, 3840, 2160, FX, args.steps, False, 64)]
            trace("build times")
            out["build_ms"] = dict(what=f"bk_build at {W}x{H} on the cube globe: [call on the host clock, best of 9] / median / device part; "
                                        "the reference's create_lensmap beside it: cpu_baseline.build_ms",
                                   **build_times(torch, blinky_amd, S, local_rank, W, H))
            out["configs_extra"] = []
            for (nm, g, l, z, w_, h_, f_, st_, rb_, rm_) in extras:
                try:
                    out["configs_extra"].append(extra_config(torch, blinky_amd, S, local_rank, nm, g, l, z, w_, h_, f_, st_, args, rubix=rb_, ring_max=rm_))
                except Exception as e:      # noqa: BLE001
                    out["configs_extra"].append({"name": nm, "error": f"{type(e).__name__}: {e}"})
            try:
                # with the launch `bench.py --gpus N` issues: 64 frames per step, stripes of equal block-map cost
                out["predicted_stripe_complete"] = dict(
                    what="rank r's stripe for N = 2 / 4 / 8 built and timed on this one GPU with the launch `bench.py --gpus N` issues (64 frames "
                         "per step, stripes cut by the block map's row costs); a step lasts as long as the slowest rank; no exchange",
                    **predicted_stripes(torch, blinky_amd, S, local_rank, GLOBE, LENS, ZOOM, W, H, 64, args.steps, k_med if F == 64 else None,
                                        resident_us_1=single_resident.get("us") if isinstance(single_resident, dict) else None),
                    frames16=predicted_stripes(torch, blinky_amd, S, local_rank, GLOBE, LENS, ZOOM, W, H, EXTRA_FRAMES, args.steps,
                                               k_med if F == EXTRA_FRAMES else None),
                    C4_trism_panini=predicted_stripes(torch, blinky_amd, S, local_rank, "trism", "panini", "f_fov 180", W, H, 64, args.steps))
            except Exception as e:      # noqa: BLE001
                out["predicted_stripe_complete"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        emit(out, args.detail)
    if comm:
        comm.synchronize()
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
