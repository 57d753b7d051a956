#!/usr/bin/env python3
"""bench.py -- Blinky warp hot path on MI355X: lensmap APPLY throughput (+ lensmap BUILD ms).

Workload (BASELINE.json metric): 3840x2160 output, cube globe (6 faces of 2160x2160), panini
lens, f_fov 180.  Synthetic data: LCG globe faces generated on the device;
the lensmap is built on the device from the bundled Lua scripts before the timed region.

A *step* = one pass of the hot path over one batch: `frames` frames (distinct resident globes,
one shared lensmap) warped by one bk_apply_device launch; with N > 1 ranks each rank owns a
stripe of output rows (it builds and keeps only that stripe of the lensmap, holds a full globe
replica) and the step ends with the frames' reassembly: one grouped RCCL send/recv in which frame f's
stripes travel to rank f % N (blinky_amd.multigpu.exchange_rotating), overlapped with the next batch's
warp.  (All frames gathered onto rank 0 - the single-display case, bounded by one GPU's xGMI ingest -
is timed too and reported as `assembled_on_rank0_mpx_s`.)

Prints ONE JSON line (rank 0).  `value` = whole-job Mpixels/s = W*H*frames*steps / time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

W, H = 3840, 2160
GLOBE, LENS, ZOOM = "cube", "panini", "f_fov 180"
ALGO_BYTES_PER_PX = 6          # 4 B lensmap index + 1 B texel + 1 B store
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(frames_budget_s=6.0):
    """The reference CPU path timed on this host's cores (1 thread: the reference is single-threaded).
    Uses oracle/_ref (unmodified fisheye.c) when the prebuilt library travelled with the repo,
    otherwise the oracle's C restatement.  Bounded sample: the 4K lensmap build once plus
    ~frames_budget_s of render_lensmap() calls."""
    import ctypes as C
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=16, help="frames per step (batch warped by one launch)")
    ap.add_argument("--variant", type=int, default=-1, help="apply kernel variant (-1 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region, compare every reassembled frame this rank owns with a full-frame warp")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BLINKY_BENCH_BACKEND=gloo + BLINKY_BENCH_ONE_GPU=1: developer smoke of the N > 1 control flow on a
        # single-GPU box (all ranks on cuda:0, stripes exchanged through host memory); never a measurement
        dist.init_process_group(os.environ.get("BLINKY_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    host_exchange = world > 1 and dist.get_backend() == "gloo"
    if os.environ.get("BLINKY_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import blinky_amd
    import scripts as S
    from blinky_amd import multigpu

    F = args.frames
    ctx = blinky_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_frames(F)
    S.configure(ctx, GLOBE, LENS, ZOOM, (W, H))
    bounds = multigpu.stripe_bounds(H, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    ctx.set_rows(r0, r1)
    if args.variant >= 0:
        ctx.set_apply_variant(args.variant)

    # ---- lensmap build (each rank: its own stripe; no exchange) ----------------------------------
    t0 = time.time()
    ctx.build()                                   # includes hiprtc compilation of the lens
    build_first_wall_ms = (time.time() - t0) * 1e3
    t0 = time.time()
    display, scale = ctx.build()                  # module cached: emit + launch only
    display = multigpu.or_display(display, world, dev)   # which plates the whole frame reads
    build_wall_ms = (time.time() - t0) * 1e3
    build_kernel_ms = ctx.last_build_ms()
    t0 = time.time()
    tile_stats = ctx.tile_stats()                 # compiles the block map (chunk lists) of the lensmap for the apply kernel
    tilemap_first_wall_ms = (time.time() - t0) * 1e3      # includes the one-off buffer allocations
    ctx.build()
    ctx.synchronize()
    t0 = time.time()
    tile_stats = ctx.tile_stats()
    tilemap_wall_ms = (time.time() - t0) * 1e3            # steady state: what a zoom / lens change costs on top of the build
    for f in range(F):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, f)
    torch.cuda.synchronize()

    rows = r1 - r0
    # Draw_TileClear stand-in: 0.  Two stripe buffers, so that batch i+1 is warped while batch i's stripes travel.
    stripes = [torch.zeros((F, rows, W), dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    stripe = stripes[0]
    nown = (F + world - 1) // world
    xdev = torch.device("cpu") if host_exchange else dev
    frames_out = [torch.zeros((nown, H, W), dtype=torch.uint8, device=xdev) for _ in range(2)] if world > 1 else None
    pending = [[], []]
    gather_list = None
    if world > 1 and rank == 0:
        hmax = max(bounds[r + 1] - bounds[r] for r in range(world))
        gather_list = [torch.empty((F, hmax, W), dtype=torch.uint8, device=dev) for r in range(world)]

    def origin(t):
        # bk_apply_device takes the address of the frame's pixel (0,0) and writes the owned rows [r0, r1) only;
        # a stripe buffer holds just those rows, so its frame origin lies r0 rows before it
        return t.data_ptr() - r0 * W

    def step(i):
        # one batch: warp this rank's stripe of F frames, then reassemble frame f on rank f % world (one grouped
        # RCCL send/recv per batch; buffers alternate, the exchange of batch i overlaps the warp of batch i+1)
        b = i & 1 if world > 1 else 0
        for w in pending[b]:
            w.wait()
        pending[b] = []
        ctx.apply_device(origin(stripes[b]), W, rows * W, frame0=(i * F) % F, nframes=F)
        if world > 1:
            src = stripes[b].cpu() if host_exchange else stripes[b]
            pending[b] = multigpu.exchange_rotating(src, bounds, rank, world, frames_out[b], wait=False)

    def drain():
        for b in range(2):
            for w in pending[b]:
                w.wait()
            pending[b] = []

    def step_root(i):
        # the single-display variant: every frame of the batch gathered onto rank 0
        ctx.apply_device(origin(stripe), W, rows * W, frame0=(i * F) % F, nframes=F)
        multigpu.gather_stripes(stripe.cpu() if host_exchange else stripe, bounds, rank, world, 0,
                                None if host_exchange else gather_list)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Should the grouped send/recv not be usable on this node, fall back to the plain gather onto rank 0 (and say so)
    # rather than lose the measurement; every rank takes the same decision.
    exchange_mode = "rotating"
    if world > 1:
        ok = 1
        try:
            step(0)
            drain()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: rotating exchange failed ({type(e).__name__}: {e}); using gather to rank 0", file=sys.stderr, flush=True)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            exchange_mode = "root"
            pending[0], pending[1] = [], []
    run_step = step if exchange_mode == "rotating" else step_root
    for i in range(args.warmup):
        run_step(i)
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    root_elapsed = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # extra (not `value`): all frames assembled on rank 0 - bounded by one GPU's xGMI ingest
        nroot = max(3, args.steps // 5)
        step_root(0)
        barrier()
        t0 = time.perf_counter()
        for i in range(nroot):
            step_root(i)
        barrier()
        t = torch.tensor([(time.perf_counter() - t0) / nroot], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        root_elapsed = float(t.item())

    # ---- the dominant kernel alone, HIP events on the launch stream (roofline) ------------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    e0.record(stream)
    for i in range(args.steps):
        ctx.apply_device(origin(stripe), W, rows * W, frame0=0, nframes=F)
    e1.record(stream)
    torch.cuda.synchronize()
    kernel_ms = e0.elapsed_time(e1) / args.steps
    stripe_mpx = W * rows * F / (kernel_ms * 1e-3) / 1e6
    if world > 1:
        t = torch.tensor([stripe_mpx], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        stripe_complete_mpx = float(t.item())
    else:
        stripe_complete_mpx = stripe_mpx
    # single-frame launches (launch-latency sensitive)
    barrier()
    e0.record(stream)
    for i in range(args.steps):
        ctx.apply_device(origin(stripe), W, rows * W, frame0=i % F, nframes=1)
    e1.record(stream)
    torch.cuda.synchronize()
    single_ms = e0.elapsed_time(e1) / args.steps

    if args.check:
        # every frame this rank ends up holding == the same frame warped whole by a full-height context
        full_ctx = blinky_amd.Context(local_rank)
        full_ctx.set_stream(stream.cuda_stream)
        full_ctx.set_frames(F)
        S.configure(full_ctx, GLOBE, LENS, ZOOM, (W, H))
        full_ctx.build()
        for f in range(F):
            for p in range(6):
                full_ctx.fill_plate_lcg(f, p, f)
        full = torch.zeros((F, H, W), dtype=torch.uint8, device=dev)
        full_ctx.apply_device(full.data_ptr(), W, H * W, frame0=0, nframes=F)
        torch.cuda.synchronize()
        if world > 1:
            if exchange_mode == "rotating":
                last = frames_out[(args.steps - 1) & 1]
                bad = [f for f in multigpu.owned_frames(F, rank, world) if not torch.equal(last[f // world].to(dev), full[f])]
            else:
                bad = []
        else:
            ctx.apply_device(origin(stripe), W, rows * W, frame0=0, nframes=F)
            torch.cuda.synchronize()
            bad = [f for f in range(F) if not torch.equal(stripe[f], full[f])]
        print(f"[check] rank {rank}: {'OK' if not bad else 'MISMATCH in frames ' + str(bad)}", file=sys.stderr, flush=True)
        if bad:
            sys.exit(3)

    if rank == 0:
        px_per_step = W * H * F
        value = px_per_step * args.steps / elapsed / 1e6
        algo_bytes = ALGO_BYTES_PER_PX * W * rows * F                      # per launch on this rank
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "apply_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get("workload") == f"{W}x{H} {GLOBE}/{LENS} x{F}" and world == 1:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "warped Mpixels/s (lensmap apply)", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} {GLOBE}/{LENS} {ZOOM}, {F} frames/step (distinct resident globes, one lensmap)",
                       "frames_per_step": F, "parallelism": f"row-stripes x{world}" + ("" if world == 1 else " + RCCL grouped send/recv: frame f reassembled on rank f%N"
                                                                     if exchange_mode == "rotating" else " + RCCL gather of every frame onto rank 0"),
                       "apply_variant": args.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms_per_launch": round(kernel_ms, 5), "algorithmic_bytes_per_launch": algo_bytes},
            "lensmap_build_ms": round(build_kernel_ms, 3),
            "lensmap_build_wall_ms": round(build_wall_ms, 2),
            "lensmap_build_first_wall_ms_incl_hiprtc": round(build_first_wall_ms, 1),
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
            "single_frame_launch_us": round(single_ms * 1e3, 2),
            "single_frame_mpx_s": round(W * rows / (single_ms * 1e-3) / 1e6, 1),
            "lens_scale": scale,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
