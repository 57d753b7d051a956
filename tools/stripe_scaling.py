#!/usr/bin/env python3
"""Predict the row-stripe scaling curve on ONE GPU (SURVEY.md 8(e) "stripe-complete" throughput).

For N in {1, 2, 4, 8} and every rank r the stripe rows [b[r], b[r+1]) are built and warped here, on the one GPU the box
has, with the very launch the N-GPU job would issue on rank r (bk_set_rows + bk_build + bk_apply_device, cold ring).  A
step of the N-GPU job lasts as long as its slowest rank, so

    predicted stripe_complete speed-up(N) = t(N=1) / max_r t(rank r of N)

Stripes: equal heights (bench.py) and, for lenses that leave part of the screen unmapped, the bounds bk_comm_rebalance /
bk_multi_rebalance would pick (equal block-map cost, multiples of 8 rows).  Nothing is exchanged: this is the number the
>= 6x target of BASELINE.json can be held against; the reassembled-frame rate is bounded by xGMI instead.

usage: python tools/stripe_scaling.py [--configs panini4k,hammer4k,c5] [--single]     (prints a table; run on the GPU box)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402
from blinky_amd import ffi  # noqa: E402

CONFIGS = {
    "panini4k": ("cube", "panini", "f_fov 180", 3840, 2160, 16, False),
    "panini4k64": ("cube", "panini", "f_fov 180", 3840, 2160, 64, False),     # what bench.py --gpus N issues: 64 frames per step
    "trism4k": ("trism", "panini", "f_fov 180", 3840, 2160, 16, False),      # C4
    "hammer4k": ("cube", "hammer", None, 3840, 2160, 16, True),
    "quincuncial4k": ("cube", "quincuncial", None, 3840, 2160, 16, True),    # C3
    "stereo1080": ("cube", "stereographic", None, 1920, 1080, 16, False),    # C2
    "c5": ("cube", "hammer", None, 7680, 4320, 64, True),
}


MAPPED_ONLY = False
JOB = False           # --job: also the wall clock per step of the two-stream job (what bench.py's `value` times), host launch cost included
LAST_JOB_US = None
SHAPE = 0          # --shape: force the block height (1 / 2 / 4 = 128x8 / 128x16 / 128x32), 0 = the cost model


def time_stripe(globe, lens, zoom, W, H, F, rows, single):
    ring_max = 16 if W > 4000 else 32
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, rows=rows, ring_bytes=1.2e9 if W < 4000 else 3.0e9, ring_max=ring_max)
    if SHAPE:
        wl.ctx.set_tile_shape(SHAPE)
        wl.tile_stats = wl.ctx.tile_stats()
    for i in range(3):
        wl.launch(i)
    launches = 30 if W < 4000 else 6
    t = wl.kernel_ms(launches=launches, repeats=5)[0]
    t1 = wl.kernel_ms(nframes=1, launches=launches, repeats=5)[0] if single else None
    global LAST_JOB_US
    LAST_JOB_US = wl.job_seconds_per_step(steps=50, repeats=9, nstreams=2) * 1e6 if JOB else None
    stats = wl.tile_stats
    wl.close()
    return t, t1, stats


def balanced_bounds(globe, lens, zoom, W, H, n):
    ctx = blinky_amd.Context(0)
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.build()
    if MAPPED_ONLY:                 # round 3's first rule: rows of equal mapped pixels
        off, _ = ctx.read_lensmap()
        ctx.close()
        cost = (off.reshape(H, W) != 0xFFFFFFFF).sum(axis=1).astype(np.uint32)
        return ffi.stripe_bounds_from_costs(cost, W, n)
    cost = ctx.row_costs()          # what the rebalance sums over the ranks: the block map's costs, row by row
    ctx.close()
    return ffi.stripe_bounds_from_costs(cost, 0, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="panini4k,hammer4k,c5")
    ap.add_argument("--single", action="store_true", help="also time single-frame launches of every stripe")
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--balanced-all", action="store_true", help="try the rebalanced stripes for fully mapped lenses too")
    ap.add_argument("--job", action="store_true", help="also the two-stream job's wall clock per step on every stripe (host launch cost included)")
    ap.add_argument("--mapped-only", action="store_true", help="balanced = equal mapped pixels (the rule before the block-map costs)")
    args = ap.parse_args()
    global SHAPE, MAPPED_ONLY, JOB
    JOB = args.job
    SHAPE = args.shape
    MAPPED_ONLY = args.mapped_only
    print(f"# {torch.cuda.get_device_name(0)}; one GPU; every rank's stripe timed in turn; us per launch (HIP events, median of 5)")
    for name in args.configs.split(","):
        globe, lens, zoom, W, H, F, unmapped = CONFIGS[name]
        base = None
        for mode in (["equal", "balanced"] if unmapped or args.balanced_all else ["equal"]):
            for n in [int(v) for v in args.ranks.split(',')]:
                if mode == "balanced" and n == 1:
                    continue
                bounds = [H * r // n for r in range(n + 1)] if mode == "equal" else balanced_bounds(globe, lens, zoom, W, H, n)
                ts, t1s, shapes, jobs = [], [], [], []
                for r in range(n):
                    t, t1, stats = time_stripe(globe, lens, zoom, W, H, F, (bounds[r], bounds[r + 1]), args.single)
                    ts.append(t * 1e3)
                    t1s.append(t1 * 1e3 if t1 else 0.0)
                    jobs.append(LAST_JOB_US)
                    shapes.append(f"128x{stats['tile_h'] % 1000}/{stats['lds_bytes_per_wave'] // 1024}K/{stats['tiles']}blk")
                if base is None:
                    base = ts[0] * n if n > 1 else ts[0]
                worst = max(ts)
                line = (f"{name:14s} {W}x{H} x{F} N={n} {mode:8s} slowest {worst:9.2f} us  fastest {min(ts):9.2f} us  "
                        f"predicted stripe_complete speed-up {base / worst:5.2f}x  ({W * H * F / worst:9.1f} Mpx/s)  per-rank us: "
                        + " ".join(f"{t:.1f}" for t in ts))
                if args.single:
                    line += "  single-frame us: " + " ".join(f"{t:.2f}" for t in t1s)
                if JOB:
                    line += "  two-stream job, wall us per step: " + " ".join(f"{t:.1f}" for t in jobs) + f" -> {W * H * F / max(jobs):9.1f} Mpx/s"
                print(line, flush=True)
                if os.environ.get("BK_VERBOSE"):
                    print("    bounds", bounds, "block shape / staging buffer per rank:", shapes)


if __name__ == "__main__":
    main()
