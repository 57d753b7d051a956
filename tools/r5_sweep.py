#!/usr/bin/env python3
"""Developer probe (round 5): the resident apply over block heights x workgroups per CU.
usage: python tools/r5_sweep.py [--lenses panini,hammer] [--shapes 0,1,2,4] [--occ 0,8,6,4,3] [--size WxH] [--flags n]"""
import argparse, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench, blinky_amd, scripts as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lenses", default="panini,hammer")
ap.add_argument("--shapes", default="0,1,2,4")
ap.add_argument("--occ", default="0,8,6,4,3")
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--frames", type=int, default=400)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--globe", default="cube")
args = ap.parse_args()
W, H = [int(v) for v in args.size.split("x")]
for lens in args.lenses.split(","):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, args.globe, lens, None if lens != "panini" else "f_fov 180", W, H, 1)
    ctx = wl.ctx
    ctx.set_ablation(args.flags)
    for shape in [int(v) for v in args.shapes.split(",")]:
        ctx.set_tile_shape(shape)
        for occ in [int(v) for v in args.occ.split(",")]:
            ctx.set_tile_shape(100 + (occ if occ else 16))
            try:
                ctx.resident_begin(idle_ms=200)
                info = ctx.resident_info()
                dst = wl.origin(wl.out[0])
                ctx.resident_wait(ctx.resident_submit(dst, W, frame=0))
                ts = []
                for rep in range(3):
                    t0 = time.perf_counter()
                    ctx.resident_wait(ctx.resident_submit_batch(dst, W, 0, frame0=(rep * args.frames) % wl.R, nframes=args.frames))
                    ts.append((time.perf_counter() - t0) / args.frames * 1e6)
                ctx.resident_end()
                print(f"SWEEP {lens:12s} {W}x{H} shape {shape} occ {occ}: {statistics.median(ts):6.2f} us/frame  [{info['workgroups']} wgs, K {info['blocks_in_registers']}, "
                      f"nq {info['chunks_per_thread']}, 128x{info['block_h']}, {info['per_cu']}/CU]", flush=True)
            except Exception as e:      # noqa: BLE001
                print(f"SWEEP {lens} shape {shape} occ {occ}: {type(e).__name__}: {e}", flush=True)
    wl.close()
