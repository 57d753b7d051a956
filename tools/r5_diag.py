#!/usr/bin/env python3
"""Developer probe (round 5): what bounds the resident apply - pipelined us/frame with the stores off (bit 4), with non-temporal instead of
write-through stores (bit 16384, timing only), and the spread of the workgroups' busy time (bit 8192).
usage: python tools/r5_diag.py [--lenses panini,hammer] [--size 3840x2160] [--frames 400]"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402


def pipelined(ctx, wl, W, N, reps=4):
    ctx.resident_begin(idle_ms=200)
    info = ctx.resident_info()
    dst = wl.origin(wl.out[0])
    ctx.resident_wait(ctx.resident_submit(dst, W, frame=0))
    out = []
    for rep in range(reps):
        t0 = time.perf_counter()
        last = ctx.resident_submit_batch(dst, W, 0, frame0=(rep * N) % wl.R, nframes=N)
        ctx.resident_wait(last)
        out.append((time.perf_counter() - t0) / N * 1e6)
    ctx.resident_end()
    return statistics.median(out), info, ctx.resident_info()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lenses", default="panini,hammer")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--flags", default="0,4,16384,8192")
    args = ap.parse_args()
    W, H = [int(v) for v in args.size.split("x")]
    for lens in args.lenses.split(","):
        wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None if lens != "panini" else "f_fov 180", W, H, 1)
        ctx = wl.ctx
        for i in range(3):
            wl.launch(i, 1)
        torch.cuda.synchronize()
        for flags in [int(v) for v in args.flags.split(",")]:
            ctx.set_ablation(flags)
            us, info, after = pipelined(ctx, wl, W, args.frames)
            extra = ""
            if flags & 8192:
                n = args.frames * 4 + 1
                lo, med, hi = after["streamed"]
                extra = f" busy us/frame per workgroup: min {lo / 100 / n:.2f} median {med / 100 / n:.2f} max {hi / 100 / n:.2f}"
            print(f"DIAG {lens:12s} {W}x{H} flags {flags:6d}: pipelined {us:6.2f} us/frame  [{info['workgroups']} wgs, K {info['blocks_in_registers']}, "
                  f"nq {info['chunks_per_thread']}, 128x{info['block_h']}, {info['per_cu']}/CU]{extra}", flush=True)
        ctx.set_ablation(0)
        wl.close()


if __name__ == "__main__":
    main()
