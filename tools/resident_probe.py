#!/usr/bin/env python3
"""Developer probe: the resident single-frame apply against one launch per frame, over the cold ring of globes.
For every lens: (a) launches: HIP events around a train of single-frame bk_apply_device launches (what `roofline.single_frame` of
bench.py has always been); (b) resident, pipelined: N frames submitted back to back, host wall clock from the first submit to the
last frame complete, per frame; (c) resident, one at a time: submit, wait, submit ... - host wall clock per frame and the device's
own figure (command seen -> frame complete in memory).
usage: python tools/resident_probe.py [--lenses a,b,c] [--size 3840x2160] [--frames 400] [--shape 0|1|2|4]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lenses", default="panini,stereographic,hammer,quincuncial,mercator")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--shape", default="0", help="block height of the staged apply: 0 = the library's choice, 1/2/4 = 128x8/16/32; a comma list tries each")
    ap.add_argument("--globe", default="cube")
    args = ap.parse_args()
    W, H = [int(v) for v in args.size.split("x")]
    N = args.frames
    for lens in args.lenses.split(","):
        wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, args.globe, lens, None if lens != "panini" else "f_fov 180", W, H, 1)
        ctx = wl.ctx
        for shape in [int(v) for v in args.shape.split(",")]:
            ctx.set_tile_shape(shape)
            for i in range(3):
                wl.launch(i, 1)
            med, lo, hi = wl.kernel_ms(nframes=1, launches=100, repeats=7)
            stats = ctx.tile_stats()
            one = []
            for i in range(200):                      # one launch, then wait for it: what a caller that needs every frame before the next pays
                t0 = time.perf_counter()
                wl.launch(i, 1)
                torch.cuda.synchronize()
                one.append((time.perf_counter() - t0) * 1e6)
            outs = [wl.origin(o) for o in wl.out]
            torch.cuda.synchronize()
            ctx.resident_begin(idle_ms=200)
            info = ctx.resident_info()
            # warm up
            ctx.resident_wait(ctx.resident_submit(outs[0], W, frame=0))
            pipe = []
            for rep in range(5):
                t0 = time.perf_counter()
                last = ctx.resident_submit_batch(outs[0], W, 0, frame0=(rep * N) % wl.R, nframes=N)     # (every frame into the same buffer)
                ctx.resident_wait(last)
                pipe.append((time.perf_counter() - t0) / N * 1e6)
            wall, dev = [], []
            for i in range(200):
                t0 = time.perf_counter()
                t = ctx.resident_submit(outs[i % 4], W, frame=(7 * i) % wl.R)
                dev.append(ctx.resident_wait(t))
                wall.append((time.perf_counter() - t0) * 1e6)
            ctx.resident_end()
            print(f"{lens:14s} {W}x{H} 128x{stats['tile_h'] % 1000}: launches {med * 1e3:6.2f} us/frame back to back (min {lo * 1e3:.2f}), {statistics.median(one):6.2f} us launch + synchronize | resident [{info['workgroups']} wgs, "
                  f"{info['blocks_in_registers']} blocks x {info['chunks_per_thread']} chunks in registers, {info['per_cu']}/CU]: pipelined {statistics.median(pipe):6.2f} us/frame "
                  f"(min {min(pipe):.2f}) | one at a time: host {statistics.median(wall):6.2f} us, device {statistics.median(dev):6.2f} us (min {min(dev):.2f})", flush=True)
        wl.close()


if __name__ == "__main__":
    main()
