#!/usr/bin/env python3
"""Developer probe: frames per block visit (the frame-group split of a batch launch; knob 300+n of bk_debug_set_tile_shape) against the
launch time: more groups = more workgroups (a 1080p batch is one half-filled round) but the block map is read once per group.
usage: python tools/fchunk_probe.py [--configs 1080p-stereographic,...] [--fchunks 0,2,4,16]     (0 = the library's rule: 8)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402

CONFIGS = {
    "1080p-stereographic": ("cube", "stereographic", None, 1920, 1080),
    "1080p-hammer": ("cube", "hammer", None, 1920, 1080),
    "1080p-panini": ("cube", "panini", "f_fov 180", 1920, 1080),
    "4k-panini": ("cube", "panini", "f_fov 180", 3840, 2160),
    "4k-hammer": ("cube", "hammer", None, 3840, 2160),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1080p-stereographic,1080p-hammer,1080p-panini,4k-panini,4k-hammer")
    ap.add_argument("--fchunks", default="0,2,4,16")
    ap.add_argument("--frames", type=int, default=16)
    args = ap.parse_args()
    for name in args.configs.split(","):
        globe, lens, zoom, W, H = CONFIGS[name]
        for fc in [int(v) for v in args.fchunks.split(",")]:
            wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, args.frames)
            wl.ctx.set_tile_shape(300 + fc)           # (300 = the rule)
            wl.ctx.set_tile_shape(0)                  # drop the block map: the next launch compiles and tunes it under this split
            for i in range(3):
                wl.launch(i)
            med, lo, _ = wl.kernel_ms(launches=40 if W < 3000 else 20, repeats=7)
            job = wl.job_seconds_per_step(steps=50, repeats=9, nstreams=2)
            st = wl.ctx.tile_stats()
            print(f"{name:22s} x{args.frames} frames/visit {fc or 8:2d}: {med * 1e3 / args.frames:7.3f} us/frame (min {lo * 1e3 / args.frames:.3f})  "
                  f"two-stream job {job * 1e6 / args.frames:7.3f} us/frame  [128x{st['tile_h'] % 1000} lds {st['lds_bytes_per_wave'] // 1024}K {st['tiles']} blocks]", flush=True)
            wl.close()


if __name__ == "__main__":
    main()
