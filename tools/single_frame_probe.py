#!/usr/bin/env python3
"""Developer probe: single-frame and 16-frame launch times of the staged apply under ablation bits, several lenses.
usage: python tools/single_frame_probe.py [--lenses a,b,c] [--flags 0,128] [--size 3840x2160] [--shapes 0]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lenses", default="panini,stereographic,hammer,mercator,gumby,quincuncial")
    ap.add_argument("--flags", default="0,128")
    ap.add_argument("--shapes", default="0")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--frames", default="1,16")
    ap.add_argument("--no-tuning", action="store_true", help="block height by the cost model alone (bk_set_blockmap_tuning 0)")
    args = ap.parse_args()
    W, H = [int(v) for v in args.size.split("x")]
    for lens in args.lenses.split(","):
        wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None if lens != "panini" else "f_fov 180", W, H, 16)
        wl.ctx.set_blockmap_tuning(not args.no_tuning)
        for shape in [int(v) for v in args.shapes.split(",")]:
            wl.ctx.set_tile_shape(shape)
            for flags in [int(v) for v in args.flags.split(",")]:
                wl.ctx.set_ablation(flags)
                line = f"{lens:14s} {W}x{H} shape {shape} flags {flags:4d}"
                for nf in [int(v) for v in args.frames.split(",")]:
                    wl.ctx.set_tile_shape(shape)                 # drops the block map: the next launch compiles (and tunes) it for nf frames
                    for i in range(3):
                        wl.launch(i, nf)
                    stats = wl.ctx.tile_stats()
                    med, lo, hi = wl.kernel_ms(nframes=nf, launches=40, repeats=7)
                    line += f" | x{nf}: {med * 1e3 / nf:7.3f} us/frame (min {lo * 1e3 / nf:.3f}) [128x{stats['tile_h'] % 1000} lds {stats['lds_bytes_per_wave'] // 1024}K]"
                print(line, flush=True)
        wl.ctx.set_ablation(0)
        wl.close()


if __name__ == "__main__":
    main()
