#!/usr/bin/env python3
"""Kernel time per frame against the size of the resident globe ring (bench workload, 16-frame launches): separates the
Infinity-Cache (256 MiB) residency of the globe lines from address-translation reach.  A frame touches 9.7 MB of globe
lines; `huge` additionally runs the 64-globe ring with every globe frame padded to a 2 MiB multiple (bk_set_frames
allocates one block; alignment of the individual frames is what changes).
usage: python tools/ring_sweep.py [lens] [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import blinky_amd
import scripts as S

lens = sys.argv[1] if len(sys.argv) > 1 else "panini"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
F = 16
for ring in (16, 20, 24, 28, 32, 40, 48, 64, 96, 128):
    ctx = blinky_amd.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_frames(ring)
    S.configure(ctx, "cube", lens, None, (W, H))
    ctx.build()
    m = ctx.traffic_model()
    for f in range(ring):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, f)
    outs = [torch.zeros((F, H, W), dtype=torch.uint8, device="cuda") for _ in range(4)]
    for i in range(8):
        ctx.apply_device(outs[i % 4].data_ptr(), W, H * W, frame0=(i * F) % ring, nframes=F)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 40
        for i in range(n):
            ctx.apply_device(outs[i % 4].data_ptr(), W, H * W, frame0=(i * F) % ring, nframes=F)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n / F * 1e3)
    touched = ring * m["unique_globe_lines"] * 128 / 2 ** 20
    print(f"ring {ring:4d} globes: {touched:7.0f} MiB of globe lines cycled, {best:.2f} us/frame", flush=True)
    ctx.close()
    del outs
    torch.cuda.empty_cache()
