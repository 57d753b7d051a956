#!/usr/bin/env python3
"""Developer probe: lensmap BUILD kernel time (bk_last_build_ms) of every bundled lens at a given size, the APPLY
time of the map it built (16-frame and single-frame launches over a cold ring of 64 globes), and the host-buffer
(PCIe-inclusive) cost of the drop-in calls bk_upload_plate / bk_apply.
usage: python tools/build_times.py [W] [H]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import blinky_amd
import scripts as S

W = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
RING, F = 64, 16
ctx = blinky_amd.Context()
ctx.set_frames(RING)
ctx.resize(W, H)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for f in range(RING):
    for p in range(6):
        ctx.fill_plate_lcg(f, p, f)
out = torch.zeros((F, H, W), dtype=torch.uint8, device="cuda")


def apply_us(nf, reps=60):
    for _ in range(3):
        ctx.apply_device(out.data_ptr(), W, H * W, 0, nf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        ctx.apply_device(out.data_ptr(), W, H * W, (r * nf) % RING, nf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / nf


rows = []
for lens in S.names("lenses"):
    try:
        info = S.configure(ctx, "cube", lens, None, (W, H))
        t0 = time.time()
        ctx.build()
        first = (time.time() - t0) * 1e3
        t0 = time.time()
        display, scale = ctx.build()
        wall = (time.time() - t0) * 1e3
        st = ctx.tile_stats()
        kms = ctx.last_build_ms()
        a16 = apply_us(F)                   # (the block map is measured for the first launch after a build: 16 frames here ...)
        st = ctx.tile_stats()
        ctx.set_tile_shape(0)               # ... and measured again for single-frame launches, as an engine context would have it
        a1 = apply_us(1)
        st1 = ctx.tile_stats()
        rows.append((lens, info.map_type, kms, wall, first, sum(display)))
        print(f"{lens:16s} map {info.map_type} build kernel {kms:8.3f} ms  wall {wall:8.2f} ms  first (hiprtc) {first:8.1f} ms  plates {sum(display)}  "
              f"apply x{F} {a16:6.2f} us/frame  x1 {a1:6.2f} us  blocks {st['tiles']} h {st['tile_h'] - 128000} slow {st['slow']} empty {st['empty']} lds {st['lds_bytes_per_wave']} | x1 map: h {st1['tile_h'] - 128000} lds {st1['lds_bytes_per_wave']}", flush=True)
    except Exception as e:      # a lens without a usable default zoom at this size
        print(f"{lens:16s} {type(e).__name__}: {str(e)[:100]}", flush=True)

# drop-in (host buffers): 6 plate uploads + one apply into a host frame, per frame
S.configure(ctx, "cube", "panini", "f_fov 180", (W, H))
ctx.build()
ps = min(W, H)
plates = np.random.default_rng(1).integers(0, 256, (6, ps, ps), dtype=np.uint8)
frame = np.zeros((H, W), np.uint8)
for rep in range(2):
    t0 = time.time()
    for p in range(6):
        ctx.upload_plate(0, p, plates[p])
    t1 = time.time()
    ctx.apply(frame, 0, W, 0, 0, False, None)
    t2 = time.time()
print(f"drop-in {W}x{H}: 6 x bk_upload_plate {(t1-t0)*1e3:.2f} ms, bk_apply to host {(t2-t1)*1e3:.2f} ms -> {W*H/(t2-t0)/1e6:.0f} Mpx/s PCIe-inclusive", flush=True)
