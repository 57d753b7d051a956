#!/usr/bin/env python3
"""Developer probe (not the bench): time the apply kernel variants on oracle-built lensmaps.
usage: python tools/apply_probe.py [lens] [W] [H] [frames] [variant...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import blinky_amd
import oracle_ffi as O

lens = sys.argv[1] if len(sys.argv) > 1 else "panini"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
F = int(sys.argv[4]) if len(sys.argv) > 4 else 16
variants = [int(v) for v in sys.argv[5:]] or [2]

t = time.time()
GLOBE = os.environ.get("BK_GLOBE", "cube")
lm = O.lensmap(GLOBE, lens, None, W, H)
print(f"oracle lensmap {lens} {W}x{H}: {time.time()-t:.2f}s nonnull={lm.nonnull}", flush=True)
RING = int(os.environ.get("BK_RING", str(F)))       # resident globes (1 = every frame reads the same globe)
SAMEOUT = int(os.environ.get("BK_SAMEOUT", "0"))       # 1 = every frame writes the same output buffer
ctx = blinky_amd.Context()
ctx.set_frames(RING)
ctx.resize(W, H)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for f in range(RING):
    for p in range(6):
        ctx.fill_plate_lcg(f, p, f)
ctx.set_lensmap(lm.offsets, lm.tints)
out = torch.zeros((F, H, W), dtype=torch.uint8, device="cuda")
ABL = [int(x) for x in os.environ.get("BK_ABLATE", "0").split(",")]
SHAPES = [int(x) for x in os.environ.get("BK_SHAPES", "0").split(",")]
WGS = [int(x) for x in os.environ.get("BK_WGS", "6").split(",")]
FCH = [int(x) for x in os.environ.get("BK_FCHUNK", "0").split(",")]      # frames per tile visit (0 = default)
REPS = int(os.environ.get("BK_REPS", "20"))
LDSKB = [int(x) for x in os.environ.get("BK_LDSKB", "0").split(",")]    # coop apply: staging buffer KiB (0 = cost model)
BCOST = [int(x) for x in os.environ.get("BK_BCOST", "-1").split(",")]   # band balance: constant cost per block in lines (-1 = default)
for v, abl, shp, wg, fc, kb, bc in [(v, a, sh, wg, fc, kb, bc) for v in variants for sh in (SHAPES if v != 0 else [0]) for a in (ABL if v != 0 else [0]) for wg in (WGS if v != 0 else [6])
                                   for fc in (FCH if v != 0 else [0]) for kb in (LDSKB if v == 2 else [0]) for bc in (BCOST if v == 2 else [-1])]:
    ctx.set_apply_variant(v)
    if v != 0:
        ctx.set_tile_shape(shp)
        ctx.set_tile_shape(100 + wg)
        ctx.set_ablation(abl)             # 2 no globe loads, 4 no stores, 8 no load pipelining
        ctx.set_tile_shape(300 + fc)
        ctx.set_tile_shape(400 + kb)
        ctx.set_tile_shape(601 + bc)
        print(f"shape {shp} ablation {abl} wgs/cu {wg} fchunk {fc} ldskb {kb} bcost {bc}; tile stats:", ctx.tile_stats(), flush=True)
        if os.environ.get("BK_MODEL"):
            m = ctx.traffic_model()
            print(f"   model: unique lines {m['unique_globe_lines']} ({m['unique_globe_lines'] * 128 / 1e6:.1f} MB/frame), staged lines {m['staged_lines']} "
                  f"(x{m['staged_lines'] / max(1, m['unique_globe_lines']):.2f}), staged chunks {m['staged_chunks']} ({m['staged_chunks'] / max(1, m['staged_lines']):.2f} per staged line), mapped px {m['mapped_pixels']}, block map {m['blockmap_bytes_per_visit'] / 1e6:.1f} MB/visit", flush=True)
    for nf in sorted(set([1, F])):
        for _ in range(3):
            ctx.apply_device(out.data_ptr(), W, H * W, 0, nf)
        torch.cuda.synchronize()
        reps = REPS
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            ctx.apply_device(out.data_ptr(), W, 0 if SAMEOUT else H * W, (r * nf) % RING, nf)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        px = W * H * nf
        print(f"variant {v} frames/launch {nf}: {ms*1e3/nf:.2f} us/frame  {px/ms/1e3:.0f} Mpx/s  "
              f"algorithmic {6*px/ms/1e9:.2f} TB/s = {6*px/ms/1e9/8*100:.1f}% of 8 TB/s", flush=True)
