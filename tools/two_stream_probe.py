#!/usr/bin/env python3
"""Developer probe: does alternating two HIP streams between consecutive batch launches hide the tail of each launch
(the last, partly filled round of workgroups) behind the ramp of the next?   python tools/two_stream_probe.py [lens] [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import blinky_amd
import oracle_ffi as O

lens = sys.argv[1] if len(sys.argv) > 1 else "panini"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H, RING, NB = 3840, 2160, 64, 4
lm = O.lensmap("cube", lens, None, W, H)
ctx = blinky_amd.Context()
ctx.set_frames(RING)
ctx.resize(W, H)
for f in range(RING):
    for p in range(6):
        ctx.fill_plate_lcg(f, p, f)
ctx.set_lensmap(lm.offsets, lm.tints)
outs = [torch.zeros((F, H, W), dtype=torch.uint8, device="cuda") for _ in range(NB)]
streams = [torch.cuda.Stream() for _ in range(4)]
ctx.set_stream(streams[0].cuda_stream)
ctx.set_tile_shape(300 + int(os.environ.get("BK_FCHUNK", "0")))      # frames per block visit (0 = default 8)
ctx.tile_stats()
ctx.apply_device(outs[0].data_ptr(), W, H * W, 0, F)
torch.cuda.synchronize()
for ns in (1, 2, 2):
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 200
        e0.record(streams[0])
        for s in streams[1:ns]:
            s.wait_event(e0)
        for i in range(steps):
            ctx.set_stream(streams[i % ns].cuda_stream)
            ctx.apply_device(outs[i % NB].data_ptr(), W, H * W, (i * F) % RING, F)
        for s in streams[1:ns]:
            ev = torch.cuda.Event()
            ev.record(s)
            streams[0].wait_event(ev)
        e1.record(streams[0])
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        print(f"{lens} x{F}: {ns} stream(s): {ms * 1e3 / F:.3f} us/frame  {W * H * F / ms / 1e3:.0f} Mpx/s", flush=True)
