#!/usr/bin/env python3
"""Developer probe for counter passes: ONE resident-apply session of N pipelined frames over the cold ring, then exit - the
session's kernel is one dispatch, so `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/resident_traffic.py hammer 2000`
gives bytes per N frames (tools/resident_pmc.sh runs the passes and divides).
usage: python tools/resident_traffic.py <lens> <frames> [W H] [flags] [shape]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402

lens, N = sys.argv[1], int(sys.argv[2])
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None if lens != "panini" else "f_fov 180", W, H, 1)
if len(sys.argv) > 5:
    wl.ctx.set_ablation(int(sys.argv[5]))
if len(sys.argv) > 6:
    wl.ctx.set_tile_shape(int(sys.argv[6]))           # 1 / 2 / 4: blocks of 128 x 8 / 16 / 32
wl.launch(0, 1)
if os.environ.get('RES_DEBUG'):
    blinky_amd.ffi.debug_set_option('print_model', 1)
torch.cuda.synchronize()
wl.ctx.resident_begin(idle_ms=100)
info = wl.ctx.resident_info()
t0 = time.perf_counter()
last = wl.ctx.resident_submit_batch(wl.origin(wl.out[0]), W, 0, frame0=0, nframes=N)
wl.ctx.resident_wait(last)
dt = time.perf_counter() - t0
wl.ctx.resident_end()
info = wl.ctx.resident_info()
print(f"RESIDENT {lens} {W}x{H} frames {N} us_per_frame {dt / N * 1e6:.3f} info {info}", flush=True)
wl.close()
