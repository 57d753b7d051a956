import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import blinky_amd, oracle_ffi as O
lib = C.CDLL(os.environ["BLINKY_HIP_LIB"])
lens, W, H, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
RING = 64
lm = O.lensmap("cube", lens, None, W, H)
ctx = blinky_amd.Context(); ctx.set_frames(RING); ctx.resize(W, H); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for f in range(RING):
    for p in range(6): ctx.fill_plate_lcg(f, p, f)
ctx.set_lensmap(lm.offsets, lm.tints)
out = torch.zeros((F, H, W), dtype=torch.uint8, device="cuda")
ctx.set_tile_shape(116)
for shp in [int(x) for x in os.environ.get("BK_SHAPES", "0").split(",")]:
  ctx.set_tile_shape(shp)
  for nf in (1, F):
    for _ in range(3): ctx.apply_device(out.data_ptr(), W, H * W, 0, nf)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    lib.bk_debug_trace(buf, 1)
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): ctx.apply_device(out.data_ptr(), W, H * W, (r * nf) % RING, nf)
    e1.record(); torch.cuda.synchronize()
    lib.bk_debug_trace(buf, 0)
    n = max(1, buf[4])
    us = e0.elapsed_time(e1) / reps * 1e3 / nf
    print(f"{lens} shape {shp} frames {nf}: {us:.2f} us/frame (instrumented); tile stats {ctx.tile_stats()}")
    print(f"   per block-frame and wave (ticks of s_memrealtime = 10 ns): wait loads + LDS write {buf[0]/n:.1f}, barrier 1 {buf[1]/n:.1f}, issue next loads + gather + store {buf[2]/n:.1f}, barrier 2 {buf[3]/n:.1f}; samples {n}")
