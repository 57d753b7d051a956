"""The forward build at 7680x4320 (28 M -> 112 M texel corners; C5's size with a forward-map lens): the GPU table against the library's own
host build of the same lens (bk_debug_host_build on a device-less context, the worker pool: tests/test_host_path_cpu.py pins that path to
the reference's goldens), entry for entry, and its time.  Developer probe; GPU box only (the host build wants its cores)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import blinky_amd as bk
import scripts as S

W, H = 7680, 4320
for lens in sys.argv[1:] or ["eckert5", "winkel2"]:
    ctx = bk.Context()
    S.configure(ctx, "cube", lens, None, (W, H))
    ctx.build()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        display, scale = ctx.build()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    device_ms = ctx.last_build_ms()
    off, tin = ctx.read_lensmap()
    flagged = ctx.last_build_fixups()
    ctx.close()
    host = bk.Context(bk.ffi.DEVICE_NONE)
    S.configure(host, "cube", lens, None, (W, H))
    t0 = time.perf_counter()
    hoff, htin, hdisplay, hscale, err = host.host_build(1)
    th = time.perf_counter() - t0
    host.close()
    same = err is None and np.array_equal(off, hoff) and np.array_equal(tin, htin) and list(display) == list(hdisplay) and scale == hscale
    print(f"{lens} {W}x{H}: bk_build {best:.3f} ms (device part {device_ms:.3f}), flagged / changed {flagged}, mapped {int((off != 0xFFFFFFFF).sum())} of {W * H}; "
          f"host pool build {th:.1f} s; tables identical: {same}", flush=True)
    assert same
