"""The drop-in's per-frame calls at 4K, PCIe included: six bk_upload_plate(_async) + bk_apply into a host frame, launch per frame against the
resident mode (profiles/r05_resident_apply.txt (10), profiles/r06_dropin_rate.txt).  Developer probe; GPU box only."""
import sys, time
import os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, blinky_amd as bk, scripts as S
import oracle_ffi as O
W, H = 3840, 2160
ctx = bk.Context()
S.configure(ctx, "cube", "panini", None, (W, H))
ctx.build()
globes = [O.lcg_globe(H, 6, f) for f in range(3)]
frame = np.zeros((H, W), np.uint8)
for mode in ("launch per frame", "resident", "resident, reserve 1"):
    ctx.set_resident_share(0, 1, 1 if "reserve" in mode else 0)
    ctx.set_resident_apply(mode != "launch per frame")
    for asyn in (False, True):
        up = ctx.upload_plate_async if asyn else ctx.upload_plate
        ts, tu, ta = [], [], []
        for i in range(12):
            g = globes[i % 3]
            t0 = time.perf_counter()
            for p in range(6): up(0, p, g[p])
            t1 = time.perf_counter()
            ctx.apply(frame)
            t2 = time.perf_counter()
            ts.append(t2 - t0); tu.append(t1 - t0); ta.append(t2 - t1)
        n = 4
        print(f"{mode:19s} {'async' if asyn else 'blocking':8s} uploads: frame {1e3 * min(ts[n:]):.2f} ms = 6 uploads {1e3 * min(tu[n:]):.2f} + bk_apply {1e3 * min(ta[n:]):.2f}  -> {W * H / min(ts[n:]) / 1e9:.1f} Gpx/s PCIe-inclusive", flush=True)
    assert O.fnv(frame) is not None
ctx.close()
