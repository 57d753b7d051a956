import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, blinky_amd as bk, scripts as S
import oracle_ffi as O
pal = O.palmap(O.synthetic_basepal())
lm = O.lensmap("cube", "panini", None, 640, 480)
W, H = lm.W, lm.H
ctx = bk.Context(); ctx.set_frames(1); ctx.resize(W, H)
ctx.set_lensmap(lm.offsets, lm.tints)
ctx.set_resident_apply(True)
pitch, x0, y0 = W + 8, 3, 2
bg = np.zeros((H + 4, pitch), np.uint8)
globe = O.lcg_globe(lm.ps, 6, 100)
for p in range(6): ctx.upload_plate(0, p, globe[p])
def step(tag, rubix):
    t0 = time.perf_counter()
    ctx.apply_begin(0, rubix, pal if rubix else None)
    t1 = time.perf_counter()
    i1 = ctx.resident_info()
    t2 = time.perf_counter()
    ctx.apply_end(bg.copy(), pitch, x0, y0)
    t3 = time.perf_counter()
    i2 = ctx.resident_info()
    keys = ("running", "workgroups", "per_cu", "launches")
    print(tag, "begin %.0f us, end %.0f us" % ((t1 - t0) * 1e6, (t3 - t2) * 1e6), {k: i1[k] for k in keys}, {k: i2[k] for k in keys}, flush=True)
for i in range(3): step("plain", False)
for i in range(4): step("rubix", True)
for i in range(2): step("plain", False)
ctx.close()
