import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_ffi as O, blinky_amd as bk, scripts as S
lm = O.lensmap("cube", "hammer", None, 960, 540)
W, H = lm.W, lm.H
m = bk.Multi([0, 0, 0])
m.load_globe(S.script("globes", "cube"), "cube"); m.load_lens(S.script("lenses", "hammer"), "hammer")
m.set_zoom(*S.zoom_args(m.ctx(0).lens_info().onload.decode())); m.resize(W, H); m.build()
m.set_resident_apply(True)
globe = O.lcg_globe(lm.ps, 6, 40)
for i in range(8):
    t0 = time.time()
    for p in range(6): m.upload_plate(0, p, globe[p])
    t1 = time.time()
    got = m.apply(np.zeros((H, W), np.uint8))
    t2 = time.time()
    print(i, f"upload {1e3*(t1-t0):.1f} ms apply {1e3*(t2-t1):.1f} ms", [(m.ctx(k).resident_info()["launches"], m.ctx(k).resident_info()["running"], m.ctx(k).resident_info()["workgroups"]) for k in range(3)], flush=True)
m.close()
