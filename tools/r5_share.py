"""What a reserved workgroup place per CU costs the resident frame, and what another context's kernels take beside the resident kernel
(profiles/r05_resident_apply.txt (8)).  Developer probe; GPU box only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402

W, H, N = 3840, 2160, 600
b = blinky_amd.Context()
S.configure(b, "cube", "panini", None, (1920, 1080))
b.build()
b.fill_plate_lcg(0, 0, seed_frame=1)
b.apply(np.zeros((1080, 1920), np.uint8))
for lens in ("panini", "hammer"):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None if lens != "panini" else "f_fov 180", W, H, 1)
    a = wl.ctx
    outs = [wl.origin(o) for o in wl.out]
    for reserve in (0, 1, 2):
        a.set_resident_share(0, 1, reserve)
        a.synchronize()
        torch.cuda.synchronize()
        a.resident_begin(idle_ms=3000)
        a.resident_wait(a.resident_submit(outs[0], W, frame=0))
        pipe = []
        for rep in range(5):
            t0 = time.perf_counter()
            a.resident_wait(a.resident_submit_batch(outs[0], W, 0, frame0=(rep * N) % wl.R, nframes=N))
            pipe.append((time.perf_counter() - t0) / N * 1e6)
        info = a.resident_info()
        line = f"{lens:8s} reserve {reserve}: {min(pipe):6.2f} us/frame  [{info['workgroups']} wgs, {info['blocks_in_registers']} blocks, {info['per_cu']}/CU]"
        if reserve:
            ts = []
            for k in range(5):
                t0 = time.perf_counter()
                for p in range(6):
                    b.fill_plate_lcg(0, p, seed_frame=5 + k)
                b.apply(np.zeros((1080, 1920), np.uint8))
                ts.append(1e3 * (time.perf_counter() - t0))
            line += f" | another context, 6 plate fills + 1080p apply + copy back: {min(ts):.2f} ms beside it (launches {a.resident_info()['launches']})"
        print(line, flush=True)
        a.resident_end()
    a.set_resident_share(0, 1, 0)
    wl.close()
ts = []
for k in range(5):
    t0 = time.perf_counter()
    for p in range(6):
        b.fill_plate_lcg(0, p, seed_frame=5 + k)
    b.apply(np.zeros((1080, 1920), np.uint8))
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"the other context alone: {min(ts):.2f} ms")
b.close()
