import sys, time, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
from blinky_amd import ffi
W, H = 3840, 2160
full = blinky_amd.Context(0); S.configure(full, "cube", "panini", "f_fov 180", (W, H)); full.build(); cost = full.row_costs(); full.close()
for n in (1, 2, 4, 8):
    bounds = ffi.stripe_bounds_from_costs(cost, 0, n) if n > 1 else [0, H]
    res = []
    for r in range(n):
        for flags in (0, 16):
            wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 1, rows=(bounds[r], bounds[r + 1]), ring_max=32)
            wl.ctx.set_ablation(flags)
            x = wl.resident_us(frames=300)
            res.append((r, flags, x["us"], x["workgroups"], x["blocks_in_registers"]))
            wl.close()
        if r >= 1: break
    print("STRIPES N", n, res, flush=True)
