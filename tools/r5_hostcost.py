import sys, time, statistics, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
from blinky_amd import ffi
ffi.debug_set_option("no_direct_submit", int(os.environ.get("NODIRECT", "0")))
W, H = 3840, 2160
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 1, rows=(270, 540), ring_max=32)
ctx = wl.ctx
obuf = torch.zeros((8, 270, W), dtype=torch.uint8, device="cuda"); base = obuf.data_ptr() - 270 * W
torch.cuda.synchronize()
ctx.resident_begin(idle_ms=200)
ctx.resident_wait(ctx.resident_submit(base, W, frame=0))
cost = []
for rep in range(20):
    t0 = time.perf_counter()
    for b in range(3):
        last = ctx.resident_submit_batch(base, W, 270 * W, frame0=0, nframes=8)
    t1 = time.perf_counter()
    ctx.resident_wait(last)
    cost.append((t1 - t0) / 24 * 1e6)
print("HOSTCOST nodirect", os.environ.get("NODIRECT", "0"), "submit us per frame (24 frames into an empty window, 3 calls):", round(statistics.median(cost), 3), "info", ctx.resident_info()["workgroups"])
ctx.resident_end(); wl.close()
