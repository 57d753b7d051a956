import sys, time, statistics, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
from blinky_amd import ffi
if os.environ.get("BLINKY_DBG_COPIES"): ffi.debug_set_option("print_model", 1)
ffi.debug_set_option("no_direct_submit", int(os.environ.get("NODIRECT", "0")))
W, H = 3840, 2160
full = blinky_amd.Context(0); S.configure(full, "cube", "panini", "f_fov 180", (W, H)); full.build(); cost = full.row_costs(); full.close()
for n in (1, 8):
    bounds = ffi.stripe_bounds_from_costs(cost, 0, n) if n > 1 else [0, H]
    r = 1 if n > 1 else 0
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 1, rows=(bounds[r], bounds[r + 1]), ring_max=32)
    x = wl.resident_us(frames=600)
    print("STRIPES2 nodirect", os.environ.get("NODIRECT", "0"), "copies", os.environ.get("BLINKY_DBG_COPIES", "8"), "N", n, x["us"], x["one_at_a_time_host_us"], x["one_at_a_time_device_us"], flush=True)
    wl.close()
