import sys, time, statistics, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
from blinky_amd import ffi
W, H, F = 3840, 2160, 64
globe, lens, zoom = sys.argv[1:4] if len(sys.argv) > 3 else ("trism", "panini", "f_fov 180")
zoom = zoom or None
full = blinky_amd.Context(0); S.configure(full, globe, lens, zoom, (W, H)); full.build(); cost = full.row_costs(); full.close()
b2 = ffi.stripe_bounds_from_costs(cost, 0, 2)
for rows in [(0, H), (b2[0], b2[1]), (b2[1], b2[2])]:
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, rows=rows, ring_max=32)
    out = []
    for shape in (0, 1, 2, 4):
        wl.ctx.set_tile_shape(shape)
        for i in range(3): wl.launch(i)
        ms = wl.kernel_ms(launches=10, repeats=5)[0]
        st = wl.ctx.tile_stats()
        out.append((shape, st["tile_h"] % 1000, round(ms * 1e3, 1)))
    print("C4PROBE", globe, lens, "rows", rows, "us per 64-frame launch by shape (0 = tuner):", out, flush=True)
    wl.close()
