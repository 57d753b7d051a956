"""Developer probe (round 5): the 64-frame C4 launch (3840x2160 trism/panini) whole and as its two cost-balanced halves,
for counter passes (tools/r5_c4_pmc.sh): same block height (128x32), no measured tuning (its timing launches would mix in)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, blinky_amd, scripts as S
from blinky_amd import ffi
W, H, F, R = 3840, 2160, 64, 64
globe, lens, zoom = "trism", "panini", "f_fov 180"
full = blinky_amd.Context(0); S.configure(full, globe, lens, zoom, (W, H)); full.build(); cost = full.row_costs(); full.close()
b2 = ffi.stripe_bounds_from_costs(cost, 0, 2)
for rows in [(0, H), (b2[0], b2[1]), (b2[1], b2[2])]:
    ctx = blinky_amd.Context(0)
    ctx.set_blockmap_tuning(False)
    ctx.set_tile_shape(4)
    ctx.set_frames(R)
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.set_rows(*rows)
    ctx.build()
    for f in range(R):
        for p in range(5):
            ctx.fill_plate_lcg(f, p, f)
    n = rows[1] - rows[0]
    out = torch.zeros((F, n, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for i in range(6):
        ctx.apply_device(out.data_ptr() - rows[0] * W, W, n * W, frame0=0, nframes=F)
    torch.cuda.synchronize()
    ctx.close()
