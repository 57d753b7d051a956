"""(r6) one frame at a time through the resident kernel: submit + wait on the C host's clock (bk_debug_resident_latency) and the device's own
figure, per lens.  Developer probe; GPU box only."""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S

W, H = 3840, 2160
for lens in sys.argv[1:] or ["panini", "hammer", "quincuncial"]:
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None, W, H, 1)
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    res = []
    for rep in range(3):
        wl.ctx.resident_begin(idle_ms=200)
        wl.ctx.resident_wait(wl.ctx.resident_submit(out.data_ptr(), W, frame=0))
        res.append(wl.ctx.resident_latency(out.data_ptr(), W, frames=300, globes=wl.R))
        wl.ctx.resident_end()
    print(f"{W}x{H} cube/{lens}: one at a time, host us / device us / difference: " + "  ".join(f"{h:.2f} / {d:.2f} / {h - d:.2f}" for h, d in res), flush=True)
    wl.close()
