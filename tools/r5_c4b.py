import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
W, H, F = 3840, 2160, 64
for globe, lens, zoom in (("trism", "panini", "f_fov 180"), ("cube", "panini", "f_fov 180"), ("cube", "hammer", None)):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, ring_max=32)
    out = []
    for knob in (16, 24, 32, 48, 64):
        wl.ctx.set_tile_shape(100 + knob)
        for i in range(3): wl.launch(i)
        out.append((knob, round(wl.kernel_ms(launches=10, repeats=5)[0] * 1e3, 1)))
    print("C4B", globe, lens, "whole frame x64, us per launch by persistent-grid workgroups per CU:", out, flush=True)
    wl.close()
