"""64-frame launches of the headline workload by frames per block visit (developer knob 300 + n) and workgroups per CU's worth of grid
(100 + n).  Developer probe; GPU box only."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
W, H = 3840, 2160
for lens, zoom in (("panini", "f_fov 180"), ("hammer", None)):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, zoom, W, H, 64)
    for i in range(2): wl.launch(i)
    base = wl.kernel_ms(launches=12, repeats=5)[0]
    print(f"{lens}: as tuned {base * 1e3:.1f} us per 64-frame launch ({base * 1e3 / 64:.3f} us/frame)", wl.ctx.tile_stats()["tile_h"] % 1000, flush=True)
    for fc in (4, 8, 16, 32, 64):
        wl.ctx.set_tile_shape(300 + fc)
        for i in range(2): wl.launch(i)
        k = wl.kernel_ms(launches=12, repeats=5)[0]
        print(f"   frames per visit {fc:2d}: {k * 1e3:.1f} us ({k * 1e3 / 64:.3f} us/frame)", flush=True)
    wl.ctx.set_tile_shape(300)
    wl.close()
