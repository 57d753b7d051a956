import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
W, H = 3840, 2160
for rubix in (True, False):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 1, rubix=rubix)
    for kb in (0, 13, 14, 15, 16, 17, 18, 19, 20):
        wl.ctx.set_tile_shape(4)
        wl.ctx.set_tile_shape(400 + kb)
        for i in range(3): wl.launch(i, 1)
        k1 = wl.kernel_ms(nframes=1, launches=30, repeats=7)[0]
        print("RUBIX" if rubix else "PLAIN", "kb", kb, "single launch", round(k1 * 1e3, 2), wl.ctx.tile_stats()["lds_bytes_per_wave"], flush=True)
    wl.close()
