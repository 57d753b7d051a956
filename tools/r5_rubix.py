"""The headline with rubix on (tint LUTs): batch, single launch, resident - and what the tint itself costs (developer bit 65536: the
chunks are staged untinted, timing only).  Developer probe; GPU box only."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
W, H = 3840, 2160
for abl in (0, 65536):
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 16, rubix=True)
    wl.ctx.set_ablation(abl)
    for i in range(3): wl.launch(i)
    k16 = wl.kernel_ms(launches=20, repeats=7)[0]
    k1 = wl.kernel_ms(nframes=1, launches=30, repeats=7)[0]
    st = wl.ctx.tile_stats()
    r = wl.resident_us(frames=400) if abl == 0 else {"us": 0, "one_at_a_time_host_us": 0, "one_at_a_time_device_us": 0}
    print(f"RUBIX 4K panini (ablation {abl}, 128x{st['tile_h'] % 1000}): x16 {k16 * 1e3 / 16:.2f} us/frame, single launch {k1 * 1e3:.2f} us, resident pipelined {r['us']:.2f} us, one at a time host {r['one_at_a_time_host_us']:.2f} device {r['one_at_a_time_device_us']:.2f}", flush=True)
    wl.close()
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 16, rubix=False)
for i in range(3): wl.launch(i)
k16 = wl.kernel_ms(launches=20, repeats=7)[0]
k1 = wl.kernel_ms(nframes=1, launches=30, repeats=7)[0]
print(f"PLAIN 4K panini: x16 {k16 * 1e3 / 16:.2f} us/frame, single launch {k1 * 1e3:.2f} us", flush=True)
wl.close()
