import sys, time, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
W, H = 3840, 2160
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, 16, rubix=True)
for i in range(3): wl.launch(i)
k16 = wl.kernel_ms(launches=20, repeats=7)[0]
k1 = wl.kernel_ms(nframes=1, launches=30, repeats=7)[0]
r = wl.resident_us(frames=400)
print(f"RUBIX 4K panini: x16 {k16 * 1e3 / 16:.2f} us/frame, single launch {k1 * 1e3:.2f} us, resident pipelined {r['us']:.2f} us, one at a time host {r['one_at_a_time_host_us']:.2f} device {r['one_at_a_time_device_us']:.2f}", flush=True)
wl.close()
