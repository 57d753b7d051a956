"""Where does the millisecond launch in the rubix train come from?  The bench's rubix line - 4K cube/panini, rubix
on, 16 frames per launch, 5 warm-up launches, then trains of 20 - with EVERY launch timed on both clocks: host wall time of the
bk_apply_device call, and HIP events around the launch on its stream.  Prints the outliers and what the context says about its
block map before / after them.  Developer probe; GPU box only."""
import statistics
import sys
import time

import os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S

W, H, F = 3840, 2160, 16
rubix = "--plain" not in sys.argv
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", "panini", "f_fov 180", W, H, F, rubix=rubix)
print("after the workload is set up:", wl.ctx.tile_stats(), flush=True)
import gc
gc_log = []
def _gc_cb(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - _t[0]) * 1e6, info["collected"]))
gc.callbacks.append(_gc_cb)
if "--no-gc" in sys.argv:
    gc.disable()
if "--like-bench" in sys.argv:
    # exactly bench.extra_config's sequence: 5 warm-up launches, then kernel_ms's repeats - with the host time of every call kept
    for i in range(5):
        wl.launch(i)
    wl.ctx.set_stream(wl.stream.cuda_stream)
    gc_log.clear()
    for rep in range(int(os.environ.get("REPEATS", "9"))):
        torch.cuda.synchronize()
        calls = []
        wl.e0.record(wl.stream)
        for i in range(20):
            t0 = time.perf_counter()
            wl.launch(i)
            calls.append((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter()
        wl.e1.record(wl.stream)
        torch.cuda.synchronize()
        tsync = (time.perf_counter() - t0) * 1e6
        print(f"repeat {rep}: {wl.e0.elapsed_time(wl.e1) * 1e3 / 20:.1f} us per launch by events; host calls max {max(calls):.1f} us (call {calls.index(max(calls))}), "
              f"sum {sum(calls):.0f} us, final synchronize {tsync:.0f} us; garbage collections during it (generation, us, collected): {[(g, round(u), c) for g, u, c in gc_log]}", flush=True)
        gc_log.clear()
    wl.close()
    sys.exit(0)
stream = wl.stream
N = 5 + 9 * 20 + 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record(stream)
for i in range(N):
    t0 = time.perf_counter()
    wl.launch(i)
    host.append((time.perf_counter() - t0) * 1e6)
    ev[i + 1].record(stream)
    if i == 4 or (i > 4 and (i - 4) % 20 == 0):
        torch.cuda.synchronize()                      # the bench synchronizes between its trains
torch.cuda.synchronize()
dev = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
med_d, med_h = statistics.median(dev[5:]), statistics.median(host[5:])
print(f"{'rubix' if rubix else 'plain'} x{F}: device median {med_d:.1f} us, host call median {med_h:.1f} us over {N - 5} launches after 5 warm-ups")
print("first 8 launches: device", [round(d, 1) for d in dev[:8]], "host", [round(h, 1) for h in host[:8]])
for i in range(5, N):
    if dev[i] > 1.5 * med_d or host[i] > max(10 * med_h, 200):
        print(f"  OUTLIER launch {i}: device {dev[i]:.1f} us, host call {host[i]:.1f} us (previous: device {dev[i - 1]:.1f}, host {host[i - 1]:.1f})")
print("at the end:", wl.ctx.tile_stats(), flush=True)
# the same trains the way bench.extra_config times them (events around 20 launches, 9 repeats): min / median / max per launch
k = wl.kernel_ms(launches=20, repeats=9)
print(f"kernel_ms(launches=20, repeats=9): median {k[0] * 1e3:.1f} min {k[1] * 1e3:.1f} max {k[2] * 1e3:.1f} us per launch")
wl.close()
