import sys, time, statistics, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, bench, blinky_amd, scripts as S
lens = sys.argv[1] if len(sys.argv) > 1 else "panini"
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
blinky_amd.ffi.debug_set_option("no_direct_submit", nd)
if os.environ.get("BLINKY_DBG_COPIES"): blinky_amd.ffi.debug_set_option("print_model", 1)
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None if lens != "panini" else "f_fov 180", 3840, 2160, 1)
ctx = wl.ctx
outs = [wl.origin(o) for o in wl.out]
torch.cuda.synchronize()
ctx.resident_begin(idle_ms=200)
ctx.resident_wait(ctx.resident_submit(outs[0], 3840, frame=0))
wall, dev, sub = [], [], []
for i in range(300):
    t0 = time.perf_counter()
    t = ctx.resident_submit(outs[i % 4], 3840, frame=(7 * i) % wl.R)
    t1 = time.perf_counter()
    dev.append(ctx.resident_wait(t))
    wall.append((time.perf_counter() - t0) * 1e6); sub.append((t1 - t0) * 1e6)
ctx.resident_wait(ctx.resident_submit(outs[0], 3840, frame=0))
ch, cd = ctx.resident_latency(outs[0], 3840, frames=300, globes=wl.R)
N = 400
t0 = time.perf_counter(); ctx.resident_wait(ctx.resident_submit_batch(outs[0], 3840, 0, frame0=0, nframes=N)); pipe = (time.perf_counter() - t0) / N * 1e6
ctx.resident_end()
print(f"LAT {lens} no_direct {nd} copies {os.environ.get('BLINKY_DBG_COPIES','-')}: one at a time host {statistics.median(wall):.2f} us (submit call {statistics.median(sub):.2f}), device {statistics.median(dev):.2f}; pipelined {pipe:.2f} us/frame; C clock: host {ch:.2f} device {cd:.2f}", flush=True)
wl.close()
