"""(r6) Is a whole-frame batch launch really slower than its two halves?  profiles/r05_c4_whole_vs_halves.txt timed
each half in a train of its own: ten launches of the SAME half over the ring touch half of every globe - 253 MB for 64 frames of
trism/panini, which fits the 256 MiB Infinity Cache, where the whole frame's 506 MB do not.  Here the halves ALTERNATE in one train
(top, bottom, top, bottom ... each pair on the next 64 globes of the ring), which is what replacing one launch by two would do.
Developer probe; GPU box only."""
import os
import statistics
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S
from blinky_amd import ffi

W, H, F = 3840, 2160, 64
RING = int(os.environ.get("RING", "128"))
for globe, lens, zoom in (("trism", "panini", "f_fov 180"), ("cube", "panini", "f_fov 180")):
    full = blinky_amd.Context(0); S.configure(full, globe, lens, zoom, (W, H)); full.build(); cost = full.row_costs(); full.close()
    b2 = ffi.stripe_bounds_from_costs(cost, 0, 2)
    whole = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, ring_max=RING, ring_bytes=0)
    top = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, rows=(b2[0], b2[1]), ring_max=RING, ring_bytes=0)
    bot = bench.OneGpuWorkload(torch, blinky_amd, S, 0, globe, lens, zoom, W, H, F, rows=(b2[1], b2[2]), ring_max=RING, ring_bytes=0)
    stream = whole.stream
    for w in (top, bot):
        w.ctx.set_stream(stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def train(fn, n=10, repeats=7):
        out = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            e0.record(stream)
            for i in range(n):
                fn(i)
            e1.record(stream)
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / n)
        return statistics.median(out)
    for w in (whole, top, bot):
        for i in range(3):
            w.launch(i)
    t_whole = train(lambda i: whole.launch(i))
    t_top = train(lambda i: top.launch(i))
    t_bot = train(lambda i: bot.launch(i))
    t_alt = train(lambda i: (top.launch(i), bot.launch(i)))
    print(f"{globe}/{lens} x{F}, ring of {whole.R} globes ({whole.R * 6 * 2160 * 2176 / 1e9:.1f} GB): whole frame {t_whole:.1f} us; each half in a train of its own "
          f"{t_top:.1f} + {t_bot:.1f} = {t_top + t_bot:.1f} us; the halves alternating in ONE train {t_alt:.1f} us per pair", flush=True)
    for w in (whole, top, bot):
        w.close()
