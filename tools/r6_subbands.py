"""(r6) sub-bands of the batch launch's XCD bands: us per launch by number of sub-bands (799 = off = the r5 grid), per map, with the
frames compared against the off setting's.  Developer probe; GPU box only.  usage: r6_subbands.py [F]"""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MAPS = [("trism", "panini", "f_fov 180", 3840, 2160), ("cube", "panini", "f_fov 180", 3840, 2160), ("cube", "hammer", None, 3840, 2160),
        ("cube", "quincuncial", None, 3840, 2160), ("cube", "stereographic", None, 1920, 1080)]
if "--8k" in sys.argv:
    MAPS = [("cube", "hammer", None, 7680, 4320)]
for (g, l, z, W, H) in MAPS:
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, g, l, z, W, H, F)
    for i in range(3):
        wl.launch(i)
    ref = None
    line = []
    for knob in (799, 700, 702, 703, 704, 706, 708):
        wl.ctx.set_tile_shape(knob)
        for i in range(3):
            wl.launch(i)
        k = wl.kernel_ms(launches=20, repeats=7)
        wl.launch(0)
        torch.cuda.synchronize()
        got = wl.out[0].clone()
        if ref is None:
            ref = got
        ok = torch.equal(got, ref)
        line.append(f"{'off' if knob == 799 else 'auto' if knob == 700 else knob - 700}: {k[0] * 1e3:.1f}{'' if ok else ' MISMATCH'}")
    st = wl.ctx.tile_stats()
    print(f"{W}x{H} {g}/{l} x{F} (128x{st['tile_h'] % 1000}, {st['tiles']} blocks): " + "  ".join(line), flush=True)
    wl.close()
