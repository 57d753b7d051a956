import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S
lens = sys.argv[1] if len(sys.argv) > 1 else "gumby"
wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None, 3840, 2160, 1)
if "--model" in sys.argv:
    blinky_amd.debug_set_option("print_model", 1)
for k in range(4):
    r = wl.resident_us(frames=300)
    print(f"RESULT session {k}: {r['us']:.2f} us/frame, {r['workgroups']} wgs x {r['blocks_in_registers']} (128x{r['block_h']})", wl.ctx.tile_stats(), flush=True)
wl.close()
