"""(r6) resident apply, pipelined us per frame and one-at-a-time host / device us, per lens and size.  Developer probe; GPU box only."""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch, bench, blinky_amd, scripts as S

CASES = [("panini", 3840, 2160), ("rectilinear", 3840, 2160), ("stereographic", 3840, 2160), ("hammer", 3840, 2160), ("quincuncial", 3840, 2160), ("mercator", 3840, 2160),
         ("gumby", 3840, 2160), ("stereographic", 1920, 1080), ("hammer", 1920, 1080), ("hammer", 7680, 4320)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]] or CASES
for lens, W, H in CASES:
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None, W, H, 1, ring_max=64 if W < 7000 else 16)
    r = [wl.resident_us(frames=600 if W < 7000 else 160) for _ in range(3)]
    print(f"{W}x{H} cube/{lens}: pipelined us/frame {' '.join('%.2f' % x['us'] for x in r)}; one at a time host {r[-1]['one_at_a_time_host_us']:.2f} device {r[-1]['one_at_a_time_device_us']:.2f}; "
          f"{r[-1]['workgroups']} wgs x {r[-1]['blocks_in_registers']} (128x{r[-1]['block_h']})", flush=True)
    wl.close()
