#!/usr/bin/env python3
"""How many pixels does the device code flag for the host fix-up, per lens?  Runs the generated build kernel on the
host (tests/hostemu) - no GPU needed.   tools/flag_probe.py [W H] [lens ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hostemu")]
import blinky_amd                       # noqa: E402
from blinky_amd import ffi              # noqa: E402
import scripts as S                     # noqa: E402
import emu                              # noqa: E402

args = sys.argv[1:]
W, H = (int(args[0]), int(args[1])) if len(args) >= 2 and args[0].isdigit() else (640, 480)
lenses = [a for a in args if not a.isdigit()] or S.LENSES
for globe in ("cube",):
    for lens in lenses:
        ctx = blinky_amd.Context(ffi.DEVICE_NONE)
        info = S.configure(ctx, globe, lens, None, (W, H))
        if info.map_type == ffi.MAP_FORWARD:
            t0 = time.time()
            xy, ok, flagged, err = emu.forward_corners(ctx)
            print(f"{globe}/{lens:14s} {W}x{H}: flagged {len(flagged):6d} of {len(ok)} corners ({len(flagged) / len(ok):.2e}) err {err}  [{time.time() - t0:.1f} s] (forward map)", flush=True)
            continue
        t0 = time.time()
        off, tin, flagged, err = emu.build_inverse(ctx)
        print(f"{globe}/{lens:14s} {W}x{H}: flagged {len(flagged):6d} of {off.size} ({len(flagged) / off.size:.2e}) err {err}  [{time.time() - t0:.1f} s]", flush=True)
