#!/usr/bin/env python3
"""Developer diagnostic: where does the GPU lensmap differ from the oracle (platform libm)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import blinky_amd, oracle_ffi as O, scripts as S
globe, lens, W, H = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
zoom = sys.argv[5] if len(sys.argv) > 5 else None
lm = O.lensmap(globe, lens, zoom, W, H)
ctx = blinky_amd.Context()
S.configure(ctx, globe, lens, zoom, (W, H))
disp, scale = ctx.build()
off, tin = ctx.read_lensmap()
bad = np.nonzero(off != lm.offsets)[0]
print(f"{globe}/{lens} {W}x{H}: scale equal {scale == lm.scale}; offset mismatches {len(bad)} of {off.size}; tint mismatches {(tin != lm.tints).sum()}; build {ctx.last_build_ms():.3f} ms")
ps = min(W, H)
for i in bad[:40]:
    ly, lx = divmod(int(i), W)
    def dec(o):
        if o == 0xFFFFFFFF: return None
        p, r = divmod(int(o), ps * ps); py, px = divmod(r, ps); return (p, px, py)
    x = (lx - W // 2) * scale; y = -(ly - H // 2) * scale
    ctx.set_host_math(False); hp = ctx.eval_host(0, x, y)
    ctx.set_host_math(True); hb = ctx.eval_host(0, x, y)
    print(f"  px({lx},{ly}) gpu {dec(off[i])} oracle {dec(lm.offsets[i])}  ray(platform) {hp}  ray(bkm) {hb}")
