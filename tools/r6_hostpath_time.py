import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, blinky_amd as bk, scripts as S, oracle_ffi as O, json
GOLD = {(r["globe"], r["lens"], r["zoom"], r["W"], r["H"]): r for r in json.load(open(os.path.join(ROOT, "tests/golden/lensmaps.json")))["lensmaps"]}
REC = """
local function down(v, n) if n == 0 then return v end return down(v, n - 1) end
local straight = %s
function %s(a, b, c) return straight(down(a, 2), b, c) end
"""
for lens, cb, W, H in (("panini", "lens_inverse", 3840, 2160), ("eckert5", "lens_forward", 3840, 2160), ("eckert5", "lens_forward", 1920, 1080)):
    ctx = bk.Context()
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(S.script("lenses", lens) + REC % (cb, cb), "rec.lua")
    cmd = ctx.lens_info().onload.decode().split()
    ctx.set_zoom(S.ZOOM_CMD[cmd[0]], int(float(cmd[1])) if len(cmd) > 1 else 0)
    ctx.resize(W, H)
    for mode in (1, 2):
        ctx.set_sequential_build(mode)
        t0 = time.time(); ctx.build(); dt = time.time() - t0
        off, tin = ctx.read_lensmap()
        rec = GOLD.get(("cube", lens, None, W, H))
        ok = rec is None or (O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"])
        print(f"{W}x{H} cube/{lens} through a recursive helper: path {ctx.last_build_path()[0]} ({'pool' if mode == 1 else 'one scan'}): {dt:.2f} s, golden {'ok' if ok else 'MISMATCH'}{'' if rec else ' (none recorded)'}", flush=True)
    ctx.close()
