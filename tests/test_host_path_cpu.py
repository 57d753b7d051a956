"""The host build paths without a GPU (bk_debug_host_build on a device-less context): what bk_build runs for scripts the GPU emitter
declines and for scripts that carry state (tests/test_host_path_gpu.py has the same through bk_build on the device).  Here the tables
are held against the goldens recorded from the unmodified reference and against the oracle's scan driven by Python callbacks (the
host-side sanitizer run of the CPU suite covers this file too)."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = {(r["globe"], r["lens"], r["zoom"], r["W"], r["H"]): r
        for r in json.load(open(os.path.join(HERE, "golden", "lensmaps.json")))["lensmaps"]}


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


def host_ctx(bk, globe, lens_src, name, zoom, W, H, rows=None):
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    ctx.load_globe(S.script("globes", globe), globe + ".lua")
    ctx.load_lens(lens_src, name)
    cmd = (zoom or ctx.lens_info().onload.decode()).split()
    ctx.set_zoom(S.ZOOM_CMD[cmd[0]], int(float(cmd[1])) if len(cmd) > 1 else 0)
    ctx.resize(W, H)
    if rows:
        ctx.set_rows(*rows)
    return ctx


@pytest.mark.parametrize("mode", [1, 2], ids=["pool", "sequential"])
@pytest.mark.parametrize("key", [("cube", "panini", None, 640, 480), ("cube", "hammer", "f_cover", 500, 300), ("fast", "panini", "f_fov 200", 640, 400),
                                 ("cube", "eckert5", None, 640, 480), ("cube", "winkel2", None, 400, 240), ("cube", "polyconic", None, 400, 300)],
                         ids=lambda k: f"{k[0]}-{k[1]}-{k[3]}x{k[4]}")
def test_host_build_equals_the_reference_goldens(bk, key, mode):
    """inverse and forward maps, the pool and the one scan: tables recorded from the unmodified fisheye.c"""
    rec = GOLD[key]
    globe, lens, zoom, W, H = key
    ctx = host_ctx(bk, globe, S.script("lenses", lens), lens + ".lua", zoom, W, H)
    off, tin, display, scale, err = ctx.host_build(mode)
    assert err is None
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"] and int((off != O.NULL).sum()) == rec["nonnull"]
    ctx.close()


def test_host_forward_build_in_stripes(bk):
    key = ("cube", "eckert5", None, 640, 480)
    rec = GOLD[key]
    parts = []
    for rows in ((0, 100), (100, 333), (333, 480)):
        ctx = host_ctx(bk, "cube", S.script("lenses", "eckert5"), "eckert5.lua", None, 640, 480, rows)
        off, tin, display, _, err = ctx.host_build(1)
        assert err is None and display[: len(rec["display"])] == rec["display"]      # (display flags are global, not per stripe)
        parts.append((off, tin))
        ctx.close()
    assert O.fnv(np.concatenate([p[0] for p in parts])) == rec["fnv_offsets"]
    assert O.fnv(np.concatenate([p[1] for p in parts])) == rec["fnv_tints"]


DROPPING = """
calls = 0
local straight = %s
function %s(a, b, c)
   calls = calls + 1
   if calls %% 7 == 0 then return nil end
   return straight(a, b, c)
end
"""


@pytest.mark.parametrize("lens,globe", [("eckert5", "cube"), ("eckert5", "fast"), ("hammer", "cube")])
def test_state_that_changes_results_follows_the_reference_call_order(bk, lens, globe):
    """every 7th callback call answers nil: WHICH calls those are is the reference's scan order - inverse maps rows from the bottom up
    (fisheye.c:2093-2103), forward maps per plate the last row's lower corners, then row by row the upper corners (2126-2217).  Expected:
    the oracle's scan driving the straight-line lens through a Python counter."""
    W, H = 96, 64
    forward = lens == "eckert5"
    cb = "lens_forward" if forward else "lens_inverse"
    twin = bk.Context(bk.ffi.DEVICE_NONE)
    twin.load_globe(S.script("globes", globe), globe + ".lua")
    twin.load_lens(S.script("lenses", lens) + ("" if forward else "\nlens_forward = nil\n"), lens + ".lua")
    info = twin.lens_info()
    calls = [0]

    def counted(which):
        def f(*a):
            calls[0] += 1
            return None if calls[0] % 7 == 0 else twin.eval_host(which, *a)
        return f
    want = O.lensmap_with_callbacks(globe, info, None if forward else counted(0), counted(1) if forward else None, "f_contain", W, H)
    twin.close()
    src = S.script("lenses", lens) + ("" if forward else "\nlens_forward = nil\n") + DROPPING % (cb, cb)
    ctx = host_ctx(bk, globe, src, "dropping.lua", "f_contain", W, H)
    assert ctx.lens_carries_state() == (True, "calls")
    off, tin, display, scale, err = ctx.host_build(0)          # what bk_build would choose for a declined script: the one scan
    assert err is None and ctx.last_build_path()[0] == 2
    np.testing.assert_array_equal(off, want.offsets)
    np.testing.assert_array_equal(tin, want.tints)
    assert scale == want.scale and display[: want.numplates] == want.display
    # ... and the pool (every worker its own copy of the state) does NOT give that table: why state-carrying scripts are scanned
    off_pool = ctx.host_build(1)[0]
    assert not np.array_equal(off_pool, want.offsets)
    ctx.close()


def test_malformed_result_and_runtime_error_on_the_host_paths(bk):
    W, H = 160, 100
    lm = O.lensmap("cube", "panini", "f_fov 180", W, H)
    ly, lx = np.divmod(np.arange(W * H), W)
    x = (lx - W // 2) * lm.scale
    y = -(ly - H // 2) * lm.scale
    key = ly * W + (W - 1 - lx)
    first = key[(x > 0.3) & (y > 0.2)].max()
    want_off = np.where(key > first, lm.offsets, O.NULL).astype(np.uint32)
    base = S.script("lenses", "panini") + "\nlocal good = lens_inverse\n"
    malformed = base + "function lens_inverse(x, y) if x > 0.3 and y > 0.2 then return x, y end return good(x, y) end\n"
    for mode in (1, 2):
        ctx = host_ctx(bk, "cube", malformed, "malformed.lua", "f_fov 180", W, H)
        off, tin, display, _, err = ctx.host_build(mode)
        assert err and "malformed result" in err and ctx.last_build_bad_key() == first + 1
        np.testing.assert_array_equal(off, want_off)
        np.testing.assert_array_equal(tin, np.where(key > first, lm.tints, 255).astype(np.uint8))
        assert display[:6] == [int(p in set((want_off[want_off != O.NULL] // (lm.ps * lm.ps)).tolist())) for p in range(6)]
        ctx.close()
    boom = base + "function lens_inverse(x, y) if x > 0.3 and y > 0.2 then local n = nil; return n + 1, 0, 1 end return good(x, y) end\n"
    for mode in (1, 2):
        ctx = host_ctx(bk, "cube", boom, "boom.lua", "f_fov 180", W, H)
        off, tin, display, _, err = ctx.host_build(mode)
        assert err and "arithmetic" in err
        assert (off == O.NULL).all() and (tin == 255).all() and display == [0] * 6
        ctx.close()
    # a forward lens: a malformed result leaves the EMPTY map (blinky_hip.h at bk_build)
    fwd = S.script("lenses", "eckert5") + "\nlocal good = lens_forward\nfunction lens_forward(x, y, z) if y > 0.9 then return 1 end return good(x, y, z) end\n"
    ctx = host_ctx(bk, "cube", fwd, "fwd_malformed.lua", None, 96, 64)
    off, tin, display, _, err = ctx.host_build(2)
    assert err and "malformed result" in err and (off == O.NULL).all() and display == [0] * 6
    ctx.close()
