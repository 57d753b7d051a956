"""The callback surface beyond what becomes GPU code (r6).  The reference lua_calls whatever a script defines
(fisheye.c:1551, 1597, 1640); the emitter (bk_emit.cpp) declines recursion, tables made at run time, functions as values and
strings.  Such scripts are not refused: bk_build evaluates their callbacks with the library's own interpreter on the host - the
worker pool when the callbacks carry no state, ONE scan in the reference's call order otherwise, for forward maps too
(fisheye.c:2126-2217) - and the tables equal what the straight-line twins' goldens (recorded from the unmodified reference) hold."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = {(r["globe"], r["lens"], r["zoom"], r["W"], r["H"]): r
        for r in json.load(open(os.path.join(HERE, "golden", "lensmaps.json")))["lensmaps"]}


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


# panini.lua with its arithmetic routed through a helper that calls ITSELF (identity after n levels)
RECURSIVE = """
local function down(v, n)
   if n == 0 then return v end
   return down(v, n - 1)
end
local straight = lens_inverse
function lens_inverse(x, y)
   return straight(down(x, 3), down(y, 2))
end
"""

# hammer.lua with its intermediate values parked in a table whose keys only exist at run time, read back through a function VALUE
RUNTIME_TABLE = """
local straight = lens_inverse
function lens_inverse(x, y)
   local t = {}
   t[x > 0 and "east" or "west"] = x
   t["y" .. ""] = y
   local pick = function(tab, k) return tab[k] end
   return straight(pick(t, "east") or pick(t, "west"), pick(t, "y"))
end
"""

# a forward lens that counts its calls (state carried from call to call) without letting the count change a result
COUNTING_FORWARD = """
calls = 0
local straight = lens_forward
function lens_forward(x, y, z)
   calls = calls + 1
   return straight(x, y, z)
end
"""

# ... and one where the count DOES change results: every 7th call answers nil (that corner's quads are skipped)
DROPPING_FORWARD = """
calls = 0
local straight = lens_forward
function lens_forward(x, y, z)
   calls = calls + 1
   if calls % 7 == 0 then return nil end
   return straight(x, y, z)
end
"""


def build(bk, globe, lens_src, name, zoom, W, H, mode=None):
    ctx = bk.Context()
    ctx.load_globe(S.script("globes", globe), globe + ".lua")
    ctx.load_lens(lens_src, name)
    info = ctx.lens_info()
    cmd = (zoom or info.onload.decode()).split()
    ctx.set_zoom(S.ZOOM_CMD[cmd[0]], int(float(cmd[1])) if len(cmd) > 1 else 0)
    ctx.resize(W, H)
    if mode is not None:
        ctx.set_sequential_build(mode)
    display, scale = ctx.build()
    off, tin = ctx.read_lensmap()
    return ctx, display, scale, off, tin


@pytest.mark.parametrize("case", ["recursive-panini", "runtime-table-hammer"])
def test_scripts_the_emitter_declines_are_evaluated_on_the_host_pool(bk, case):
    key, extra, construct = {"recursive-panini": (("cube", "panini", None, 640, 480), RECURSIVE, "recursion"),
                             "runtime-table-hammer": (("cube", "hammer", None, 960, 540), RUNTIME_TABLE, "GPU callback")}[case]
    rec = GOLD[key]
    globe, lens, zoom, W, H = key
    src = S.script("lenses", lens) + extra
    ctx, display, scale, off, tin = build(bk, globe, src, case + ".lua", zoom, W, H)
    with pytest.raises(bk.BlinkyError, match=construct):
        ctx.kernel_source()                                  # the emitter really declines it ...
    path, why = ctx.last_build_path()
    assert path == 1 and construct in why and "worker pool" in why, (path, why)        # ... and says so, by construct
    assert ctx.lens_carries_state()[0] is False
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"] and int((off != O.NULL).sum()) == rec["nonnull"]
    # the table is an ordinary lensmap: the apply kernels warp through it, and the frame is the reference's
    for p in range(6):
        ctx.fill_plate_lcg(0, p, seed_frame=0)
    assert O.fnv(ctx.apply(np.zeros((H, W), np.uint8))) == rec["fnv_frame"]
    # mode 2: the same script as ONE scan in the reference's order - the same table
    ctx.set_sequential_build(2)
    ctx.build()
    assert ctx.last_build_path()[0] == 2
    off2, tin2 = ctx.read_lensmap()
    np.testing.assert_array_equal(off2, off)
    np.testing.assert_array_equal(tin2, tin)
    ctx.close()


def test_a_straight_line_lens_still_goes_through_the_kernels(bk):
    ctx, *_ = build(bk, "cube", S.script("lenses", "panini"), "panini.lua", None, 320, 200)
    assert ctx.last_build_path() == (0, "")
    ctx.close()


def test_declined_script_with_a_malformed_result_keeps_what_the_scan_had_set(bk):
    """the pool evaluates every pixel; what the reference's scan (rows from the bottom up, pixels left to right) had not reached
    when the malformed result ended it is taken away again - the kernels' rule (tests/test_build_gpu.py), on the host path"""
    W, H = 320, 200
    lm = O.lensmap("cube", "panini", "f_fov 180", W, H)
    ly, lx = np.divmod(np.arange(W * H), W)
    x = (lx - W // 2) * lm.scale
    y = -(ly - H // 2) * lm.scale
    bad = (x > 0.3) & (y > 0.2)
    key = ly * W + (W - 1 - lx)
    first = key[bad].max()
    want_off = np.where(key > first, lm.offsets, O.NULL).astype(np.uint32)
    src = S.script("lenses", "panini") + RECURSIVE + """
local good = lens_inverse
function lens_inverse(x, y)
   if x > 0.3 and y > 0.2 then return x, y end
   return good(x, y)
end
"""
    for mode in (1, 2):                                      # the pool, then the one sequential scan: the same partial table
        ctx = bk.Context()
        ctx.load_globe(S.script("globes", "cube"), "cube.lua")
        ctx.load_lens(src, "malformed_recursive.lua")
        ctx.set_zoom(bk.ffi.ZOOM_FOV, 180)
        ctx.resize(W, H)
        ctx.set_sequential_build(mode)
        with pytest.raises(bk.BlinkyError, match="malformed result"):
            ctx.build()
        assert ctx.last_build_path()[0] == mode and ctx.last_build_bad_key() == first + 1
        off, tin = ctx.read_lensmap()
        np.testing.assert_array_equal(off, want_off)
        np.testing.assert_array_equal(tin, np.where(key > first, lm.tints, 255).astype(np.uint8))
        ctx.close()


def test_declined_script_with_a_runtime_error_draws_nothing(bk):
    src = S.script("lenses", "panini") + RECURSIVE + """
local good = lens_inverse
function lens_inverse(x, y)
   if x > 0.3 and y > 0.2 then local n = nil; return n + 1, 0, 1 end
   return good(x, y)
end
"""
    ctx = bk.Context()
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, "boom.lua")
    ctx.set_zoom(bk.ffi.ZOOM_FOV, 180)
    ctx.resize(160, 100)
    with pytest.raises(bk.BlinkyError, match="arithmetic"):
        ctx.build()
    off, tin = ctx.read_lensmap()
    assert (off == O.NULL).all() and (tin == 255).all()
    ctx.close()


def test_counting_forward_lens_is_scanned_in_the_reference_order(bk):
    """a forward lens whose callback counts its calls: bk_lens_carries_state finds the counter, bk_build scans the corners on the
    host in the reference's order, and - the count changing nothing - the table is the golden of the straight-line eckert5"""
    key = ("cube", "eckert5", None, 640, 480)
    rec = GOLD[key]
    ctx, display, scale, off, tin = build(bk, "cube", S.script("lenses", "eckert5") + COUNTING_FORWARD, "counting.lua", None, 640, 480)
    assert ctx.lens_carries_state() == (True, "calls")
    path, why = ctx.last_build_path()
    assert path == 2 and "calls" in why, (path, why)
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"]
    ctx.close()


@pytest.mark.parametrize("globe", ["cube", "fast"])
def test_forward_lens_whose_state_changes_results_equals_the_oracle_scan(bk, globe):
    """every 7th lens_forward call answers nil: which corners those are depends on the ORDER of the calls (fisheye.c:2126-2217: per
    plate the last row's lower corners, then row by row upwards the upper corners left to right).  Expected: the oracle's forward
    scan driving the straight-line lens through a Python counter.  (`fast` adds a globe_plate script to the texels' own-plate tests.)"""
    W, H = 96, 64
    twin = bk.Context(bk.ffi.DEVICE_NONE)
    twin.load_globe(S.script("globes", globe), globe + ".lua")
    twin.load_lens(S.script("lenses", "eckert5"), "eckert5.lua")
    info = twin.lens_info()
    calls = [0]

    def fwd(x, y, z):
        calls[0] += 1
        return None if calls[0] % 7 == 0 else twin.eval_host(1, x, y, z)
    want = O.lensmap_with_callbacks(globe, info, None, fwd, "f_contain", W, H)
    twin.close()
    ctx, display, scale, off, tin = build(bk, globe, S.script("lenses", "eckert5") + DROPPING_FORWARD, "dropping.lua", None, W, H)
    assert ctx.last_build_path()[0] == 2
    np.testing.assert_array_equal(off, want.offsets)
    np.testing.assert_array_equal(tin, want.tints)
    assert scale == want.scale and display[: want.numplates] == want.display
    # ... and the GPU build (mode 0: every call sees calls == 1) differs, which is why the default scans on the host
    ctx.set_sequential_build(0)
    ctx.build()
    assert ctx.last_build_path()[0] == 0
    assert not np.array_equal(ctx.read_lensmap()[0], want.offsets)
    ctx.close()


@pytest.mark.parametrize("key", [("cube", "eckert5", None, 640, 480), ("cube", "winkel2", None, 400, 240), ("cube", "polyconic", None, 400, 300)],
                         ids=lambda k: f"{k[1]}-{k[3]}x{k[4]}")
def test_host_forward_scan_equals_the_reference_goldens(bk, key):
    """mode 2 on forward lenses: the host restatement of resume_lensmap_forward + draw_quad against tables recorded from the
    unmodified reference"""
    rec = GOLD[key]
    globe, lens, zoom, W, H = key
    ctx, display, scale, off, tin = build(bk, globe, S.script("lenses", lens), lens + ".lua", zoom, W, H, mode=2)
    assert ctx.last_build_path()[0] == 2
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"] and int((off != O.NULL).sum()) == rec["nonnull"]
    ctx.close()


def test_declined_forward_lens_on_the_pool_and_in_stripes(bk):
    """a forward lens the emitter declines and that carries no state: corners on the worker pool, quads drawn in the reference's
    order; three stripe contexts concatenate to the whole table (stripe-filtered commit, display flags global)"""
    key = ("cube", "eckert5", None, 640, 480)
    rec = GOLD[key]
    src = S.script("lenses", "eckert5") + """
local function down(v, n) if n == 0 then return v end return down(v, n - 1) end
local straight = lens_forward
function lens_forward(x, y, z) return straight(down(x, 2), y, z) end
"""
    ctx, display, scale, off, tin = build(bk, "cube", src, "recursive_forward.lua", None, 640, 480)
    path, why = ctx.last_build_path()
    assert path == 1 and "recursion" in why
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"] and display[: len(rec["display"])] == rec["display"]
    ctx.close()
    parts = []
    for r0, r1 in ((0, 160), (160, 320), (320, 480)):
        c = bk.Context()
        c.load_globe(S.script("globes", "cube"), "cube.lua")
        c.load_lens(src, "recursive_forward.lua")
        c.set_zoom(bk.ffi.ZOOM_CONTAIN, 0)
        c.resize(640, 480)
        c.set_rows(r0, r1)
        d, _ = c.build()
        assert d[: len(rec["display"])] == rec["display"]
        parts.append(c.read_lensmap())
        c.close()
    assert O.fnv(np.concatenate([p[0] for p in parts])) == rec["fnv_offsets"]
    assert O.fnv(np.concatenate([p[1] for p in parts])) == rec["fnv_tints"]
