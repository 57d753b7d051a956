"""Parity of the HIP lensmap BUILD (bk_build, replacing create_lensmap / resume_lensmap_inverse /
resume_lensmap_forward, fisheye.c:2084-2397) against the CPU oracle and the reference goldens.
Offsets, tints, display flags and scale must be bit-identical."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "lensmaps.json")))["lensmaps"]


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


def build(bk, globe, lens, zoom, W, H, rows=None, grid=None):
    ctx = bk.Context()
    if grid:
        ctx.set_rubixgrid(*grid)
    S.configure(ctx, globe, lens, zoom, (W, H))
    if rows:
        ctx.set_rows(*rows)
    display, scale = ctx.build()
    off, tin = ctx.read_lensmap()
    return ctx, display, scale, off, tin


@pytest.mark.parametrize("rec", GOLD, ids=lambda r: f"{r['globe']}-{r['lens']}-{r['zoom']}-{r['W']}x{r['H']}")
def test_build_equals_reference_golden(bk, rec):
    """every golden recorded from the unmodified reference, BASELINE.json's full 4K sizes included"""
    ctx, display, scale, off, tin = build(bk, rec["globe"], rec["lens"], rec["zoom"], rec["W"], rec["H"])
    nplates = len(rec["display"])
    assert repr(scale) == rec["scale"]
    assert display[:nplates] == rec["display"]
    assert int((off != O.NULL).sum()) == rec["nonnull"]
    assert O.fnv(tin) == rec["fnv_tints"]
    assert O.fnv(off) == rec["fnv_offsets"]         # bit-exact, no exceptions
    # the exactness bookkeeping: what the device could not decide on its own libm went through the host
    # interpreter (platform libm); that must stay a vanishing fraction of the table
    flagged, changed = ctx.last_build_fixups()
    # (lines of symmetry at most - quincuncial's diagonals and axes - never areas)
    # (eckert4 is the exception: 20 Newton steps per pixel make the first-order bounds blow up near the poles, ~6 % flagged)
    limit = off.size // 8 if rec["lens"] == "eckert4" else max(64, 8 * (rec["W"] + rec["H"]))
    assert changed <= flagged <= limit, (flagged, changed)
    # and the whole path: GPU-built map applied on the GPU to the LCG globe == the reference's frame
    for p in range(nplates):
        ctx.fill_plate_lcg(0, p, 0)
    frame = ctx.apply(np.zeros((rec["H"], rec["W"]), np.uint8))
    assert O.fnv(frame) == rec["fnv_frame"]
    ctx.close()


@pytest.mark.parametrize("globe,lens,W,H,N", [("cube", "eckert5", 3840, 2160, 8), ("cube", "winkel2", 1920, 1080, 3),
                                               ("cube", "quincuncial", 3840, 2160, 8)])
def test_stripe_local_builds_concatenate_to_the_reference_table(bk, globe, lens, W, H, N):
    """Multi-GPU build at BASELINE sizes: rank r builds only rows [H*r/N, H*(r+1)/N) - for the forward map by replicated
    evaluation with a stripe-filtered commit (every rank walks all plate texels, keeps the writes that land in its rows;
    fisheye.c:2126-2338) - and the stripes put end to end are the unmodified reference's table, display flags OR-ed."""
    rec = next(r for r in GOLD if (r["globe"], r["lens"], r["zoom"], r["W"], r["H"]) == (globe, lens, None, W, H))
    offs, tins, disp = [], [], [0] * 6
    for r in range(N):
        ctx, display, scale, off, tin = build(bk, globe, lens, None, W, H, rows=(H * r // N, H * (r + 1) // N))
        assert repr(scale) == rec["scale"]
        disp = [a | b for a, b in zip(disp, display)]
        offs.append(off)
        tins.append(tin)
        ctx.close()
    off, tin = np.concatenate(offs), np.concatenate(tins)
    assert disp[: len(rec["display"])] == rec["display"]
    assert int((off != O.NULL).sum()) == rec["nonnull"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"]


@pytest.mark.parametrize("cfg", [
    ("cube", "panini", "f_fov 90", 200, 150),
    ("cube", "panini", "f_vfov 100", 257, 129),
    ("trism", "stereographic", "f_fov 200", 320, 240),
    ("trism", "hammer", "f_cover", 300, 300),
    ("trism", "quincuncial", None, 256, 256),
    ("trism", "eckert5", None, 200, 120),           # forward map
    ("cube", "eckert5", "f_cover", 160, 120),
    ("cube", "stereographic", None, 1, 1),          # degenerate sizes
    ("cube", "hammer", None, 7, 3),
])
def test_build_equals_oracle_arrays(bk, cfg):
    lm = O.lensmap(*cfg)
    ctx, display, scale, off, tin = build(bk, *cfg)
    assert scale == lm.scale
    assert display[: lm.numplates] == lm.display
    np.testing.assert_array_equal(off, lm.offsets)
    np.testing.assert_array_equal(tin, lm.tints)
    ctx.close()


@pytest.mark.parametrize("cfg", [
    ("cube", "quincuncial", None, 3840, 2160),
    ("cube", "stereographic", None, 1920, 1080),
    ("trism", "panini", None, 960, 540),
    ("cube", "hammer", "f_cover", 500, 300),
    ("cube", "eckert5", None, 320, 240),
])
def test_gpu_equals_portable_libm_oracle_exactly(bk, cfg):
    """With the host side switched to the portable libm as well, the whole GPU result is a pure
    function of the scripts: it must equal the oracle built on the same libm with NO exceptions."""
    lm = O.lensmap(*cfg, portable=True)
    ctx = bk.Context()
    ctx.set_host_math(True)
    S.configure(ctx, cfg[0], cfg[1], cfg[2], (cfg[3], cfg[4]))
    display, scale = ctx.build()
    off, tin = ctx.read_lensmap()
    assert scale == lm.scale and display[: lm.numplates] == lm.display
    np.testing.assert_array_equal(off, lm.offsets)
    np.testing.assert_array_equal(tin, lm.tints)
    ctx.close()


@pytest.mark.parametrize("globe,lens,W,H", [("cube", "quincuncial", 3840, 2160), ("cube", "eckert4", 800, 400), ("cube", "eckert5", 640, 480),
                                             ("fast", "panini", 640, 400)])
def test_flagged_entries_through_the_compiled_host_module(bk, globe, lens, W, H, request):
    """the same goldens with the flagged entries re-derived by the compiled host module (forced and waited for) and by the
    script interpreter alone: same table, same counts"""
    import shutil
    if not (shutil.which("c++") or shutil.which("g++") or shutil.which("clang++")):
        pytest.skip("no host C++ compiler on this box")
    rec = next(r for r in GOLD if (r["globe"], r["lens"], r["W"], r["H"]) == (globe, lens, W, H))
    request.addfinalizer(lambda: bk.debug_set_option("host_module", 0))
    counts = {}
    for mode in (1, 2):
        bk.debug_set_option("host_module", mode)
        ctx, display, scale, off, tin = build(bk, globe, lens, rec["zoom"], W, H)
        assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
        assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"]
        counts[mode] = ctx.last_build_fixups()
        assert ctx.build_breakdown()["compiled_host_module"] == (mode == 1 and counts[mode][0] > 0)
        ctx.close()
    assert counts[1] == counts[2]


MALFORMED = """
local good = lens_inverse
function lens_inverse(x, y)
   if x > 0.3 and y > 0.2 then
      return x, y            -- two values: LUAtoC_lens_inverse's status -1 (fisheye.c:1579-1584)
   end
   return good(x, y)
end
"""


@pytest.mark.parametrize("ranks", [1, 3])
def test_malformed_result_keeps_what_the_reference_scan_had_set(bk, ranks):
    """A lens_inverse that returns a malformed result ends the reference's scan - rows from the bottom up, pixels left to right
    (fisheye.c:2093-2103) - at that pixel and keeps what it had set (2113-2115).  bk_build / bk_multi_build report BK_E_SCRIPT
    and leave exactly that table: the oracle's panini table up to the first failing pixel of the scan, NULL from there on, and
    the display flags of what is left."""
    W, H = 320, 200
    lm = O.lensmap("cube", "panini", "f_fov 180", W, H)
    ly, lx = np.divmod(np.arange(W * H), W)
    x = (lx - W // 2) * lm.scale
    y = -(ly - H // 2) * lm.scale
    bad = (x > 0.3) & (y > 0.2)
    key = ly * W + (W - 1 - lx)                     # larger = earlier in the reference's scan
    first = key[bad].max()
    want_off = np.where(key > first, lm.offsets, O.NULL).astype(np.uint32)
    want_tin = np.where(key > first, lm.tints, 255).astype(np.uint8)
    kept_plates = sorted(set((want_off[want_off != O.NULL] // (lm.ps * lm.ps)).tolist()))
    src = S.script("lenses", "panini") + MALFORMED
    if ranks == 1:
        ctx = bk.Context()
        ctx.load_globe(S.script("globes", "cube"), "cube.lua")
        ctx.load_lens(src, "panini_malformed.lua")
        ctx.set_zoom(bk.ffi.ZOOM_FOV, 180)
        ctx.resize(W, H)
        with pytest.raises(bk.BlinkyError, match="malformed result"):
            ctx.build()
        assert ctx.last_build_bad_key() == first + 1
        off, tin = ctx.read_lensmap()
        frame = ctx.apply(np.zeros((H, W), np.uint8))          # the partial table is a valid lensmap: it can be applied
        assert (frame.reshape(-1)[want_off == O.NULL] == 0).all()
        ctx.close()
    else:
        m = bk.Multi([0] * ranks)
        m.load_globe(S.script("globes", "cube"), "cube.lua")
        m.load_lens(src, "panini_malformed.lua")
        m.set_zoom(bk.ffi.ZOOM_FOV, 180)
        m.resize(W, H)
        with pytest.raises(bk.BlinkyError, match="malformed result"):
            m.build()
        parts = [m.ctx(r).read_lensmap() for r in range(ranks)]
        off, tin = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        # (bk_truncate_build is idempotent: asking again with the same pixel returns each stripe's display flags)
        shown = sorted(set(i for r in range(ranks) for i, d in enumerate(m.ctx(r).truncate_build(int(first) + 1)) if d))
        assert shown == kept_plates
        m.close()
    np.testing.assert_array_equal(off, want_off)
    np.testing.assert_array_equal(tin, want_tin)
    assert 0 < int((off != O.NULL).sum()) < lm.nonnull


COUNTER = """
count = 0
local good = lens_inverse
function lens_inverse(x, y)
   count = count + 1
   if count % 7 == 0 then return nil end
   return good(x, y)
end
"""


# the same counter kept in a LOCAL of the script and advanced by a function defined inside the callback
COUNTER_LOCAL = """
local count = 0
local good = lens_inverse
function lens_inverse(x, y)
   local function dropped()
      count = count + 1
      return count % 7 == 0
   end
   if dropped() then return nil end
   return good(x, y)
end
"""


@pytest.mark.parametrize("host_module", [0, 2], ids=["compiled", "interpreter"])
@pytest.mark.parametrize("counter", ["global", "local"])
def test_sequential_build_reproduces_a_script_that_counts_pixels(bk, host_module, counter, request):
    """bk_set_sequential_build(1), the default: a lens whose callback carries state from pixel to pixel - here a counter that drops every 7th
    pixel it is asked for - is built as ONE scan in the reference's order (rows from the bottom up, pixels left to right,
    fisheye.c:2093-2103) on the host.  Expected: the oracle's panini table with the entry of the k-th scanned pixel gone where
    k % 7 == 0.  The parallel GPU build (mode 0) gives every pixel count = 1 instead: documented, and shown here."""
    W, H = 200, 120
    lm = O.lensmap("cube", "panini", "f_fov 180", W, H)
    ly, lx = np.divmod(np.arange(W * H), W)
    k = (H - 1 - ly) * W + lx + 1                       # the reference's scan reaches this pixel k-th
    want = np.where(k % 7 == 0, O.NULL, lm.offsets).astype(np.uint32)
    request.addfinalizer(lambda: bk.debug_set_option("host_module", 0))
    bk.debug_set_option("host_module", host_module)
    ctx = bk.Context()
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(S.script("lenses", "panini") + (COUNTER if counter == "global" else COUNTER_LOCAL), "counter.lua")
    ctx.set_zoom(bk.ffi.ZOOM_FOV, 180)
    ctx.resize(W, H)
    assert ctx.lens_carries_state() == (True, "count")
    ctx.set_sequential_build(0)
    ctx.build()                                         # mode 0: parallel, per-pixel state -> every pixel sees count == 1
    off, _ = ctx.read_lensmap()
    np.testing.assert_array_equal(off, lm.offsets)
    ctx.set_sequential_build(1)                         # (the default since round 4)
    display, scale = ctx.build()
    off, tin = ctx.read_lensmap()
    np.testing.assert_array_equal(off, want)
    np.testing.assert_array_equal(tin, np.where(k % 7 == 0, 255, lm.tints).astype(np.uint8))
    assert scale == lm.scale and display[: lm.numplates] == lm.display
    ctx.close()


@pytest.mark.parametrize("cfg", [("cube", "eckert4", None, 800, 400), ("cube", "quincuncial", None, 640, 480), ("fast", "panini", "f_fov 200", 640, 400)])
def test_sequential_build_equals_the_reference_goldens(bk, cfg):
    """mode 2: every inverse lens through the one-scan host build - eckert4's per-row cache carried exactly as the reference
    carries it - gives the golden tables"""
    globe, lens, zoom, W, H = cfg
    rec = next(r for r in GOLD if (r["globe"], r["lens"], r["zoom"], r["W"], r["H"]) == cfg)
    ctx = bk.Context()
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.set_sequential_build(2)
    display, scale = ctx.build()
    off, tin = ctx.read_lensmap()
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"]
    assert ctx.last_build_fixups() == (0, 0)            # nothing to flag: the platform libm computed every entry
    ctx.close()


def test_exact_ties_are_resolved_on_the_platform_libm(bk):
    """cube/quincuncial at 3840x2160 (BASELINE.json configs[2]): at a few pixels 2*atan2(r,1) - pi/2 cancels to
    exactly 0 on glibc and to +-1 ulp on any other correct libm, which moves u*ps across an integer.  The device
    flags those pixels, the host interpreter re-derives them on the platform libm, and the table equals the
    unmodified reference's; with the host switched to the portable libm the same pixels are flagged and nothing
    changes (the result is then a pure function of the scripts)."""
    rec = next(r for r in GOLD if r["lens"] == "quincuncial" and r["W"] == 3840)
    ctx, _, _, off, _ = build(bk, rec["globe"], rec["lens"], rec["zoom"], rec["W"], rec["H"])
    flagged, changed = ctx.last_build_fixups()
    assert O.fnv(off) == rec["fnv_offsets"]
    assert changed >= 1 and flagged >= changed
    ctx.close()
    ctx = bk.Context()
    ctx.set_host_math(True)
    S.configure(ctx, rec["globe"], rec["lens"], rec["zoom"], (rec["W"], rec["H"]))
    ctx.build()
    flagged_p, changed_p = ctx.last_build_fixups()
    assert flagged_p == flagged and changed_p == 0
    ctx.close()


def test_rubixgrid_variants(bk):
    for grid in [(3, 2.0, 1.0), (10, 4.0, 1.0), (5, 1.0, 0.5)]:
        lm = O.lensmap("cube", "panini", None, 320, 240, grid=grid)
        ctx, _, _, off, tin = build(bk, "cube", "panini", None, 320, 240, grid=grid)
        np.testing.assert_array_equal(tin, lm.tints)
        np.testing.assert_array_equal(off, lm.offsets)
        ctx.close()


@pytest.mark.parametrize("lens", ["hammer", "eckert5"])       # inverse and forward (stripe-filtered commit)
def test_row_stripes_build_the_same_table(bk, lens):
    W, H = 480, 270
    lm = O.lensmap("cube", lens, None, W, H)
    bounds = [0, 33, 34, 200, 270]
    alldisp = [0] * 6
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        ctx, display, _, off, tin = build(bk, "cube", lens, None, W, H, rows=(r0, r1))
        np.testing.assert_array_equal(off, lm.offsets.reshape(H, W)[r0:r1].ravel())
        np.testing.assert_array_equal(tin, lm.tints.reshape(H, W)[r0:r1].ravel())
        alldisp = [a | b for a, b in zip(alldisp, display)]
        ctx.close()
    assert alldisp == lm.display                   # the OR over stripes is the reference's display[]


@pytest.mark.parametrize("lens", S.LENSES)
def test_device_callbacks_bit_equal_host_interpreter(bk, lens):
    """The generated device code and the host interpreter walk the same AST with the same portable
    libm: raw callback results must be bit-identical for every shipped lens (also checks bkm.h
    device == host)."""
    ctx = bk.Context()
    ctx.set_host_math(True)
    info = S.configure(ctx, "cube", lens, None, (640, 480))
    rng = np.random.default_rng(5)
    if info.has_inverse:
        w = info.lens_width or 6.0
        h = info.lens_height or 4.0
        args = np.concatenate([rng.uniform(-0.6, 0.6, (700, 2)) * [w, h], [[0.0, 0.0], [w / 2, 0.0], [0.0, h / 2], [1e-9, -1e-9]]])
        d_out, d_n = ctx.eval_device(0, args)
        h_out, h_n = ctx.eval_host_many(0, args)
        np.testing.assert_array_equal(d_n, h_n)
        assert d_out.view(np.uint64).tolist() == h_out.view(np.uint64).tolist() or \
            np.array_equal(d_out[~np.isnan(h_out)].view(np.uint64), h_out[~np.isnan(h_out)].view(np.uint64))
    if info.has_forward:
        v = rng.normal(size=(600, 3))
        v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
        d_out, d_n = ctx.eval_device(1, v)
        h_out, h_n = ctx.eval_host_many(1, v)
        np.testing.assert_array_equal(d_n, h_n)
        m = ~np.isnan(h_out)
        np.testing.assert_array_equal(np.isnan(d_out), np.isnan(h_out))
        assert np.array_equal(d_out[m].view(np.uint64), h_out[m].view(np.uint64))
    ctx.close()


@pytest.mark.parametrize("lens", S.LENSES)
def test_every_shipped_lens_builds_the_oracle_table(bk, lens):
    """All 31 lens scripts (21 inverse, 10 forward-only), cube and trism globes, small frame: the GPU build
    against the oracle's fisheye.c restatement whose lens callbacks are evaluated by the host interpreter
    on the platform libm (i.e. the way the reference's Lua VM would).  The tables must be identical."""
    for globe, (W, H) in (("cube", (160, 120)), ("trism", (96, 128))):
        hostctx = bk.Context(bk.ffi.DEVICE_NONE)            # interpreter only, platform libm
        info = S.configure(hostctx, globe, lens, None, (W, H))
        zoom = info.onload.decode()
        inv = (lambda x, y: hostctx.eval_host(0, x, y)) if info.has_inverse else None
        fwd = (lambda x, y, z: hostctx.eval_host(1, x, y, z)) if info.has_forward else None
        lm = O.lensmap_with_callbacks(globe, info, inv, fwd, zoom, W, H)
        ctx, display, scale, off, tin = build(bk, globe, lens, None, W, H)
        assert lm.built, (lens, globe)
        assert scale == lm.scale, (lens, globe)
        assert display[: lm.numplates] == lm.display, (lens, globe)
        bad = int((off != lm.offsets).sum())
        assert bad == 0, f"{lens}/{globe}: {bad} of {off.size} lensmap entries differ"
        np.testing.assert_array_equal(tin, lm.tints)
        ctx.close()
        hostctx.close()


def test_globe_plate_override_fast_globe(bk):
    """fast.lua's globe_plate (nil for z <= 0) runs on the device; check against the host interpreter
    evaluating the same script for plate choice, then the table for structure."""
    ctx = bk.Context()
    ctx.set_host_math(True)
    S.configure(ctx, "fast", "panini", "f_fov 200", (320, 200))
    rng = np.random.default_rng(9)
    v = rng.normal(size=(500, 3))
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    d_out, d_n = ctx.eval_device(2, v)
    h_out, h_n = ctx.eval_host_many(2, v)
    np.testing.assert_array_equal(d_n, h_n)
    assert (d_n[v[:, 2] <= 0] == -1).all()
    m = h_n == 1
    np.testing.assert_array_equal(d_out[m, 0], h_out[m, 0])
    display, _ = ctx.build()
    off, _ = ctx.read_lensmap()
    assert display[:2] == [1, 1]
    mapped = off != O.NULL
    assert 0 < mapped.sum() < off.size              # fov 200: rays behind the viewer stay NULL
    assert (off[mapped] // (200 * 200) <= 1).all()
    ctx.close()


def test_module_cache_round_trip_builds_the_same_table(bk, tmp_path, monkeypatch, request):
    """a lens module loaded back from BLINKY_HIP_CACHE builds the identical lensmap"""
    monkeypatch.setenv("BLINKY_HIP_CACHE", str(tmp_path))
    bk.debug_set_option("no_memcache", 1)                      # (otherwise the second context is served from the process' own cache)
    request.addfinalizer(lambda: bk.debug_set_option("no_memcache", 0))
    tables = []
    for i in range(2):
        ctx = bk.Context()
        S.configure(ctx, "cube", "quincuncial", None, (400, 300))
        ctx.build()
        assert ctx.module_from_cache() == (i == 1)
        tables.append(ctx.read_lensmap())
        ctx.close()
    np.testing.assert_array_equal(tables[0][0], tables[1][0])
    np.testing.assert_array_equal(tables[0][1], tables[1][1])


def test_async_compile_answers_pending_and_keeps_the_previous_lensmap(bk, tmp_path, monkeypatch):
    """bk_set_async_compile: a lens that still has to go through hiprtc makes bk_build return BK_PENDING at once and leaves
    the previous lensmap in place; a later call finds the module and builds."""
    import time
    monkeypatch.setenv("BLINKY_HIP_CACHE", str(tmp_path))
    ctx = bk.Context()
    S.configure(ctx, "cube", "panini", None, (320, 240))
    ctx.build()
    before = ctx.read_lensmap()[0].copy()
    ctx.set_async_compile(True)
    # a lens nobody has compiled yet in this process or on disk: the constant below ends up in the generated kernel
    unique = 1.0 + (int(time.time() * 1e6) % 100000) * 1e-9
    ctx.load_lens(f"lens_width = 4 lens_height = 3 function lens_inverse(x,y) local k = {unique!r} "
                  "return latlon_to_ray(y * k * 0.5, x * k * 0.5) end", "unique.lua")
    ctx.set_zoom(bk.ffi.ZOOM_CONTAIN)
    t0 = time.perf_counter()
    assert ctx.build_nowait() is None
    assert time.perf_counter() - t0 < 0.1                       # it did not wait for hiprtc
    np.testing.assert_array_equal(ctx.read_lensmap()[0], before)    # the old table is still what bk_apply would use
    deadline = time.time() + 120
    while (res := ctx.build_nowait()) is None:
        assert time.time() < deadline
        time.sleep(0.01)
    display, scale = res
    assert scale == 4 / 320
    assert not np.array_equal(ctx.read_lensmap()[0], before)
    ctx.close()


def test_generic_for_strings_and_type_on_the_device(bk):
    """`for .. in ipairs/pairs`, string constants (==, ~=, per-pixel string state) and type(): device results bit-equal to
    the host interpreter's"""
    from test_frontend import GENERIC_FOR_LENS
    ctx = bk.Context()
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube")
    ctx.load_lens(GENERIC_FOR_LENS, "gf.lua")
    ctx.resize(64, 48)
    rng = np.random.default_rng(4)
    args = np.concatenate([rng.uniform(-2.5, 2.5, (300, 2)), [[1.95, 0.0], [0.0, 0.0]]])
    d_out, d_n = ctx.eval_device(0, args)
    # (a host interpreter carries the global `mode` from call to call; on the device it is per-pixel state - so every
    #  argument is compared with a FRESH interpreter)
    flips = args[:, 0] > 1.9
    fresh = []
    for a in args:
        c2 = bk.Context(bk.ffi.DEVICE_NONE)
        c2.set_host_math(True)
        c2.load_globe(S.script("globes", "cube"), "cube")
        c2.load_lens(GENERIC_FOR_LENS, "gf.lua")
        fresh.append(c2.eval_host(0, *a))
        c2.close()
    for i, r in enumerate(fresh):
        if r is None:
            assert d_n[i] == -1, (i, args[i])
        else:
            assert d_n[i] == 3 and d_out[i, :3].tobytes() == np.array(r).tobytes(), (i, args[i])
    assert (d_n[flips] == -1).all() and (d_n[~flips] == 3).all()
    ctx.close()


def test_script_runtime_errors_surface_as_errors(bk):
    ctx = bk.Context()
    ctx.load_globe(S.script("globes", "cube"), "cube")
    ctx.load_lens("lens_width = 2 lens_height = 2 function lens_inverse(x,y) return x, y end", "two.lua")   # 2 values
    ctx.set_zoom(bk.ffi.ZOOM_CONTAIN)
    ctx.resize(64, 48)
    with pytest.raises(bk.BlinkyError, match="malformed result"):
        ctx.build()
    off, _ = ctx.read_lensmap()
    assert (off == O.NULL).all()
    ctx.load_lens("lens_width = 2 lens_height = 2 function lens_inverse(x,y) return x + undefined_global, 0, 1 end", "nil.lua")
    with pytest.raises(bk.BlinkyError, match="arithmetic on a non-number"):
        ctx.build()
    ctx.load_lens("lens_width = 2 lens_height = 2 function lens_inverse(x,y) while true do end end", "loop.lua")
    with pytest.raises(bk.BlinkyError, match="iteration budget"):
        ctx.build()
    # and an invalid zoom leaves an empty map, like the reference (create_lensmap returns early)
    S.configure(ctx, "cube", "quincuncial", "f_fov 90")
    with pytest.raises(bk.BlinkyError, match="max_fov"):
        ctx.build()
    assert (ctx.read_lensmap()[0] == O.NULL).all()
    ctx.close()


@pytest.mark.parametrize("lens,W,H", [("quincuncial", 640, 480), ("stereographic", 640, 400), ("hammer", 640, 360), ("winkeltripel", 480, 300),
                                      ("fisheye1", 400, 400), ("sinusoidal", 480, 300), ("gins8", 480, 300)])
def test_flag_and_fix_up_end_to_end_against_a_stand_in_libm(bk, lens, W, H, monkeypatch, request):
    """The whole mechanism on the GPU, with the libm discrepancy scaled up until it bites thousands of times: the host
    interpreter runs on a stand-in libm 2^-30 away from bkm.h (bk_set_host_math(ctx, 30)), the kernels are generated with
    BK_LIBM_REL = 2^-30 to match, and the table bk_build delivers - device results, flagged entries re-derived on the host
    and patched - must be, entry for entry, what the host interpreter alone derives for every pixel."""
    bk.debug_set_option("libm_rel_log2", 30)
    request.addfinalizer(lambda: bk.debug_set_option("libm_rel_log2", 0))
    monkeypatch.setenv("BLINKY_HIP_CACHE", "off")
    ctx = bk.Context()
    ctx.set_host_math(30)
    info = S.configure(ctx, "cube", lens, None, (W, H))
    ctx.build()
    off, tin = ctx.read_lensmap()
    flagged, changed = ctx.last_build_fixups()
    if info.map_type == 1:                                   # inverse map: every pixel has a host-side value to compare with
        hoff, htin = ctx.host_entries(np.arange(W * H, dtype=np.uint32))
        np.testing.assert_array_equal(off, hoff)
        np.testing.assert_array_equal(tin, htin)
        assert flagged > 100, (flagged, changed)              # the scaled-up discrepancy does bite
    # and the same map built with the kernels' normal assumption (2^-50) on the same stand-in libm is NOT that table where
    # the stand-in disagrees with bkm.h by more than the kernels allow for: the flags are what makes the difference
    bk.debug_set_option("libm_rel_log2", 0)
    ctx2 = bk.Context()
    ctx2.set_host_math(30)
    S.configure(ctx2, "cube", lens, None, (W, H))
    ctx2.build()
    off2, tin2 = ctx2.read_lensmap()
    f2, c2 = ctx2.last_build_fixups()
    assert f2 <= flagged
    print(f"{lens}: 2^-30 kernels flagged {flagged} changed {changed}; 2^-50 kernels flagged {f2} changed {c2}, entries that differ between the two tables {int((off != off2).sum())}")
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("lens,W,H", [("eckert5", 480, 300), ("winkel2", 400, 240)])
def test_forward_build_that_flags_entries_is_redone_pass_by_pass(bk, lens, W, H, monkeypatch, request):
    """(r6) A forward build submits its three passes in one go and looks at the flag counters once, at the end; a build whose corner pass
    DID flag something has to throw that away and go pass by pass (host answers patched in between), and so has the next build of the
    same lens.  Stand-in libm 2^-16 away, kernels told so: a corner lands within 0.02 pixels of a pixel edge often enough for thousands of flags.  Expected table: the host interpreter's own
    forward build on the same stand-in libm (bk_debug_host_build on a device-less context)."""
    bk.debug_set_option("libm_rel_log2", 16)
    request.addfinalizer(lambda: bk.debug_set_option("libm_rel_log2", 0))
    monkeypatch.setenv("BLINKY_HIP_CACHE", "off")
    ctx = bk.Context()
    ctx.set_host_math(16)
    S.configure(ctx, "cube", lens, None, (W, H))
    display1 = ctx.build()
    off1, tin1 = ctx.read_lensmap()
    flagged1, changed1 = ctx.last_build_fixups()
    assert flagged1 > 100, (flagged1, changed1)
    display2 = ctx.build()                                   # (straight to the careful path this time)
    off2, tin2 = ctx.read_lensmap()
    assert ctx.last_build_fixups() == (flagged1, changed1) and display2 == display1
    np.testing.assert_array_equal(off2, off1)
    np.testing.assert_array_equal(tin2, tin1)
    host = bk.Context(bk.ffi.DEVICE_NONE)
    host.set_host_math(16)
    S.configure(host, "cube", lens, None, (W, H))
    hoff, htin, hdisplay, _, err = host.host_build(1)
    assert err is None
    np.testing.assert_array_equal(off1, hoff)
    np.testing.assert_array_equal(tin1, htin)
    host.close()
    # ... and the ordinary case - nothing flagged - twice over: the one-submission path, the same table both times, the goldens' table
    bk.debug_set_option("libm_rel_log2", 0)
    plain = bk.Context()
    S.configure(plain, "cube", lens, None, (W, H))
    plain.build()
    a = plain.read_lensmap()
    assert plain.last_build_fixups()[0] == 0
    plain.build()
    b = plain.read_lensmap()
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    lm = O.lensmap("cube", lens, None, W, H)
    np.testing.assert_array_equal(a[0], lm.offsets)
    np.testing.assert_array_equal(a[1], lm.tints)
    ctx.close()
    plain.close()


def test_one_context_through_lenses_globes_sizes_and_stripes(bk):
    """(r6) bk_build keeps things between builds now - the generated translation unit while no script state has moved, the rubix bitmap per
    platesize and grid, the forward build's uv / quotient tables per platesize, whether a forward lens' last build flagged anything - and
    decides per build which of them still hold.  One context driven through what an engine session does: f_lens, f_globe, a window
    resize, f_rubixgrid, zoom commands, a stripe set and lifted again, forward and inverse maps in turn, the same build twice; every table
    against the oracle's."""
    ctx = bk.Context()
    steps = [("cube", "panini", "f_fov 180", 640, 400, None, None), ("cube", "eckert5", None, 640, 400, None, None),
             ("cube", "eckert5", None, 640, 400, None, None),                       # (nothing changed: kept source, kept tables)
             ("cube", "eckert5", None, 500, 300, None, None),                       # a new platesize under the same lens
             ("trism", "eckert5", None, 500, 300, None, None),                      # another globe: the tiles' flags are the plates'
             ("trism", "winkel2", None, 500, 300, (100, 233), None),                # a stripe of a forward map
             ("trism", "winkel2", None, 500, 300, None, (6, 3.0, 2.0)),             # the grid changes under a kept source
             ("cube", "hammer", "f_cover", 500, 300, None, (6, 3.0, 2.0)), ("cube", "hammer", "f_contain", 500, 300, None, None),
             ("cube", "winkel2", None, 640, 400, None, None), ("cube", "panini", "f_fov 120", 640, 400, (0, 17), None),
             ("cube", "panini", "f_fov 120", 640, 400, None, None)]
    loaded = (None, None)
    for k, (globe, lens, zoom, W, H, rows, grid) in enumerate(steps):
        if globe != loaded[0]:
            ctx.load_globe(S.script("globes", globe), globe + ".lua")
        if (globe, lens) != loaded:
            ctx.load_lens(S.script("lenses", lens), lens + ".lua")                  # (the reference reloads the lens after f_globe too)
        loaded = (globe, lens)
        cmd = (zoom or ctx.lens_info().onload.decode()).split()
        ctx.set_zoom(S.ZOOM_CMD[cmd[0]], int(float(cmd[1])) if len(cmd) > 1 else 0)
        ctx.resize(W, H)
        ctx.set_rows(*(rows or (0, H)))
        g = grid or (10, 4.0, 1.0)
        ctx.set_rubixgrid(*g)
        display, scale = ctx.build()
        off, tin = ctx.read_lensmap()
        lm = O.lensmap(globe, lens, zoom, W, H, grid=g)
        r0, r1 = rows or (0, H)
        np.testing.assert_array_equal(off, lm.offsets.reshape(H, W)[r0:r1].ravel(), err_msg=f"step {k}: {steps[k]}")
        np.testing.assert_array_equal(tin, lm.tints.reshape(H, W)[r0:r1].ravel(), err_msg=f"step {k}: {steps[k]}")
        assert scale == lm.scale
        if not rows:
            assert display[: lm.numplates] == lm.display, (k, display, lm.display)
    ctx.close()


def _random_globe(rng):
    """a globe script with 2-6 plates in general position: random forward / up vectors (the loader makes right and up from them,
    not normalised - fisheye.c:1818-1850), fields of view from narrow to wider than a half-space's"""
    n = int(rng.integers(2, 7))
    rows = []
    for _ in range(n):
        f = rng.normal(size=3)
        f *= rng.choice([0.25, 1.0, 1.0, 3.0]) / np.linalg.norm(f)                 # forward vectors of different lengths, too
        u = rng.normal(size=3)
        rows.append("   {{%r, %r, %r}, {%r, %r, %r}, %r}," % (*f.tolist(), *u.tolist(), float(rng.choice([40, 75, 90, 110, 128, 150]))))
    return "plates = {\n" + "\n".join(rows) + "\n}\n"


@pytest.mark.parametrize("seed", range(12))
def test_forward_tiles_taken_on_trust_equal_the_texel_by_texel_build(bk, seed, request):
    """(r6) bk_forward_tiles lets the quad pass skip the ownership test for tiles of 16 x 16 texels that lie inside their plate's region by a
    margin (an argument about affine functions and float rounding, bk_build_kernels.h).  Here the argument is put to globes it was not
    written with in mind - plates in general position, skewed up vectors, forward vectors of length 0.25 to 3, overlapping and gappy fields
    of view - and the table must be the one the texel-by-texel build gives ("forward_careful"), entry for entry."""
    rng = np.random.default_rng(7000 + seed)
    globe = _random_globe(rng)
    lens = ["eckert5", "winkel2", "sinusoidal"][seed % 3]
    W, H = [(640, 400), (500, 500), (800, 320)][seed % 3]
    tables = []
    for careful in (0, 1):
        bk.debug_set_option("forward_careful", careful)
        request.addfinalizer(lambda: bk.debug_set_option("forward_careful", 0))
        ctx = bk.Context()
        ctx.load_globe(globe, "random.lua")
        ctx.load_lens(S.script("lenses", lens), lens + ".lua")
        cmd = ctx.lens_info().onload.decode().split()
        ctx.set_zoom(S.ZOOM_CMD[cmd[0]], int(float(cmd[1])) if len(cmd) > 1 else 0)
        ctx.resize(W, H)
        display, scale = ctx.build()
        tables.append((ctx.read_lensmap(), display, scale))
        taken, total = ctx.forward_tiles()
        if careful:
            assert taken == -1
        else:
            assert 0 < taken <= total, (taken, total)     # (most globes: some tiles inside a plate's region, some across a border; two plates back to back: all inside)
            print(f"seed {seed}: {taken} of {total} tiles taken on trust")
        ctx.close()
    bk.debug_set_option("forward_careful", 0)
    (a, da, sa), (b, db, sb) = tables
    assert da == db and sa == sb
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert int((a[0] != O.NULL).sum()) > W * H // 50, "a globe that shows nothing tests nothing"


def test_functions_defined_inside_a_callback_build_the_same_table_on_the_gpu(bk):
    """tests/test_frontend.py's pair of scripts - the same arithmetic written plainly and with local functions / closures / chunk
    locals as scratch - through the GPU build, and the second one through the one-scan host build as well"""
    from test_frontend import INTEGRALS_HIGHER_ORDER, INTEGRALS_PLAIN, NESTED_LENS, PLAIN_LENS
    _same_table_on_the_gpu(bk, ((PLAIN_LENS, 0), (NESTED_LENS, 0), (NESTED_LENS, 1)))
    # ... and functions passed as arguments (the callee generated once per function it is handed)
    _same_table_on_the_gpu(bk, ((INTEGRALS_PLAIN, 0), (INTEGRALS_HIGHER_ORDER, 0), (INTEGRALS_HIGHER_ORDER, 2)))


def _same_table_on_the_gpu(bk, variants):
    tables = []
    for body, sequential in variants:
        ctx = bk.Context()
        ctx.load_globe(S.script("globes", "cube"), "cube.lua")
        ctx.load_lens(body, "nested.lua")
        ctx.set_zoom(bk.ffi.ZOOM_CONTAIN, 0)
        ctx.resize(640, 400)
        ctx.set_sequential_build(sequential)
        display, scale = ctx.build()
        tables.append(ctx.read_lensmap() + (display, scale))
        ctx.close()
    assert (tables[0][0] != O.NULL).sum() > 150000 and sum(tables[0][2]) == 6
    for t in tables[1:]:
        np.testing.assert_array_equal(t[0], tables[0][0])
        np.testing.assert_array_equal(t[1], tables[0][1])
        assert t[2:] == tables[0][2:]


@pytest.mark.parametrize("pair", ["records_and_matrices", "varargs", "constant_objects"])
def test_round3_constructs_build_the_same_table_on_the_gpu(bk, pair):
    """tests/test_frontend.py's remaining differential pairs - records + matrices, vararg helpers / select, method calls on constant
    objects (one through a metatable) and objects as arguments - through bk_build ON THE DEVICE (round 3 pinned them on the host
    emulation of the generated code only), the construct version through the one-scan host build as well (fisheye.c:1545-1588)"""
    import test_frontend as F
    plain, fancy = {"records_and_matrices": (F.ROTATION_PLAIN, F.ROTATION_TABLES), "varargs": (F.VARARGS_PLAIN, F.VARARGS_LENS),
                    "constant_objects": (F.OBJECT_PLAIN, F.OBJECT_LENS)}[pair]
    _same_table_on_the_gpu(bk, ((plain, 0), (fancy, 0), (fancy, 2)))


@pytest.mark.parametrize("name", ["measured_profile", "thin_lens_object", "integrated_arc", "uses_shared"])
def test_example_lenses_build_the_oracle_table_on_the_gpu(bk, name, monkeypatch):
    """examples/lenses/*.lua (a profile read from a file, an object with methods + a record per pixel, a higher-order integrator with
    nested functions and varargs, a required helper module with a rotation matrix) through bk_build on the device, against the oracle's
    fisheye.c restatement whose callbacks the host interpreter evaluates on the platform libm - the way the reference's Lua VM would
    (fisheye.c:1545-1588, 2084-2124): offsets, tints, display flags and scale identical, on the cube and on the trism globe"""
    root = os.path.dirname(HERE)
    monkeypatch.chdir(root)                                   # (the examples' io.open / require paths are relative to the repository root)
    body = open(os.path.join(root, "examples", "lenses", name + ".lua")).read()
    for globe, (W, H) in (("cube", (640, 400)), ("trism", (200, 256))):
        hostctx = bk.Context(bk.ffi.DEVICE_NONE)
        hostctx.load_globe(S.script("globes", globe), globe + ".lua")
        hostctx.load_lens(body, name + ".lua")
        hostctx.resize(W, H)
        info = hostctx.lens_info()
        lm = O.lensmap_with_callbacks(globe, info, lambda x, y: hostctx.eval_host(0, x, y), None, info.onload.decode(), W, H)
        ctx = bk.Context()
        ctx.load_globe(S.script("globes", globe), globe + ".lua")
        ctx.load_lens(body, name + ".lua")
        ctx.set_zoom(*S.zoom_args(info.onload.decode()))
        ctx.resize(W, H)
        display, scale = ctx.build()
        off, tin = ctx.read_lensmap()
        assert lm.built and scale == lm.scale and display[: lm.numplates] == lm.display, (name, globe)
        assert (off != O.NULL).sum() > W * H // 4, (name, globe)
        bad = int((off != lm.offsets).sum())
        assert bad == 0, f"{name}/{globe}: {bad} of {off.size} lensmap entries differ"
        np.testing.assert_array_equal(tin, lm.tints)
        ctx.close()
        hostctx.close()
