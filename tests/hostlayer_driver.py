"""Scenarios for tests/test_host_layer.py, each run in a process of its own (the C host layer, like fisheye.c, keeps its
state in file-scope statics):   python tests/hostlayer_driver.py <scenario> <basedir>
frames   - whole F_RenderView frames through libhosttest.so compared byte for byte with the oracle (single context, or
           several stripe contexts when BLINKY_HIP_DEVICES lists more than one device)
async    - with asynchronous lens compilation (the default) no frame waits for hiprtc: the previous lensmap keeps being
           drawn until the new module is ready
complete - tab completion of f_lens / f_globe
malformed - a lens whose lens_inverse returns a malformed result for part of the screen: the reference's scan stops there and keeps
           what it had set (fisheye.c:2113-2115); the host layer must draw that partial table and render exactly its plates"""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]
HOSTLIB = os.path.join(HERE, "host", "libhosttest.so")


def load():
    h = C.CDLL(HOSTLIB)
    h.hosttest_console.restype = C.c_char_p
    h.hosttest_plate_fov.restype = C.c_double
    return h


def render(h, O, globe, lens, zoom, W, H, x0, y0, extra, rubix=False, bg=7, fidx=2, pal=None, order_of=None):
    """one F_RenderView; returns (frame as rendered, the oracle's frame, plates the host rendered, seconds)"""
    lm = O.lensmap(globe, lens, zoom, W, H)
    olm = O.lensmap(*order_of) if order_of else lm               # whose display[] decides which plates get painted
    h.hosttest_resize(W, H, x0, y0, extra)
    order = [i for i, d in enumerate(olm.display) if d]
    arr = (C.c_int * len(order))(*order)
    pitch, vh = h.hosttest_rowbytes(), h.hosttest_vidheight()
    out = np.zeros((vh, pitch), np.uint8)
    t0 = time.perf_counter()
    n = h.hosttest_frame(arr, len(order), fidx, bg, out.ctypes.data_as(C.c_void_p))
    dt = time.perf_counter() - t0
    want = np.full((vh, pitch), bg, np.uint8)
    # Draw_TileClear repaints [0,vid.width) x [0,vid.height); the plate renders are overwritten by it
    O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, fidx), want, pitch, x0, y0, rubix, pal)
    return out[:, : pitch - extra], want[:, : pitch - extra], n, len(order), dt, lm


def scenario_frames(base):
    import blinky_amd  # noqa: F401  (loads torch's HIP runtime before libblinkyhip, see blinky_amd/ffi.py)
    import oracle_ffi as O
    os.environ["BLINKY_HIP_SYNC_COMPILE"] = "1"
    h = load()
    assert h.hosttest_init(base.encode()) == 1, h.hosttest_console().decode()
    pal = O.palmap(O.synthetic_basepal())

    def frame(*a, **kw):
        got, want, n, nwant, _, lm = render(h, O, *a, pal=pal, **kw)
        assert n == nwant, "the host must render exactly the plates the lensmap uses (display[])"
        np.testing.assert_array_equal(got, want)
        return lm

    # defaults of F_Init: cube / panini / f_fov 180
    frame("cube", "panini", None, 320, 200, 8, 4, 16)
    assert abs(h.hosttest_plate_fov(0) - float(O.globe_plates("cube")[0][9])) == 0     # fisheye_plate_fov = plate fov
    h.hosttest_cmd(b"f_rubix")                  # same lens, rubix overlay on
    frame("cube", "panini", None, 320, 200, 8, 4, 16, rubix=True)
    h.hosttest_cmd(b"f_rubix")
    # a lens whose onload changes the zoom, another globe, a resize, an odd origin
    h.hosttest_cmd(b"f_globe trism")
    h.hosttest_cmd(b"f_lens hammer")
    frame("trism", "hammer", None, 322, 203, 3, 1, 5)
    h.hosttest_cmd(b"f_fov 150")
    frame("trism", "hammer", "f_fov 150", 322, 203, 3, 1, 5)
    h.hosttest_cmd(b"f_globe cube")             # forward-only lens
    h.hosttest_cmd(b"f_lens eckert5")
    frame("cube", "eckert5", None, 200, 120, 0, 0, 0)
    # an invalid lens blanks the view (fisheye.c:737-741, 2372): only the cleared background remains
    h.hosttest_console_clear()
    h.hosttest_cmd(b"f_lens doesnotexist")
    h.hosttest_resize(200, 120, 0, 0, 0)
    out = np.zeros((h.hosttest_vidheight(), h.hosttest_rowbytes()), np.uint8)
    h.hosttest_frame((C.c_int * 1)(0), 0, 0, 9, out.ctypes.data_as(C.c_void_p))
    assert (out == 9).all()
    assert "not a valid lens" in h.hosttest_console().decode()
    h.hosttest_shutdown()
    print("frames ok")


def scenario_async(base):
    import blinky_amd  # noqa: F401
    import oracle_ffi as O
    os.environ.pop("BLINKY_HIP_SYNC_COMPILE", None)
    os.environ["BLINKY_HIP_CACHE"] = os.path.join(base, "fresh-cache")     # nothing compiled yet
    h = load()
    assert h.hosttest_init(base.encode()) == 1, h.hosttest_console().decode()
    W, H = 2040, 1200                          # the engine's largest mode (MAXWIDTH x MAXHEIGHT)
    worst, pending_frames = 0.0, 0
    # panini compiles in the background: until it is ready nothing is drawn (there is no previous lensmap), then it appears
    t0 = time.perf_counter()
    while True:
        got, want, n, nwant, dt, _ = render(h, O, "cube", "panini", None, W, H, 0, 0, 0)
        worst = max(worst, dt)
        if np.array_equal(got, want):
            break
        assert (got == 7).all(), "while the first lens compiles only the cleared background may be shown"
        pending_frames += 1
        assert time.perf_counter() - t0 < 120
    first_ready = time.perf_counter() - t0
    # now switch lens: the frames in between keep showing panini, then hammer takes over
    h.hosttest_cmd(b"f_lens hammer")
    shown_old = 0
    t0 = time.perf_counter()
    while True:
        got, want_old, _, _, dt, _ = render(h, O, "cube", "panini", None, W, H, 0, 0, 0)
        worst = max(worst, dt)
        if np.array_equal(got, want_old):
            shown_old += 1
            assert time.perf_counter() - t0 < 120
            continue
        got, want_new, n, nwant, dt, _ = render(h, O, "cube", "hammer", None, W, H, 0, 0, 0)
        worst = max(worst, dt)
        np.testing.assert_array_equal(got, want_new)
        assert n == nwant
        break
    h.hosttest_shutdown()
    print(f"async ok: {pending_frames} blank frames while panini compiled ({first_ready * 1e3:.0f} ms), {shown_old} panini frames while "
          f"hammer compiled, slowest F_RenderView {worst * 1e3:.1f} ms")
    assert pending_frames >= 1 and shown_old >= 1, "hiprtc was expected to take longer than one frame"
    assert worst < 0.25, f"a frame stalled for {worst * 1e3:.0f} ms"


MALFORMED_TAIL = """
local good = lens_inverse
function lens_inverse(x, y)
   if x > 0.3 and y > 0.2 then
      return x, y            -- two values: LUAtoC_lens_inverse's status -1 (fisheye.c:1579-1584)
   end
   return good(x, y)
end
"""


def scenario_malformed(base):
    import blinky_amd  # noqa: F401
    import oracle_ffi as O
    import scripts as S
    os.environ["BLINKY_HIP_SYNC_COMPILE"] = "1"
    with open(os.path.join(base, "lua-scripts", "lenses", "panini_malformed.lua"), "w") as f:
        f.write(S.script("lenses", "panini") + MALFORMED_TAIL)
    h = load()
    assert h.hosttest_init(base.encode()) == 1, h.hosttest_console().decode()
    W, H, fidx, bg = 320, 200, 2, 7
    # the previous lens shows every plate of its own; the malformed one must not inherit those flags
    h.hosttest_cmd(b"f_lens hammer")
    h.hosttest_resize(W, H, 0, 0, 0)
    pitch, vh = h.hosttest_rowbytes(), h.hosttest_vidheight()
    out = np.zeros((vh, pitch), np.uint8)
    h.hosttest_frame((C.c_int * 6)(0, 1, 2, 3, 4, 5), 6, fidx, bg, out.ctypes.data_as(C.c_void_p))
    h.hosttest_console_clear()
    h.hosttest_cmd(b"f_lens panini_malformed")
    h.hosttest_cmd(b"f_fov 180")
    lm = O.lensmap("cube", "panini", "f_fov 180", W, H)
    ly, lx = np.divmod(np.arange(W * H), W)
    x, y = (lx - W // 2) * lm.scale, -(ly - H // 2) * lm.scale
    key = ly * W + (W - 1 - lx)                     # larger = earlier in the reference's scan (rows from the bottom up)
    first = key[(x > 0.3) & (y > 0.2)].max()
    off = np.where(key > first, lm.offsets, O.NULL).astype(np.uint32)
    tin = np.where(key > first, lm.tints, 255).astype(np.uint8)
    plates = sorted(set((off[off != O.NULL] // (lm.ps * lm.ps)).tolist()))
    assert 0 < len(plates) < sum(lm.display), "the scenario is meant to cut the table before every plate of panini is in use"
    arr = (C.c_int * len(plates))(*plates)
    n = h.hosttest_frame(arr, len(plates), fidx, bg, out.ctypes.data_as(C.c_void_p))
    assert n == len(plates), f"the host rendered {n} plates, the partial table uses {plates}"
    want = np.full((vh, pitch), bg, np.uint8)
    O.apply(off, tin, W, H, O.lcg_globe(lm.ps, lm.numplates, fidx), want, pitch, 0, 0, False, None)
    np.testing.assert_array_equal(out, want)
    assert "malformed" in h.hosttest_console().decode() or "lens_inverse" in h.hosttest_console().decode()
    h.hosttest_shutdown()
    print("malformed ok")


def scenario_complete(base):
    os.environ["BLINKY_HIP_DEVICE"] = "none"
    h = load()
    h.hosttest_init(base.encode())
    buf = C.create_string_buffer(4096)
    n = h.hosttest_complete(b"f_lens", b"pa", buf, 4096)
    assert (n, buf.value.decode().split()) == (1, ["panini"]), (n, buf.value)
    n = h.hosttest_complete(b"f_lens", b"e", buf, 4096)
    assert buf.value.decode().split() == ["eckert1", "eckert4", "eckert5", "equirect"]
    n = h.hosttest_complete(b"f_globe", b"", buf, 4096)
    assert (n, buf.value.decode().split()) == (6, ["cube", "cube_corner", "cube_edge", "fast", "tetra", "trism"]), buf.value
    assert h.hosttest_complete(b"f_fov", b"", buf, 4096) == -1       # no completion registered for other commands
    print("complete ok")


REFLIB = os.path.join(ROOT, "oracle", "_ref", "libhosttest_ref.so")


def scenario_differential(base, seed_range="0:4", frames="1"):
    """The same random session - console commands, resizes, frames - put to the product's host layer (libhosttest.so) and to the
    UNMODIFIED reference behind the same engine stand-in (oracle/_ref/libhosttest_ref.so: fisheye.c's own F_Init / commands /
    F_RenderView): after every step the console text and the config must be identical, and after every frame the screen, the number of
    plate views the engine was asked for and their fov.  frames = "0": commands only (no GPU needed)."""
    import scripts as S
    with_frames = frames == "1"
    if with_frames:
        import blinky_amd  # noqa: F401
        os.environ["BLINKY_HIP_SYNC_COMPILE"] = "1"
    else:
        os.environ["BLINKY_HIP_DEVICE"] = "none"
    lo, hi = [int(v) for v in seed_range.split(":")]
    mine, ref = load(), C.CDLL(REFLIB)
    ref.hosttest_console.restype = C.c_char_p
    ref.hosttest_plate_fov.restype = C.c_double
    both = (mine, ref)
    for h in both:
        assert h.hosttest_init(base.encode()) == 1
    ref.hosttest_ref_build_to_completion()
    buf = C.create_string_buffer(8192)

    def config(h):
        h.hosttest_writeconfig(buf, 8192)
        return buf.value.decode()

    def console(h):
        t = h.hosttest_console().decode()
        h.hosttest_console_clear()
        return t

    def same_text(what):
        a, b = console(mine), console(ref)
        assert a == b, f"{what}: console differs\n--- product ---\n{a}\n--- reference ---\n{b}"
        a, b = config(mine), config(ref)
        assert a == b, f"{what}: config differs\n--- product ---\n{a}\n--- reference ---\n{b}"

    same_text("after F_Init")
    nframes = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(4000 + seed)
        W, H, x0, y0, extra = 320, 200, 0, 0, 0
        for h in both:
            h.hosttest_resize(W, H, x0, y0, extra)
        for step in range(int(rng.integers(10, 24))):
            k = rng.random()
            deg = int(rng.choice([10, 60, 90, 120, 150, 180, 181, 200, 270, 360, 400]))
            if k < 0.16:
                cmd = "f_lens " + (str(rng.choice(S.LENSES)) if rng.random() < 0.93 else "nosuchlens")
            elif k < 0.26:
                cmd = "f_globe " + (str(rng.choice(S.GLOBES)) if rng.random() < 0.93 else "nosuchglobe")
            elif k < 0.42:
                cmd = str(rng.choice([f"f_fov {deg}", f"f_vfov {deg}", "f_cover", "f_contain", "f_fov", "f_vfov", f"f_fov {deg}.7"]))
            elif k < 0.50:
                cmd = str(rng.choice(["f_rubix", f"f_rubixgrid {int(rng.integers(1, 20))} {float(rng.choice([0.5, 1, 2, 4]))} {float(rng.choice([0, 0.5, 1]))}",
                                      "f_rubixgrid", "f_rubixgrid 3 2"]))
            elif k < 0.56:
                cmd = str(rng.choice(["f_help", "fisheye", "f_lens", "f_globe", "f_shortcutkeys", "f_saveglobe", f"f_saveglobe shot{step}",
                                      f"f_saveglobe full{step} 1"]))
            elif k < 0.66:
                W, H = [(320, 200), (200, 120), (333, 217), (160, 240), (400, 300), (64, 48)][int(rng.integers(0, 6))]
                x0, y0, extra = int(rng.integers(0, 9)), int(rng.integers(0, 5)), int(rng.integers(0, 17))
                for h in both:
                    h.hosttest_resize(W, H, x0, y0, extra)
                cmd = None
            else:
                cmd = "frame"
            what = f"seed {seed} step {step}: {cmd or f'resize {W}x{H}+{x0}+{y0} pitch+{extra}'}"
            if cmd == "frame":
                if not with_frames:
                    continue
                fidx, bg = int(rng.integers(0, 5)), int(rng.integers(0, 256))
                order = (C.c_int * 6)(0, 1, 2, 3, 4, 5)
                outs, ns = [], []
                for h in both:
                    pitch, vh = h.hosttest_rowbytes(), h.hosttest_vidheight()
                    out = np.zeros((vh, pitch), np.uint8)
                    ns.append(h.hosttest_frame(order, 6, fidx, bg, out.ctypes.data_as(C.c_void_p)))
                    outs.append(out[:, : pitch - extra] if extra else out)
                # (also when the zoom cannot be computed: the reference returns from create_lensmap before it resets the display flags,
                #  fisheye.c:2376-2384, and goes on rendering the PREVIOUS lensmap's plates into a lensmap that shows none of them)
                assert ns[0] == ns[1], f"{what}: the engine was asked for {ns[0]} plate views, the reference asks for {ns[1]}"
                for i in range(ns[0]):
                    assert mine.hosttest_plate_fov(i) == ref.hosttest_plate_fov(i), f"{what}: plate view {i}: fisheye_plate_fov differs"
                if not np.array_equal(outs[0], outs[1]):
                    bad = np.argwhere(outs[0] != outs[1])
                    raise AssertionError(f"{what}: {len(bad)} screen bytes differ, first at (y, x) = {tuple(bad[0])}; config:\n{config(ref)}")
                # the globe screenshots a pending f_saveglobe wrote during this frame (WritePCXplate, fisheye.c:1396-1484)
                nf = [h.hosttest_num_files() for h in both]
                assert nf[0] == nf[1], f"{what}: {nf[0]} files written, the reference writes {nf[1]}"
                for i in range(nf[0]):
                    got = []
                    for h in both:
                        name, data = C.create_string_buffer(128), np.zeros(1 << 20, np.uint8)
                        n = h.hosttest_file(i, name, data.ctypes.data_as(C.c_void_p), data.size)
                        got.append((name.value, bytes(data[:n])))
                    # (a plate the lensmap does not show is not rendered, and its pixels in a screenshot are whatever the globe buffer held: in the
                    #  reference uninitialised memory after a resize - compared only when every plate was rendered this frame)
                    if got[0][0] != got[1][0]:
                        raise AssertionError(f"{what}: file {i} is called {got[0][0]}, the reference's {got[1][0]}")
                    if got[0] != got[1] and ns[0] == nf[0]:
                        a, b = np.frombuffer(got[0][1], np.uint8), np.frombuffer(got[1][1], np.uint8)
                        m = min(a.size, b.size)
                        d = np.flatnonzero(a[:m] != b[:m])
                        raise AssertionError(f"{what}: file {i} ({got[0][0]} / {got[1][0]}) differs: {a.size} / {b.size} bytes, {d.size} of the common "
                                             f"{m} differ, first at {d[:5]}: {a[d[:5]]} vs {b[d[:5]]}; {W}x{H}; config:\n{config(ref)}")
                for h in both:
                    h.hosttest_clear_files()
                nframes += 1
            elif cmd:
                for h in both:
                    h.hosttest_cmd(cmd.encode())
            same_text(what)
    mine.hosttest_shutdown()
    print(f"differential ok: seeds {lo}..{hi - 1}, {nframes} frames compared")


if __name__ == "__main__":
    {"frames": scenario_frames, "async": scenario_async, "complete": scenario_complete, "malformed": scenario_malformed,
     "differential": scenario_differential}[sys.argv[1]](*sys.argv[2:])
