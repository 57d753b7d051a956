"""The compiled host module (include/blinky_hip.h: bk_set_host_compile / bk_host_module_ready): the generated lens code,
compiled for the HOST against the platform libm and dlopen'ed, re-derives the entries a build flags.  It must give, entry for
entry, what the script interpreter gives (the two are interchangeable inside bk_build: whichever is there answers) - and
both equal the CPU oracle on the platform libm.  No GPU needed: a BK_DEVICE_NONE context generates and compiles."""
import shutil

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.skipif(not (shutil.which("c++") or shutil.which("g++") or shutil.which("clang++")), reason="no host C++ compiler")


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


@pytest.fixture(autouse=True)
def _modes(bk, tmp_path_factory, monkeypatch):
    monkeypatch.setenv("BLINKY_HIP_CACHE", str(tmp_path_factory.getbasetemp() / "hostmod_cache"))
    yield
    bk.debug_set_option("host_module", 0)


def host_ctx(bk, globe, lens, zoom, W, H):
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    info = S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.calc_zoom()
    return ctx, info


def both_ways(bk, ctx, fn, ids):
    bk.debug_set_option("host_module", 2)             # the interpreter
    a = fn(ids)
    bk.debug_set_option("host_module", 1)             # the compiled module (waited for)
    b = fn(ids)
    bk.debug_set_option("host_module", 0)
    return a, b


@pytest.mark.parametrize("lens", S.LENSES)
def test_module_equals_interpreter_on_every_lens(bk, lens):
    W, H = 640, 400
    ctx, info = host_ctx(bk, "cube", lens, None, W, H)
    rng = np.random.default_rng(7)
    if info.has_inverse:
        # random pixels plus whole rows and columns through the centre: the symmetry lines where results hinge on libm's last bits
        ids = np.concatenate([rng.integers(0, W * H, 3000), np.arange(W) + (H // 2) * W, np.arange(H) * W + W // 2,
                              np.arange(min(W, H)) * (W + 1)]).astype(np.uint32)
        ids.sort()
        (off_i, tin_i), (off_m, tin_m) = both_ways(bk, ctx, ctx.host_entries, ids)
        np.testing.assert_array_equal(off_m, off_i)
        np.testing.assert_array_equal(tin_m, tin_i)
        assert ctx.host_module_ready()
    if info.has_forward:
        ps = min(W, H)
        ids = np.sort(rng.integers(0, 6 * (ps + 1) * (ps + 1), 3000).astype(np.uint32))
        (sx_i, sy_i, ok_i), (sx_m, sy_m, ok_m) = both_ways(bk, ctx, ctx.host_corners, ids)
        np.testing.assert_array_equal(ok_m, ok_i)
        np.testing.assert_array_equal(sx_m, sx_i)
        np.testing.assert_array_equal(sy_m, sy_i)
    ctx.close()


@pytest.mark.parametrize("cfg", [("cube", "quincuncial", None, 480, 480), ("trism", "panini", "f_fov 200", 400, 300),
                                 ("fast", "stereographic", None, 320, 320), ("tetra", "debug", None, 256, 256)])
def test_module_equals_the_platform_oracle_on_whole_tables(bk, cfg):
    """every entry of the table, derived by the compiled module alone, is the CPU oracle's (platform libm) entry"""
    globe, lens, zoom, W, H = cfg
    lm = O.lensmap(globe, lens, zoom, W, H)
    ctx, _ = host_ctx(bk, globe, lens, zoom, W, H)
    bk.debug_set_option("host_module", 1)
    off, tin = ctx.host_entries(np.arange(W * H, dtype=np.uint32))
    np.testing.assert_array_equal(off, lm.offsets)
    np.testing.assert_array_equal(tin, lm.tints)
    ctx.close()


def test_module_is_cached_on_disk_and_optional(bk, tmp_path, monkeypatch):
    monkeypatch.setenv("BLINKY_HIP_CACHE", str(tmp_path))

    def variant(lens, tag):                      # a source no other test of this process has compiled (line numbers move)
        ctx = bk.Context(bk.ffi.DEVICE_NONE)
        ctx.load_globe(S.script("globes", "cube"), "cube.lua")
        ctx.load_lens("\n" * tag + "-- (variant)\n" + S.script("lenses", lens), lens + "_variant.lua")
        ctx.set_zoom(bk.ffi.ZOOM_CONTAIN)
        ctx.resize(320, 200)
        ctx.calc_zoom()
        return ctx
    ctx = variant("gumby", 3)
    assert ctx.host_module_ready(wait=True)
    files = [f.name for f in tmp_path.iterdir()]
    assert any(f.startswith("bk_host_") and f.endswith(".so") for f in files), files
    assert not any("build" in f for f in files), files            # the scratch directory is gone
    # the shared object is sealed (SHA-256 trailer) and private; a tampered one is not loaded: the next process compiles again
    import hashlib
    import stat
    (so,) = [f for f in tmp_path.iterdir() if f.name.startswith("bk_host_")]
    blob = so.read_bytes()
    assert stat.S_IMODE(so.stat().st_mode) == 0o600 and blob[-40:-32] == b"BKSHA256" and blob[-32:] == hashlib.sha256(blob[:-40]).digest()
    assert len(so.stem) == len("bk_host_") + 32
    # host math switched to the portable libm: the module (platform libm only) must not answer
    ctx.set_host_math(True)
    assert not ctx.host_module_ready(wait=True)
    ctx.close()
    # no compiler: the interpreter answers, nothing fails
    monkeypatch.setenv("BLINKY_HIP_HOSTCXX", "off")
    ctx = variant("fahey", 5)
    assert not ctx.host_module_ready(wait=True)
    off, tin = ctx.host_entries(np.arange(1000, dtype=np.uint32))
    assert off.shape == (1000,)
    ctx.close()
