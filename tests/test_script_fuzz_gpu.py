"""Differential fuzz of the script surface: random lens scripts inside the supported Lua subset are run by the
host interpreter (portable libm) and by the generated device code; every raw result of lens_inverse /
lens_forward must be bit-identical (values, NaNs, nil vs numbers, count of results).  The shipped lenses only
exercise the constructs their authors happened to use; this walks the emitter through operator precedence,
short-circuit and/or, script functions with upvalues, numeric for / while / repeat loops, local array tables,
multiple assignment and multiple returns."""
import os

import numpy as np
import pytest

import scripts as S

from scriptgen import Gen, WideGen

pytestmark = pytest.mark.gpu


# found by a campaign over build seeds 138..938 (round 3): forward scripts whose NaN projections give a quad one bound at INT_MIN
# and the other at exactly 0 - it passes the reference's size check (abs(INT_MIN) == INT_MIN) and is scanned over 2^31 rows.
# The oracle does scan them (tens of seconds per such quad on one core), so only the two cheapest stay in the suite; the others -
# 242, 383, 608, 917 - pass as well (BLINKY_FUZZ_BUILD_SEEDS=242:243 ...).
EXTRA_BUILD_SEEDS = [368, 605]


def _seeds(default, env, extra=()):
    """the committed seed range, or (developer campaigns) `lo:hi` from the environment"""
    v = os.environ.get(env)
    if not v:
        return list(range(default)) + list(extra)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seeds(60, "BLINKY_FUZZ_EVAL_SEEDS"))
def test_random_scripts_device_equals_host_interpreter(seed):
    forward = seed % 3 == 2
    _device_equals_host_interpreter(Gen(1000 + seed).script(forward), seed, forward)


@pytest.mark.parametrize("seed", _seeds(48, "BLINKY_FUZZ_WIDE_EVAL_SEEDS"))
def test_wide_random_scripts_device_equals_host_interpreter(seed):
    """the generator widened by round 3's constructs (tests/scriptgen.py WideGen: nested functions, functions as arguments, records,
    matrices, varargs / select, constant tables and objects as arguments, # of a constant table), lens_inverse and lens_forward, on
    the DEVICE (round 3 had them on the host emulation of the generated code only)"""
    forward = seed % 3 == 2
    _device_equals_host_interpreter(WideGen((13000 if forward else 9000) + seed).script(forward), seed, forward)


def _device_equals_host_interpreter(src, seed, forward):
    import blinky_amd
    ctx = blinky_amd.Context()
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, f"fuzz{seed}.lua")
    ctx.resize(64, 48)
    rng = np.random.default_rng(seed)
    if forward:
        a = rng.normal(size=(300, 3))
        a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
        which = 1
    else:
        a = np.concatenate([rng.uniform(-3, 3, (280, 2)), [[0, 0], [1, -1], [1e-12, 2.5], [-3, 3]]])
        which = 0
    d_out, d_n = ctx.eval_device(which, a)
    h_out, h_n = ctx.eval_host_many(which, a)
    np.testing.assert_array_equal(d_n, h_n, err_msg=src)
    nan_d, nan_h = np.isnan(d_out), np.isnan(h_out)
    np.testing.assert_array_equal(nan_d, nan_h, err_msg=src)
    assert np.array_equal(d_out[~nan_h].view(np.uint64), h_out[~nan_h].view(np.uint64)), src
    assert (d_n > 0).any(), "degenerate script: every point returned nil\n" + src
    ctx.close()


@pytest.mark.parametrize("seed", _seeds(18, "BLINKY_FUZZ_BUILD_SEEDS", EXTRA_BUILD_SEEDS))
def test_random_scripts_build_the_oracle_table(seed):
    forward = seed % 3 == 2
    _builds_the_oracle_table(Gen(5000 + seed).script(forward), seed, forward)


@pytest.mark.parametrize("seed", _seeds(24, "BLINKY_FUZZ_WIDE_BUILD_SEEDS"))
def test_wide_random_scripts_build_the_oracle_table(seed):
    """the widened generator through the whole GPU build, inverse and forward"""
    forward = seed % 3 == 2
    _builds_the_oracle_table(WideGen(17000 + seed).script(forward), seed, forward)


DECLINED_WRAP = """
local function bk_down(v, n) if n == 0 then return v end return bk_down(v, n - 1) end
local bk_inner = %s
function %s(a, b, c) return bk_inner(bk_down(a, 2), b, c) end
"""


@pytest.mark.parametrize("mode", ["sequential", "declined"])
@pytest.mark.parametrize("seed", _seeds(12, "BLINKY_FUZZ_HOST_SEEDS"))
def test_random_scripts_through_the_host_paths_build_the_oracle_table(seed, mode):
    """(r6) the same random scripts through the HOST evaluation of the callbacks: `sequential` = bk_set_sequential_build(2), one scan in the
    reference's order (inverse: compiled host module or interpreter; forward: the interpreter in resume_lensmap_forward's call order);
    `declined` = the callback routed through a recursive helper, which the GPU emitter declines - the interpreter on the worker pool, or
    the one scan when the script carries state.  Same oracle, same verdicts (built / malformed / run-time error), same tables."""
    forward = seed % 3 == 2
    gen = (WideGen if seed % 2 else Gen)(21000 + seed)
    src = gen.script(forward)
    if mode == "declined":
        cb = "lens_forward" if forward else "lens_inverse"
        src = src + DECLINED_WRAP % (cb, cb)
    _builds_the_oracle_table(src, seed, forward, host_path=mode)


def _builds_the_oracle_table(src, seed, forward, host_path=None):
    """The whole build on random scripts: the GPU lensmap (inverse map for inverse scripts, the forward scatter for
    forward scripts - whose garbage projections push draw_quad through NaNs, huge coordinates and degenerate quads)
    against the oracle's fisheye.c restatement with its callbacks evaluated by the host interpreter on the same
    portable libm: offsets, tints, display flags, scale and the built / not-built verdict must be identical."""
    import blinky_amd
    import oracle_ffi as O
    if forward:
        W, H = 72, 48
    else:
        W, H = 96, 64
        src = src.replace('onload = "f_fov 90"', 'lens_width = 5\nlens_height = 3.5\nonload = "f_contain"')
    host = blinky_amd.Context(blinky_amd.ffi.DEVICE_NONE)
    host.set_host_math(True)
    host.load_globe(S.script("globes", "cube"), "cube.lua")
    host.load_lens(src, f"fuzz{seed}.lua")
    host.resize(W, H)
    info = host.lens_info()
    inv = (lambda x, y: host.eval_host(0, x, y)) if info.has_inverse else None
    fwd = (lambda x, y, z: host.eval_host(1, x, y, z)) if info.has_forward else None
    lm = O.lensmap_with_callbacks("cube", info, inv, fwd, info.onload.decode(), W, H, portable=True)
    ctx = blinky_amd.Context()
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, f"fuzz{seed}.lua")
    ctx.set_zoom(*S.zoom_args(info.onload.decode()))
    ctx.resize(W, H)
    if host_path == "sequential":
        ctx.set_sequential_build(2)
    try:
        display, scale = ctx.build()
        built = True
    except blinky_amd.ffi.BlinkyError:
        built = False
    assert built == lm.built, src
    if host_path and (built or ctx.last_build_path()[0]):
        # (a build that failed before the callbacks were looked at - calc_zoom - has no path; otherwise it must be one of the host's)
        path, why = ctx.last_build_path()
        assert path in ((2,) if host_path == "sequential" else (1, 2)), (path, why, src)
        assert host_path == "sequential" or "recursion" in why, why
    if built:
        off, tin = ctx.read_lensmap()
        assert scale == lm.scale or (scale != scale and lm.scale != lm.scale), src      # (a NaN scale builds an empty map, as in the reference)
        bad = int((off != lm.offsets).sum())
        assert bad == 0, f"{bad} of {off.size} entries differ\n{src}"
        np.testing.assert_array_equal(tin, lm.tints)
        assert display[: lm.numplates] == lm.display
    ctx.close()
    host.close()
