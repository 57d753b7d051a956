"""The script fuzz of tests/test_script_fuzz_gpu.py on the CPU, with its generator widened by the constructs round 3 added to the code
generator: functions defined inside the callback (`local function`, `local f = function`, one inside another) that close over its
parameters, locals and tables; script functions and builtins passed as arguments, passed on, and given local names; locals of the chunk
that the callback assigns; the length of a constant table of the chunk.  Every random script is run by the host interpreter (portable
libm) and by the generated code on the host emulation (tests/hostemu: the translation unit hiprtc gets, compiled by g++): every raw
result of lens_inverse - values bit for bit, NaNs, nil against numbers, the count of results - must agree on every pixel."""
import os

import numpy as np
import pytest

import scripts as S
from scriptgen import WideGen


def _seeds():
    v = os.environ.get("BLINKY_FUZZ_CPU_SEEDS", "0:24")
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seeds())
def test_random_scripts_generated_code_equals_host_interpreter(seed):
    """BLINKY_FUZZ_CPU_SEEDS=lo:hi runs a longer campaign"""
    import blinky_amd as bk
    from hostemu import emu
    src = WideGen(9000 + seed).script(False).replace('onload = "f_fov 90"', 'lens_width = 5\nlens_height = 3.5\nonload = "f_contain"')
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, f"wide{seed}.lua")
    ctx.set_zoom(bk.ffi.ZOOM_CONTAIN, 0)
    ctx.resize(48, 32)
    v = emu.inverse_values(ctx)
    assert (v["err"] == 0).all(), src
    xy = np.stack([v["x"], v["y"]], axis=1)
    h_out, h_n = ctx.eval_host_many(0, xy)
    np.testing.assert_array_equal(v["nret"], h_n, err_msg=src)
    d_out = v["val"][:, : h_out.shape[1]]
    used = np.arange(h_out.shape[1])[None, :] < h_n[:, None]
    nan_d, nan_h = np.isnan(d_out) & used, np.isnan(h_out) & used
    np.testing.assert_array_equal(nan_d, nan_h, err_msg=src)
    same = d_out.view(np.uint64) == h_out.view(np.uint64)
    assert (same | ~used | nan_h).all(), src
    assert (h_n > 0).any(), "degenerate script: every pixel returned nil\n" + src


@pytest.mark.parametrize("seed", [s for s in _seeds() if s % 3 == 2])
def test_random_forward_scripts_generated_code_equals_host_interpreter(seed):
    """the same for lens_forward: the generated function on random unit rays (as floats, the way the build hands them over)"""
    import blinky_amd as bk
    from hostemu import emu
    src = WideGen(13000 + seed).script(True)
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, f"widef{seed}.lua")
    ctx.set_zoom(bk.ffi.ZOOM_FOV, 90)
    ctx.resize(48, 32)
    rng = np.random.default_rng(seed)
    rays = rng.normal(size=(400, 3))
    rays = (rays / np.linalg.norm(rays, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    try:
        v = emu.forward_values(ctx, rays)
    except bk.BlinkyError as e:
        if "scale" in str(e) or "zoom" in str(e).lower() or "fov" in str(e).lower():
            pytest.skip("the random lens_forward gives this zoom no usable scale: " + str(e))      # (calc_zoom's verdict, not the generated code's)
        raise
    assert (v["err"] == 0).all(), src
    h_out, h_n = ctx.eval_host_many(1, rays)
    np.testing.assert_array_equal(v["nret"], h_n, err_msg=src)
    d_out = v["val"][:, : h_out.shape[1]]
    used = np.arange(h_out.shape[1])[None, :] < h_n[:, None]
    nan_d, nan_h = np.isnan(d_out) & used, np.isnan(h_out) & used
    np.testing.assert_array_equal(nan_d, nan_h, err_msg=src)
    assert ((d_out.view(np.uint64) == h_out.view(np.uint64)) | ~used | nan_h).all(), src


SINCOS_MEMORY = {
    # the operand is assigned between the two calls
    "reassigned": "local t = x\n   local a = sin(t)\n   t = t + y\n   local b = cos(t)\n   return a, b, t",
    # ... through a multiple assignment
    "swapped": "local t, u = x, y\n   local a = cos(t)\n   t, u = u, t\n   local b = sin(t) + cos(u)\n   return a, b, 0",
    # the first call before a loop that changes the operand, the second in its condition and body
    "loop": "local t = x\n   local a = sin(t)\n   local n = 0\n   while cos(t) > -0.5 and n < 6 do\n      t = t + 0.7\n      n = n + sin(t)\n   end\n   return a, n, cos(t)",
    # short-circuit branches are scopes of their own
    "shortcut": "local t = x * 2\n   local a = (y > 0 and sin(t)) or cos(t)\n   local b = cos(t) + sin(t)\n   return a, b, 0",
    # a global and an upvalue of the chunk as operands, assigned in between
    "global": "g = x\n   local a = sin(g)\n   g = g * y\n   local b = cos(g)\n   up = y\n   local c = cos(up)\n   up = up + 1\n   return a, b, c + sin(up)",
    # a record field
    "record": "local p = {u = x, v = y}\n   local a = sin(p.u)\n   p.u = p.u + p.v\n   local b = cos(p.u) + cos(p.v)\n   p.v = 0.25\n   return a, b, sin(p.v)",
    # a script function that assigns the operand behind the caller's back
    "callee": "local t = x\n   local function bump()\n      t = t + y\n      return 1\n   end\n   local a = sin(t)\n   local k = bump()\n   local b = cos(t) * k\n   return a, b, sin(t)",
    # the operand of a call statement's argument, then again after it
    "repeat": "local t = x\n   local a = 0\n   repeat\n      a = a + sin(t) * cos(t)\n      t = t - 0.3\n   until cos(t) < 0.2 or a > 3\n   return a, sin(t), cos(t)",
    # if / elseif conditions are evaluated in nested scopes
    "elseif": "local t = x + y\n   local a = 0\n   if sin(t) > 0.5 then\n      a = cos(t)\n   elseif cos(t) > 0.5 then\n      a = sin(t) * 2\n   else\n      t = t * 2\n      a = sin(t) + cos(t)\n   end\n   return a, cos(t), sin(t)",
}


@pytest.mark.parametrize("name", sorted(SINCOS_MEMORY))
def test_sincos_memory_is_forgotten_when_the_operand_changes(name):
    """sin(v) and cos(v) of one operand share a reduction from one plain assignment to the next (bk_emit.cpp: SinCos) - as long as nothing
    assigned v, no scope was left and no script function ran in between.  Every way of breaking that, against the host interpreter."""
    import blinky_amd as bk
    from hostemu import emu
    src = ("g = 0\nlocal up = 0\nmax_fov = 360\nmax_vfov = 180\nlens_width = 5\nlens_height = 3.5\nonload = \"f_contain\"\n"
           "function lens_inverse(x, y)\n   " + SINCOS_MEMORY[name] + "\nend\n")
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    ctx.set_host_math(True)
    ctx.load_globe(S.script("globes", "cube"), "cube.lua")
    ctx.load_lens(src, name + ".lua")
    ctx.set_zoom(bk.ffi.ZOOM_CONTAIN, 0)
    ctx.resize(40, 28)
    assert "bk_f_sincos(" in ctx.kernel_source(compile=False)
    v = emu.inverse_values(ctx)
    assert (v["err"] == 0).all(), src
    xy = np.stack([v["x"], v["y"]], axis=1)
    h_out, h_n = ctx.eval_host_many(0, xy)
    np.testing.assert_array_equal(v["nret"], h_n, err_msg=src)
    d_out = v["val"][:, : h_out.shape[1]]
    used = np.arange(h_out.shape[1])[None, :] < h_n[:, None]
    same = (d_out.view(np.uint64) == h_out.view(np.uint64)) | (np.isnan(d_out) & np.isnan(h_out))
    assert (same | ~used).all(), src
