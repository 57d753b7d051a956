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

pytestmark = pytest.mark.gpu

UNARY = ["math.sin", "math.cos", "math.atan", "math.sqrt", "math.abs", "math.exp", "math.floor", "math.ceil", "math.tanh",
         "math.asin", "math.acos", "math.log", "math.tan", "math.sinh", "math.cosh"]
BINARY_FN = ["math.atan2", "math.min", "math.max", "math.pow", "math.fmod"]
BINOPS = ["+", "-", "*", "/", "%", "^"]
CMPS = ["<", "<=", ">", ">=", "==", "~="]


class Gen:
    def __init__(self, seed):
        self.r = np.random.default_rng(seed)
        self.n = 0

    def pick(self, xs):
        return xs[int(self.r.integers(0, len(xs)))]

    def num(self):
        k = self.r.integers(0, 6)
        if k == 0:
            return str(int(self.r.integers(-3, 9)))
        if k == 1:
            return "math.pi"
        if k == 2:
            return f"{self.r.uniform(-2, 2):.6g}"
        if k == 3:
            return f"{10.0 ** self.r.uniform(-3, 3):.4e}"
        return f"{self.r.uniform(0, 1):.9f}"

    def expr(self, vars_, depth):
        if depth <= 0 or self.r.random() < 0.2:
            return self.pick(vars_) if self.r.random() < 0.65 else self.num()
        k = self.r.integers(0, 10)
        a = self.expr(vars_, depth - 1)
        if k <= 3:
            return f"({a} {self.pick(BINOPS)} {self.expr(vars_, depth - 1)})"
        if k == 4:
            return f"(- {a})"
        if k == 5:
            return f"{self.pick(UNARY)}({a})"
        if k == 6:
            return f"{self.pick(BINARY_FN)}({a}, {self.expr(vars_, depth - 1)})"
        if k == 7:        # short-circuit value selection (the Lua idiom `c and a or b`)
            return f"(({self.cond(vars_, depth - 1)}) and {a} or {self.expr(vars_, depth - 1)})"
        if k == 8:
            return f"helper({a}, {self.expr(vars_, depth - 1)})"
        return f"(({a}) * 0.5 + {self.pick(vars_)})"

    def cond(self, vars_, depth):
        c = f"{self.expr(vars_, depth)} {self.pick(CMPS)} {self.expr(vars_, depth)}"
        k = self.r.integers(0, 5)
        if k == 0:
            return f"not ({c})"
        if k == 1:
            return f"({c}) and ({self.expr(vars_, depth)} {self.pick(CMPS)} {self.num()})"
        if k == 2:
            return f"({c}) or ({self.expr(vars_, depth)} {self.pick(CMPS)} {self.num()})"
        return c

    def block(self, vars_, depth, indent):
        out = []
        vars_ = list(vars_)
        for _ in range(int(self.r.integers(1, 4))):
            k = self.r.integers(0, 9)
            pad = "  " * indent
            if k <= 2:
                self.n += 1
                v = f"v{self.n}"
                out.append(f"{pad}local {v} = {self.expr(vars_, depth)}")
                vars_.append(v)
            elif k == 3 and len(vars_) > 2:
                out.append(f"{pad}{self.pick(vars_[2:])} = {self.expr(vars_, depth)}")
            elif k == 4:
                out.append(f"{pad}if {self.cond(vars_, depth - 1)} then")
                out += self.block(vars_, depth - 1, indent + 1)[0]
                if self.r.random() < 0.5:
                    out.append(f"{pad}elseif {self.cond(vars_, depth - 1)} then")
                    out += self.block(vars_, depth - 1, indent + 1)[0]
                if self.r.random() < 0.6:
                    out.append(f"{pad}else")
                    out += self.block(vars_, depth - 1, indent + 1)[0]
                out.append(f"{pad}end")
            elif k == 5:
                self.n += 1
                acc, i = f"acc{self.n}", f"i{self.n}"
                out.append(f"{pad}local {acc} = {self.expr(vars_, 1)}")
                step = self.pick(["", ", 2", ", -1"])
                lo, hi = (1, int(self.r.integers(2, 7))) if step != ", -1" else (int(self.r.integers(2, 7)), 1)
                out.append(f"{pad}for {i} = {lo}, {hi}{step} do {acc} = {acc} * 0.75 + {self.expr(vars_ + [i], 2)} end")
                vars_.append(acc)
            elif k == 6:
                self.n += 1
                w, c = f"w{self.n}", f"c{self.n}"
                out.append(f"{pad}local {w}, {c} = {self.expr(vars_, 2)}, 0")
                if self.r.random() < 0.5:
                    out.append(f"{pad}while {c} < {int(self.r.integers(1, 6))} do {w} = math.cos({w}) + {self.pick(vars_)} * 0.125; {c} = {c} + 1 end")
                else:
                    out.append(f"{pad}repeat {w} = {w} * 0.5 + {self.expr(vars_, 1)}; {c} = {c} + 1 until {c} >= {int(self.r.integers(1, 5))} or {w} > 1e6")
                vars_.append(w)
            elif k == 7:      # a local array table: constant and computed indices, element stores, the length operator
                self.n += 1
                t, i = f"t{self.n}", f"j{self.n}"
                size = int(self.r.integers(2, 5))
                elems = [self.expr(vars_, 2) for _ in range(size)]
                if elems[-1].startswith("helper("):
                    elems[-1] = f"({elems[-1]})"          # a script function in the last slot must be truncated to one value
                out.append(f"{pad}local {t} = {{{', '.join(elems)}}}")
                out.append(f"{pad}{t}[{int(self.r.integers(1, size + 1))}] = {self.expr(vars_, 2)}")
                out.append(f"{pad}for {i} = 1, #{t} do {t}[{i}] = {t}[{i}] + {t}[({i} % #{t}) + 1] * 0.5 end")
                vars_ += [f"{t}[{c}]" for c in range(1, size + 1)]
            else:
                self.n += 1
                a, b = f"p{self.n}", f"q{self.n}"
                out.append(f"{pad}local {a}, {b} = pair({self.expr(vars_, 2)}, {self.expr(vars_, 2)})")
                vars_ += [a, b]
        return out, vars_

    def script(self, forward):
        args = ["x", "y", "z"] if forward else ["x", "y"]
        pre = []
        start = list(args)
        if forward and self.r.random() < 0.7:
            # an equirectangular base perturbed by bounded noise keeps the scatter on the screen
            pre = ["  local lat, lon = ray_to_latlon(x, y, z)"]
            start += ["lat", "lon"]
        body, vars_ = self.block(start, 3, 1)
        body = pre + body
        if pre:
            rets = f"lon + 0.3 * math.sin({self.expr(vars_, 2)}), lat + 0.2 * math.cos({self.expr(vars_, 2)})"
        else:
            rets = ", ".join(self.expr(vars_, 2) for _ in range(2 if forward else 3))
        name = "lens_forward" if forward else "lens_inverse"
        return "\n".join([
            "local bias = 0.25",
            "local function helper(a, b) if a > b then return a - b * bias end return (a + b) * 0.5 end",
            "local function pair(a, b) return a + b, a * b - bias end",
            f"function {name}({', '.join(args)})",
            *body,
            f"  if ({self.cond(vars_, 1)}) and {'z < -0.6' if forward else 'x > 1.5'} then return nil end",
            f"  return {rets}",
            "end",
            "max_fov = 360", "max_vfov = 180", 'onload = "f_fov 90"',
        ])


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
    import blinky_amd
    forward = seed % 3 == 2
    src = Gen(1000 + seed).script(forward)
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
    """The whole build on random scripts: the GPU lensmap (inverse map for inverse scripts, the forward scatter for
    forward scripts - whose garbage projections push draw_quad through NaNs, huge coordinates and degenerate quads)
    against the oracle's fisheye.c restatement with its callbacks evaluated by the host interpreter on the same
    portable libm: offsets, tints, display flags, scale and the built / not-built verdict must be identical."""
    import blinky_amd
    import oracle_ffi as O
    forward = seed % 3 == 2
    src = Gen(5000 + seed).script(forward)
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
    try:
        display, scale = ctx.build()
        built = True
    except blinky_amd.ffi.BlinkyError:
        built = False
    assert built == lm.built, src
    if built:
        off, tin = ctx.read_lensmap()
        assert scale == lm.scale or (scale != scale and lm.scale != lm.scale), src      # (a NaN scale builds an empty map, as in the reference)
        bad = int((off != lm.offsets).sum())
        assert bad == 0, f"{bad} of {off.size} entries differ\n{src}"
        np.testing.assert_array_equal(tin, lm.tints)
        assert display[: lm.numplates] == lm.display
    ctx.close()
    host.close()
