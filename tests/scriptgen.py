"""Random lens scripts for the differential script fuzz (tests/test_script_fuzz_gpu.py on the device, tests/test_script_fuzz_cpu.py on the
host emulation of the generated code).  `Gen` walks the emitter through operator precedence, short-circuit and/or, script functions with
upvalues, numeric for / while / repeat loops, local array tables, multiple assignment and multiple returns; `WideGen` adds what round 3
taught the code generator: functions defined inside the callback, functions passed as arguments, chunk locals as scratch, records,
matrices, varargs / select, constant tables and objects (methods, one through a metatable) as arguments, the length of a constant table."""
import numpy as np

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


class WideGen(Gen):
    def block(self, vars_, depth, indent):
        out, vars_ = super().block(vars_, depth, indent)
        pad = "  " * indent
        plain = [v for v in vars_ if "[" not in v and "." not in v and "(" not in v]
        for _ in range(int(self.r.integers(1, 3))):
            k = int(self.r.integers(0, 10))
            self.n += 1
            n = self.n
            if k == 0:        # a function defined here, closing over everything in sight, called twice
                out.append(f"{pad}local function f{n}(a, b) local s = a * 0.5 + {self.expr(vars_, 2)} if s > b then return s - b, a end return {self.expr(vars_ + ['a', 'b', 's'], 2)}, b end")
                out.append(f"{pad}local r{n}, s{n} = f{n}({self.expr(vars_, 2)}, {self.pick(plain)})")
                out.append(f"{pad}local u{n} = f{n}(r{n}, s{n})")
                vars_ += [f"r{n}", f"s{n}", f"u{n}"]
            elif k == 1:      # local f = function, writing an enclosing local and a table of the enclosing function
                out.append(f"{pad}local m{n} = {{{self.expr(vars_, 1)}, {self.expr(vars_, 1)}, 0}}")
                out.append(f"{pad}local k{n} = 0")
                out.append(f"{pad}local g{n} = function(i, v) m{n}[i] = v * 0.5 + m{n}[(i % #m{n}) + 1] k{n} = k{n} + 1 return m{n}[i] end")
                out.append(f"{pad}local w{n} = g{n}(1, {self.expr(vars_, 2)}) + g{n}(3, {self.pick(plain)}) + k{n}")
                vars_ += [f"w{n}", f"m{n}[2]", f"k{n}"]
            elif k == 2:      # a function inside a function inside the callback
                out.append(f"{pad}local function o{n}(a)")
                out.append(f"{pad}  local function inner(b) return (a + b) * 0.5 + {self.pick(plain)} end")
                out.append(f"{pad}  local acc = 0 for i = 1, 3 do acc = acc + inner(i * a) end return acc")
                out.append(f"{pad}end")
                out.append(f"{pad}local z{n} = o{n}({self.expr(vars_, 2)})")
                vars_.append(f"z{n}")
            elif k == 3:      # functions as arguments: a script function, a builtin, passed on once more
                fn = self.pick(["helper2", "math.sin", "math.cos", "wave", "math.abs", "lib.tri"])
                out.append(f"{pad}local h{n} = {self.pick(['apply1', 'twice'])}({fn if fn != 'helper2' else 'wave'}, {self.expr(vars_, 2)})")
                out.append(f"{pad}local e{n} = fold(helper, {self.expr(vars_, 1)}, {self.pick(plain)})")
                vars_ += [f"h{n}", f"e{n}"]
            elif k == 4:      # a local of the chunk as scratch, a local name for a builtin
                out.append(f"{pad}scratch = {self.expr(vars_, 2)}")
                out.append(f"{pad}local sn{n} = math.sin")
                out.append(f"{pad}local c{n} = sn{n}(scratch) + scratch * 0.25")
                vars_.append(f"c{n}")
            elif k == 6:      # a record: fields read, written, swapped; an unnamed field is nil
                out.append(f"{pad}local rec{n} = {{a = {self.expr(vars_, 2)}, b = {self.expr(vars_, 1)}, c = 0}}")
                out.append(f"{pad}rec{n}.c = rec{n}.a * 0.5 + rec{n}.b")
                out.append(f"{pad}rec{n}.a, rec{n}.b = rec{n}.b, rec{n}.a")
                out.append(f"{pad}if rec{n}.missing ~= nil then rec{n}.c = 0 end")
                vars_ += [f"rec{n}.a", f"rec{n}.b", f"rec{n}.c"]
            elif k == 7:      # a matrix: constant and computed indices, element stores, both lengths, reached from a function defined here
                out.append(f"{pad}local mat{n} = {{{{{self.expr(vars_, 1)}, ({self.expr(vars_, 1)})}}, {{{self.expr(vars_, 1)}, 1}}, {{0.5, {self.pick(plain)}}}}}")
                out.append(f"{pad}local function cell{n}(i, j) return mat{n}[i][j] end")
                out.append(f"{pad}for i = 1, #mat{n} do for j = 1, #mat{n}[i] do mat{n}[i][j] = mat{n}[i][j] * 0.5 + cell{n}((i % #mat{n}) + 1, j) * 0.25 end end")
                vars_ += [f"mat{n}[1][2]", f"mat{n}[3][1]", f"cell{n}(2, 2)"]
            elif k == 8:      # vararg helpers: counted, indexed from both ends, passed on, spread over locals
                out.append(f"{pad}local va{n}, vb{n} = spread({self.expr(vars_, 1)}, {self.pick(plain)}, {self.expr(vars_, 1)})")
                out.append(f"{pad}local vc{n} = total({self.pick(plain)}, va{n}, ({self.expr(vars_, 2)})) + (select(-1, vb{n}, {self.pick(plain)}))")
                vars_ += [f"va{n}", f"vb{n}", f"vc{n}"]
            elif k == 9:      # a constant object: methods (one through a metatable), the object and a plain table as arguments
                out.append(f"{pad}local ob{n} = gadget:bend({self.expr(vars_, 2)}) + gadget:base() + lookup(knots, {self.pick(plain)})")
                out.append(f"{pad}local kn{n} = knots")
                out.append(f"{pad}local oc{n} = using(gadget, {self.pick(plain)}) + kn{n}[2]")
                vars_ += [f"ob{n}", f"oc{n}"]
            else:             # a constant table of the chunk: indexed, its length
                out.append(f"{pad}local q{n} = math.abs({self.pick(plain)}) if not (q{n} < 100) then q{n} = 1 end       -- (a NaN or huge index would be a nil element)")
                out.append(f"{pad}local d{n} = knots[(math.floor(q{n} * 3) % #knots) + 1] + #knots")
                vars_.append(f"d{n}")
        return out, vars_

    def script(self, forward):
        text = super().script(forward)
        head = "\n".join([
            "local scratch = 0.125",
            "local knots = {0.1, 0.35, 0.7, 1.3}",
            "local function wave(t) return math.sin(t * 1.5) * 0.5 + t * 0.25 end",
            "local lib = {tri = function(t) return math.abs(t - math.floor(t + 0.5)) end}",
            "local function apply1(f, a) return f(a) + 0.5 end",
            "local function twice(f, a) return apply1(f, apply1(f, a)) end",
            "local Gadget = {offset = 0.375}",
            "Gadget.__index = Gadget",
            "function Gadget:base() return self.offset + self.gain end",
            "local gadget = setmetatable({gain = 1.25, taps = {0.5, 0.25}}, Gadget)",
            "function gadget:bend(v) return v * self.gain + self.taps[2] * math.sin(v) + self:base() end",
            "local function lookup(t, v) if v > 0 then return t[1] + #t end return t[#t] end",
            "local function using(g, v) return g:bend(v) * 0.5 + g.taps[1] end",
            "local function total(...) local s = 0 for i = 1, select('#', ...) do s = s * 0.5 + (select(i, ...)) end return s end",
            "local function spread(first, ...) local n = select('#', ...) return first + n, total(...) end",
        ])
        tail = "local function fold(f, a, b) local s = a for i = 1, 3 do s = f(s, b) * 0.5 + s * 0.25 end return s end"
        # (helper and pair are defined by the base script; fold needs helper, so it goes after them)
        text = text.replace("local function pair(a, b)", tail + "\nlocal function pair(a, b)", 1)
        return head + "\n" + text
