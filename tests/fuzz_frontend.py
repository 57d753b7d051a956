"""TEST INFRASTRUCTURE: token-level mutants of the bundled lens scripts through the Lua front-end on a BK_DEVICE_NONE context - parser,
chunk execution, callback evaluation, calc_zoom, the HIP code generator, the carries-state walk.  A broken script has to come back as an
error (the engine prints it and keeps running, fisheye.c:1670-1680), never as a crash: run in a process of its own by
tests/test_frontend.py; `python tests/fuzz_frontend.py LO HI` runs a longer campaign."""
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

OPS = ["+", "-", "*", "/", "^", "%", "..", "==", "~=", "<", "<=", ">", ">=", "and", "or"]
KEYWORDS = {"function", "end", "local", "return", "if", "then", "else", "elseif", "for", "do", "while", "repeat", "until", "in", "not",
            "and", "or", "nil", "true", "false", "break"}
SPLICE = ["(", ")", "{", "}", "end", "function", "local", "return", ",", "=", "...", ":", ".", "'", '"', "[[", "--[[", "#", "::", "goto", "\0"]


def mutant(seed, scripts):
    rng = random.Random(seed)
    src = scripts.script("lenses", rng.choice(scripts.LENSES))
    toks = re.findall(r"--[^\n]*|[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+|==|~=|<=|>=|\.\.\.?|\s+|.", src, re.S)
    idx = [i for i, t in enumerate(toks) if not t.isspace() and not t.startswith("--")]
    names = [t for t in toks if re.match(r"[A-Za-z_]\w*$", t)]
    for _ in range(rng.randint(1, 3)):
        i = rng.choice(idx)
        t = toks[i]
        if re.match(r"\d|\.\d", t):
            toks[i] = rng.choice(["0", "1", "-1", "0.5", "1e308", "1e-320", "nil", "(0/0)", "math.huge", "'s'", "{}", "true"])
        elif t in OPS:
            toks[i] = rng.choice(OPS)
        elif re.match(r"[A-Za-z_]", t) and t not in KEYWORDS:
            toks[i] = rng.choice(names + ["nil", "x", "y", "lens_inverse", "print", "math", "pi"])
        elif rng.random() < 0.3:
            toks[i] = ""
        elif rng.random() < 0.2:
            toks[i] = t + " " + rng.choice(SPLICE) + " "
    return "".join(toks)


def run(lo, hi):
    import blinky_amd as bk
    import scripts
    loaded = rejected = 0
    for seed in range(lo, hi):
        ctx = bk.Context(bk.ffi.DEVICE_NONE)
        ctx.load_globe(scripts.script("globes", "cube"), "cube")
        try:
            ctx.load_lens(mutant(seed, scripts), "fuzz.lua")
        except bk.BlinkyError:
            rejected += 1
            continue
        loaded += 1
        ctx.set_zoom(3, 0)
        ctx.resize(64, 48)
        for call in ([ctx.eval_host, 0, 0.1, 0.2], [ctx.eval_host, 0, 0.0, 0.0], [ctx.eval_host, 1, 0.1, 0.2, 0.9], [ctx.calc_zoom],
                     [ctx.kernel_source, False], [ctx.lens_carries_state]):
            try:
                call[0](*call[1:])
            except bk.BlinkyError:
                pass
    print("fuzz_frontend seeds %d:%d loaded %d rejected %d" % (lo, hi, loaded, rejected))


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]))
