"""TEST INFRASTRUCTURE: compile the HIP translation unit libblinkyhip generates for a lens + globe as host C++
(g++, -ffp-contract=off like the hiprtc build) and run the inverse build kernel serially.  Gives CPU tests and
tools/flag_probe.py the device code's lensmap AND the list of pixels it flags for the host fix-up, without a GPU."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "blinky_amd", "csrc")
_cache = {}


def compile_source(src, defines=()):
    """generated source text -> loaded shared object (defines: extra -D options, e.g. a wider BK_LIBM_REL)"""
    key = hashlib.sha1((" ".join(defines) + src + open(os.path.join(CSRC, "bk_device_rt.h")).read() +
                        open(os.path.join(CSRC, "bk_build_kernels.h")).read() +
                        open(os.path.join(HERE, "emu_driver.inc")).read() + open(os.path.join(HERE, "emu_prelude.h")).read()).encode()).hexdigest()[:16]
    if key in _cache:
        return _cache[key]
    d = os.path.join(tempfile.gettempdir(), "bk_hostemu")
    os.makedirs(d, exist_ok=True)
    cpp, so = os.path.join(d, key + ".cpp"), os.path.join(d, key + ".so")
    if not os.path.exists(so):
        # per-process scratch names + atomic rename: pytest-xdist workers may want the same key at once
        cpp = os.path.join(d, "%s.%d.cpp" % (key, os.getpid()))
        tmp = "%s.%d.tmp" % (so, os.getpid())
        with open(cpp, "w") as f:
            f.write('#include "emu_prelude.h"\n' + src + "\n" + open(os.path.join(HERE, "emu_driver.inc")).read())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w",
                               *["-D" + d for d in defines], "-I", HERE, "-I", CSRC, "-o", tmp, cpp])
        os.replace(tmp, so)
        os.unlink(cpp)
    lib = C.CDLL(so)
    _cache[key] = lib
    return lib


def _field_offsets():
    """byte offsets of the pointer / count fields of BkBuildParams (bk_build_params.h), found by compiling a probe"""
    if "off" in _cache:
        return _cache["off"]
    d = os.path.join(tempfile.gettempdir(), "bk_hostemu")
    os.makedirs(d, exist_ok=True)
    src = os.path.join(d, "offs.%d.c" % os.getpid())
    with open(src, "w") as f:
        f.write('#include <stddef.h>\n#include <stdio.h>\n#include "bk_build_params.h"\nint main(){'
                'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(BkBuildParams, offsets), offsetof(BkBuildParams, tints),'
                'offsetof(BkBuildParams, display), offsetof(BkBuildParams, err), offsetof(BkBuildParams, flag_list),'
                'offsetof(BkBuildParams, flag_count), offsetof(BkBuildParams, flag_cap), sizeof(BkBuildParams), offsetof(BkBuildParams, corner_xy), offsetof(BkBuildParams, corner_ok));return 0;}')
    exe = os.path.join(d, "offs.%d" % os.getpid())
    subprocess.check_call(["gcc", "-I", CSRC, "-o", exe, src])
    vals = [int(v) for v in subprocess.check_output([exe]).split()]
    _cache["off"] = dict(zip(["offsets", "tints", "display", "err", "flag_list", "flag_count", "flag_cap", "size", "corner_xy", "corner_ok"], vals))
    return _cache["off"]


def build_inverse(ctx, defines=()):
    """ctx: a configured blinky_amd Context (BK_DEVICE_NONE is enough; bk_resize done).  Runs the generated
    bk_build_inverse on the host.  Returns (offsets in DEVICE layout uint32 [rows*W], tints, flagged ids, err bits)."""
    lib = compile_source(ctx.kernel_source(), defines)
    fo = _field_offsets()
    bp = ctx.build_params()
    assert len(bp) == fo["size"] == lib.emu_sizeof_params()
    W, H, ps, r0, r1 = ctx.size()
    n = W * (r1 - r0)
    off = np.full(n, 0xFFFFFFFF, np.uint32)
    tin = np.full(n, 255, np.uint8)
    misc = np.zeros(8, np.int32)                 # display[6], err, flag_count
    cap = n
    flags = np.zeros((cap, 4), np.uint32)

    def put(name, addr):
        C.memmove(C.addressof(bp) + fo[name], C.byref(C.c_uint64(addr)), 8)
    put("offsets", off.ctypes.data)
    put("tints", tin.ctypes.data)
    put("display", misc.ctypes.data)
    put("err", misc.ctypes.data + 24)
    put("flag_count", misc.ctypes.data + 28)
    put("flag_list", flags.ctypes.data)
    C.memmove(C.addressof(bp) + fo["flag_cap"], C.byref(C.c_uint32(cap)), 4)
    first_bad = np.zeros(2, np.uint32)           # (first_bad is BkBuildParams' last member: a script that returns a malformed result writes here)
    C.memmove(C.addressof(bp) + fo["size"] - 8, C.byref(C.c_uint64(first_bad.ctypes.data)), 8)
    lib.emu_build_inverse(bp)
    nf = int(misc[7])
    return off, tin, flags[:nf, 0].copy(), int(misc[6])


def device_to_reference_layout(off, ps):
    """device (16x8-tile) offsets -> plate*ps*ps + py*ps + px (bk_texel_coords, bk_build_params.h)"""
    gp, ph = (ps + 63) & ~63, (ps + 7) & ~7
    out = np.full(off.shape, 0xFFFFFFFF, np.uint32)
    m = off != 0xFFFFFFFF
    o = off[m].astype(np.uint64)
    p = o // (gp * ph)
    rem = o - p * (gp * ph)
    tile = rem >> 7
    tpr = gp >> 4
    ty, tx = tile // tpr, tile % tpr
    px = tx * 16 + (rem & 15)
    py = ty * 8 + ((rem >> 4) & 7)
    out[m] = (p * ps * ps + py * ps + px).astype(np.uint32)
    return out


def forward_corners(ctx, defines=()):
    """the generated bk_forward_corners on the host: (corner_xy int32 [n,2], corner_ok uint8 [n], flagged ids, err)"""
    lib = compile_source(ctx.kernel_source(), defines)
    fo = _field_offsets()
    bp = ctx.build_params()
    W, H, ps, r0, r1 = ctx.size()
    nplates = len(ctx.globe())
    n = nplates * (ps + 1) * (ps + 1)
    xy = np.zeros((n, 2), np.int32)
    ok = np.zeros(n, np.uint8)
    misc = np.zeros(8, np.int32)
    flags = np.zeros((n, 4), np.uint32)

    def put(name, addr):
        C.memmove(C.addressof(bp) + fo[name], C.byref(C.c_uint64(addr)), 8)
    put("corner_xy", xy.ctypes.data)
    put("corner_ok", ok.ctypes.data)
    put("display", misc.ctypes.data)
    put("err", misc.ctypes.data + 24)
    put("flag_count", misc.ctypes.data + 28)
    put("flag_list", flags.ctypes.data)
    C.memmove(C.addressof(bp) + fo["flag_cap"], C.byref(C.c_uint32(n)), 4)
    lib.emu_forward_corners(bp)
    return xy, ok, flags[: int(misc[7]), 0].copy(), int(misc[6])


def inverse_values(ctx, defines=()):
    """the generated lens_inverse alone over every pixel: dict(nret [n], val [n,8], bound [n,8], tag [n,8], flag [n], err [n],
    x [n], y [n]) - val +- bound is the device code's claim about what ANY libm within BK_LIBM_REL of bkm.h returns
    (valid where flag == 0: no decision inside the script was uncertain)"""
    lib = compile_source(ctx.kernel_source(), defines)
    bp = ctx.build_params()
    W, H, ps, r0, r1 = ctx.size()
    n = W * (r1 - r0)
    out = dict(nret=np.zeros(n, np.int32), val=np.zeros((n, 8)), bound=np.zeros((n, 8)), tag=np.zeros((n, 8), np.int32),
               flag=np.zeros(n, np.int32), err=np.zeros(n, np.int32))
    lib.emu_inverse_values(bp, *[C.c_void_p(out[k].ctypes.data) for k in ("nret", "val", "bound", "tag", "flag", "err")])
    scale = ctx.calc_zoom()
    ly, lx = np.divmod(np.arange(n), W)
    out["x"] = (lx - W // 2).astype(np.float64) * scale
    out["y"] = (-(ly + r0 - H // 2)).astype(np.float64) * scale
    return out


def forward_values(ctx, rays, defines=()):
    """the generated lens_forward alone on the given rays [n,3] (float32 values, as the build hands them over): the same
    dict as inverse_values without x / y"""
    lib = compile_source(ctx.kernel_source(), defines)
    bp = ctx.build_params()
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    n = len(rays)
    out = dict(nret=np.zeros(n, np.int32), val=np.zeros((n, 8)), bound=np.zeros((n, 8)), tag=np.zeros((n, 8), np.int32),
               flag=np.zeros(n, np.int32), err=np.zeros(n, np.int32))
    lib.emu_forward_values(bp, n, C.c_void_p(rays.ctypes.data),
                           *[C.c_void_p(out[k].ctypes.data) for k in ("nret", "val", "bound", "tag", "flag", "err")])
    return out


def draw_quad(ctx, corners, defines=()):
    """the generated code's bk_draw_quad alone on an empty W x H table (ctx: a configured FORWARD-lens context): uint8 mask [H, W]"""
    lib = compile_source(ctx.kernel_source(), defines)
    bp = ctx.build_params()
    W, H, ps, r0, r1 = ctx.size()
    keys = np.zeros(W * H, np.uint32)
    c = np.ascontiguousarray(corners, np.int32)
    lib.emu_draw_quad(bp, C.c_void_p(c.ctypes.data), C.c_void_p(keys.ctypes.data))
    return (keys != 0).astype(np.uint8).reshape(H, W)
