"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when present, to
oracle/_ref/libref.so (the unmodified reference).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
ORACLE_BKM_SO = os.path.join(ROOT, "oracle", "liboracle_bkm.so")   # same oracle on the portable libm (bkm.h)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
NULL = 0xFFFFFFFF

if not os.path.exists(ORACLE_SO) or not os.path.exists(ORACLE_BKM_SO):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so", "liboracle_bkm.so"])
_o = C.CDLL(ORACLE_SO)
_ob = C.CDLL(ORACLE_BKM_SO)
_ob.okpy_lensmap.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double,
                             C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double),
                             C.POINTER(C.c_int), C.POINTER(C.c_int)]
_o.ok_fnv1a64.restype = C.c_uint64
_o.ok_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
_o.ok_lcg_fill_plate.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
_o.okpy_lensmap.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double,
                            C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double),
                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
_o.okpy_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_void_p]
_o.okpy_palmap.argtypes = [C.c_void_p, C.c_void_p]
_o.okpy_globe.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % _o.ok_fnv1a64(_p(a), a.nbytes)


class Lensmap:
    def __init__(self, W, H, offsets, tints, display, scale, numplates, map_type, built):
        self.W, self.H = W, H
        self.ps = min(W, H)
        self.offsets, self.tints = offsets, tints
        self.display, self.scale, self.numplates, self.map_type, self.built = display, scale, numplates, map_type, built

    @property
    def nonnull(self):
        return int((self.offsets != NULL).sum())


def lensmap(globe, lens, zoom, W, H, grid=(10, 4.0, 1.0), portable=False):
    """Oracle lensmap for 'f_globe G; f_lens L; zoom' at WxH (zoom None = the lens' onload).
    portable=True: the liboracle_bkm.so build (every libm call = the GPU kernels' bkm.h function)."""
    off = np.empty(W * H, np.uint32)
    tin = np.empty(W * H, np.uint8)
    disp = (C.c_int * 6)()
    scale, npl, mt = C.c_double(), C.c_int(), C.c_int()
    rc = (_ob if portable else _o).okpy_lensmap(globe.encode(), lens.encode(), zoom.encode() if zoom else None, W, H,
                         int(grid[0]), float(grid[1]), float(grid[2]), _p(off), _p(tin), disp,
                         C.byref(scale), C.byref(npl), C.byref(mt))
    if rc < 0:
        raise KeyError(f"oracle has no transliteration of {globe}/{lens}")
    return Lensmap(W, H, off, tin, list(disp)[: npl.value], scale.value, npl.value, mt.value, rc == 1)


def globe_plates(name):
    buf = np.zeros((6, 13), np.float32)
    n = C.c_int()
    if not _o.okpy_globe(name.encode(), _p(buf), C.byref(n)):
        raise KeyError(name)
    return buf[: n.value]


def lcg_globe(ps, nplates=6, frame=0):
    """SURVEY.md 8(d) synthetic globe: uint8 [6][ps][ps], plates >= nplates left zero."""
    g = np.zeros((6, ps, ps), np.uint8)
    for p in range(nplates):
        _o.ok_lcg_fill_plate(_p(g[p]), ps * ps, p, frame)
    return g


def apply(offsets, tints, W, rows, globe, dst, pitch=None, x0=0, y0=0, rubix_on=False, pal=None):
    pitch = dst.shape[-1] if pitch is None else pitch
    pal = None if pal is None else np.ascontiguousarray(pal, np.uint8)
    _o.okpy_apply(_p(np.ascontiguousarray(offsets, np.uint32)),
                  _p(np.ascontiguousarray(tints, np.uint8)) if tints is not None else None, W, rows,
                  _p(np.ascontiguousarray(globe)), _p(dst), pitch, x0, y0, int(rubix_on), _p(pal))
    return dst


def time_apply_mt(offsets, tints, W, rows, globe, reps, nthreads):
    """best wall seconds of the row-parallel (pthreads) restatement of render_lensmap over `reps` calls"""
    dst = np.zeros((rows, W), np.uint8)
    _o.okpy_time_apply_mt.restype = C.c_double
    return _o.okpy_time_apply_mt(_p(np.ascontiguousarray(offsets, np.uint32)), _p(np.ascontiguousarray(tints, np.uint8)), W, rows,
                                 _p(np.ascontiguousarray(globe)), _p(dst), W, reps, nthreads), dst


def palmap(basepal):
    out = np.empty((6, 256), np.uint8)
    _o.okpy_palmap(_p(np.ascontiguousarray(basepal, np.uint8)), _p(out))
    return out


def synthetic_basepal():
    """pal[i] = (i*37) mod 256, i < 768"""
    return ((np.arange(768) * 37) % 256).astype(np.uint8)


# ---- oracle/_ref: the unmodified reference -------------------------------------------
def have_ref():
    return os.path.exists(REF_SO)


_r = None


def ref_run(globe, lens, zoom, W, H, frame_index=0, rubix_on=False, grid=None, want_frame=True):
    """Run the UNMODIFIED engine/NQ/fisheye.c (oracle/_ref).  Returns (Lensmap, frame)."""
    global _r
    if _r is None:
        _r = C.CDLL(REF_SO)
    off = np.empty(W * H, np.uint32)
    tin = np.empty(W * H, np.uint8)
    disp = (C.c_int * 6)()
    scale, npl = C.c_double(), C.c_int()
    frame = np.zeros((H, W), np.uint8) if want_frame else None
    ok = _r.ref_run(globe.encode(), lens.encode(), zoom.encode() if zoom else None, W, H, _p(off), _p(tin),
                    disp, C.byref(scale), C.byref(npl), _p(frame), frame_index, int(rubix_on),
                    grid.encode() if grid else None, 0)
    lm = Lensmap(W, H, off, tin, list(disp)[: npl.value], scale.value, npl.value, None, ok == 1)
    return lm, frame


def ref_palettes():
    global _r
    if _r is None:
        _r = C.CDLL(REF_SO)
    out = np.empty((6, 256), np.uint8)
    _r.ref_palettes(_p(out))
    return out


def ref_palettes_of(basepal):
    """the unmodified reference's create_palmap on any 768-byte base palette"""
    global _r
    if _r is None:
        _r = C.CDLL(REF_SO)
    out = np.empty((6, 256), np.uint8)
    _r.ref_palettes_of(_p(np.ascontiguousarray(basepal, np.uint8)), _p(out))
    return out


def ref_saveglobe(name, with_margins, frame_index, plate, ps):
    """the reference's own f_saveglobe (cmd_saveglobe + save_globe + WritePCXplate) on the state the last
    ref_run left behind, plates = LCG(frame_index): (file name, file bytes) of plate `plate`"""
    buf = np.empty(ps * ps * 2 + 1000, np.uint8)
    nm = C.create_string_buffer(64)
    n = _r.ref_saveglobe(name.encode(), int(with_margins), frame_index, plate, _p(buf), buf.size, nm)
    return nm.value.decode(), (buf[:n].copy() if n >= 0 else None)


def pcx_plate(globe, ps, plate, with_margins, plate_pixels, basepal):
    """oracle restatement of WritePCXplate: the file bytes"""
    out = np.empty(ps * ps * 2 + 1000, np.uint8)
    n = _o.okpy_pcx_plate(globe.encode(), ps, plate, int(with_margins), _p(np.ascontiguousarray(plate_pixels, np.uint8)),
                          _p(np.ascontiguousarray(basepal, np.uint8)), _p(out))
    assert n > 0
    return out[:n].copy()


_o.okpy_eval.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]


def eval_lens(lens, which, x, y, z=0.0):
    """hand-transliterated callback (platform libm): tuple of results, None for nil"""
    out = np.zeros(3)
    rc = _o.okpy_eval(lens.encode(), which, x, y, z, _p(out))
    if rc < 0:
        raise KeyError(lens)
    if rc == 0:
        return None
    return tuple(out[:3] if which == 0 else out[:2])


def lens_def(lens):
    hi, hf, mf, mv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    w, h = C.c_double(), C.c_double()
    buf = C.create_string_buffer(128)
    _o.okpy_lens_def.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)] * 2 + [C.c_char_p, C.c_int]
    if not _o.okpy_lens_def(lens.encode(), C.byref(hi), C.byref(hf), C.byref(mf), C.byref(mv), C.byref(w), C.byref(h), buf, 128):
        raise KeyError(lens)
    return dict(has_inverse=hi.value, has_forward=hf.value, max_fov=mf.value, max_vfov=mv.value,
                lens_width=w.value, lens_height=h.value, onload=buf.value.decode())


_INV_CB = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double))
_FWD_CB = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double))


def lensmap_with_callbacks(globe, info, eval_inverse, eval_forward, zoom, W, H, portable=False):
    """The oracle's fisheye.c restatement (platform libm; portable=True: the bkm.h build) driven by arbitrary lens callbacks:
    eval_inverse(x, y) -> (rx, ry, rz) | None;  eval_forward(x, y, z) -> (x, y) | None.
    `info` carries the lens globals (map_type, max_fov, max_vfov, lens_width, lens_height)."""
    def inv(x, y, out):
        r = eval_inverse(x, y)
        if r is None:
            return 0
        if len(r) != 3:
            return -1
        out[0], out[1], out[2] = r
        return 1

    def fwd(x, y, z, out):
        r = eval_forward(x, y, z)
        if r is None:
            return 0
        if len(r) != 2:
            return -1
        out[0], out[1] = r
        return 1

    inv_c = _INV_CB(inv) if eval_inverse else _INV_CB()
    fwd_c = _FWD_CB(fwd) if eval_forward else _FWD_CB()
    off = np.empty(W * H, np.uint32)
    tin = np.empty(W * H, np.uint8)
    disp = (C.c_int * 6)()
    scale, npl = C.c_double(), C.c_int()
    lib = _ob if portable else _o
    lib.okpy_lensmap_cb.argtypes = [C.c_char_p, _INV_CB, _FWD_CB, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_double), C.POINTER(C.c_int)]
    rc = lib.okpy_lensmap_cb(globe.encode(), inv_c, fwd_c, info.map_type, info.max_fov, info.max_vfov,
                            info.lens_width, info.lens_height, zoom.encode() if zoom else None, W, H,
                            _p(off), _p(tin), disp, C.byref(scale), C.byref(npl))
    if rc < 0:
        raise KeyError(globe)
    return Lensmap(W, H, off, tin, list(disp)[: npl.value], scale.value, npl.value, info.map_type, rc == 1)
