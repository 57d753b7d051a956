"""ctypes binding of include/blinky_hip.h and include/blinky_hip_debug.h (1:1, no logic of its own)."""
import ctypes as C
import os

import numpy as np

try:
    # torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first in a process
    # is the only one that sees the GPU, so when torch is installed it must be loaded before
    # libblinkyhip.so pulls in the system runtime (bench.py / tests share device memory with torch).
    import torch  # noqa: F401
except ImportError:  # the C host layer (blinky_amd/host) does not need torch at all
    torch = None

LIB_PATH = os.environ.get("BLINKY_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libblinkyhip.so")   # (BLINKY_HIP_LIB: developer A/B builds)
if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C blinky_amd/csrc`).  There is no CPU fallback."
    )
lib = C.CDLL(LIB_PATH)

MAX_PLATES = 6
NULL_OFFSET = 0xFFFFFFFF
DEVICE_NONE = -2
OK = 0
PENDING = 1
MAP_NONE, MAP_INVERSE, MAP_FORWARD = 0, 1, 2
ZOOM_NONE, ZOOM_FOV, ZOOM_VFOV, ZOOM_COVER, ZOOM_CONTAIN = range(5)


class Plate(C.Structure):
    _fields_ = [("forward", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float), ("dist", C.c_float)]


class LensInfo(C.Structure):
    _fields_ = [("map_type", C.c_int), ("has_inverse", C.c_int), ("has_forward", C.c_int),
                ("max_fov", C.c_int), ("max_vfov", C.c_int),
                ("lens_width", C.c_double), ("lens_height", C.c_double), ("onload", C.c_char * 128)]


_vp, _i, _sz, _d = C.c_void_p, C.c_int, C.c_size_t, C.c_double
_SIGS = {
    "bk_create": (_vp, [_i]),
    "bk_destroy": (None, [_vp]),
    "bk_last_error": (C.c_char_p, [_vp]),
    "bk_set_stream": (_i, [_vp, _vp]),
    "bk_synchronize": (_i, [_vp]),
    "bk_load_globe": (_i, [_vp, C.c_char_p, _sz, C.c_char_p]),
    "bk_load_lens": (_i, [_vp, C.c_char_p, _sz, C.c_char_p]),
    "bk_clear_lens": (_i, [_vp]),
    "bk_clear_globe": (_i, [_vp]),
    "bk_get_lens_info": (_i, [_vp, C.POINTER(LensInfo)]),
    "bk_get_globe": (_i, [_vp, C.POINTER(Plate), C.POINTER(_i)]),
    "bk_set_globe_plates": (_i, [_vp, C.POINTER(Plate), _i]),
    "bk_resize": (_i, [_vp, _i, _i]),
    "bk_set_rows": (_i, [_vp, _i, _i]),
    "bk_set_frames": (_i, [_vp, _i]),
    "bk_set_zoom": (_i, [_vp, _i, _i]),
    "bk_set_rubixgrid": (_i, [_vp, _i, _d, _d]),
    "bk_build": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
    "bk_calc_zoom": (_i, [_vp, C.POINTER(_d)]),
    "bk_last_build_fixups": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "bk_last_build_bad_key": (C.c_uint, [_vp]),
    "bk_set_sequential_build": (_i, [_vp, _i]),
    "bk_last_build_path": (_i, [_vp, C.c_char_p, _sz]),
    "bk_lens_carries_state": (_i, [_vp, C.c_char_p, _sz]),
    "bk_truncate_build": (_i, [_vp, C.c_uint, C.POINTER(_i)]),
    "bk_set_cache_dir": (_i, [C.c_char_p]),
    "bk_set_async_compile": (_i, [_vp, _i]),
    "bk_set_lensmap": (_i, [_vp, _vp, _vp]),
    "bk_read_lensmap": (_i, [_vp, _vp, _vp]),
    "bk_upload_plate": (_i, [_vp, _i, _i, _vp, _i]),
    "bk_upload_plate_async": (_i, [_vp, _i, _i, _vp, _i]),
    "bk_globe_device_ptr": (_vp, [_vp, _i]),
    "bk_fill_plate_lcg": (_i, [_vp, _i, _i, C.c_uint32]),
    "bk_apply": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "bk_apply_device": (_i, [_vp, _i, _i, _vp, _i, _sz, _i, _i, _i, _vp]),
    "bk_apply_begin": (_i, [_vp, _i, _i, _vp]),
    "bk_apply_resident_begin": (_i, [_vp, _i, _vp, _d]),
    "bk_apply_resident_submit": (_i, [_vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint64)]),
    "bk_apply_resident_submit_batch": (_i, [_vp, _i, _i, _vp, _i, _sz, _i, _i, C.POINTER(C.c_uint64)]),
    "bk_apply_resident_wait": (_i, [_vp, C.c_uint64, C.POINTER(_d)]),
    "bk_apply_resident_end": (_i, [_vp]),
    "bk_apply_resident_info": (_i, [_vp, C.POINTER(_i)]),
    "bk_apply_end": (_i, [_vp, _vp, _i, _i, _i]),
    "bk_create_palmap": (None, [_vp, _vp]),
    "bk_get_size": (_i, [_vp] + [C.POINTER(_i)] * 5),
    "bk_version": (C.c_char_p, []),
    "bk_set_apply_variant": (_i, [_vp, _i]),
    "bk_set_resident_share": (_i, [_vp, _i, _i, _i]),
    "bk_set_resident_apply": (_i, [_vp, _i]),
    "bk_multi_set_resident_apply": (_i, [_vp, _i]),
    "bk_set_blockmap_tuning": (_i, [_vp, _i]),
    "bk_set_host_compile": (_i, [_i]),
    "bk_host_module_ready": (_i, [_vp, _i]),
    "bk_debug_module_from_cache": (_i, [_vp]),
    "bk_debug_forward_tiles": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "bk_debug_set_option": (_i, [C.c_char_p, _i]),
    "bk_debug_build_breakdown": (_i, [_vp, C.POINTER(_d)]),
    "bk_multi_lensmap_valid": (_i, [_vp]),
    "bk_last_build_ms": (_d, [_vp]),
    "bk_globe_pitch": (_i, [_vp]),
    "bk_globe_rows": (_i, [_vp]),
    "bk_globe_texel_offset": (C.c_uint32, [_vp, _i, _i, _i]),
    "bk_download_plate": (_i, [_vp, _i, _i, _vp, _i]),
    "bk_save_plate": (_i, [_vp, _i, _i, _i, _vp, _i]),
    "bk_debug_tile_stats": (_i, [_vp, C.POINTER(_i)]),
    "bk_debug_traffic_model": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "bk_debug_band_balance": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "bk_debug_stream_mix": (_i, [_vp, _sz, _i, _i, C.POINTER(_d)]),
    "bk_debug_fnv1a64": (_i, [_vp, _sz, C.POINTER(C.c_uint64)]),
    "bk_debug_host_build": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i), C.POINTER(_d)]),
    "bk_debug_resident_latency": (_i, [_vp, _i, _vp, _i, _i, C.POINTER(_d), C.POINTER(_d)]),
    "bk_debug_build_params": (_i, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "bk_debug_host_entries": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "bk_debug_host_corners": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "bk_debug_xcd_of_workgroups": (_i, [_vp, C.POINTER(_i), _i]),
    "bk_debug_set_ablation": (_i, [_vp, _i]),
    "bk_debug_set_tile_shape": (_i, [_vp, _i]),
    "bk_debug_kernel_source": (_i, [_vp, C.c_char_p, _sz, C.POINTER(_sz), _i]),
    "bk_debug_eval": (_i, [_vp, _i, C.POINTER(_d), _i, C.POINTER(_d), C.POINTER(_i)]),
    "bk_script_console": (C.c_char_p, [_vp]),
    "bk_set_host_math": (_i, [_vp, _i]),
    "bk_debug_eval_device": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "bk_dev_alloc": (_vp, [_vp, _sz]),
    "bk_dev_free": (None, [_vp, _vp]),
    "bk_dev_read": (_i, [_vp, _vp, _vp, _sz]),
    # multi-GPU (bk_comm.cpp)
    "bk_comm_unique_id": (_i, [_vp]),
    "bk_comm_create": (_vp, [_vp, _i, _i, _vp]),
    "bk_comm_destroy": (None, [_vp]),
    "bk_comm_last_error": (C.c_char_p, [_vp]),
    "bk_comm_stripe": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "bk_comm_restripe": (_i, [_vp]),
    "bk_comm_rebalance": (_i, [_vp]),
    "bk_debug_stripe_bounds": (_i, [C.POINTER(C.c_uint32), _i, _i, _i, C.POINTER(_i)]),
    "bk_debug_row_costs": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "bk_multi_rebalance": (_i, [_vp, C.POINTER(_i)]),
    "bk_comm_or_display": (_i, [_vp, C.POINTER(_i)]),
    "bk_comm_gather": (_i, [_vp, _vp, _i, _i, _vp, _sz, _i]),
    "bk_comm_exchange_rotating": (_i, [_vp, _vp, _i, _vp, _sz, _i]),
    "bk_comm_wait": (_i, [_vp, _i]),
    "bk_comm_synchronize": (_i, [_vp]),
    "bk_create_multi": (_vp, [_i, C.POINTER(_i)]),
    "bk_destroy_multi": (None, [_vp]),
    "bk_multi_last_error": (C.c_char_p, [_vp]),
    "bk_multi_size": (_i, [_vp]),
    "bk_multi_ctx": (_vp, [_vp, _i]),
    "bk_multi_comm": (_vp, [_vp, _i]),
    "bk_multi_uses_rccl": (_i, [_vp]),
    "bk_multi_load_globe": (_i, [_vp, C.c_char_p, _sz, C.c_char_p]),
    "bk_multi_load_lens": (_i, [_vp, C.c_char_p, _sz, C.c_char_p]),
    "bk_multi_clear_lens": (_i, [_vp]),
    "bk_multi_clear_globe": (_i, [_vp]),
    "bk_multi_resize": (_i, [_vp, _i, _i]),
    "bk_multi_set_frames": (_i, [_vp, _i]),
    "bk_multi_set_zoom": (_i, [_vp, _i, _i]),
    "bk_multi_set_rubixgrid": (_i, [_vp, _i, _d, _d]),
    "bk_multi_upload_plate": (_i, [_vp, _i, _i, _vp, _i]),
    "bk_multi_fill_plate_lcg": (_i, [_vp, _i, _i, C.c_uint32]),
    "bk_multi_synchronize": (_i, [_vp]),
    "bk_multi_build": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
    "bk_multi_apply": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "bk_multi_apply_stripes": (_i, [_vp, _i, _i, C.POINTER(_vp), _i, _vp]),
    "bk_multi_wait": (_i, [_vp, _i]),
    "bk_multi_gather": (_i, [_vp, C.POINTER(_vp), _i, _i, _vp, _sz, _i]),
    "bk_multi_exchange_rotating": (_i, [_vp, C.POINTER(_vp), _i, C.POINTER(_vp), _sz, _i]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library mismatch
    _fn.restype, _fn.argtypes = _res, _args

EXPORTS = tuple(_SIGS)


class BlinkyError(RuntimeError):
    pass


def debug_set_option(name, value):
    """process-wide developer / test switch (include/blinky_hip_debug.h: bk_debug_set_option)"""
    if lib.bk_debug_set_option(name.encode(), int(value)) != OK:
        raise ValueError(f"bk_debug_set_option: unknown option {name!r}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """Thin object wrapper over a ``bk_ctx*``."""

    def __init__(self, device=-1, _borrowed=None):
        self._owned = _borrowed is None
        self._h = lib.bk_create(device) if _borrowed is None else _borrowed
        if not self._h:
            raise BlinkyError(lib.bk_last_error(None).decode())

    def close(self):
        if self._h and self._owned:
            lib.bk_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise BlinkyError(f"[{rc}] {lib.bk_last_error(self._h).decode()}")

    # lifecycle
    def set_stream(self, stream_handle):
        self._chk(lib.bk_set_stream(self._h, stream_handle))

    def synchronize(self):
        self._chk(lib.bk_synchronize(self._h))

    # scripts
    def load_globe(self, src, name="globe"):
        b = src.encode() if isinstance(src, str) else src
        self._chk(lib.bk_load_globe(self._h, b, len(b), name.encode()))

    def load_lens(self, src, name="lens"):
        b = src.encode() if isinstance(src, str) else src
        self._chk(lib.bk_load_lens(self._h, b, len(b), name.encode()))

    def lens_info(self):
        info = LensInfo()
        self._chk(lib.bk_get_lens_info(self._h, C.byref(info)))
        return info

    def globe(self):
        plates = (Plate * MAX_PLATES)()
        n = _i()
        self._chk(lib.bk_get_globe(self._h, plates, C.byref(n)))
        return list(plates)[: n.value]

    def set_globe_plates(self, plates):
        arr = (Plate * len(plates))(*plates)
        self._chk(lib.bk_set_globe_plates(self._h, arr, len(plates)))

    # geometry
    def resize(self, w, h):
        self._chk(lib.bk_resize(self._h, w, h))

    def set_rows(self, r0, r1):
        self._chk(lib.bk_set_rows(self._h, r0, r1))

    def set_frames(self, n):
        self._chk(lib.bk_set_frames(self._h, n))

    def set_zoom(self, ztype, fov=0):
        self._chk(lib.bk_set_zoom(self._h, ztype, fov))

    def set_rubixgrid(self, numcells, cell, pad):
        self._chk(lib.bk_set_rubixgrid(self._h, numcells, cell, pad))

    def size(self):
        v = [_i() for _ in range(5)]
        self._chk(lib.bk_get_size(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    # build
    def build(self):
        disp = (_i * MAX_PLATES)()
        scale = _d()
        self._chk(lib.bk_build(self._h, disp, C.byref(scale)))
        return list(disp), scale.value

    def set_async_compile(self, on):
        self._chk(lib.bk_set_async_compile(self._h, int(on)))

    def build_nowait(self):
        """bk_build under bk_set_async_compile: None while the lens is still compiling, else (display, scale)"""
        disp = (_i * MAX_PLATES)()
        scale = _d()
        rc = lib.bk_build(self._h, disp, C.byref(scale))
        if rc == PENDING:
            return None
        self._chk(rc)
        return list(disp), scale.value

    def calc_zoom(self):
        scale = _d()
        self._chk(lib.bk_calc_zoom(self._h, C.byref(scale)))
        return scale.value

    def last_build_fixups(self):
        """(entries the last build re-derived on the host, entries that changed)"""
        a, b = _i(), _i()
        self._chk(lib.bk_last_build_fixups(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def build_breakdown(self):
        out = (_d * 6)()
        self._chk(lib.bk_debug_build_breakdown(self._h, out))
        return dict(build_ms=out[0], host_eval_ms=out[1], flagged=int(out[2]), pool_threads=int(out[3]), kernel_wall_ms=out[4],
                    retries=int(out[5]) % 1000, compiled_host_module=out[5] >= 1000)

    def host_build(self, mode=0):
        """the host build paths (bk_debug_host_build; works without a device): (offsets in the reference layout, tints, display, scale, error
        text or None) - a malformed result / run-time error comes back as text with the truncated / empty table in place"""
        W, H, ps, r0, r1 = self.size()
        off = np.empty((r1 - r0) * W, np.uint32)
        tin = np.empty((r1 - r0) * W, np.uint8)
        disp = (_i * MAX_PLATES)()
        scale = _d()
        rc = lib.bk_debug_host_build(self._h, int(mode), _ptr(off), _ptr(tin), disp, C.byref(scale))
        err = None
        if rc != OK:
            err = lib.bk_last_error(self._h).decode(errors="replace")
            if rc != -3:                                     # (BK_E_SCRIPT: the table is in place)
                raise BlinkyError(f"[{rc}] {err}")
        return off, tin, list(disp), scale.value, err

    def last_build_path(self):
        """(path, why) of the last build(): 0 GPU kernels, 1 host worker pool, 2 one sequential host scan"""
        buf = C.create_string_buffer(1024)
        rc = lib.bk_last_build_path(self._h, buf, 1024)
        if rc < 0:
            self._chk(rc)
        return rc, buf.value.decode(errors="replace")

    def set_sequential_build(self, mode):
        self._chk(lib.bk_set_sequential_build(self._h, int(mode)))

    def lens_carries_state(self):
        """(bool, name of the first script global a callback reads before assigning it)"""
        buf = C.create_string_buffer(128)
        rc = lib.bk_lens_carries_state(self._h, buf, 128)
        if rc < 0:
            self._chk(rc)
        return bool(rc), buf.value.decode()

    def last_build_bad_key(self):
        return int(lib.bk_last_build_bad_key(self._h))

    def truncate_build(self, bad_key):
        out = (_i * MAX_PLATES)()
        self._chk(lib.bk_truncate_build(self._h, bad_key, out))
        return list(out)

    def last_build_ms(self):
        return lib.bk_last_build_ms(self._h)

    def set_lensmap(self, offsets, tints=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        if tints is not None:
            tints = np.ascontiguousarray(tints, dtype=np.uint8)
        self._chk(lib.bk_set_lensmap(self._h, _ptr(offsets), _ptr(tints)))

    def read_lensmap(self):
        w, h, ps, r0, r1 = self.size()
        off = np.empty((r1 - r0) * w, np.uint32)
        tin = np.empty((r1 - r0) * w, np.uint8)
        self._chk(lib.bk_read_lensmap(self._h, _ptr(off), _ptr(tin)))
        return off, tin

    # globe
    def upload_plate(self, frame, plate, src, pitch=None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if pitch is None:
            pitch = src.shape[-1]
        self._chk(lib.bk_upload_plate(self._h, frame, plate, _ptr(src), pitch))

    def upload_plate_async(self, frame, plate, src, pitch=None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self._chk(lib.bk_upload_plate_async(self._h, frame, plate, _ptr(src), src.shape[-1] if pitch is None else pitch))

    def globe_device_ptr(self, frame=0):
        return lib.bk_globe_device_ptr(self._h, frame)

    def fill_plate_lcg(self, frame, plate, seed_frame=None):
        self._chk(lib.bk_fill_plate_lcg(self._h, frame, plate, frame if seed_frame is None else seed_frame))

    # apply
    def apply(self, dst, frame=0, pitch=None, x0=0, y0=0, rubix_on=False, pal=None):
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        if pitch is None:
            pitch = dst.shape[-1]
        if pal is not None:
            pal = np.ascontiguousarray(pal, dtype=np.uint8)
        self._chk(lib.bk_apply(self._h, frame, _ptr(dst), pitch, x0, y0, int(rubix_on), _ptr(pal)))
        return dst

    def apply_begin(self, frame=0, rubix_on=False, pal=None):
        if pal is not None:
            pal = np.ascontiguousarray(pal, dtype=np.uint8)
        self._chk(lib.bk_apply_begin(self._h, frame, int(rubix_on), _ptr(pal)))

    def apply_end(self, dst, pitch=None, x0=0, y0=0):
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        self._chk(lib.bk_apply_end(self._h, _ptr(dst), dst.shape[-1] if pitch is None else pitch, x0, y0))
        return dst

    def apply_device(self, dst_ptr, pitch, frame_stride, frame0=0, nframes=1, x0=0, y0=0,
                     rubix_on=False, pal=None):
        # (the palette's pointer is kept between calls: `a.ctypes.data_as` builds a handful of container objects per call, and enough
        #  of those wake CPython's cyclic collector in the middle of a caller's launch train - 1 ms for a young generation, 35-40 ms
        #  for a full one with torch loaded: the "millisecond launch" of the rubix bench line, tools/r6_rubix_outlier.py)
        if pal is None:
            pp = None
        else:
            cached = getattr(self, "_pal_ptr", None)
            if cached is not None and cached[0] is pal:
                pp = cached[2]
            else:
                arr = np.ascontiguousarray(pal, dtype=np.uint8)
                pp = _ptr(arr)
                self._pal_ptr = (pal, arr, pp)
        self._chk(lib.bk_apply_device(self._h, frame0, nframes, dst_ptr, pitch, frame_stride, x0, y0,
                                      int(rubix_on), pp))

    # ---- the resident single-frame apply (bk_apply_resident_*): one kernel stays on the device, frames are commands
    def resident_begin(self, rubix_on=False, pal=None, idle_ms=0.0):
        if pal is not None:
            pal = np.ascontiguousarray(pal, dtype=np.uint8)
        self._chk(lib.bk_apply_resident_begin(self._h, int(rubix_on), _ptr(pal), float(idle_ms)))

    def resident_submit(self, dst_ptr, pitch, frame=0, x0=0, y0=0):
        t = C.c_uint64(0)
        self._chk(lib.bk_apply_resident_submit(self._h, frame, dst_ptr, pitch, x0, y0, C.byref(t)))
        return t.value

    def resident_submit_batch(self, dst_ptr, pitch, frame_stride, frame0=0, nframes=1, x0=0, y0=0):
        t = C.c_uint64(0)
        self._chk(lib.bk_apply_resident_submit_batch(self._h, frame0, nframes, dst_ptr, pitch, frame_stride, x0, y0, C.byref(t)))
        return t.value

    def resident_wait(self, ticket):
        """-> device microseconds from the kernel seeing the command to the frame complete in memory"""
        us = _d(0)
        self._chk(lib.bk_apply_resident_wait(self._h, ticket, C.byref(us)))
        return us.value

    def resident_end(self):
        self._chk(lib.bk_apply_resident_end(self._h))

    def resident_latency(self, dst_ptr, pitch, frames=200, globes=1):
        """(host us, device us): medians of `frames` submit + wait cycles timed by the C host itself (bk_debug_resident_latency)"""
        h, d = _d(), _d()
        self._chk(lib.bk_debug_resident_latency(self._h, frames, dst_ptr, pitch, globes, C.byref(h), C.byref(d)))
        return h.value, d.value

    def resident_info(self):
        out = (_i * 12)()
        self._chk(lib.bk_apply_resident_info(self._h, out))
        return dict(running=bool(out[0]), workgroups=out[1], blocks_in_registers=out[2], chunks_per_thread=out[3], block_h=out[4],
                    per_cu=out[5], launches=out[6], pending=out[7], streamed=(out[8], out[9], out[10]), laggard=out[11] // 10000, laggards=out[11] % 10000)

    def host_module_ready(self, wait=False):
        return bool(lib.bk_host_module_ready(self._h, int(wait)))

    def set_blockmap_tuning(self, measured):
        self._chk(lib.bk_set_blockmap_tuning(self._h, int(measured)))

    def set_ablation(self, bits):
        self._chk(lib.bk_debug_set_ablation(self._h, bits))

    def tile_stats(self):
        out = (_i * 6)()
        self._chk(lib.bk_debug_tile_stats(self._h, out))
        return dict(tiles=out[0], slow=out[1], empty=out[2], lds_bytes_per_wave=out[3], tile_h=out[4], lines=out[5])

    def row_costs(self):
        """what every row of this stripe costs the apply (uint32 [H], other stripes' rows 0): the input of the rebalance"""
        _, H = self.size()[:2]
        out = np.zeros(H, np.uint32)
        self._chk(lib.bk_debug_row_costs(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def build_params(self):
        """raw BkBuildParams bytes (tests/hostemu)"""
        need = _sz()
        self._chk(lib.bk_debug_build_params(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        self._chk(lib.bk_debug_build_params(self._h, buf, need.value, None))
        return buf

    def host_entries(self, ids):
        """the host fix-up's value for the given pixel indices: (offsets in the reference layout, tints)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.empty(len(ids), np.uint32)
        tin = np.empty(len(ids), np.uint8)
        self._chk(lib.bk_debug_host_entries(self._h, _ptr(ids), len(ids), _ptr(off), _ptr(tin)))
        return off, tin

    def host_corners(self, ids):
        """the host fix-up's value for the given texel corners of the forward build: (sx, sy, ok)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        sx = np.empty(len(ids), np.int32)
        sy = np.empty(len(ids), np.int32)
        ok = np.empty(len(ids), np.uint8)
        self._chk(lib.bk_debug_host_corners(self._h, _ptr(ids), len(ids), _ptr(sx), _ptr(sy), _ptr(ok)))
        return sx, sy, ok

    def xcd_of_workgroups(self, n):
        out = (_i * n)()
        self._chk(lib.bk_debug_xcd_of_workgroups(self._h, out, n))
        return list(out)

    def traffic_model(self):
        out = (C.c_uint64 * 8)()
        self._chk(lib.bk_debug_traffic_model(self._h, out))
        return dict(unique_globe_lines=out[0], staged_lines=out[1], staged_chunks=out[2], blockmap_bytes_per_visit=out[3],
                    mapped_pixels=out[4], frames_per_visit=out[5], blocks=out[6], block_height=out[7])

    def stream_mix(self, nbytes, period, writes):
        """GB/s of a plain streaming kernel with `writes` KiB written per `period` KiB read (calibration for bench.py)"""
        out = _d()
        self._chk(lib.bk_debug_stream_mix(self._h, nbytes, period, writes, C.byref(out)))
        return out.value

    def band_balance(self):
        """how the staged apply splits the live blocks over the 8 XCDs: band starts, whether equal-count bands would be uneven, band costs"""
        out = (C.c_uint32 * 18)()
        self._chk(lib.bk_debug_band_balance(self._h, out))
        return dict(starts=list(out[:9]), live_blocks=out[8], equal_count_bands_uneven=bool(out[9]), band_cost=list(out[10:18]))

    def set_tile_shape(self, lw):
        self._chk(lib.bk_debug_set_tile_shape(self._h, lw))

    def globe_pitch(self):
        return lib.bk_globe_pitch(self._h)

    def globe_rows(self):
        return lib.bk_globe_rows(self._h)

    def globe_texel_offset(self, plate, px, py):
        return lib.bk_globe_texel_offset(self._h, plate, px, py)

    def save_plate(self, frame, plate, with_margins):
        _, _, ps, _, _ = self.size()
        out = np.empty((ps, ps), np.uint8)
        self._chk(lib.bk_save_plate(self._h, frame, plate, int(with_margins), _ptr(out), ps))
        return out

    def download_plate(self, frame, plate):
        _, _, ps, _, _ = self.size()
        out = np.empty((ps, ps), np.uint8)
        self._chk(lib.bk_download_plate(self._h, frame, plate, _ptr(out), ps))
        return out

    def kernel_source(self, compile=False):
        need = _sz()
        self._chk(lib.bk_debug_kernel_source(self._h, None, 0, C.byref(need), 0))
        buf = C.create_string_buffer(need.value)
        self._chk(lib.bk_debug_kernel_source(self._h, buf, need.value, None, int(compile)))
        return buf.value.decode()

    def eval_host(self, which, *args):
        """which: 0 lens_inverse(x,y), 1 lens_forward(x,y,z), 2 globe_plate(x,y,z) -> tuple or None (nil)"""
        a = (_d * len(args))(*args)
        out = (_d * 8)()
        n = _i()
        self._chk(lib.bk_debug_eval(self._h, which, a, len(args), out, C.byref(n)))
        return None if n.value < 0 else tuple(out[: n.value])

    def eval_device(self, which, args):
        """args: float64 array [n, nargs] -> (out [n, 8] float64, nout [n] int32)"""
        args = np.ascontiguousarray(args, dtype=np.float64)
        n, nargs = args.shape
        out = np.empty((n, 8), np.float64)
        nout = np.empty(n, np.int32)
        self._chk(lib.bk_debug_eval_device(self._h, which, _ptr(args), nargs, n, _ptr(out), _ptr(nout)))
        return out, nout

    def eval_host_many(self, which, args):
        args = np.ascontiguousarray(args, dtype=np.float64)
        out = np.full((len(args), 8), np.nan)
        nout = np.empty(len(args), np.int32)
        for i, a in enumerate(args):
            r = self.eval_host(which, *a)
            if r is None:
                nout[i] = -1
            else:
                nout[i] = len(r)
                out[i, : len(r)] = r
        return out, nout

    def set_host_math(self, portable):
        self._chk(lib.bk_set_host_math(self._h, int(portable)))

    def console(self):
        return lib.bk_script_console(self._h).decode()

    def forward_tiles(self):
        """(tiles of the last forward build whose texels' ownership was taken on bk_forward_tiles' word, or -1; tiles in all)"""
        a, b = _i(), _i()
        self._chk(lib.bk_debug_forward_tiles(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def module_from_cache(self):
        return bool(lib.bk_debug_module_from_cache(self._h))

    def dev_alloc(self, nbytes):
        p = lib.bk_dev_alloc(self._h, nbytes)
        if not p:
            raise BlinkyError("bk_dev_alloc failed: " + lib.bk_last_error(self._h).decode())
        return p

    def dev_free(self, p):
        lib.bk_dev_free(self._h, p)

    def dev_read(self, p, nbytes):
        out = np.empty(nbytes, np.uint8)
        self._chk(lib.bk_dev_read(self._h, _ptr(out), p, nbytes))
        return out

    def set_apply_variant(self, v):
        self._chk(lib.bk_set_apply_variant(self._h, v))

    def set_resident_share(self, part=0, parts=1, reserve_slots_per_cu=0):
        """how many of a CU's workgroup places the resident kernel may take (bk_set_resident_share)"""
        self._chk(lib.bk_set_resident_share(self._h, part, parts, reserve_slots_per_cu))

    def set_resident_apply(self, on=True):
        """bk_apply / bk_upload_plate* through the resident kernel (bk_set_resident_apply)"""
        self._chk(lib.bk_set_resident_apply(self._h, int(bool(on))))


def comm_unique_id():
    """rank 0: the 128-byte id every rank passes to Comm (ncclGetUniqueId)"""
    buf = (C.c_uint8 * 128)()
    if lib.bk_comm_unique_id(buf) != OK:
        raise BlinkyError(lib.bk_comm_last_error(None).decode())
    return bytes(buf)


class Comm:
    """One rank of the stripe exchange (``bk_comm*``): RCCL send/recv on the context's stream."""

    def __init__(self, ctx, nranks, rank, unique_id=None):
        idbuf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self._h = lib.bk_comm_create(ctx._h, nranks, rank, idbuf)
        if not self._h:
            raise BlinkyError(lib.bk_comm_last_error(None).decode())
        self.nranks, self.rank = nranks, rank

    def close(self):
        if self._h:
            lib.bk_comm_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != OK:
            raise BlinkyError(f"[{rc}] {lib.bk_comm_last_error(self._h).decode()}")

    def stripe(self, rank):
        a, b = _i(), _i()
        self._chk(lib.bk_comm_stripe(self._h, rank, C.byref(a), C.byref(b)))
        return a.value, b.value

    def rebalance(self):
        """collective: stripes of equal work instead of equal height (then build again); returns this rank's new rows"""
        self._chk(lib.bk_comm_rebalance(self._h))
        return self.stripe(self.rank) if hasattr(self, "rank") else None

    def or_display(self, display):
        arr = (_i * MAX_PLATES)(*display)
        self._chk(lib.bk_comm_or_display(self._h, arr))
        return list(arr)

    def gather(self, stripe_ptr, nframes, root, frames_ptr, frame_stride, slot=0):
        self._chk(lib.bk_comm_gather(self._h, stripe_ptr, nframes, root, frames_ptr, frame_stride, slot))

    def exchange_rotating(self, stripe_ptr, nframes, frames_ptr, frame_stride, slot=0):
        self._chk(lib.bk_comm_exchange_rotating(self._h, stripe_ptr, nframes, frames_ptr, frame_stride, slot))

    def wait(self, slot=0):
        self._chk(lib.bk_comm_wait(self._h, slot))

    def synchronize(self):
        self._chk(lib.bk_comm_synchronize(self._h))


class Multi:
    """N stripe contexts + communicators driven by one host thread (``bk_multi*``)."""

    def __init__(self, devices):
        arr = (_i * len(devices))(*devices)
        self._h = lib.bk_create_multi(len(devices), arr)
        if not self._h:
            raise BlinkyError(lib.bk_multi_last_error(None).decode())
        self.n = len(devices)

    def close(self):
        if self._h:
            lib.bk_destroy_multi(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != OK:
            raise BlinkyError(f"[{rc}] {lib.bk_multi_last_error(self._h).decode()}")

    def ctx(self, i):
        return Context(_borrowed=lib.bk_multi_ctx(self._h, i))

    def set_resident_apply(self, on=True):
        self._chk(lib.bk_multi_set_resident_apply(self._h, int(bool(on))))

    def uses_rccl(self):
        return bool(lib.bk_multi_uses_rccl(self._h))

    def load_globe(self, src, name="globe"):
        b = src.encode()
        self._chk(lib.bk_multi_load_globe(self._h, b, len(b), name.encode()))

    def load_lens(self, src, name="lens"):
        b = src.encode()
        self._chk(lib.bk_multi_load_lens(self._h, b, len(b), name.encode()))

    def resize(self, w, h):
        self._chk(lib.bk_multi_resize(self._h, w, h))

    def rebalance(self):
        """stripes of equal work (mapped pixels) instead of equal height; returns the N+1 row bounds.  build() again afterwards."""
        out = (_i * (self.n + 1))()
        self._chk(lib.bk_multi_rebalance(self._h, out))
        return list(out)

    def set_frames(self, n):
        self._chk(lib.bk_multi_set_frames(self._h, n))

    def set_zoom(self, ztype, fov=0):
        self._chk(lib.bk_multi_set_zoom(self._h, ztype, fov))

    def set_rubixgrid(self, numcells, cell, pad):
        self._chk(lib.bk_multi_set_rubixgrid(self._h, numcells, cell, pad))

    def upload_plate(self, frame, plate, src, pitch=None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self._chk(lib.bk_multi_upload_plate(self._h, frame, plate, _ptr(src), src.shape[-1] if pitch is None else pitch))

    def fill_plate_lcg(self, frame, plate, seed_frame=None):
        self._chk(lib.bk_multi_fill_plate_lcg(self._h, frame, plate, frame if seed_frame is None else seed_frame))

    def synchronize(self):
        self._chk(lib.bk_multi_synchronize(self._h))

    def build(self):
        disp = (_i * MAX_PLATES)()
        scale = _d()
        self._chk(lib.bk_multi_build(self._h, disp, C.byref(scale)))
        return list(disp), scale.value

    def apply(self, dst, frame=0, pitch=None, x0=0, y0=0, rubix_on=False, pal=None):
        assert dst.dtype == np.uint8 and dst.flags.c_contiguous
        if pal is not None:
            pal = np.ascontiguousarray(pal, dtype=np.uint8)
        self._chk(lib.bk_multi_apply(self._h, frame, _ptr(dst), dst.shape[-1] if pitch is None else pitch, x0, y0, int(rubix_on), _ptr(pal)))
        return dst

    def apply_stripes(self, stripe_ptrs, frame0=0, nframes=1, rubix_on=False, pal=None):
        arr = (_vp * self.n)(*stripe_ptrs)
        if pal is not None:
            pal = np.ascontiguousarray(pal, dtype=np.uint8)
        self._chk(lib.bk_multi_apply_stripes(self._h, frame0, nframes, arr, int(rubix_on), _ptr(pal)))

    def gather(self, stripe_ptrs, nframes, root, frames_ptr, frame_stride, slot=0):
        arr = (_vp * self.n)(*stripe_ptrs)
        self._chk(lib.bk_multi_gather(self._h, arr, nframes, root, frames_ptr, frame_stride, slot))

    def exchange_rotating(self, stripe_ptrs, nframes, frames_ptrs, frame_stride, slot=0):
        a = (_vp * self.n)(*stripe_ptrs)
        b = (_vp * self.n)(*frames_ptrs)
        self._chk(lib.bk_multi_exchange_rotating(self._h, a, nframes, b, frame_stride, slot))

    def wait(self, slot=0):
        self._chk(lib.bk_multi_wait(self._h, slot))


def stripe_bounds_from_costs(row_cost, W, nranks):
    """the stripe bounds bk_comm_rebalance / bk_multi_rebalance derive from per-row costs (host logic, no device); W = 0: the
    costs are complete (Context.row_costs()), W >= 1: mapped pixels per row, priced like the direct-gather apply's rows"""
    row_cost = np.ascontiguousarray(row_cost, dtype=np.uint32)
    out = (_i * (nranks + 1))()
    rc = lib.bk_debug_stripe_bounds(row_cost.ctypes.data_as(C.POINTER(C.c_uint32)), len(row_cost), W, nranks, out)
    if rc != OK:
        raise BlinkyError(f"[{rc}] bk_debug_stripe_bounds")
    return list(out)


def create_palmap(basepal):
    basepal = np.ascontiguousarray(basepal, dtype=np.uint8)
    out = np.empty((MAX_PLATES, 256), np.uint8)
    lib.bk_create_palmap(_ptr(basepal), _ptr(out))
    return out


def fnv1a64(a):
    """FNV-1a-64 of a host array, as the hex string tests/golden/lensmaps.json records (bk_debug_fnv1a64; ctypes drops the GIL,
    so a thread pool hashes a batch of frames in parallel)"""
    a = np.ascontiguousarray(a)
    out = C.c_uint64()
    rc = lib.bk_debug_fnv1a64(_ptr(a), a.nbytes, C.byref(out))
    if rc != OK:
        raise BlinkyError(f"[{rc}] bk_debug_fnv1a64")
    return "%016x" % out.value
