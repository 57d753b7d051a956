"""blinky_amd -- MI355X-native implementation of Blinky's globe->screen warp.

The product is the C-ABI library ``libblinkyhip.so`` (include/blinky_hip.h) plus the C
host layer in ``blinky_amd/host``.  This Python package is only the ctypes binding that
tests/ and bench.py use; it never computes anything itself and has no CPU fallback:
importing :mod:`blinky_amd.ffi` raises if the HIP library has not been built.
"""
from .ffi import Context, Comm, Multi, BlinkyError, lib, LIB_PATH, debug_set_option  # noqa: F401
