"""The C-ABI boundary: libblinkyhip.so loads and exports every function include/blinky_hip.h
declares (no compute calls - this runs without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "blinky_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["bk_create", "bk_destroy", "bk_load_lens", "bk_load_globe", "bk_resize", "bk_build",
                 "bk_upload_plate", "bk_apply", "bk_apply_device", "bk_read_lensmap", "bk_last_error"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "blinky_amd", "libblinkyhip.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = [n for n in declared_functions() if n not in exported]
    assert not missing, f"declared in blinky_hip.h but not exported: {missing}"


def test_ctypes_binding_covers_the_header():
    import blinky_amd.ffi as ffi
    assert sorted(ffi.EXPORTS) == declared_functions()


def test_no_cpu_fallback_without_gpu():
    """Without a usable GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import blinky_amd
    with pytest.raises(blinky_amd.BlinkyError, match="no CPU fallback"):
        blinky_amd.Context()


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under blinky_amd/ or include/ may mention it."""
    bad = []
    for base in ("blinky_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle[/_.]|liboracle|okpy_|ok_state", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
