"""The C-ABI boundary: libblinkyhip.so loads and exports every function include/blinky_hip.h
declares (no compute calls - this runs without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(headers=("blinky_hip.h", "blinky_hip_debug.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["bk_create", "bk_destroy", "bk_load_lens", "bk_load_globe", "bk_resize", "bk_build",
                 "bk_upload_plate", "bk_apply", "bk_apply_device", "bk_read_lensmap", "bk_last_error"]:
        assert must in names


def test_debug_entry_points_live_in_their_own_header():
    """the drop-in boundary (blinky_hip.h) declares no developer knob or test hook; those are blinky_hip_debug.h's, and the
    product reads no test switches from the environment"""
    assert not [n for n in declared_functions(("blinky_hip.h",)) if n.startswith("bk_debug_")]
    dbg = declared_functions(("blinky_hip_debug.h",))
    assert dbg and all(n.startswith("bk_debug_") for n in dbg), dbg
    for dp, _, files in os.walk(os.path.join(ROOT, "blinky_amd", "csrc")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h")) and f != "bk_embed.cpp":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for env in re.findall(r'getenv\("([A-Z_]+)"\)', txt):
                    # (LUA_PATH / LUA_PATH_5_2: where Lua's `require` looks, as the reference's VM reads them - loadlib.c)
                    assert env in ("BLINKY_HIP_CACHE", "XDG_CACHE_HOME", "HOME", "BLINKY_HIP_FIXUP_THREADS", "BLINKY_HIP_COMM", "BLINKY_HIP_COMM_TIMEOUT", "BLINKY_HIP_HOSTCXX", "BLINKY_HIP_KEEP_HW_QUEUES", "GPU_MAX_HW_QUEUES", "PATH",
                                   "LUA_PATH", "LUA_PATH_5_2"), (f, env)


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


ENGINE = "/root/reference/engine"


@pytest.mark.ref
@pytest.mark.skipif(not os.path.isdir(ENGINE + "/NQ"), reason="needs the reference engine headers (/root/reference)")
def test_host_layer_compiles_inside_the_engine_tree(tmp_path):
    """The seam a TyrQuake maintainer uses: blinky_amd/host/fisheye_hip.c compiled with
    -DBLINKY_IN_ENGINE against the engine's own headers, in place of NQ/fisheye.c (engine/Makefile:818,834-841), must
    define exactly the symbols of engine/include/fisheye.h:4-9 plus fisheye_plate_fov (NQ/fisheye.c:299, read by
    common/r_main.c:417-418) - and nothing else with external linkage."""
    obj = str(tmp_path / "fisheye_hip.o")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror=implicit-function-declaration", "-fcommon", "-DNQ_HACK", "-DELF",
                           "-DBLINKY_IN_ENGINE", "-I", ENGINE + "/include", "-I", ENGINE + "/NQ", "-I", os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "blinky_amd", "host", "fisheye_hip.c"), "-o", obj])
    out = subprocess.check_output(["nm", "--defined-only", "--extern-only", obj], text=True)
    # ("C" = tentative definitions the engine's own headers leave in every translation unit under -fcommon, e.g. cvar_tree:
    #  NQ/fisheye.c has the same ones)
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip() and line.split()[-2] != "C")
    common = set(line.split()[-1] for line in out.splitlines() if line.strip() and line.split()[-2] == "C")
    data = {"fisheye_enabled", "fisheye_plate_fov"}        # (tentative definitions here as in NQ/fisheye.c:297-299)
    assert sorted(set(defined) - data) == sorted(["F_Init", "F_Shutdown", "F_RenderView", "F_WriteConfig"]), defined
    assert data <= set(defined) | common
    # what it needs from the engine is what NQ/fisheye.c needs plus libblinkyhip's C ABI - no HIP, no C++
    und = subprocess.check_output(["nm", "--undefined-only", obj], text=True)
    undefined = set(line.split()[-1] for line in und.splitlines() if line.strip())
    engine_ok = {"Cmd_AddCommand", "Cmd_SetCompletion", "Cmd_ExecuteString", "Cmd_Argc", "Cmd_Argv", "Con_Printf", "COM_ScanDir", "COM_WriteFile",
                 "Q_atof", "Q_atoi", "Z_Malloc", "Z_Free", "STree_AllocInit", "STree_InsertAlloc", "Hunk_TempAlloc", "LittleShort", "AngleVectors", "VectorMA",
                 "VectorNormalize", "CrossProduct", "R_PushDlights", "R_RenderView", "R_ViewChanged", "R_SetVrect", "Draw_TileClear",
                 "D_EnableBackBufferAccess", "D_DisableBackBufferAccess", "vid", "r_refdef", "scr_vrect", "sb_lines", "host_basepal", "com_basedir",
                 "com_gamedir", "Sys_Error", "Cvar_RegisterVariable", "Cvar_SetValue", "key_dest", "Key_SetBinding", "Cbuf_AddText", "Cbuf_InsertText"}
    header = set(declared_functions(("blinky_hip.h",)))
    stray = sorted(u for u in undefined if u not in engine_ok and u not in header and not u.startswith("_") and
                   u not in ("fopen", "fclose", "fread", "fwrite", "fprintf", "snprintf", "sprintf", "strcpy", "strcmp", "strncmp", "strlen", "strchr", "strrchr",
                             "strtok", "strcat", "strncpy", "memcpy", "memset", "malloc", "free", "atoi", "getenv", "fseek", "ftell", "realloc", "puts", "printf",
                             "putchar", "calloc", "memmove", "strstr", "stderr", "stdout", "abs", "sqrt", "tan", "atan", "fputc", "fputs", "opendir", "readdir", "closedir", "strtol", "strerror", "fflush"))
    assert not stray, stray
