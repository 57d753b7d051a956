"""The C host layer blinky_amd/host/fisheye_hip.c (the drop-in for engine/NQ/fisheye.c) driven through
an engine stand-in (tests/host/engine_stub.c): console commands, config persistence, and - on the GPU -
a whole F_RenderView frame compared byte for byte with the oracle."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import scripts as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTLIB = os.path.join(ROOT, "tests", "host", "libhosttest.so")


def game_dir(tmp_path):
    for kind in ("lenses", "globes"):
        d = tmp_path / "lua-scripts" / kind
        d.mkdir(parents=True)
        for n in S.names(kind):
            (d / f"{n}.lua").write_text(S.script(kind, n))
    return str(tmp_path)


def build_hostlib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host")], stdout=subprocess.DEVNULL)


def test_console_commands_and_config_without_gpu(tmp_path):
    """Command semantics of fisheye.c:916-1176 and F_WriteConfig (683-696); runs in a subprocess because
    the host layer, like fisheye.c, keeps its state in file-scope statics."""
    build_hostlib()
    base = game_dir(tmp_path)
    code = textwrap.dedent(f"""
        import ctypes as C
        h = C.CDLL({HOSTLIB!r})
        h.hosttest_console.restype = C.c_char_p
        h.hosttest_init({base!r}.encode())
        buf = C.create_string_buffer(4096)
        def cfg():
            h.hosttest_writeconfig(buf, 4096); return buf.value.decode()
        print("CFG0<<" + cfg() + ">>")
        h.hosttest_cmd(b"f_lens hammer")          # its onload is f_contain
        h.hosttest_cmd(b"f_globe trism")
        h.hosttest_cmd(b"f_rubixgrid 5 2 0.5")
        print("CFG1<<" + cfg() + ">>")
        h.hosttest_cmd(b"f_vfov 100.9")           # (int)Q_atof
        h.hosttest_cmd(b"f_lens nosuchlens")
        h.hosttest_cmd(b"f_fov")
        h.hosttest_cmd(b"f_rubix")
        h.hosttest_cmd(b"f_shortcutkeys")
        print("CFG2<<" + cfg() + ">>")
        print("CON<<" + h.hosttest_console().decode() + ">>")
    """)
    env = dict(os.environ, BLINKY_HIP_DEVICE="none")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    cfg0 = out.split("CFG0<<")[1].split(">>")[0]
    cfg1 = out.split("CFG1<<")[1].split(">>")[0]
    cfg2 = out.split("CFG2<<")[1].split(">>")[0]
    con = out.split("CON<<")[1].split(">>")[0]
    # F_Init defaults (fisheye.c:668-672)
    assert cfg0 == 'fisheye 1\nf_lens "panini"\nf_globe "cube"\nf_rubixgrid 10 4.000000 1.000000\nf_fov 180\n'
    assert cfg1 == 'fisheye 1\nf_lens "hammer"\nf_globe "trism"\nf_rubixgrid 5 2.000000 0.500000\nf_contain\n'
    assert cfg2 == 'fisheye 1\nf_lens ""\nf_globe "trism"\nf_rubixgrid 5 2.000000 0.500000\nf_vfov 100\n'
    assert "f_lens hammer; f_contain" in con
    assert "not a valid lens" in con
    assert "Zoom currently: f_vfov 100" in con
    assert "Rubix is ON" in con
    assert '[engine] bind 1 "f_lens panini"' in con and '[engine] bind p "f_globe fast"' in con


@pytest.mark.gpu
def test_full_frames_through_the_c_host_layer(tmp_path):
    import blinky_amd  # noqa: F401  (loads torch's HIP runtime before libblinkyhip, see blinky_amd/ffi.py)
    import oracle_ffi as O
    build_hostlib()
    base = game_dir(tmp_path)
    h = C.CDLL(HOSTLIB)
    h.hosttest_console.restype = C.c_char_p
    h.hosttest_plate_fov.restype = C.c_double
    assert h.hosttest_init(base.encode()) == 1, h.hosttest_console().decode()
    pal = O.palmap(O.synthetic_basepal())

    def frame(globe, lens, zoom, W, H, x0, y0, extra, rubix=False, bg=7, fidx=2):
        lm = O.lensmap(globe, lens, zoom, W, H)
        h.hosttest_resize(W, H, x0, y0, extra)
        order = [i for i, d in enumerate(lm.display) if d]
        arr = (C.c_int * len(order))(*order)
        pitch, vh = h.hosttest_rowbytes(), h.hosttest_vidheight()
        out = np.zeros((vh, pitch), np.uint8)
        n = h.hosttest_frame(arr, len(order), fidx, bg, out.ctypes.data_as(C.c_void_p))
        assert n == len(order), "the host must render exactly the plates the lensmap uses (display[])"
        want = np.full((vh, pitch), bg, np.uint8)
        # Draw_TileClear repaints [0,vid.width) x [0,vid.height); the plate renders are overwritten by it
        O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, fidx), want, pitch, x0, y0, rubix, pal)
        np.testing.assert_array_equal(out[:, : pitch - extra], want[:, : pitch - extra])
        return lm

    # defaults of F_Init: cube / panini / f_fov 180
    lm = frame("cube", "panini", None, 320, 200, 8, 4, 16)
    assert abs(h.hosttest_plate_fov(0) - float(O.globe_plates("cube")[0][9])) == 0     # fisheye_plate_fov = plate fov
    # same lens, rubix overlay on
    h.hosttest_cmd(b"f_rubix")
    frame("cube", "panini", None, 320, 200, 8, 4, 16, rubix=True)
    h.hosttest_cmd(b"f_rubix")
    # a lens whose onload changes the zoom, another globe, a resize, an odd origin
    h.hosttest_cmd(b"f_globe trism")
    h.hosttest_cmd(b"f_lens hammer")
    frame("trism", "hammer", None, 322, 203, 3, 1, 5)
    h.hosttest_cmd(b"f_fov 150")
    frame("trism", "hammer", "f_fov 150", 322, 203, 3, 1, 5)
    # forward-only lens
    h.hosttest_cmd(b"f_globe cube")
    h.hosttest_cmd(b"f_lens eckert5")
    frame("cube", "eckert5", None, 200, 120, 0, 0, 0)
    # an invalid lens blanks the view (fisheye.c:737-741, 2372): only the cleared background remains
    h.hosttest_console_clear()
    h.hosttest_cmd(b"f_lens doesnotexist")
    h.hosttest_resize(200, 120, 0, 0, 0)
    out = np.zeros((h.hosttest_vidheight(), h.hosttest_rowbytes()), np.uint8)
    h.hosttest_frame((C.c_int * 1)(0), 0, 0, 9, out.ctypes.data_as(C.c_void_p))
    assert (out == 9).all()
    assert "not a valid lens" in h.hosttest_console().decode()
    h.hosttest_shutdown()
