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


def run_scenario(name, tmp_path, env=None, timeout=900):
    build_hostlib()
    base = game_dir(tmp_path)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostlayer_driver.py"), name, base], env=e,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_tab_completion_of_lens_and_globe_names(tmp_path):
    """Cmd_SetCompletion("f_lens" / "f_globe", ...) (fisheye.c:661-663, 1105-1117, 1163-1175)"""
    assert "complete ok" in run_scenario("complete", tmp_path)


@pytest.mark.gpu
def test_full_frames_through_the_c_host_layer(tmp_path):
    assert "frames ok" in run_scenario("frames", tmp_path)


@pytest.mark.gpu
def test_full_frames_with_the_warp_spread_over_three_stripe_contexts(tmp_path):
    """BLINKY_HIP_DEVICES=0,0,0: F_RenderView through bk_multi (three stripe contexts on the one GPU)"""
    assert "frames ok" in run_scenario("frames", tmp_path, {"BLINKY_HIP_DEVICES": "0,0,0"})


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0,0"], ids=["one-context", "three-stripes"])
def test_a_malformed_lens_result_draws_the_partial_table_and_renders_its_plates(tmp_path, devices):
    """the library's BK_E_SCRIPT-with-the-reference's-truncated-table (fisheye.c:2113-2115) is not discarded by F_RenderView: the
    display flags become those of the partial table"""
    assert "malformed ok" in run_scenario("malformed", tmp_path, {"BLINKY_HIP_DEVICES": devices} if devices else None)


@pytest.mark.gpu
def test_no_frame_waits_for_hiprtc(tmp_path):
    """asynchronous lens compilation (default in the host layer): the render loop never stalls on `f_lens`"""
    out = run_scenario("async", tmp_path)
    assert "async ok" in out
