"""The C host layer (blinky_amd/host/fisheye_hip.c: F_Init, the console commands, F_RenderView, F_WriteConfig) against the UNMODIFIED
reference behind the same engine stand-in (oracle/_ref/libhosttest_ref.so = tests/host/engine_stub.c + engine/NQ/fisheye.c itself): random
sessions of console commands, resizes and frames, everything a user can observe compared after every step - console text, config, and
with a GPU the screen, the plate views the engine is asked to render and their fov (fisheye.c:683-811, 916-1176).  Each session runs in a
process of its own (tests/hostlayer_driver.py differential).  BLINKY_HOST_CAMPAIGN=lo:hi runs a longer developer campaign."""
import os
import subprocess
import sys

import pytest

from test_host_layer import ROOT, build_hostlib, game_dir

REFLIB = os.path.join(ROOT, "oracle", "_ref", "libhosttest_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref/libhosttest_ref.so not built (needs /root/reference)")


request_cleanup = []


@pytest.fixture(autouse=True)
def _cleanup():
    yield
    while request_cleanup:
        request_cleanup.pop()()


def run(tmp_path, seeds, frames, timeout=1500, devices=None):
    build_hostlib()
    # (the reference formats script paths into fixed 100-byte buffers, fisheye.c:1665, 1758: the game directory has to be short - pytest's
    #  own tmp_path under xdist is not)
    import pathlib
    import shutil
    import tempfile
    short = pathlib.Path(tempfile.mkdtemp(prefix="bk", dir="/tmp"))
    request_cleanup.append(lambda: shutil.rmtree(short, ignore_errors=True))
    base = game_dir(short)
    env = dict(os.environ)
    if devices:
        env["BLINKY_HIP_DEVICES"] = devices          # several stripe contexts (bk_multi): rows split, rebalanced, reassembled
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostlayer_driver.py"), "differential", base, seeds, frames],
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "differential ok" in r.stdout
    return r.stdout


@needs_ref
@pytest.mark.ref
def test_console_sessions_equal_the_reference(tmp_path):
    run(tmp_path, os.environ.get("BLINKY_HOST_CAMPAIGN", "0:40"), "0")


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0,0"], ids=["one-context", "three-stripes"])
def test_frames_of_random_sessions_equal_the_reference(tmp_path, devices):
    out = run(tmp_path, os.environ.get("BLINKY_HOST_CAMPAIGN", "0:8"), "1", devices=devices or os.environ.get("BLINKY_HIP_DEVICES"))
    assert " 0 frames compared" not in out
