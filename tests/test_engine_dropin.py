"""The drop-in inside the engine it is a drop-in for.  oracle/_ref/engine holds the reference's TyrQuake engine (NQ client, software
renderer) built headless twice from the sources where they lie (oracle/Makefile, target _ref_engine): `tq_ref` with the UNMODIFIED
engine/NQ/fisheye.c, `tq_hip` with blinky_amd/host/fisheye_hip.c + libblinkyhip.so in its place - every other object file is the same, down
to the display-less video driver (oracle/ref/engine/headless.c) that logs a hash of every frame the engine presents.  Both run the same
console script on a generated game directory (oracle/ref/engine/mkgame.py: one textured, lit room; the real id1/pak0.pak is not in the
reference tree) and everything a player could see is compared: every presented frame (the real R_RenderView renders the plates, the real
screen / status bar / console code draws around and over the warped view), the console text, the files written (config.cfg, screenshots,
f_saveglobe's plate images).

Without a GPU: that both engines link with nothing unresolved, and console sessions (no map, so no frame is warped)."""
import os
import pathlib
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = os.path.join(ROOT, "oracle", "_ref", "engine")
TQ_REF = os.path.join(ENGINE, "tq_ref")
TQ_HIP = os.path.join(ENGINE, "tq_hip")
GAME = os.path.join(ENGINE, "game")
needs_engines = pytest.mark.skipif(not (os.path.exists(TQ_REF) and os.path.exists(TQ_HIP) and os.path.isdir(GAME)),
                                   reason="oracle/_ref/engine not built (needs /root/reference: make -C oracle _ref _ref_engine)")

CONNECT = ["host_framerate 0.05",      # every frame advances the game by the same 50 ms whatever the wall clock says
           "scr_conspeed 1000000",      # the console leaves the screen at once instead of scrolling out by wall-clock time
           "con_notifytime -1",         # no notify lines: they expire by wall-clock time
           "map box"] + ["wait"] * 8


def run_engine(binary, script, size=None, env_extra=None, timeout=600, prepare=None):
    """-> (stdout, [frame log lines], {file name: bytes of what the engine wrote into its game directory})"""
    # the reference formats script paths into fixed 100-byte buffers (fisheye.c:1665, 1758): a short game directory
    base = pathlib.Path(tempfile.mkdtemp(prefix="bq", dir="/tmp"))
    try:
        shutil.copytree(GAME, base / "g")
        game = base / "g"
        (game / "id1" / "session.cfg").write_text("\n".join(script) + "\n")
        if prepare:
            prepare(game)
        # the engine writes (config.cfg, screenshots, plate images) into $HOME/.blinky/<game> (common/common.c:2045)
        env = dict(os.environ, HOME=str(base), BLINKY_HEADLESS_LOG=str(base / "frames.log"), BLINKY_HEADLESS_FRAMES="2000", BLINKY_HIP_SYNC_COMPILE="1")
        env.pop("BLINKY_HIP_DEVICES", None)
        env.setdefault("BLINKY_HIP_CACHE", os.path.join(tempfile.gettempdir(), "bq_hipcache"))     # compiled lenses shared by the sessions
        if size:
            env["BLINKY_HEADLESS_SIZE"] = size
        env.update(env_extra or {})
        r = subprocess.run([binary, "-basedir", ".", "-noconinput", "+exec", "session.cfg"], cwd=game, env=env, capture_output=True,
                           timeout=timeout)
        assert r.returncode == 0, (binary, r.returncode, r.stdout[-3000:], r.stderr[-3000:])
        frames = (base / "frames.log").read_text().splitlines() if (base / "frames.log").exists() else []
        out_dir = base / ".blinky" / "id1"
        written = {p.name: p.read_bytes() for p in out_dir.iterdir() if p.is_file()} if out_dir.is_dir() else {}
        if (game / "palette").exists():                                    # f_dumppal writes into the working directory (fisheye.c:916-931)
            written["palette"] = (game / "palette").read_bytes()
        return r.stdout.decode("latin-1"), frames, written
    finally:
        shutil.rmtree(base, ignore_errors=True)


def console_text(stdout):
    """what the engine printed, without the build stamp and without the product's 'no device' notes of the CPU sessions"""
    keep = []
    for line in stdout.splitlines():
        if line.startswith("Exe: ") or "this context has no device" in line:
            continue
        keep.append(line)
    return keep


@needs_engines
@pytest.mark.ref
def test_both_engines_link_and_the_drop_in_resolves_everything_from_the_engine():
    """tq_hip = the reference's engine objects + fisheye_hip.o + libblinkyhip.so, nothing unresolved, no Lua library (engine/Makefile:818,
    838-840 link one for fisheye.c); the only dynamic symbols beyond libc / libm are the C ABI's"""
    undefined = subprocess.run(["nm", "-D", "--undefined-only", TQ_HIP], capture_output=True, text=True, check=True).stdout.split("\n")
    names = [line.split()[-1].split("@")[0] for line in undefined if line.strip()]
    abi = sorted(n for n in names if n.startswith("bk_"))
    assert abi and all(n.startswith("bk_") for n in abi)
    assert not [n for n in names if n.startswith("lua")]
    header = open(os.path.join(ROOT, "include", "blinky_hip.h")).read()
    assert all(n + "(" in header for n in abi), [n for n in abi if n + "(" not in header]       # only the PUBLIC header's entry points
    needed = subprocess.run(["readelf", "-d", TQ_HIP], capture_output=True, text=True, check=True).stdout
    libs = sorted(line.split("[")[1].split("]")[0] for line in needed.splitlines() if "(NEEDED)" in line)
    assert libs == ["libblinkyhip.so", "libc.so.6", "libm.so.6"], libs
    defined = subprocess.run(["nm", TQ_HIP], capture_output=True, text=True, check=True).stdout
    for sym in ("F_Init", "F_Shutdown", "F_RenderView", "F_WriteConfig", "fisheye_enabled", "fisheye_plate_fov"):   # engine/include/fisheye.h:4-9
        assert any(line.split()[-1] == sym and line.split()[-2] in "TBD" for line in defined.splitlines() if len(line.split()) >= 3), sym


CONSOLE_SESSION = [
    "f_help", "fisheye", "f_lens", "f_globe", "f_fov", "f_vfov", "f_rubixgrid", "f_rubix", "f_rubix",
    "f_lens hammer", "f_lens", "f_fov", "f_lens quincuncial", "f_vfov", "f_lens eckert5", "f_lens nosuchlens", "f_lens", "f_lens panini",
    "f_globe trism", "f_globe", "f_globe nosuchglobe", "f_globe", "f_globe tetra", "f_fov 120", "f_fov", "f_vfov 75.9", "f_vfov", "f_cover",
    "f_fov", "f_contain", "f_vfov", "f_rubixgrid 4 8.5 2", "f_rubixgrid", "f_rubixgrid 1 2", "f_saveglobe", "f_shortcutkeys", "bind 3",
    "bind y", "f_shortcutkeys", "bind 3", "bind 9", "f_dumppal", "fisheye 0", "fisheye", "fisheye 1", "f_lens stereographic", "f_globe cube",
    "toggleconsole", "quit",
]


@needs_engines
@pytest.mark.ref
def test_console_sessions_in_the_real_engine_equal_the_reference():
    """the 13 fisheye commands through the engine's own Cmd_* / Cbuf / key binding code, and the config the engine writes on quit"""
    ref_out, _, ref_files = run_engine(TQ_REF, CONSOLE_SESSION)
    hip_out, _, hip_files = run_engine(TQ_HIP, CONSOLE_SESSION, env_extra={"BLINKY_HIP_DEVICE": "none"})
    assert "f_lens hammer; f_contain" in ref_out and "not a valid lens" in ref_out and "Enabled Fisheye shortcut keys" in ref_out
    assert console_text(hip_out) == console_text(ref_out)
    assert "config.cfg" in ref_files and b"f_lens \"stereographic\"" in ref_files["config.cfg"]
    assert hip_files.keys() == ref_files.keys()
    assert hip_files["config.cfg"] == ref_files["config.cfg"]
    assert len(ref_files["palette"].splitlines()) == 256 and hip_files["palette"] == ref_files["palette"]


FRAME_SESSION = CONNECT + [
    "wait", "f_lens hammer", "wait", "wait", "f_globe trism", "wait", "f_fov 120", "wait", "f_vfov 90", "wait", "f_cover", "wait",
    "f_lens quincuncial", "wait", "f_rubix", "wait", "f_rubixgrid 4 8 2", "wait", "f_rubix", "wait",
    "viewsize 120", "wait", "wait", "viewsize 60", "wait", "wait", "viewsize 100", "wait",
    "f_globe cube_edge", "f_lens stereographic", "wait", "+left", "wait", "wait", "wait", "-left", "+lookup", "wait", "wait", "-lookup",
    "f_lens eckert5", "wait", "f_lens nosuchlens", "wait", "wait", "f_lens panini", "wait", "f_globe nosuchglobe", "wait", "f_globe tetra",
    "wait", "f_fov 400", "wait", "f_globe fast", "f_lens fisheye1", "wait", "f_globe cube", "f_lens winkeltripel", "wait",
    "headless_size 512 384", "wait", "wait", "f_lens hammer", "wait", "headless_size 700 300", "wait", "f_saveglobe small", "wait",
    "headless_size 640 480", "wait",
    "f_saveglobe plate", "wait", "f_saveglobe full 1", "wait", "screenshot", "wait",
    "fisheye 0", "wait", "wait", "fisheye 1", "wait", "wait",
    "toggleconsole", "quit",            # (no frame with the console down: its input cursor blinks by wall-clock time, console.c)
]


@needs_engines
@pytest.mark.gpu
@pytest.mark.parametrize("size", ["320x200", "640x480", "1024x600"])
def test_every_frame_of_a_session_in_the_real_engine_equals_the_reference(size):
    ref_out, ref_frames, ref_files = run_engine(TQ_REF, FRAME_SESSION, size)
    hip_out, hip_frames, hip_files = run_engine(TQ_HIP, FRAME_SESSION, size)
    assert len(ref_frames) > 40 and len(set(ref_frames[i].split()[-1] for i in range(len(ref_frames)))) > 25      # the session does show things
    assert console_text(hip_out) == console_text(ref_out)
    assert len(hip_frames) == len(ref_frames)
    different = [(a, b) for a, b in zip(ref_frames, hip_frames) if a != b]
    assert not different, different[:5]
    assert sorted(hip_files) == sorted(ref_files) and len([n for n in ref_files if n.endswith(".pcx")]) >= 13
    for name in ref_files:
        assert hip_files[name] == ref_files[name], name


@needs_engines
@pytest.mark.gpu
def test_the_engine_session_with_the_warp_spread_over_three_stripe_contexts():
    ref_out, ref_frames, ref_files = run_engine(TQ_REF, FRAME_SESSION, "640x480")
    hip_out, hip_frames, hip_files = run_engine(TQ_HIP, FRAME_SESSION, "640x480", env_extra={"BLINKY_HIP_DEVICES": "0,0,0"})
    assert hip_frames == ref_frames
    assert console_text(hip_out) == console_text(ref_out)
    assert {n: hip_files[n] for n in ref_files} == ref_files


@needs_engines
@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"BLINKY_HIP_RESIDENT": "1"}, {"BLINKY_HIP_RESIDENT": "1", "BLINKY_HIP_RESERVE_SLOTS": "1"},
                                 {"BLINKY_HIP_RESIDENT": "1", "BLINKY_HIP_DEVICES": "0,0,0"}],
                         ids=["resident", "resident-reserve-1-slot", "resident-three-stripes"])
def test_the_engine_session_through_the_resident_apply(env):
    """BLINKY_HIP_RESIDENT=1: F_RenderView's per-frame calls (bk_upload_plate_async for every displayed plate, bk_apply) go through the
    resident kernel - plates re-tiled on the host and moved by DMA, a frame is a command, the frame comes back by DMA - and every
    presented frame, the console text and the written files still equal the unmodified engine's (fisheye.c:698-811, 2406-2450)"""
    ref_out, ref_frames, ref_files = run_engine(TQ_REF, FRAME_SESSION, "640x480")
    hip_out, hip_frames, hip_files = run_engine(TQ_HIP, FRAME_SESSION, "640x480", env_extra=env)
    different = [(a, b) for a, b in zip(ref_frames, hip_frames) if a != b]
    assert len(hip_frames) == len(ref_frames) and not different, different[:5]
    assert console_text(hip_out) == console_text(ref_out)
    assert {n: hip_files[n] for n in ref_files} == ref_files


# ---- random sessions -----------------------------------------------------------------------------------------------------------

def random_session(seed):
    """a seed is a whole session in the engine: frame size, and a script of console commands with frames in between"""
    import random
    sys_path_scripts = __import__("scripts")
    rng = random.Random(seed)
    size = rng.choice(["320x200", "400x300", "512x384", "640x400", "640x480", "800x600", "854x480", "1024x600", "1280x720"])
    lenses, globes = sys_path_scripts.LENSES, sys_path_scripts.GLOBES
    script = list(CONNECT)
    held = set()
    for _ in range(rng.randint(12, 28)):
        kind = rng.random()
        if kind < 0.25:
            script.append("f_lens " + rng.choice(lenses))
        elif kind < 0.35:
            script.append("f_globe " + rng.choice(globes))
        elif kind < 0.50:
            script.append(rng.choice(["f_fov %d" % rng.choice([30, 90, 120, 150, 180, 200, 270, 359, 400]),
                                      "f_vfov %d" % rng.choice([45, 90, 120, 170, 181]), "f_cover", "f_contain"]))
        elif kind < 0.58:
            script.append("viewsize %d" % rng.choice([30, 50, 70, 90, 100, 110, 120]))
        elif kind < 0.64:
            script.append("f_rubix")
        elif kind < 0.68:
            script.append("f_rubixgrid %d %g %g" % (rng.randint(1, 12), rng.choice([1, 2.5, 4, 8]), rng.choice([0.5, 1, 2])))
        elif kind < 0.78:
            key = rng.choice(["left", "right", "lookup", "lookdown"])
            script.append(("-" if key in held else "+") + key)
            held.symmetric_difference_update({key})
        elif kind < 0.82:
            script.append("fisheye %d" % rng.randint(0, 1))
        elif kind < 0.86:
            # (after a resize the reference's plate memory is a fresh malloc: what f_saveglobe writes for a plate the lensmap does not
            #  show is then uninitialised memory, fisheye.c:712-727 - only asked for while the first, zero-filled allocation is in use)
            if not any(c.startswith("headless_size") for c in script):
                script.append("f_saveglobe s%d %d" % (len(script), rng.randint(0, 1)))
        elif kind < 0.89:
            script.append("screenshot")
        elif kind < 0.92:
            script.append("f_lens no_such_lens" if rng.random() < 0.5 else "f_globe no_such_globe")
        elif kind < 0.96:
            script.append("headless_size %d %d" % rng.choice([(320, 200), (400, 300), (640, 360), (512, 512), (800, 450), (333, 241)]))
        script.extend(["wait"] * rng.randint(1, 3))
    script.extend("-" + k for k in sorted(held))
    script.extend(["wait", "toggleconsole", "quit"])
    return size, script


def _seeds(default):
    lo, hi = os.environ.get("BLINKY_ENGINE_CAMPAIGN", default).split(":")
    return range(int(lo), int(hi))


@needs_engines
@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(_seeds("0:6")))
def test_random_sessions_in_the_real_engine_equal_the_reference(seed):
    """BLINKY_ENGINE_CAMPAIGN=lo:hi runs a longer developer campaign"""
    size, script = random_session(seed)
    devices = {"BLINKY_HIP_DEVICES": "0,0"} if seed % 3 == 2 else {"BLINKY_HIP_RESIDENT": "1"} if seed % 3 == 1 else None
    ref_out, ref_frames, ref_files = run_engine(TQ_REF, script, size)
    hip_out, hip_frames, hip_files = run_engine(TQ_HIP, script, size, env_extra=devices)
    assert console_text(hip_out) == console_text(ref_out), (seed, size)
    different = [(a, b) for a, b in zip(ref_frames, hip_frames) if a != b]
    assert len(hip_frames) == len(ref_frames) and not different, (seed, size, different[:3])
    assert sorted(hip_files) == sorted(ref_files), (seed, size)
    for name in ref_files:
        assert hip_files[name] == ref_files[name], (seed, size, name)


@needs_engines
@pytest.mark.ref
def test_a_lens_that_requires_a_helper_file_loads_inside_the_engine():
    """the script library's require / dofile resolve against the engine's working directory, like the reference's Lua (loadlib.c): a lens
    next to the bundled ones that keeps its arithmetic in lua-scripts/lenses/shared/optics.lua"""
    helper = "local M = {}\nfunction M.squash(v) return v / (1 + v * v) end\nprint('optics loaded')\nreturn M\n"
    lens = ("local optics = require 'lua-scripts.lenses.shared.optics'\nmax_fov = 200\nmax_vfov = 200\nonload = 'f_fov 120'\n"
            "function lens_inverse(x, y) return optics.squash(x), optics.squash(y), 1 end\n")

    def add_files(game):
        (game / "lua-scripts" / "lenses" / "shared").mkdir()
        (game / "lua-scripts" / "lenses" / "shared" / "optics.lua").write_text(helper)
        (game / "lua-scripts" / "lenses" / "with_helper.lua").write_text(lens)

    out, _, _ = run_engine(TQ_HIP, ["f_lens with_helper", "f_lens", "f_fov", "toggleconsole", "quit"], env_extra={"BLINKY_HIP_DEVICE": "none"},
                           prepare=add_files)
    text = console_text(out)
    # (the module's print() lands where the reference's Lua would put it: on stdout, in the middle of cmd_lens' "f_lens <name>; <onload>" line)
    assert "f_lens with_helperoptics loaded" in text and "; f_fov 120" in text and "Currently: with_helper" in text and "Zoom currently: f_fov 120" in text


def random_console_session(seed):
    """console commands only (no map: the engine draws its full-screen console and never warps a frame) - every lens and globe by name,
    the zoom commands with and without arguments, the rubix commands, key bindings, invalid names"""
    import random
    names = __import__("scripts")
    rng = random.Random(seed)
    script = []
    for _ in range(rng.randint(15, 40)):
        script.append(rng.choice([
            "f_lens " + rng.choice(names.LENSES), "f_globe " + rng.choice(names.GLOBES), "f_lens", "f_globe", "f_fov", "f_vfov",
            "f_fov %s" % rng.choice(["90", "180.7", "-5", "abc", "1e3", "360"]), "f_vfov %d" % rng.choice([0, 60, 179, 400]), "f_cover", "f_contain",
            "f_rubix", "f_rubixgrid", "f_rubixgrid %d %s %s" % (rng.randint(0, 20), rng.choice(["4", "2.5", "x"]), rng.choice(["1", "0.25"])),
            "f_rubixgrid 3", "f_help", "fisheye", "fisheye %d" % rng.randint(0, 2), "f_shortcutkeys", "bind %d" % rng.randint(1, 9), "bind y",
            "f_saveglobe", "f_lens no_such_lens", "f_globe no_such_globe", "f_lens \"\"", "f_dumppal",
            "fisheye %s" % rng.choice(["abc", "-1", "7", "0x10", "1.9"]), "f_fov 90 extra words", "f_cover 12", "f_contain now", "f_lens PANINI",
            "f_lens hammer hammer", "f_globe cube edge", "f_vfov", "f_fov 0", "f_vfov -0.5", "f_rubixgrid -1 -1 -1", "f_rubixgrid 1e9 1e-9 0",
            "f_help me", "f_shortcutkeys on", "f_dumppal twice", "f_lens %s; f_fov %d" % (rng.choice(names.LENSES), rng.randint(1, 400)),
        ]))
    return script + ["toggleconsole", "quit"]


@needs_engines
@pytest.mark.ref
@pytest.mark.parametrize("seed", list(_seeds("0:12")) if "BLINKY_ENGINE_CONSOLE_CAMPAIGN" not in os.environ else
                         list(range(*[int(v) for v in os.environ["BLINKY_ENGINE_CONSOLE_CAMPAIGN"].split(":")])))
def test_random_console_sessions_in_the_real_engine_equal_the_reference(seed):
    script = random_console_session(seed)
    ref_out, _, ref_files = run_engine(TQ_REF, script)
    hip_out, _, hip_files = run_engine(TQ_HIP, script, env_extra={"BLINKY_HIP_DEVICE": "none"})
    assert console_text(hip_out) == console_text(ref_out), seed
    assert hip_files == ref_files, seed
