#!/usr/bin/env python3
"""Frame times of the reference's engine with its own fisheye.c and with the drop-in (oracle/_ref/engine/tq_ref, tq_hip: the same
engine objects, headless; tests/test_engine_dropin.py) on the generated one-room map: wall-clock milliseconds per frame - from the
frame's first call into a driver to its presentation; the engine itself never runs more than 72 frames a second - while the view turns, per lens, and the first frame after a lens change (the reference spreads its lensmap build over frames,
1/60 s of it per frame, fisheye.c:645, 813-826; the drop-in builds on the GPU inside the frame).

usage: tools/engine_fps.py [WxH=1920x1080] [frames=40] [lens ...]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_engine_dropin as T                                                              # noqa: E402


def session(lens, frames):
    return ["host_framerate 0.05", "scr_conspeed 1000000", "con_notifytime -1", "viewsize 120", "map box"] + ["wait"] * 8 + \
        ["f_lens " + lens, "+left"] + ["wait"] * frames + ["-left", "toggleconsole", "quit"]


def times_of(binary, script, size, extra=None):
    import tempfile
    path = tempfile.mktemp(prefix="bqtimes")
    env = {"BLINKY_HEADLESS_TIMES": path}
    env.update(extra or {})
    try:
        T.run_engine(binary, script, size, env_extra=env)
        return [float(x) for x in open(path).read().split()]
    finally:
        if os.path.exists(path):
            os.remove(path)


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    lenses = sys.argv[3:] or ["panini", "hammer", "stereographic", "quincuncial"]
    print("# %s, one-room map, view turning, status bar off; ms per presented frame: median of the last %d" % (size, frames // 2))
    for lens in lenses:
        row = [lens]
        for name, binary, extra in (("reference", T.TQ_REF, None), ("drop-in", T.TQ_HIP, None),
                                    ("drop-in, lens compiled before", T.TQ_HIP, None),
                                    ("drop-in on the resident apply (BLINKY_HIP_RESIDENT=1, a place per CU reserved)", T.TQ_HIP, {"BLINKY_HIP_RESIDENT": "1"}),
                                    ("resident, whole chip (BLINKY_HIP_RESERVE_SLOTS=0)", T.TQ_HIP, {"BLINKY_HIP_RESIDENT": "1", "BLINKY_HIP_RESERVE_SLOTS": "0"})):
            t = times_of(binary, session(lens, frames), size, extra)
            steady = t[-(frames // 2):]
            change = max(t[8:8 + frames // 2]) if len(t) > 8 + frames // 2 else float("nan")
            row.append("%s: %.2f ms/frame (slowest frame after the lens change %.1f ms)" % (name, statistics.median(steady), change))
        print("  ".join(row))


if __name__ == "__main__":
    main()
