#!/usr/bin/env python3
"""Generate tests/golden/saveglobe.json from the UNMODIFIED reference (oracle/_ref): the PCX files
cmd_saveglobe + save_globe + WritePCXplate (fisheye.c:1120-1136, 1396-1484) hand to COM_WriteFile, with the
plates holding the SURVEY.md 8(d) LCG stream of `frame`.  Run in the build container:
    make -C oracle _ref && python tests/golden/make_saveglobe_golden.py
Each record: file name, length and FNV-1a-64 of the file, and the count of 0xFE mask bytes it would hold
without RLE escapes (nmask, from the oracle's view of the same plate, for diagnosis)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ffi as O  # noqa: E402

CONFIGS = [("cube", "hammer", 160, 120, 2), ("trism", "hammer", 200, 150, 1), ("fast", "hammer", 180, 96, 3),
           ("cube", "hammer", 640, 480, 0)]


def main():
    out = []
    for globe, lens, W, H, frame in CONFIGS:
        lm, _ = O.ref_run(globe, lens, None, W, H, want_frame=False)
        ps = min(W, H)
        for wm in (0, 1):
            for plate in range(lm.numplates):
                name, data = O.ref_saveglobe("shot", wm, frame, plate, ps)
                rec = dict(globe=globe, lens=lens, W=W, H=H, frame=frame, with_margins=wm, plate=plate, name=name,
                           length=int(data.size), fnv=O.fnv(data))
                print(rec)
                out.append(rec)
    doc = dict(source="oracle/_ref = unmodified /root/reference/engine/NQ/fisheye.c (f_saveglobe path), gcc 11.4 -O2, glibc 2.35",
               files=out)
    with open(os.path.join(HERE, "saveglobe.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
