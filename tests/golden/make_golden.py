#!/usr/bin/env python3
"""Generate tests/golden/lensmaps.json from the UNMODIFIED reference (oracle/_ref).

Run in the build container (needs /root/reference):
    make -C oracle _ref && python tests/golden/make_golden.py
Each record pins, for one (globe, lens, zoom, W, H): lens.scale, display flags, the
number of non-NULL entries, and FNV-1a-64 of the uint32 offset table, of the tint table
and of one warped frame over the SURVEY.md 8(d) LCG globe (frame 0, rubix off).
The GPU box has no /root/reference; tests there compare against this file."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_ffi as O  # noqa: E402

CONFIGS = [
    # BASELINE.json configs[0] (C1) and scaled-down / full versions of C2..C5
    ("cube", "panini", None, 640, 480),
    ("cube", "stereographic", None, 1920, 1080),
    ("cube", "quincuncial", None, 640, 480),
    ("cube", "quincuncial", None, 3840, 2160),
    ("trism", "panini", None, 960, 540),
    ("trism", "panini", None, 3840, 2160),
    ("cube", "hammer", None, 960, 540),
    ("cube", "hammer", None, 3840, 2160),
    ("cube", "panini", None, 3840, 2160),
    ("cube", "eckert5", None, 640, 480),          # forward map
    ("cube", "panini", "f_fov 120", 322, 203),    # odd sizes, W%4 != 0
    ("cube", "hammer", "f_cover", 500, 300),
    ("cube", "stereographic", "f_vfov 90", 300, 500),   # portrait: ps = W
    ("cube", "hammer", None, 7680, 4320),         # BASELINE.json configs[4] (C5); its 64-frame batch below
    # the second batch of hand transliterations (oracle/oracle_lenses.c): 16 lenses and 4 globes in all come from the
    # unmodified reference
    ("cube", "rectilinear", None, 640, 400),
    ("cube", "equirect", None, 640, 320),
    ("cube", "mercator", None, 600, 400),
    ("cube", "cylinder", None, 600, 400),
    ("cube", "miller", None, 640, 480),
    ("cube", "fisheye1", None, 512, 512),
    ("cube", "cubestereo", None, 640, 400),
    ("cube", "mollweide", None, 800, 400),
    ("cube", "eckert4", None, 800, 400),          # per-row cache in script globals
    ("cube", "winkeltripel", None, 800, 500),     # Newton iteration with `break`
    ("cube", "winkeltripel", None, 1920, 1080),
    ("cube", "debug", None, 600, 400),            # plate_to_ray, math.modf, table.unpack
    ("tetra", "debug", None, 512, 512),
    ("tetra", "panini", None, 640, 400),          # plate vectors computed by the globe script
    ("fast", "panini", "f_fov 200", 640, 400),    # globe_plate override
    ("fast", "stereographic", None, 512, 512),
    # the third batch: all 31 lenses and all 6 globes are now recorded from the unmodified reference
    ("cube", "cube", None, 640, 480),
    ("cube", "eckert1", None, 400, 240),          # (forward maps: the plate texels are scattered, sizes kept small)
    ("cube", "fahey", None, 640, 400),
    ("cube", "fisheye2", None, 512, 512),
    ("cube", "gallstereo", None, 640, 400),
    ("cube", "gins8", None, 400, 240),
    ("cube", "gumby", None, 640, 400),
    ("cube", "kavrayskiy7", None, 400, 240),
    ("cube", "larrivee", None, 400, 240),
    ("cube", "polyconic", None, 400, 300),
    ("cube", "sinusoidal", None, 400, 240),
    ("cube", "vandergrinten", None, 600, 600),
    ("cube", "wagner6", None, 400, 240),
    ("cube", "winkel1", None, 400, 240),
    ("cube", "winkel2", None, 400, 240),
    ("cube_edge", "panini", None, 640, 400),
    ("cube_corner", "stereographic", None, 640, 400),
    # round 3: the forward map at BASELINE sizes (the reference's own scatter takes 1.7 s at 1080p and ~7 s at 4K) - the
    # write-order keys, the 20-pixel guards and the stripe-filtered commit above 0.3 Mpx
    ("cube", "eckert5", None, 1920, 1080),
    ("cube", "eckert5", None, 3840, 2160),
    ("cube", "winkel2", None, 1920, 1080),
    ("cube", "winkel2", None, 3840, 2160),
    ("trism", "eckert5", None, 1920, 1080),
    ("cube", "sinusoidal", None, 2560, 1440),
]
# configs whose record also carries `fnv_frames`: one hash per frame of a batch over the LCG globes 0..n-1
# (SURVEY.md 8(d)).  Frame 0 comes from the unmodified reference; the others from the oracle's render_lensmap
# restatement applied to the REFERENCE's lensmap (building the 8K map 64 times over would take ten minutes).
# r6: the bench headline's own launch (4K cube/panini x 64) and C4 (4K trism/panini x 64) carry all 64 frames.
BATCH = {("cube", "hammer", None, 7680, 4320): 64, ("cube", "panini", None, 3840, 2160): 64, ("trism", "panini", None, 3840, 2160): 64}
# configs whose record also carries `fnv_frame_rubix`: frame 0 warped by the unmodified reference with f_rubix on (grid 10/4/1,
# the reference's own create_palmap over the synthetic base palette: fisheye.c:2416-2419)
RUBIX = {("cube", "panini", None, 3840, 2160), ("trism", "panini", None, 3840, 2160), ("cube", "panini", None, 640, 480)}


def main():
    # incremental by default: configurations already recorded are kept (`--all` derives every record again)
    path = os.path.join(HERE, "lensmaps.json")
    have = {}
    if "--all" not in sys.argv and os.path.exists(path):
        for r in json.load(open(path))["lensmaps"]:
            have[(r["globe"], r["lens"], r["zoom"], r["W"], r["H"])] = r
    out = []
    for globe, lens, zoom, W, H in CONFIGS:
        key = (globe, lens, zoom, W, H)
        if key in have and len(have[key].get("fnv_frames", [])) >= BATCH.get(key, 0) and (key not in RUBIX or "fnv_frame_rubix" in have[key]):
            out.append(have[key])
            continue
        lm, frame = O.ref_run(globe, lens, zoom, W, H)
        rec = dict(globe=globe, lens=lens, zoom=zoom, W=W, H=H, built=bool(lm.built), scale=repr(lm.scale),
                   display=lm.display, nonnull=lm.nonnull, fnv_offsets=O.fnv(lm.offsets),
                   fnv_tints=O.fnv(lm.tints), fnv_frame=O.fnv(frame))
        nb = BATCH.get((globe, lens, zoom, W, H), 0)
        if nb:
            import numpy as np
            hashes = []
            for f in range(nb):
                fr = O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, f), np.zeros((H, W), np.uint8))
                hashes.append(O.fnv(fr))
            assert hashes[0] == rec["fnv_frame"]            # the restatement's frame 0 IS the reference's
            rec["fnv_frames"] = hashes
        if key in RUBIX:
            lm2, frame2 = O.ref_run(globe, lens, zoom, W, H, rubix_on=True)
            assert O.fnv(lm2.offsets) == rec["fnv_offsets"] and O.fnv(lm2.tints) == rec["fnv_tints"]
            rec["fnv_frame_rubix"] = O.fnv(frame2)
        if key in have:                                     # an upgraded record must agree with what was pinned before
            old = have[key]
            assert all(rec[k] == old[k] for k in ("scale", "display", "nonnull", "fnv_offsets", "fnv_tints", "fnv_frame")), key
            assert rec.get("fnv_frames", [])[: len(old.get("fnv_frames", []))] == old.get("fnv_frames", []), key
        print(rec)
        out.append(rec)
    pal = O.ref_palettes()
    doc = dict(source="oracle/_ref = unmodified /root/reference/engine/NQ/fisheye.c, gcc 11.4 -O2, glibc 2.35",
               lensmaps=out, fnv_palettes=O.fnv(pal))
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
