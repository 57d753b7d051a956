"""The CPU oracle (oracle/oracle.c) against the goldens recorded from the UNMODIFIED reference
(tests/golden/lensmaps.json, made by tests/golden/make_golden.py from oracle/_ref), and - in
the build container - directly against oracle/_ref on extra configurations."""
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "lensmaps.json")))


@pytest.mark.parametrize("rec", GOLD["lensmaps"], ids=lambda r: f"{r['globe']}-{r['lens']}-{r['zoom']}-{r['W']}x{r['H']}")
def test_oracle_matches_reference_golden(rec):
    lm = O.lensmap(rec["globe"], rec["lens"], rec["zoom"], rec["W"], rec["H"])
    assert lm.built == rec["built"]
    assert repr(lm.scale) == rec["scale"]
    assert lm.display == rec["display"]
    assert lm.nonnull == rec["nonnull"]
    assert O.fnv(lm.offsets) == rec["fnv_offsets"]
    assert O.fnv(lm.tints) == rec["fnv_tints"]
    globe = O.lcg_globe(lm.ps, lm.numplates, 0)
    frame = np.zeros((rec["H"], rec["W"]), np.uint8)
    O.apply(lm.offsets, lm.tints, rec["W"], rec["H"], globe, frame)
    assert O.fnv(frame) == rec["fnv_frame"]


@pytest.mark.parametrize("rec", [r for r in GOLD["lensmaps"] if "fnv_frame_rubix" in r or ("fnv_frames" in r and r["W"] <= 3840)],
                         ids=lambda r: f"{r['globe']}-{r['lens']}-{r['W']}x{r['H']}")
def test_oracle_matches_reference_golden_batch_and_rubix_frames(rec):
    """the other frames of a golden batch (LCG globes 1.., oracle gather over the lensmap) and the f_rubix frame the unmodified
    reference warped with its own palettes (fisheye.c:2416-2419): a sample of the batch, every rubix record"""
    W, H = rec["W"], rec["H"]
    lm = O.lensmap(rec["globe"], rec["lens"], rec["zoom"], W, H)
    assert O.fnv(lm.offsets) == rec["fnv_offsets"]
    for f in ([1, 17, len(rec["fnv_frames"]) - 1] if "fnv_frames" in rec else []):
        frame = O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, f), np.zeros((H, W), np.uint8))
        assert O.fnv(frame) == rec["fnv_frames"][f], f
    if "fnv_frame_rubix" in rec:
        frame = O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, 0), np.zeros((H, W), np.uint8), W, 0, 0, True,
                        O.palmap(O.synthetic_basepal()))
        assert O.fnv(frame) == rec["fnv_frame_rubix"]


def test_palmap_golden():
    assert O.fnv(O.palmap(O.synthetic_basepal())) == GOLD["fnv_palettes"]


def test_lcg_stream_known_answers():
    # s0 = 0x9E3779B9*(p+1+6*frame); s <- s*1664525+1013904223; texel = s>>24
    g = O.lcg_globe(4, 2, 0)
    s = (0x9E3779B9 * 1) & 0xFFFFFFFF
    want = []
    for _ in range(16):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        want.append(s >> 24)
    assert g[0].ravel().tolist() == want
    assert g[2].sum() == 0


ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@ref
@pytest.mark.ref
@pytest.mark.parametrize("cfg", [
    ("cube", "panini", "f_fov 90", 200, 150),
    ("cube", "panini", "f_vfov 100", 257, 129),
    ("trism", "stereographic", "f_fov 200", 320, 240),
    ("trism", "hammer", "f_cover", 300, 300),
    ("trism", "quincuncial", None, 256, 256),
    ("trism", "eckert5", None, 200, 120),
    ("cube", "eckert5", "f_cover", 160, 120),
    # the second batch of transliterations (oracle_lenses.c): loops with break, repeat-until, a cache in script
    # globals, plate_to_ray + math.modf + table.unpack, computed plate vectors, a globe_plate override
    ("cube", "rectilinear", None, 320, 200),
    ("cube", "equirect", None, 320, 160),
    ("cube", "mercator", None, 300, 200),
    ("trism", "cylinder", None, 300, 200),
    ("cube", "miller", None, 320, 240),
    ("cube", "fisheye1", None, 256, 256),
    ("cube", "cubestereo", None, 320, 200),
    ("cube", "mollweide", None, 400, 200),
    ("cube", "mollweide", "f_fov 300", 300, 200),       # zoom through lens_forward (repeat ... until)
    ("cube", "eckert4", None, 400, 200),
    ("trism", "eckert4", "f_fov 200", 320, 200),
    ("cube", "winkeltripel", None, 400, 250),
    ("trism", "winkeltripel", "f_fov 250", 320, 200),
    ("cube", "debug", None, 300, 200),
    ("trism", "debug", None, 300, 200),
    ("tetra", "debug", None, 256, 256),
    ("tetra", "panini", None, 320, 200),
    ("tetra", "hammer", None, 300, 150),
    ("fast", "panini", "f_fov 200", 320, 200),
    ("fast", "stereographic", None, 300, 300),
    # ... and the rest: every shipped lens and globe goes through the unmodified reference
    ("cube", "cube", None, 320, 240),
    ("cube", "cube", "f_fov 300", 320, 240),           # zoom through its lens_forward
    ("cube", "eckert1", None, 200, 120),
    ("cube", "fahey", None, 320, 200),
    ("cube", "fisheye2", None, 256, 256),
    ("cube", "gallstereo", None, 320, 200),
    ("cube", "gins8", None, 200, 120),
    ("cube", "gumby", None, 320, 200),
    ("cube", "kavrayskiy7", None, 200, 120),
    ("cube", "larrivee", None, 200, 120),
    ("cube", "polyconic", None, 200, 150),
    ("cube", "sinusoidal", None, 200, 120),
    ("cube", "vandergrinten", None, 300, 300),
    ("trism", "vandergrinten", "f_fov 200", 300, 200),
    ("cube", "wagner6", None, 200, 120),
    ("cube", "winkel1", None, 200, 120),
    ("cube", "winkel2", None, 200, 120),
    ("cube_edge", "panini", None, 320, 200),
    ("cube_corner", "stereographic", None, 320, 200),
    ("cube_corner", "hammer", None, 300, 150),
])
def test_oracle_equals_unmodified_reference(cfg):
    lm_ref, frame_ref = O.ref_run(*cfg, rubix_on=True)
    lm = O.lensmap(*cfg)
    assert lm.built == lm_ref.built
    assert lm.scale == lm_ref.scale
    assert lm.display == lm_ref.display
    np.testing.assert_array_equal(lm.offsets, lm_ref.offsets)
    np.testing.assert_array_equal(lm.tints, lm_ref.tints)
    W, H = cfg[3], cfg[4]
    frame = np.zeros((H, W), np.uint8)
    O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, 0), frame,
            rubix_on=True, pal=O.palmap(O.synthetic_basepal()))
    np.testing.assert_array_equal(frame, frame_ref)


@ref
@pytest.mark.ref
def test_rubixgrid_variants_equal_reference():
    for grid in ["3 2 1", "10 4 1", "5 1 0.5"]:
        lm_ref, _ = O.ref_run("cube", "panini", None, 160, 120, grid=grid, want_frame=False)
        n, c, p = grid.split()
        lm = O.lensmap("cube", "panini", None, 160, 120, grid=(int(n), float(c), float(p)))
        np.testing.assert_array_equal(lm.tints, lm_ref.tints)


@ref
@pytest.mark.ref
def test_zoom_failures_match_reference():
    # quincuncial has no lens_forward: f_fov cannot scale (fisheye.c:1341-1345) -> nothing built
    lm_ref, _ = O.ref_run("cube", "quincuncial", "f_fov 90", 64, 48, want_frame=False)
    lm = O.lensmap("cube", "quincuncial", "f_fov 90", 64, 48)
    assert not lm_ref.built and not lm.built
    assert lm.nonnull == 0 and lm_ref.nonnull == 0
    # fov beyond max_fov (fisheye.c:1306)
    lm_ref, _ = O.ref_run("cube", "panini", "f_fov 400", 64, 48, want_frame=False)
    lm = O.lensmap("cube", "panini", "f_fov 400", 64, 48)
    assert not lm_ref.built and not lm.built
