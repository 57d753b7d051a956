"""Pinning the oracle wider than the hand-picked list of test_oracle_golden.py: randomised configurations - any shipped globe x
any shipped lens x any zoom command and angle x any frame size x any rubix grid - run through BOTH the oracle's restatement
(oracle/oracle.c + oracle_lenses.c) and oracle/_ref, the unmodified fisheye.c driven through F_Init -> console commands ->
F_RenderView (fisheye.c:698-811, 916-1176, 2367-2424): lensmap offsets, tints, scale, display flags, the built / zoom-failed
verdict and the rubix frame must be identical.  CPU only; needs /root/reference (oracle/_ref is built from it).  A seed is a whole
configuration; BLINKY_ORACLE_CAMPAIGN=lo:hi runs a longer developer campaign."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def _seeds():
    v = os.environ.get("BLINKY_ORACLE_CAMPAIGN")
    if not v:
        return range(60)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


def configuration(seed):
    rng = np.random.default_rng(3000 + seed)
    globe = str(rng.choice(S.GLOBES))
    lens = str(rng.choice(S.LENSES))
    deg = int(rng.choice([10, 45, 60, 90, 100, 120, 150, 179, 180, 181, 200, 270, 359, 360])) if rng.random() < 0.6 else int(rng.integers(1, 400))
    zoom = [None, None, f"f_fov {deg}", f"f_vfov {deg}", "f_cover", "f_contain"][int(rng.integers(0, 6))]
    if rng.random() < 0.5:
        W, H = int(rng.integers(8, 400)), int(rng.integers(8, 300))
    else:
        W, H = [(320, 200), (640, 480), (400, 300), (256, 256), (300, 500), (333, 217)][int(rng.integers(0, 6))]
    if os.environ.get("BLINKY_ORACLE_CAMPAIGN_SIZES") == "big":      # developer campaign at HD sizes (seconds per configuration)
        W, H = [(1280, 720), (1920, 1080), (1080, 1920), (1600, 1200), (2040, 1200)][int(rng.integers(0, 5))]
    grid = None if rng.random() < 0.5 else (int(rng.integers(1, 24)), float(rng.choice([0.5, 1, 2, 4, 7.5])), float(rng.choice([0, 0.25, 1, 3])))
    return globe, lens, zoom, W, H, grid


@ref
@pytest.mark.ref
@pytest.mark.parametrize("seed", _seeds())
def test_oracle_equals_unmodified_reference_on_a_random_configuration(seed):
    globe, lens, zoom, W, H, grid = configuration(seed)
    cfg = f"seed {seed}: {globe}/{lens} {zoom or 'onload'} {W}x{H} grid {grid}"
    gstr = None if grid is None else f"{grid[0]} {grid[1]} {grid[2]}"
    lm_ref, frame_ref = O.ref_run(globe, lens, zoom, W, H, rubix_on=True, grid=gstr)
    lm = O.lensmap(globe, lens, zoom, W, H, grid or (10, 4.0, 1.0))
    assert lm.built == lm_ref.built, cfg
    if not lm.built:
        return
    assert lm.scale == lm_ref.scale or (lm.scale != lm.scale and lm_ref.scale != lm_ref.scale), f"{cfg}: {lm.scale!r} != {lm_ref.scale!r}"
    assert lm.display == lm_ref.display, cfg
    bad = np.flatnonzero(lm.offsets != lm_ref.offsets)
    assert bad.size == 0, f"{cfg}: {bad.size} offsets differ, first at (y, x) = {divmod(int(bad[0]), W)}"
    np.testing.assert_array_equal(lm.tints, lm_ref.tints, err_msg=cfg)
    frame = np.zeros((H, W), np.uint8)
    O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, lm.numplates, 0), frame, rubix_on=True, pal=O.palmap(O.synthetic_basepal()))
    np.testing.assert_array_equal(frame, frame_ref, err_msg=cfg)


@ref
@pytest.mark.ref
def test_rubix_palettes_of_random_base_palettes():
    """create_palmap / find_closest_pal_index (fisheye.c:835-908) on base palettes built to tie: greys, a few levels, repeated colours -
    the unmodified reference, the oracle's restatement and the product's bk_create_palmap give the same six look-up tables"""
    import blinky_amd
    O.ref_run("cube", "panini", None, 64, 48, want_frame=False)           # F_Init
    rng = np.random.default_rng(5)
    for i in range(80):
        kind = i % 4
        if kind == 0:
            pal = rng.integers(0, 256, 768, dtype=np.uint8)
        elif kind == 1:
            pal = np.repeat(rng.integers(0, 256, 256, dtype=np.uint8), 3)
        elif kind == 2:
            pal = (rng.integers(0, 4, 768) * 85).astype(np.uint8)
        else:
            pal = np.tile(rng.integers(0, 256, 48, dtype=np.uint8), 16)
        want = O.ref_palettes_of(pal)
        np.testing.assert_array_equal(O.palmap(pal), want, err_msg=f"oracle, palette {i}")
        np.testing.assert_array_equal(blinky_amd.ffi.create_palmap(pal), want, err_msg=f"bk_create_palmap, palette {i}")
