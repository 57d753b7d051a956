"""Randomised configurations of the HIP lensmap BUILD (bk_build, replacing create_lensmap / resume_lensmap_inverse / _forward,
fisheye.c:2084-2397) against the CPU oracle on the platform libm: any shipped globe x any shipped lens x any zoom command and
angle x any frame size x any rubix grid x any stripe.  What this exercises beyond the goldens is the exactness machinery at sizes
and zooms nobody picked: every pixel whose texel depends on libm's last bits has to be flagged and settled on the host.  A seed is
a whole configuration; BLINKY_BUILD_CAMPAIGN=lo:hi runs a longer developer campaign.  Bit-exact."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.gpu


def _seeds():
    v = os.environ.get("BLINKY_BUILD_CAMPAIGN")
    if not v:
        return range(24)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seeds())
def test_random_build_configuration(seed):
    import blinky_amd as bk
    rng = np.random.default_rng(7000 + seed)
    globe = str(rng.choice(S.GLOBES))
    lens = str(rng.choice(S.LENSES))
    deg = int(rng.choice([10, 45, 60, 90, 100, 120, 150, 179, 180, 181, 200, 270, 359, 360])) if rng.random() < 0.6 else int(rng.integers(1, 400))
    zoom = [None, None, f"f_fov {deg}", f"f_vfov {deg}", "f_cover", "f_contain"][int(rng.integers(0, 6))]
    if rng.random() < 0.5:
        W, H = int(rng.integers(8, 640)), int(rng.integers(8, 420))
    else:
        W, H = [(320, 200), (640, 480), (400, 300), (512, 512), (300, 500), (854, 480), (333, 217)][int(rng.integers(0, 7))]
    if os.environ.get("BLINKY_BUILD_CAMPAIGN_SIZES") == "big":      # developer campaign at BASELINE's frame sizes (the oracle takes seconds)
        W, H = [(1920, 1080), (2560, 1440), (3840, 2160), (1080, 1920), (3440, 1440)][int(rng.integers(0, 5))]
    grid = (10, 4.0, 1.0) if rng.random() < 0.5 else (int(rng.integers(1, 24)), float(rng.choice([0.5, 1, 2, 4, 7.5])), float(rng.choice([0, 0.25, 1, 3])))
    r0, r1 = 0, H
    if rng.random() < 0.35:
        r0 = int(rng.integers(0, H - 1))
        r1 = int(rng.integers(r0 + 1, H + 1))
    cfg = f"seed {seed}: {globe}/{lens} {zoom or 'onload'} {W}x{H} grid {grid} rows [{r0},{r1})"
    lm = O.lensmap(globe, lens, zoom, W, H, grid)
    ctx = bk.Context()
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.set_rubixgrid(*grid)
    ctx.set_rows(r0, r1)
    try:
        display, scale = ctx.build()
        built = True
    except bk.ffi.BlinkyError:
        built = False
    assert built == lm.built, cfg
    if built:
        off, tin = ctx.read_lensmap()
        assert scale == lm.scale or (scale != scale and lm.scale != lm.scale), f"{cfg}: scale {scale!r} != {lm.scale!r}"
        want_off = lm.offsets.reshape(H, W)[r0:r1].ravel()
        bad = np.flatnonzero(off != want_off)
        assert bad.size == 0, f"{cfg}: {bad.size} of {off.size} offsets differ, first at (y, x) = {divmod(int(bad[0]), W)} (stripe row): {off[bad[0]]} != {want_off[bad[0]]}; fixups {ctx.last_build_fixups()}"
        np.testing.assert_array_equal(tin, lm.tints.reshape(H, W)[r0:r1].ravel(), err_msg=cfg)
        if (r0, r1) == (0, H):
            assert display[: lm.numplates] == lm.display, cfg
        else:                                                   # a stripe sees the plates its own rows read (forward maps: all of them)
            assert all(d <= w for d, w in zip(display[: lm.numplates], lm.display)), cfg
    ctx.close()
