"""Randomised configurations of the multi-GPU schedule (bk_multi_*: row stripes, stripe-local build, stripe apply, the gather onto
one rank and the rotating exchange - SURVEY.md 8(e)) against the CPU oracle: any number of ranks from 2 to 8, any frame size that has a
row per rank, stripes equal or cut by work, batches of 1..12 frames, rubix on and off, two buffer pairs in flight.  On this one-GPU box
the ranks are stripe contexts on device 0 and the transport is device-to-device copies: the SCHEDULE (who sends which rows of which
frame where) is the code the RCCL transport runs too.  BLINKY_COMM_CAMPAIGN=lo:hi runs a longer developer campaign.  Byte-exact."""
import os

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.gpu


def _seeds():
    v = os.environ.get("BLINKY_COMM_CAMPAIGN")
    if not v:
        return range(12)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seeds())
def test_random_multi_gpu_configuration(seed):
    import blinky_amd as bk
    import torch
    rng = np.random.default_rng(6000 + seed)
    N = int(rng.integers(2, 9))
    globe = str(rng.choice(["cube", "trism", "tetra", "cube_edge"]))
    lens = str(rng.choice(["panini", "hammer", "stereographic", "quincuncial", "mercator", "eckert5", "fisheye1", "cube"]))
    W = int(rng.integers(64, 640))
    H = int(rng.integers(max(48, 8 * N), 420))
    F = int(rng.integers(1, 13))
    rubix = bool(rng.random() < 0.4)
    rebalance = bool(rng.random() < 0.5)
    root = int(rng.integers(0, N))
    cfg = f"seed {seed}: {N} ranks {globe}/{lens} {W}x{H} x{F} rubix {rubix} rebalance {rebalance} root {root}"
    lm = O.lensmap(globe, lens, None, W, H)
    if not lm.built:
        pytest.skip("the lens' own zoom does not build at this size")
    m = bk.Multi([0] * N)
    m.set_frames(F)
    m.load_globe(S.script("globes", globe), globe)
    m.load_lens(S.script("lenses", lens), lens)
    info = m.ctx(0).lens_info()
    m.set_zoom(*S.zoom_args(info.onload.decode()))
    m.resize(W, H)
    display, scale = m.build()
    assert scale == lm.scale and display[: lm.numplates] == lm.display, cfg
    bounds = [H * r // N for r in range(N + 1)]
    if rebalance:
        bounds = m.rebalance()
        assert bounds[0] == 0 and bounds[-1] == H and all(b1 > b0 for b0, b1 in zip(bounds, bounds[1:])), (cfg, bounds)
        m.build()
    for r in range(N):
        off, tin = m.ctx(r).read_lensmap()
        np.testing.assert_array_equal(off, lm.offsets.reshape(H, W)[bounds[r]:bounds[r + 1]].ravel(), err_msg=f"{cfg}: stripe {r}")
        np.testing.assert_array_equal(tin, lm.tints.reshape(H, W)[bounds[r]:bounds[r + 1]].ravel(), err_msg=f"{cfg}: stripe {r}")
    for f in range(F):
        for p in range(6):
            m.fill_plate_lcg(f, p, 17 * seed + f)
    pal = O.palmap(O.synthetic_basepal())
    want = [O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, 17 * seed + f), np.zeros((H, W), np.uint8), rubix_on=rubix, pal=pal)
            for f in range(F)]
    nown = (F + N - 1) // N
    stripes = [[torch.zeros((F, bounds[r + 1] - bounds[r], W), dtype=torch.uint8, device="cuda") for r in range(N)] for _ in range(2)]
    frames = [[torch.zeros((nown, H, W), dtype=torch.uint8, device="cuda") for r in range(N)] for _ in range(2)]
    gathered = torch.zeros((F, H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for step in range(3):                                   # two buffer pairs in flight, as bench.py's steps use them
        b = step & 1
        m.wait(b)
        m.apply_stripes([t.data_ptr() for t in stripes[b]], frame0=0, nframes=F, rubix_on=rubix, pal=pal)
        m.exchange_rotating([t.data_ptr() for t in stripes[b]], F, [t.data_ptr() for t in frames[b]], H * W, slot=b)
    m.synchronize()
    for b in range(2):
        for f in range(F):
            np.testing.assert_array_equal(frames[b][f % N][f // N].cpu().numpy(), want[f], err_msg=f"{cfg}: rotating exchange, buffer {b} frame {f}")
    m.wait(0)
    m.gather([t.data_ptr() for t in stripes[0]], F, root, gathered.data_ptr(), H * W, slot=0)
    m.synchronize()
    for f in range(F):
        np.testing.assert_array_equal(gathered[f].cpu().numpy(), want[f], err_msg=f"{cfg}: gather onto rank {root}, frame {f}")
    m.close()
