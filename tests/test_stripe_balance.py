"""The host logic of bk_comm_rebalance / bk_multi_rebalance (stripes of equal work instead of equal height), without a GPU:
the bounds derived from per-row costs."""
import numpy as np
import pytest

import oracle_ffi as O


@pytest.fixture(scope="module")
def ffi():
    import blinky_amd
    return blinky_amd.ffi


def shares(cost, W, bounds):
    c = cost.astype(np.int64) + max(1, W // 32)
    return [int(c[a:b].sum()) for a, b in zip(bounds, bounds[1:])]


@pytest.mark.parametrize("lens,W,H", [("hammer", 960, 540), ("quincuncial", 640, 480), ("panini", 960, 540), ("fisheye1", 640, 400)])
@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_bounds_split_the_mapped_pixels_evenly(ffi, lens, W, H, n):
    lm = O.lensmap("cube", lens, None, W, H)
    cost = (lm.offsets.reshape(H, W) != 0xFFFFFFFF).sum(axis=1).astype(np.uint32)
    b = ffi.stripe_bounds_from_costs(cost, W, n)
    assert len(b) == n + 1 and b[0] == 0 and b[-1] == H
    assert all(y1 - y0 >= 8 for y0, y1 in zip(b, b[1:]))                      # nobody is left without rows
    assert all(y % 8 == 0 for y in b[1:-1])                                   # the apply's smallest block height
    sh = shares(cost, W, b)
    rows8 = 8 * (int(cost.max()) + W // 32)                                   # the granularity: 8 of the fullest rows
    assert max(sh) <= sum(sh) / n + rows8, (b, sh)
    eq = [H * r // n for r in range(n + 1)]
    assert max(sh) <= max(shares(cost, W, eq)) + rows8                         # never worse than equal heights (to the granularity)


def test_degenerate_inputs(ffi):
    assert ffi.stripe_bounds_from_costs(np.zeros(100, np.uint32), 640, 4) == [0, 24, 48, 72, 100]     # nothing mapped: equal shares of the rows
    assert ffi.stripe_bounds_from_costs(np.full(10, 7, np.uint32), 64, 10) == list(range(11))         # as many ranks as rows
    b = ffi.stripe_bounds_from_costs(np.r_[np.zeros(90, np.uint32), np.full(10, 1000, np.uint32)], 640, 4)
    assert b[0] == 0 and b[-1] == 100 and b == sorted(b) and all(y1 - y0 >= 8 for y0, y1 in zip(b, b[1:]))   # all the work in the last rows
    with pytest.raises(ffi.BlinkyError):
        ffi.stripe_bounds_from_costs(np.ones(4, np.uint32), 64, 5)                                     # more ranks than rows


def test_complete_costs_are_taken_as_they_are(ffi):
    """W = 0: the costs already hold everything (the staged apply's block-map costs) - no per-row base is added"""
    cost = np.r_[np.full(64, 10, np.uint32), np.full(64, 30, np.uint32)]
    assert ffi.stripe_bounds_from_costs(cost, 0, 2) == [0, 88, 128]           # 640 + 24 * 30 = 1360 of 2560: the nearest multiple of 8 rows to the half
    assert ffi.stripe_bounds_from_costs(cost, 32 * 1000, 2) == [0, 64, 128]   # a base of 1000 a row drowns the difference
    assert ffi.stripe_bounds_from_costs(np.zeros(64, np.uint32), 0, 4) == [0, 16, 32, 48, 64]
