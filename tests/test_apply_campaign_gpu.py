"""Randomised configurations of the HIP lensmap APPLY (bk_apply_device, replacing render_lensmap, fisheye.c:2406-2424) against
the CPU oracle: frame sizes down to one pixel and off every alignment, stripes cut at any row, batches that wrap the globe ring,
unaligned pitches and origins, both kernels, forced and measured block shapes, small staging buffers, rubix.  A seed is a whole
configuration; the committed range runs in seconds, BLINKY_APPLY_CAMPAIGN=lo:hi runs a longer developer campaign.  Byte-exact."""
import os

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

SIZES_W = [1, 2, 3, 5, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 320, 333, 517, 640, 701]
SIZES_H = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 100, 199, 200, 240, 301, 400]


def _seeds():
    v = os.environ.get("BLINKY_APPLY_CAMPAIGN")
    if not v:
        return range(24)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


def _table(rng, W, H, ps):
    """a table of offsets no particular lens produces: random texels, runs along rows / columns, or a smooth (lens-like) walk over a
    plate; NULL pixels sprinkled, in blobs, or none; random tints"""
    n = W * H
    kind = rng.choice(["random", "rows", "columns", "smooth"])
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == "random":
        off = rng.integers(0, 6 * ps * ps, n, dtype=np.uint32)
    elif kind == "rows":
        y = rng.integers(0, ps, H)[:, None]
        p = rng.integers(0, 6, H)[:, None]
        x = (xx * int(rng.integers(1, 4)) + rng.integers(0, ps, H)[:, None]) % ps
        off = (p * ps * ps + y * ps + x).astype(np.uint32).reshape(-1)
    elif kind == "columns":
        x = rng.integers(0, ps, W)[None, :]
        y = (yy * int(rng.integers(1, 3)) + rng.integers(0, ps, W)[None, :]) % ps
        p = rng.integers(0, 6, W)[None, :]
        off = (p * ps * ps + y * ps + x).astype(np.uint32).reshape(-1)
    else:
        a, b, c, d = rng.uniform(-1.5, 1.5, 4)                        # an affine walk, wrapped: magnifying, minifying, rotated
        x = np.floor(xx * a + yy * b + rng.uniform(0, ps)).astype(np.int64) % ps
        y = np.floor(xx * c + yy * d + rng.uniform(0, ps)).astype(np.int64) % ps
        p = (xx * 6 // max(W, 1)) % 6
        off = (p * ps * ps + y * ps + x).astype(np.uint32).reshape(-1)
    nulls = rng.choice(["none", "sprinkled", "blob", "most"])
    if nulls == "sprinkled":
        off[rng.random(n) < 0.07] = O.NULL
    elif nulls == "blob":
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(1, max(W, H))
        off[(((yy - cy) ** 2 + (xx - cx) ** 2) > r * r).reshape(-1)] = O.NULL
    elif nulls == "most":
        off[rng.random(n) < 0.9] = O.NULL
    tints = rng.integers(0, 6, n).astype(np.uint8)
    tints[rng.random(n) < 0.5] = 255
    return off, tints, f"{kind}/{nulls}"


@pytest.mark.parametrize("seed", _seeds())
def test_random_apply_configuration(seed):
    import blinky_amd as bk
    import torch
    rng = np.random.default_rng(9000 + seed)
    W = int(rng.choice(SIZES_W)) if rng.random() < 0.7 else int(rng.integers(1, 720))
    H = int(rng.choice(SIZES_H)) if rng.random() < 0.7 else int(rng.integers(1, 420))
    ps = min(W, H)
    R = int(rng.integers(1, 10))                                      # resident globes
    nf = int(rng.integers(1, 2 * R + 2)) if rng.random() < 0.7 else 1  # frames per launch (may wrap the ring more than once)
    frame0 = int(rng.integers(0, R))
    off, tints, what = _table(rng, W, H, ps)
    r0, r1 = 0, H
    if H > 1 and rng.random() < 0.35:
        r0 = int(rng.integers(0, H - 1))
        r1 = int(rng.integers(r0 + 1, H + 1))
    variant = 2 if rng.random() < 0.75 else 0
    shape = int(rng.choice([0, 0, 1, 2, 4]))
    ldskb = int(rng.choice([0, 0, 0, 1, 4, 16, 48]))
    tuning = bool(rng.random() < 0.5)
    ablation = int(rng.choice([0, 0, 0, 32, 128, 4096, 32 + 4096, 2048, 64]))
    pitch = W + int(rng.integers(0, 8))
    x0 = int(rng.integers(0, pitch - W + 1))
    y0 = int(rng.integers(0, 4))
    rubix = bool(rng.random() < 0.4)
    pal = O.palmap(((np.arange(768) * int(rng.integers(1, 250)) + 11) % 256).astype(np.uint8))
    flip = bool(rng.random() < 0.35)                                  # (r5) f_rubix switched between launches: the block map changes flavour (tinted <-> plain)
    cfg = (f"seed {seed}: {W}x{H} {what} rows [{r0},{r1}) ring {R} frames {nf} from {frame0} variant {variant} shape {shape} lds {ldskb}K "
           f"tuning {tuning} ablation {ablation} pitch {pitch} origin ({x0},{y0}) rubix {rubix} flip {flip}")

    ctx = bk.Context()
    ctx.set_frames(R)
    ctx.resize(W, H)
    ctx.set_rows(r0, r1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_apply_variant(variant)
    ctx.set_blockmap_tuning(tuning)
    if shape:
        ctx.set_tile_shape(shape)
    if ldskb:
        ctx.set_tile_shape(400 + ldskb)
    ctx.set_ablation(ablation)
    globes = [O.lcg_globe(ps, 6, 31 * seed + f) for f in range(R)]
    for f in range(R):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(off.reshape(H, W)[r0:r1].ravel(), tints.reshape(H, W)[r0:r1].ravel())
    FH = H + y0 + 2
    rubix0 = rubix
    for rep in range(3 if flip else 2):                               # (the second launch runs on the block map the first one compiled / measured)
        rubix = (not rubix0) if (flip and rep == 1) else rubix0
        out = torch.full((nf, FH, pitch), 77, dtype=torch.uint8, device="cuda")
        ctx.apply_device(out.data_ptr(), pitch, FH * pitch, frame0=frame0, nframes=nf, x0=x0, y0=y0, rubix_on=rubix, pal=pal)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for f in range(nf):
            full = np.full((FH, pitch), 77, np.uint8)
            O.apply(off, tints, W, H, globes[(frame0 + f) % R], full, pitch, x0, y0, rubix, pal)
            want = np.full((FH, pitch), 77, np.uint8)
            want[y0 + r0:y0 + r1] = full[y0 + r0:y0 + r1]             # a stripe context writes its rows only
            if not np.array_equal(got[f], want):
                bad = np.argwhere(got[f] != want)
                raise AssertionError(f"{cfg}: launch {rep} frame {f}: {len(bad)} bytes differ, first at (y, x) = {tuple(bad[0])}: "
                                     f"got {got[f][tuple(bad[0])]} want {want[tuple(bad[0])]}")
    ctx.close()
