"""Randomised configurations of the RESIDENT apply (bk_apply_resident_*, and the drop-in's calls riding on it: bk_set_resident_apply +
bk_upload_plate + bk_apply) against the CPU oracle - the per-frame call of F_RenderView (fisheye.c:803 -> render_lensmap 2406-2424).
A seed is a whole configuration: frame size, a table no particular lens produces (tests/test_apply_campaign_gpu.py), a stripe of the rows,
globes in the ring, rubix with a random palette (the tint is applied to the staged chunk: a tinted block map), forced block heights,
a reserved place per CU, pipelined submissions that wrap the ring, unaligned pitch and origin; then the same context in drop-in mode with
fresh plates.  The committed range runs in seconds; BLINKY_RESIDENT_CAMPAIGN=lo:hi runs a developer campaign.  Byte-exact."""
import os

import numpy as np
import pytest

import oracle_ffi as O
from test_apply_campaign_gpu import SIZES_H, SIZES_W, _table

pytestmark = pytest.mark.gpu


def _seeds():
    v = os.environ.get("BLINKY_RESIDENT_CAMPAIGN")
    if not v:
        return range(16)
    lo, hi = [int(x) for x in v.split(":")]
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seeds())
def test_random_resident_configuration(seed):
    import blinky_amd as bk
    import torch
    rng = np.random.default_rng(77000 + seed)
    W = int(rng.choice(SIZES_W)) if rng.random() < 0.6 else int(rng.integers(1, 900))
    H = int(rng.choice(SIZES_H)) if rng.random() < 0.6 else int(rng.integers(1, 520))
    ps = min(W, H)
    R = int(rng.integers(1, 6))
    nf = int(rng.integers(1, 3 * R + 2))                               # frames submitted back to back (wrap the ring)
    off, tints, what = _table(rng, W, H, ps)
    r0, r1 = 0, H
    if H > 1 and rng.random() < 0.35:
        r0 = int(rng.integers(0, H - 1))
        r1 = int(rng.integers(r0 + 1, H + 1))
    shape = int(rng.choice([0, 0, 0, 1, 2, 4]))
    reserve = int(rng.choice([0, 0, 0, 1, 3]))
    ablation = int(rng.choice([0, 0, 0, 8, 16, 2048, 4096]))           # every frame on its own / no stride / not the one-block form / no table forms
    pitch = W + int(rng.integers(0, 9))
    x0 = int(rng.integers(0, pitch - W + 1))
    y0 = int(rng.integers(0, 4))
    rubix = bool(rng.random() < 0.5)
    pal = O.palmap(((np.arange(768) * int(rng.integers(1, 250)) + 11) % 256).astype(np.uint8))
    dropin = bool(rng.random() < 0.5)
    cfg = (f"seed {seed}: {W}x{H} {what} rows [{r0},{r1}) ring {R} frames {nf} shape {shape} reserve {reserve} ablation {ablation} "
           f"pitch {pitch} origin ({x0},{y0}) rubix {rubix} then drop-in {dropin}")

    ctx = bk.Context()
    ctx.set_frames(R)
    ctx.resize(W, H)
    ctx.set_rows(r0, r1)
    if shape:
        ctx.set_tile_shape(shape)
    ctx.set_ablation(ablation)
    ctx.set_resident_share(0, 1, reserve)
    globes = [O.lcg_globe(ps, 6, 17 * seed + f) for f in range(R)]
    for f in range(R):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(off.reshape(H, W)[r0:r1].ravel(), tints.reshape(H, W)[r0:r1].ravel())
    FH = H + y0 + 2

    def want_for(globe, rbx):
        full = np.full((FH, pitch), 77, np.uint8)
        O.apply(off, tints, W, H, globe, full, pitch, x0, y0, rbx, pal)
        want = np.full((FH, pitch), 77, np.uint8)
        want[y0 + r0:y0 + r1] = full[y0 + r0:y0 + r1]
        return want

    out = torch.full((nf, FH, pitch), 77, dtype=torch.uint8, device="cuda")
    ctx.synchronize()
    torch.cuda.synchronize()
    ctx.resident_begin(rubix, pal if rubix else None, idle_ms=2000)
    last = 0
    for f in range(nf):
        last = ctx.resident_submit(out[f].data_ptr(), pitch, frame=f % R, x0=x0, y0=y0)
    ctx.resident_wait(last)
    info = ctx.resident_info()
    ctx.resident_end()
    got = out.cpu().numpy()
    for f in range(nf):
        want = want_for(globes[f % R], rubix)
        if not np.array_equal(got[f], want):
            bad = np.argwhere(got[f] != want)
            raise AssertionError(f"{cfg}: session {info}: frame {f}: {len(bad)} bytes differ, first at (y, x) = {tuple(bad[0])}: got {got[f][tuple(bad[0])]} want {want[tuple(bad[0])]}")
    if dropin:
        # the drop-in's calls on the same context: fresh plates by DMA, the frame into a pitched host buffer, rubix switched in between
        ctx.set_resident_apply(True)
        for i, rbx in enumerate((rubix, not rubix, not rubix)):
            g = O.lcg_globe(ps, 6, 1000 + 3 * seed + i)
            for p in range(6):
                (ctx.upload_plate_async if i % 2 else ctx.upload_plate)(0, p, g[p])
            frame = np.full((FH, pitch), 77, np.uint8)
            ctx.apply(frame, 0, pitch, x0, y0, rbx, pal if rbx else None)
            want = want_for(g, rbx)
            if not np.array_equal(frame, want):
                bad = np.argwhere(frame != want)
                raise AssertionError(f"{cfg}: drop-in call {i} rubix {rbx}: {len(bad)} bytes differ, first at (y, x) = {tuple(bad[0])}")
    ctx.close()
