"""The N > 1 path on CPU: two gloo ranks shard a frame by row stripes with blinky_amd.multigpu (the code
bench.py runs over RCCL), each producing its stripe from ITS OWN stripe-local lensmap, and rank 0
reassembles.  No GPU here, so the per-stripe warp is done by the oracle - what is under test is the
partition, the stripe-local build semantics, the gather and the display[] reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, lens, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import importlib.util
    spec = importlib.util.spec_from_file_location("multigpu", os.path.join(ROOT, "blinky_amd", "multigpu.py"))
    multigpu = importlib.util.module_from_spec(spec)     # (blinky_amd/__init__ needs the HIP library; this module does not)
    spec.loader.exec_module(multigpu)
    import oracle_ffi as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = multigpu.stripe_bounds(H, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    lm = O.lensmap("cube", lens, None, W, H)             # stand-in for bk_set_rows + bk_build on this rank
    off = lm.offsets.reshape(H, W)[r0:r1]
    tin = lm.tints.reshape(H, W)[r0:r1]
    # display flags of THIS stripe only: plates its rows reference
    ps = lm.ps
    disp = [0] * 6
    for p in np.unique(off[off != O.NULL] // (ps * ps)):
        disp[int(p)] = 1
    disp = multigpu.or_display(disp, world)
    F = 3
    stripes = np.zeros((F, r1 - r0, W), np.uint8)
    for f in range(F):
        O.apply(off, tin, W, r1 - r0, O.lcg_globe(ps, 6, f), stripes[f])
    got = multigpu.gather_stripes(torch.from_numpy(stripes), bounds, rank, world, dst=0)
    if rank == 0:
        frames = multigpu.assemble(got).numpy()
        want = np.zeros((F, H, W), np.uint8)
        for f in range(F):
            O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(ps, 6, f), want[f])
        ok = np.array_equal(frames, want) and disp[: lm.numplates] == lm.display
        open(out_path, "w").write("ok" if ok else "MISMATCH")
    else:
        assert got is None
    # the batch exchange: frame f reassembled on rank f % world
    mine = multigpu.owned_frames(F, rank, world)
    outb = torch.zeros((max(1, (F + world - 1) // world), H, W), dtype=torch.uint8)
    multigpu.exchange_rotating(torch.from_numpy(stripes), bounds, rank, world, outb)
    ok2 = True
    for f in mine:
        want = np.zeros((H, W), np.uint8)
        O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(ps, 6, f), want)
        ok2 = ok2 and np.array_equal(outb[f // world].numpy(), want)
    open(out_path + f".rot{rank}", "w").write("ok" if ok2 else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 270), (3, 271)])      # even and uneven stripes
def test_stripes_gather_to_the_full_frame(tmp_path, world, H):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), H, 480, "hammer", out), nprocs=world, join=True)
    assert open(out).read() == "ok"
    for r in range(world):
        assert open(out + f".rot{r}").read() == "ok"        # every rank holds its frames of the batch, complete


def test_stripe_bounds_cover_every_row_once():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("multigpu", os.path.join(ROOT, "blinky_amd", "multigpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for H in (1, 7, 1080, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            b = m.stripe_bounds(H, n)
            assert b[0] == 0 and b[-1] == H and all(x <= y for x, y in zip(b, b[1:]))
            assert max(y - x for x, y in zip(b, b[1:])) - min(y - x for x, y in zip(b, b[1:])) <= 1
