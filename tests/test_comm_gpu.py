"""Multi-GPU behind the C ABI (include/blinky_hip.h: bk_comm_*, bk_multi_*), on the one GPU the test box has.

RCCL refuses the same device twice in one communicator, so a one-GPU box cannot run two RCCL ranks; what runs here:
  * the schedule (who sends which rows where, uneven stripes, rotating roots, slots) over bk_multi's copy transport with
    two and three "devices" that are all GPU 0 - from a plain C host (tests/host/multi_test.c) and from Python;
  * librccl itself being found and answering (bk_comm_unique_id), and the single-rank communicator.
The N > 1 RCCL transport posts exactly the same per-rank op lists (bk_comm.cpp: ops_gather / ops_rotating -> post_rccl);
bench.py runs it when the driver launches N ranks."""
import os
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
import scripts as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


def oracle_frames(globe, lens, W, H, F):
    lm = O.lensmap(globe, lens, None, W, H)
    return lm, [O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, f), np.zeros((H, W), np.uint8)) for f in range(F)]


@pytest.mark.parametrize("devs", ["0,0", "0,0,0"])
def test_c_host_spreads_the_warp_over_several_ranks(tmp_path, devs):
    """tests/host/multi_test.c: bk_create_multi -> stripe-wise build -> host frame, gather onto rank 0, rotating exchange"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host"), "multi_test"], stdout=subprocess.DEVNULL)
    globe, lens, W, H, F = "cube", "hammer", 322, 203, 5           # W % 4 != 0, H % N != 0: uneven stripes
    (tmp_path / "g.lua").write_text(S.script("globes", globe))
    (tmp_path / "l.lua").write_text(S.script("lenses", lens))
    out = tmp_path / "frames.bin"
    r = subprocess.run([os.path.join(ROOT, "tests", "host", "multi_test"), str(tmp_path / "g.lua"), str(tmp_path / "l.lua"),
                        str(W), str(H), str(F), devs, str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lm, want = oracle_frames(globe, lens, W, H, F)
    head = r.stdout.splitlines()[0].split()
    assert head[1] == str(len(devs.split(","))) and head[3] == "0"            # ranks, copy transport
    assert float(head[5]) == lm.scale and head[7] == "".join(str(d) for d in lm.display + [0] * (6 - len(lm.display)))
    got = np.fromfile(out, np.uint8).reshape(1 + 2 * F, H, W)
    np.testing.assert_array_equal(got[0], want[0], err_msg="bk_multi_apply (host frame)")
    for f in range(F):
        np.testing.assert_array_equal(got[1 + f], want[f], err_msg=f"gather, frame {f}")
        np.testing.assert_array_equal(got[1 + F + f], want[f], err_msg=f"rotating exchange, frame {f}")


def test_multi_from_python_with_rubix_and_double_buffering(bk):
    import torch
    globe, lens, W, H, F, N = "trism", "panini", 480, 270, 6, 3
    m = bk.Multi([0] * N)
    assert not m.uses_rccl()
    m.set_frames(F)
    m.load_globe(S.script("globes", globe), globe)
    m.load_lens(S.script("lenses", lens), lens)
    m.set_zoom(bk.ffi.ZOOM_FOV, 180)
    m.resize(W, H)
    display, scale = m.build()
    lm = O.lensmap(globe, lens, None, W, H)
    assert scale == lm.scale and display[: lm.numplates] == lm.display
    # every stripe context holds exactly its rows of the oracle's table
    bounds = [H * r // N for r in range(N + 1)]
    for r in range(N):
        off, tin = m.ctx(r).read_lensmap()
        np.testing.assert_array_equal(off, lm.offsets.reshape(H, W)[bounds[r]:bounds[r + 1]].ravel())
    for f in range(F):
        for p in range(6):
            m.fill_plate_lcg(f, p, f)
    pal = O.palmap(O.synthetic_basepal())
    want = [O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, f), np.zeros((H, W), np.uint8), rubix_on=True, pal=pal) for f in range(F)]
    np.testing.assert_array_equal(m.apply(np.zeros((H, W), np.uint8), frame=2, rubix_on=True, pal=pal), want[2])
    # two buffer pairs in flight (slots 0 and 1), as bench.py's steps use them
    stripes = [[torch.zeros((F, bounds[r + 1] - bounds[r], W), dtype=torch.uint8, device="cuda") for r in range(N)] for _ in range(2)]
    frames = [[torch.zeros(((F + N - 1) // N, H, W), dtype=torch.uint8, device="cuda") for r in range(N)] for _ in range(2)]
    torch.cuda.synchronize()
    for step in range(4):
        b = step & 1
        m.wait(b)
        m.apply_stripes([t.data_ptr() for t in stripes[b]], frame0=0, nframes=F, rubix_on=True, pal=pal)
        m.exchange_rotating([t.data_ptr() for t in stripes[b]], F, [t.data_ptr() for t in frames[b]], H * W, slot=b)
    m.synchronize()
    for b in range(2):
        for f in range(F):
            np.testing.assert_array_equal(frames[b][f % N][f // N].cpu().numpy(), want[f], err_msg=f"buffer {b} frame {f}")
    m.close()


def test_c4_eight_stripes_full_size(bk):
    """BASELINE.json configs[3] as stated: 3840x2160 trism/panini row-striped over EIGHT ranks (270 rows each; here eight
    stripe contexts on the one GPU, copy transport): stripe-local build, stripe apply, gather onto rank 0 and the rotating
    exchange.  The concatenated stripe tables hash to the reference's 4K golden and every reassembled frame to the golden
    frame (frame 0: the unmodified reference's; the others: the oracle applied to the golden-checked table)."""
    import json
    import torch
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lensmaps.json")))["lensmaps"]
    rec = next(r for r in gold if r["globe"] == "trism" and r["lens"] == "panini" and r["W"] == 3840)
    globe, lens, W, H, F, N = "trism", "panini", 3840, 2160, 3, 8
    m = bk.Multi([0] * N)
    m.set_frames(F)
    m.load_globe(S.script("globes", globe), globe)
    m.load_lens(S.script("lenses", lens), lens)
    m.set_zoom(bk.ffi.ZOOM_FOV, 180)
    m.resize(W, H)
    display, scale = m.build()
    assert repr(scale) == rec["scale"] and display[: len(rec["display"])] == rec["display"]
    bounds = [H * r // N for r in range(N + 1)]
    offs, tins = [], []
    for r in range(N):
        assert m.ctx(r).size()[3:5] == (bounds[r], bounds[r + 1])
        off, tin = m.ctx(r).read_lensmap()
        assert off.size == (bounds[r + 1] - bounds[r]) * W
        offs.append(off)
        tins.append(tin)
    off, tin = np.concatenate(offs), np.concatenate(tins)
    assert O.fnv(off) == rec["fnv_offsets"] and O.fnv(tin) == rec["fnv_tints"] and int((off != O.NULL).sum()) == rec["nonnull"]
    for f in range(F):
        for p in range(5):
            m.fill_plate_lcg(f, p, f)
    want = [O.apply(off, tin, W, H, O.lcg_globe(2160, 5, f), np.zeros((H, W), np.uint8)) for f in range(F)]
    assert O.fnv(want[0]) == rec["fnv_frame"]
    # host frame (what the engine drop-in does with N GPUs)
    assert O.fnv(m.apply(np.zeros((H, W), np.uint8), frame=0)) == rec["fnv_frame"]
    # device frames: gather onto rank 0, then the rotating exchange
    stripes = [torch.zeros((F, bounds[r + 1] - bounds[r], W), dtype=torch.uint8, device="cuda") for r in range(N)]
    gathered = torch.zeros((F, H, W), dtype=torch.uint8, device="cuda")
    frames = [torch.zeros(((F + N - 1) // N, H, W), dtype=torch.uint8, device="cuda") for r in range(N)]
    torch.cuda.synchronize()
    m.apply_stripes([t.data_ptr() for t in stripes], frame0=0, nframes=F)
    m.gather([t.data_ptr() for t in stripes], F, 0, gathered.data_ptr(), H * W, slot=0)
    m.exchange_rotating([t.data_ptr() for t in stripes], F, [t.data_ptr() for t in frames], H * W, slot=1)
    m.synchronize()
    for f in range(F):
        np.testing.assert_array_equal(gathered[f].cpu().numpy(), want[f], err_msg=f"gather, frame {f}")
        np.testing.assert_array_equal(frames[f % N][f // N].cpu().numpy(), want[f], err_msg=f"rotating exchange, frame {f}")
    assert O.fnv(gathered[0].cpu().numpy()) == rec["fnv_frame"]
    m.close()


def test_row_costs_of_both_apply_variants(bk):
    """what the rebalance sums over the ranks: the direct-gather apply prices a row by its mapped pixels (+ W/32), the staged apply by
    the costs of the row's blocks in its block map - zero for no row, larger where the row touches more of the globe; rows of other
    stripes are 0"""
    globe, lens, W, H = "cube", "hammer", 640, 400
    lm = O.lensmap(globe, lens, None, W, H)
    mapped = (lm.offsets.reshape(H, W) != 0xFFFFFFFF).sum(axis=1)
    ctx = bk.Context()
    S.configure(ctx, globe, lens, None, (W, H))
    ctx.set_rows(96, 304)
    ctx.build()
    ctx.set_apply_variant(0)
    c0 = ctx.row_costs()
    np.testing.assert_array_equal(c0[96:304], mapped[96:304] + W // 32)
    assert not c0[:96].any() and not c0[304:].any()
    ctx.set_apply_variant(2)
    c2 = ctx.row_costs().astype(np.int64)
    assert (c2[96:304] > 0).all() and not c2[:96].any() and not c2[304:].any()
    # rows of one block row share their blocks' cost: constant over runs of 8 rows at least
    assert all(len(set(c2[y:y + 8])) == 1 for y in range(96, 304, 8))
    # and the staged apply still warps the stripe after its block map was compiled for the costs alone
    for p in range(6):
        ctx.fill_plate_lcg(0, p, 5)
    want = O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, 5), np.zeros((H, W), np.uint8))
    got = ctx.apply(np.zeros((H, W), np.uint8))
    np.testing.assert_array_equal(got[96:304], want[96:304])
    assert not got[:96].any() and not got[304:].any()
    ctx.close()


def test_stripes_of_equal_work(bk):
    """bk_multi_rebalance: hammer's ellipse leaves the top and bottom stripes of an equal-height split nearly empty; cut by
    what the rows cost the apply instead (its block map's costs), every stripe gets its share, and the reassembled frames are still the oracle's."""
    import torch
    globe, lens, W, H, F, N = "cube", "hammer", 640, 400, 4, 4
    m = bk.Multi([0] * N)
    m.set_frames(F)
    m.load_globe(S.script("globes", globe), globe)
    m.load_lens(S.script("lenses", lens), lens)
    m.set_zoom(bk.ffi.ZOOM_CONTAIN)
    m.resize(W, H)
    m.build()
    lm = O.lensmap(globe, lens, None, W, H)
    mapped = (lm.offsets.reshape(H, W) != 0xFFFFFFFF).sum(axis=1)
    equal = [H * r // N for r in range(N + 1)]
    share_eq = [mapped[equal[r]:equal[r + 1]].sum() for r in range(N)]
    assert max(share_eq) > 1.25 * (mapped.sum() / N)                      # the reason to do it
    cost = sum(m.ctx(r).row_costs().astype(np.int64) for r in range(N))   # every stripe prices its own rows from its block map
    assert (cost > 0).all()                                               # (empty rows too: they are dealt out as well)
    cost_eq = [cost[equal[r]:equal[r + 1]].sum() for r in range(N)]
    bounds = m.rebalance()
    assert bounds[0] == 0 and bounds[-1] == H and bounds == sorted(bounds) and all(b % 8 == 0 for b in bounds[1:-1])
    assert bounds == bk.ffi.stripe_bounds_from_costs(cost.astype(np.uint32), 0, N)
    share = [cost[bounds[r]:bounds[r + 1]].sum() for r in range(N)]
    assert max(share) < 1.12 * (sum(share) / N) and max(share) < max(cost_eq), (bounds, share, cost_eq)
    px = [mapped[bounds[r]:bounds[r + 1]].sum() for r in range(N)]
    assert max(px) < max(share_eq), (px, share_eq)                        # and nobody holds as many pixels as the fullest equal stripe did
    with pytest.raises(bk.BlinkyError):                                   # the stripes' lensmaps are gone: build again
        m.apply(np.zeros((H, W), np.uint8))
    m.build()
    for r in range(N):
        off, tin = m.ctx(r).read_lensmap()
        np.testing.assert_array_equal(off, lm.offsets.reshape(H, W)[bounds[r]:bounds[r + 1]].ravel())
        assert m.ctx(r).size()[3:] == (bounds[r], bounds[r + 1])
    for f in range(F):
        for p in range(6):
            m.fill_plate_lcg(f, p, f)
    want = [O.apply(lm.offsets, lm.tints, W, H, O.lcg_globe(lm.ps, 6, f), np.zeros((H, W), np.uint8)) for f in range(F)]
    np.testing.assert_array_equal(m.apply(np.zeros((H, W), np.uint8), frame=1), want[1])
    stripes = [torch.zeros((F, bounds[r + 1] - bounds[r], W), dtype=torch.uint8, device="cuda") for r in range(N)]
    frames = [torch.zeros(((F + N - 1) // N, H, W), dtype=torch.uint8, device="cuda") for r in range(N)]
    torch.cuda.synchronize()
    m.apply_stripes([t.data_ptr() for t in stripes], frame0=0, nframes=F)
    m.exchange_rotating([t.data_ptr() for t in stripes], F, [t.data_ptr() for t in frames], H * W, slot=0)
    m.synchronize()
    for f in range(F):
        np.testing.assert_array_equal(frames[f % N][f // N].cpu().numpy(), want[f], err_msg=f"frame {f}")
    # a new size starts from equal shares again
    m.resize(320, 200)
    assert [m.ctx(r).size()[3] for r in range(N)] == [200 * r // N for r in range(N)]
    m.close()


def test_rccl_is_reachable_and_a_single_rank_communicator_works(bk):
    import torch
    uid = bk.ffi.comm_unique_id()                    # dlopen(librccl) + ncclGetUniqueId
    assert len(uid) == 128 and any(uid)
    lm, want = oracle_frames("cube", "panini", 320, 240, 3)
    ctx = bk.Context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_frames(3)
    S.configure(ctx, "cube", "panini", None, (320, 240))
    comm = bk.Comm(ctx, 1, 0)
    assert comm.stripe(0) == (0, 240)
    display, _ = ctx.build()
    assert comm.or_display(display) == display
    for f in range(3):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, f)
    stripe = torch.zeros((3, 240, 320), dtype=torch.uint8, device="cuda")
    frames = torch.zeros((3, 240, 320), dtype=torch.uint8, device="cuda")
    ctx.apply_device(stripe.data_ptr(), 320, 240 * 320, frame0=0, nframes=3)
    comm.exchange_rotating(stripe.data_ptr(), 3, frames.data_ptr(), 240 * 320, slot=2)
    comm.synchronize()
    for f in range(3):
        np.testing.assert_array_equal(frames[f].cpu().numpy(), want[f])
    comm.close()
    ctx.close()
