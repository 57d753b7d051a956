"""Parity of the RESIDENT single-frame apply (bk_apply_resident_begin / submit / wait / end: one kernel stays on the device with the
block map in its registers, a frame is a command written to pinned host memory) against the CPU oracle's render_lensmap restatement
(fisheye.c:2406-2424, one frame per F_RenderView call: fisheye.c:803), through the C ABI.  Byte-exact, background untouched."""
import ctypes
import json
import os
import time

import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = {(r["globe"], r["lens"], r["zoom"], r["W"], r["H"]): r
        for r in json.load(open(os.path.join(HERE, "golden", "lensmaps.json")))["lensmaps"]}


@pytest.fixture(scope="module")
def bk():
    import blinky_amd
    return blinky_amd


@pytest.fixture(scope="module")
def hip():
    h = ctypes.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return h


def make_ctx(bk, lm, nframes=1, rows=None):
    ctx = bk.Context()
    ctx.set_frames(nframes)
    ctx.resize(lm.W, lm.H)
    if rows:
        ctx.set_rows(*rows)
    return ctx


def background(H, pitch, extra=7):
    return (np.arange((H + extra) * pitch, dtype=np.uint32) * 7 % 251).astype(np.uint8).reshape(H + extra, pitch)


CONFIGS = [
    ("cube", "panini", None, 640, 480),
    ("cube", "hammer", None, 960, 540),                 # 30 % unmapped
    ("cube", "quincuncial", None, 640, 480),
    ("trism", "panini", None, 960, 540),
    ("cube", "panini", "f_fov 120", 322, 203),          # W % 4 != 0
    ("cube", "stereographic", "f_vfov 90", 300, 500),   # portrait, one NULL pixel in the middle
    ("cube", "eckert5", None, 640, 480),                # forward-built map (ragged mapped region)
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"{c[0]}-{c[1]}-{c[3]}x{c[4]}")
@pytest.mark.parametrize("rubix", [False, True])
def test_resident_frames_match_oracle(bk, hip, cfg, rubix):
    """three globes, three frames submitted back to back into buffers with pitch > W and an origin; then globe 0 is REWRITTEN behind the
    running kernel's back (a DMA, no kernel boundary) and warped again: the kernel must not serve stale texels from a cache"""
    import torch
    lm = O.lensmap(*cfg)
    W, H = lm.W, lm.H
    F = 3
    globes = [O.lcg_globe(lm.ps, 6, 3 + f) for f in range(F)]
    pal = O.palmap(O.synthetic_basepal())
    ctx = make_ctx(bk, lm, nframes=F)
    for f in range(F):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    pitch, x0, y0 = W + 24, 5, 3
    bg = background(H, pitch)
    outs = [torch.from_numpy(bg.copy()).cuda() for _ in range(F + 1)]
    nbytes = 6 * ctx.globe_pitch() * ctx.globe_rows()
    raw1 = np.empty(nbytes, np.uint8)
    assert hip.hipMemcpy(raw1.ctypes.data, ctx.globe_device_ptr(1), nbytes, 2) == 0
    torch.cuda.synchronize()
    ctx.resident_begin(rubix, pal, idle_ms=2000)
    info = ctx.resident_info()
    assert info["running"] and info["workgroups"] > 1 and info["launches"] == 1
    tickets = [ctx.resident_submit(outs[f].data_ptr(), pitch, frame=f, x0=x0, y0=y0) for f in range(F)]
    assert tickets == [1, 2, 3]
    us = [ctx.resident_wait(t) for t in reversed(tickets)]
    assert all(0 < u < 1e6 for u in us)
    # globe 0 <- globe 1's bytes by DMA while the kernel stays where it is
    assert hip.hipMemcpy(ctx.globe_device_ptr(0), raw1.ctypes.data, nbytes, 1) == 0
    t = ctx.resident_submit(outs[F].data_ptr(), pitch, frame=0, x0=x0, y0=y0)
    ctx.resident_wait(t)
    assert ctx.resident_info()["launches"] == 1, "the kernel was meant to stay on the device through all of this"
    # ... and bk_upload_plate / _async while the session runs: re-tiled on the host and moved by DMA, the kernel stays (the calls
    # used to end the session, and every frame after an upload paid a relaunch)
    extra = torch.from_numpy(bg.copy()).cuda()
    for p in range(6):
        (ctx.upload_plate if p % 2 else ctx.upload_plate_async)(0, p, globes[2][p])
    ctx.synchronize()
    ctx.resident_wait(ctx.resident_submit(extra.data_ptr(), pitch, frame=0, x0=x0, y0=y0))
    assert ctx.resident_info()["launches"] == 1 and ctx.resident_info()["running"]
    ctx.resident_end()
    assert not ctx.resident_info()["running"]
    for f in range(F + 1):
        want = O.apply(lm.offsets, lm.tints, W, H, globes[f if f < F else 1], bg.copy(), pitch, x0, y0, rubix, pal)
        np.testing.assert_array_equal(outs[f].cpu().numpy(), want, err_msg=f"frame {f}")
    np.testing.assert_array_equal(extra.cpu().numpy(), O.apply(lm.offsets, lm.tints, W, H, globes[2], bg.copy(), pitch, x0, y0, rubix, pal), err_msg="after the uploads")
    ctx.close()


@pytest.mark.parametrize("kind", ["random", "rows", "affine", "sparse"])
def test_resident_arbitrary_tables(bk, kind):
    """tables no lens would make - scrambled offsets (blocks without a chunk list: the direct-gather path), runs, 90 % NULL - and a
    stripe of the rows"""
    import torch
    W, H = 517, 260
    ps = H
    rng = np.random.default_rng(11)
    n = 6 * ps * ps
    if kind == "random":
        off = rng.integers(0, n, W * H).astype(np.uint32)
    elif kind == "rows":
        off = ((np.arange(W * H, dtype=np.int64) * 3 + 1000) % n).astype(np.uint32)
    elif kind == "affine":
        yy, xx = np.mgrid[0:H, 0:W]
        off = (((yy * 7 // 8) % ps) * ps + (xx * 3 // 7) % ps + 2 * ps * ps).astype(np.uint32).ravel()
    else:
        off = rng.integers(0, n, W * H).astype(np.uint32)
        off[rng.random(W * H) < 0.9] = O.NULL
    tints = rng.integers(0, 6, W * H).astype(np.uint8)
    tints[rng.random(W * H) < 0.5] = 255
    globe = O.lcg_globe(ps, 6, 1)
    pal = O.palmap(O.synthetic_basepal())

    class LM:
        pass
    lm = LM()
    lm.W, lm.H, lm.ps = W, H, ps
    for rows, rubix in (((0, H), False), ((64, 201), True)):
        ctx = make_ctx(bk, lm, rows=rows)
        for p in range(6):
            ctx.upload_plate(0, p, globe[p])
        r0, r1 = rows
        ctx.set_lensmap(off.reshape(H, W)[r0:r1].copy(), tints.reshape(H, W)[r0:r1].copy())
        bg = background(H, W, 0)
        out = torch.from_numpy(bg.copy()).cuda()
        torch.cuda.synchronize()
        ctx.resident_begin(rubix, pal, idle_ms=2000)
        ctx.resident_wait(ctx.resident_submit(out.data_ptr(), W))
        ctx.resident_end()
        o2, t2 = off.copy().reshape(H, W), tints.copy().reshape(H, W)
        o2[:r0] = O.NULL
        o2[r1:] = O.NULL
        want = O.apply(o2.ravel(), t2.ravel(), W, H, globe, bg.copy(), W, 0, 0, rubix, pal)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        ctx.close()


def test_resident_kernel_leaves_when_idle_and_comes_back(bk):
    """nobody can hang the device: without a submission the kernel exits after idle_ms; the next submit starts it again; any
    other device entry point of the context ends the session first"""
    import torch
    lm = O.lensmap("cube", "panini", None, 640, 480)
    globe = O.lcg_globe(lm.ps, 6, 0)
    ctx = make_ctx(bk, lm)
    for p in range(6):
        ctx.upload_plate(0, p, globe[p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    want = O.apply(lm.offsets, lm.tints, lm.W, lm.H, globe, np.zeros((lm.H, lm.W), np.uint8))
    outs = [torch.zeros((lm.H, lm.W), dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=10)
    ctx.resident_wait(ctx.resident_submit(outs[0].data_ptr(), lm.W))
    time.sleep(0.2)
    assert not ctx.resident_info()["running"]
    torch.cuda.synchronize()                          # (returns: nothing of ours is on the device any more)
    ctx.resident_wait(ctx.resident_submit(outs[1].data_ptr(), lm.W))
    assert ctx.resident_info()["launches"] == 2
    # another entry point of the context: the session ends by itself, and a submit after it brings the kernel back
    got = ctx.apply(np.zeros((lm.H, lm.W), np.uint8))
    np.testing.assert_array_equal(got, want)
    assert not ctx.resident_info()["running"]
    ctx.resident_wait(ctx.resident_submit(outs[2].data_ptr(), lm.W))
    assert ctx.resident_info()["launches"] == 3
    ctx.resident_end()
    for o in outs:
        np.testing.assert_array_equal(o.cpu().numpy(), want)
    with pytest.raises(bk.BlinkyError):
        ctx.resident_wait(10 ** 9)
    ctx.close()


def test_resident_submit_without_a_lensmap_is_an_error_not_a_launch(bk):
    """a session outlives the calls that end the kernel: after bk_set_rows / bk_resize (the tables are freed, nothing is built) a submit
    must answer BK_E_STATE like bk_apply_device does - not compile a block map from freed tables; once there is a lensmap again the
    same session resumes transparently (bk_api.cpp:539)"""
    import torch
    lm = O.lensmap("cube", "panini", None, 640, 480)
    W, H = lm.W, lm.H
    globe = O.lcg_globe(lm.ps, 6, 0)
    ctx = make_ctx(bk, lm)
    for p in range(6):
        ctx.upload_plate(0, p, globe[p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=500)
    ctx.resident_wait(ctx.resident_submit(out.data_ptr(), W))
    ctx.set_rows(120, 360)                            # ends the kernel, frees the tables: no lensmap for these rows yet
    with pytest.raises(bk.BlinkyError):
        ctx.resident_submit(out.data_ptr(), W)
    with pytest.raises(bk.BlinkyError):
        ctx.resident_submit_batch(out.data_ptr(), W, H * W, frame0=0, nframes=2)
    assert not ctx.resident_info()["running"]
    ctx.set_lensmap(lm.offsets.reshape(H, W)[120:360].copy(), lm.tints.reshape(H, W)[120:360].copy())
    out.zero_()
    torch.cuda.synchronize()
    ctx.resident_wait(ctx.resident_submit(out.data_ptr(), W))      # the session comes back by itself
    ctx.resident_end()
    want = np.zeros((H, W), np.uint8)
    want[120:360] = O.apply(lm.offsets, lm.tints, W, H, globe, np.zeros((H, W), np.uint8))[120:360]
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    ctx.resize(W + 64, H)                             # a new size: nothing built for it
    with pytest.raises(bk.BlinkyError):
        ctx.resident_submit(out.data_ptr(), W + 64)
    ctx.close()


def test_resident_frame_is_in_memory_when_wait_returns_while_later_frames_stream(bk, hip):
    """bk_apply_resident_wait(n) promises frame n COMPLETE IN MEMORY while frames n+1.. are still being warped (the flag is raised two
    staging barriers after the frame's last write-through store, without a drain): a DMA reads every frame back the moment its wait
    returns, 48 frames streaming behind one another (bk_apply_resident.inc:407)"""
    import torch
    lm = O.lensmap("cube", "hammer", None, 960, 540)
    W, H = lm.W, lm.H
    F = 4
    globes = [O.lcg_globe(lm.ps, 6, 10 + f) for f in range(F)]
    ctx = make_ctx(bk, lm, nframes=F)
    for f in range(F):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    want = [O.apply(lm.offsets, lm.tints, W, H, globes[f], np.full((H, W), 9, np.uint8)) for f in range(F)]
    N = 48
    outs = [torch.full((H, W), 9, dtype=torch.uint8, device="cuda") for _ in range(N)]
    torch.cuda.synchronize()
    host = np.empty((H, W), np.uint8)
    ctx.resident_begin(idle_ms=2000)
    for rnd in range(3):
        tickets = []
        for i in range(24):                          # 24 in flight, then top up as the reads go on
            tickets.append(ctx.resident_submit(outs[i].data_ptr(), W, frame=(i + rnd) % F))
        for i in range(N):
            ctx.resident_wait(tickets[i])
            assert hip.hipMemcpy(host.ctypes.data, outs[i].data_ptr(), W * H, 2) == 0      # a DMA, not a kernel: it does not wait for the resident kernel
            np.testing.assert_array_equal(host, want[(i + rnd) % F], err_msg=f"round {rnd} frame {i} read right after its wait")
            if i + 24 < N:
                tickets.append(ctx.resident_submit(outs[i + 24].data_ptr(), W, frame=(i + 24 + rnd) % F))
        # (the buffers are overwritten by the next round's frames: every mapped pixel changes with the globe, the background stays 9)
    assert ctx.resident_info()["launches"] == 1
    ctx.resident_end()
    ctx.close()


def test_resident_pipelined_submissions(bk):
    """100 frames over a ring of 8 globes into 4 rotating buffers, up to 32 in flight: every frame's bytes, in order"""
    import torch
    lm = O.lensmap("cube", "hammer", None, 960, 540)
    F = 8
    globes = [O.lcg_globe(lm.ps, 6, f) for f in range(F)]
    ctx = make_ctx(bk, lm, nframes=F)
    for f in range(F):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    want = [O.apply(lm.offsets, lm.tints, lm.W, lm.H, globes[f], np.zeros((lm.H, lm.W), np.uint8)) for f in range(F)]
    outs = [torch.zeros((lm.H, lm.W), dtype=torch.uint8, device="cuda") for _ in range(100)]
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=2000)
    tickets = [ctx.resident_submit(outs[i].data_ptr(), lm.W, frame=(i * 3) % F) for i in range(100)]
    assert tickets == list(range(1, 101))
    ctx.resident_wait(tickets[-1])
    ctx.resident_end()
    for i in range(100):
        np.testing.assert_array_equal(outs[i].cpu().numpy(), want[(i * 3) % F], err_msg=f"submission {i}")
    ctx.close()


RES_4K = [("cube", "panini", None, 3840, 2160), ("cube", "hammer", None, 3840, 2160),
          ("cube", "quincuncial", None, 3840, 2160),       # C3 (BASELINE.json configs[2])
          ("trism", "panini", None, 3840, 2160)]           # C4's map (configs[3]) on one GPU


@pytest.mark.parametrize("shape", [0, 1], ids=["shape-chosen", "128x8-blocks"])
@pytest.mark.parametrize("key", RES_4K, ids=lambda k: f"{k[0]}-{k[1]}")
def test_resident_4k_frame_hash_equals_reference_golden(bk, key, shape):
    """BASELINE.json's full size: the frame hash recorded from the unmodified reference, through the resident kernel, on the GPU-built
    lensmap; prints what the session looked like (workgroups, blocks held in registers) and the device time of a frame"""
    import torch
    import scripts as S
    rec = GOLD[key]
    globe, lens, zoom, W, H = key
    ctx = bk.Context()
    ctx.set_frames(2)
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.build()
    for f in range(2):
        for p in range(len(rec["display"])):
            ctx.fill_plate_lcg(f, p, seed_frame=0)
    out = torch.zeros((2, H, W), dtype=torch.uint8, device="cuda")
    if shape:
        ctx.set_tile_shape(shape)             # 128x8 blocks: 8160 of them - far more than a workgroup's share in registers
    ctx.synchronize()
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=500)
    info = ctx.resident_info()
    assert info["block_h"] == (8 if shape else info["block_h"])
    us = [ctx.resident_wait(ctx.resident_submit(out[i % 2].data_ptr(), W, frame=i % 2)) for i in range(20)]
    ctx.resident_end()
    print(f"\nresident {lens} {W}x{H}: {info}, device us per frame (one at a time): min {min(us):.2f} median {sorted(us)[len(us) // 2]:.2f}")
    for i in range(2):
        assert O.fnv(out[i].cpu().numpy()) == rec["fnv_frame"]
    ctx.close()


@pytest.mark.parametrize("reserve", [0, 1], ids=["whole-chip", "a-place-per-cu-reserved"])
def test_drop_in_calls_on_a_session_that_fills_the_chip(bk, reserve):
    """bk_set_resident_apply at 3840x2160 cube/panini: 2040 blocks on 2048 workgroups - not a free place on the chip.  bk_upload_plate +
    bk_apply into a pitched host frame must still run on ONE launch of the kernel, every call in milliseconds: the plates travel by DMA and
    so does the frame (through the pinned frame; a 2-D copy into a pageable buffer is a shader copy for small frames and waited for the
    kernel's idle exit - 200 ms a frame - in round 5's first version).  With a place per CU reserved (bk_set_resident_share) the plates are
    re-tiled by a kernel and the frame is copied straight out, BESIDE the resident kernel.  Frame hash = the unmodified reference's; also
    with rubix on."""
    import scripts as S
    key = ("cube", "panini", None, 3840, 2160)
    rec = GOLD[key]
    W, H = 3840, 2160
    ctx = bk.Context()
    S.configure(ctx, *key[:3], (W, H))
    ctx.build()
    off, tin = ctx.read_lensmap()
    ctx.set_resident_share(0, 1, reserve)
    ctx.set_resident_apply(True)
    globe = O.lcg_globe(min(W, H), 6, 0)                     # (the plate size of the reference: the frame's smaller side)
    pal = O.palmap(O.synthetic_basepal())
    pitch, x0, y0 = W + 16, 5, 3
    for i in range(6):
        for p in range(6):
            ctx.upload_plate(0, p, globe[p])
        frame = np.full((H + 4, pitch), 9, np.uint8)
        t0 = time.time()
        ctx.apply(frame, pitch=pitch, x0=x0, y0=y0)
        dt = time.time() - t0
        info = ctx.resident_info()
        assert info["running"] and info["launches"] == 1 and (info["workgroups"] == 2048 if reserve == 0 else info["per_cu"] <= 7), info
        assert i == 0 or dt < 0.05, f"bk_apply took {dt * 1e3:.1f} ms beside a resident kernel that holds every place"
        assert O.fnv(np.ascontiguousarray(frame[y0:y0 + H, x0:x0 + W])) == rec["fnv_frame"]
        assert (frame[:y0] == 9).all() and (frame[:, :x0] == 9).all() and (frame[:, x0 + W:] == 9).all()
    want = np.zeros((H, W), np.uint8)
    O.apply(off, tin, W, H, globe, want, W, 0, 0, True, pal)
    for i in range(3):
        frame = np.zeros((H, W), np.uint8)
        t0 = time.time()
        ctx.apply(frame, rubix_on=True, pal=pal)
        dt = time.time() - t0
        info = ctx.resident_info()
        assert info["running"] and info["launches"] == 2, info
        assert i == 0 or dt < 0.05, f"rubix: bk_apply took {dt * 1e3:.1f} ms"
        np.testing.assert_array_equal(frame, want)
    ctx.close()


def test_a_launch_of_the_other_flavour_in_the_middle_of_a_session(bk):
    """a plain session with frames still pending, then bk_apply_device with rubix on (the block map changes flavour: tinted), then the
    session goes on (the map changes back): the pending frames are finished first, every frame is the oracle's, nothing recurses"""
    import torch
    lm = O.lensmap("cube", "panini", None, 640, 480)
    W, H = lm.W, lm.H
    pal = O.palmap(O.synthetic_basepal())
    ctx = make_ctx(bk, lm, nframes=3)
    globes = [O.lcg_globe(lm.ps, 6, 40 + f) for f in range(3)]
    for f in range(3):
        for p in range(6):
            ctx.upload_plate(f, p, globes[f][p])
    ctx.set_lensmap(lm.offsets, lm.tints)
    outs = torch.zeros((8, H, W), dtype=torch.uint8, device="cuda")
    tinted = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ctx.synchronize()
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=2000)
    for i in range(4):
        ctx.resident_submit(outs[i].data_ptr(), W, frame=i % 3)                 # (not waited for)
    ctx.apply_device(tinted.data_ptr(), W, H * W, frame0=1, nframes=1, rubix_on=True, pal=pal)      # ends the session: pending frames first
    torch.cuda.synchronize()
    last = 0
    for i in range(4, 8):
        last = ctx.resident_submit(outs[i].data_ptr(), W, frame=i % 3)          # the session comes back, on a plain map again
    ctx.resident_wait(last)
    ctx.resident_end()
    for i in range(8):
        np.testing.assert_array_equal(outs[i].cpu().numpy(), O.apply(lm.offsets, lm.tints, W, H, globes[i % 3], np.zeros((H, W), np.uint8)), err_msg=f"frame {i}")
    np.testing.assert_array_equal(tinted.cpu().numpy(), O.apply(lm.offsets, lm.tints, W, H, globes[1], np.zeros((H, W), np.uint8), W, 0, 0, True, pal))
    ctx.close()


def test_resident_c5_8k_frames_equal_reference_golden(bk):
    """BASELINE.json configs[4] (7680x4320 cube/hammer) through the resident kernel: frames 0, 1, 5 and 63 of the 64-frame golden
    batch (frame 0 recorded from the unmodified reference, the others from the oracle's gather over the reference's lensmap), submitted
    back to back - the deep form: more blocks per workgroup than any 4K map"""
    import torch
    import scripts as S
    key = ("cube", "hammer", None, 7680, 4320)
    rec = GOLD[key]
    globe, lens, zoom, W, H = key
    picks = [0, 1, 5, 63]
    ctx = bk.Context()
    ctx.set_frames(len(picks))
    S.configure(ctx, globe, lens, zoom, (W, H))
    ctx.build()
    for i, f in enumerate(picks):
        for p in range(6):
            ctx.fill_plate_lcg(i, p, seed_frame=f)
    out = torch.zeros((len(picks), H, W), dtype=torch.uint8, device="cuda")
    ctx.synchronize()
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=1000)
    info = ctx.resident_info()
    last = ctx.resident_submit_batch(out.data_ptr(), W, H * W, frame0=0, nframes=len(picks))
    ctx.resident_wait(last)
    # ... and once more one at a time into the same buffers (a drained pipeline starts differently from a streaming one)
    out2 = torch.zeros_like(out)
    torch.cuda.synchronize()                     # (the fill is a kernel on torch's stream, running BESIDE the resident kernel: a frame submitted before it
    #                                               has finished is overwritten by its zeros - seen once the resident frame got faster than the fill, r6)
    for i in range(len(picks)):
        ctx.resident_wait(ctx.resident_submit(out2[i].data_ptr(), W, frame=i))
    ctx.resident_end()
    print(f"\nresident C5 {W}x{H}: {info}")
    for i, f in enumerate(picks):
        assert O.fnv(out[i].cpu().numpy()) == rec["fnv_frames"][f], f"frame {f} (pipelined)"
        assert O.fnv(out2[i].cpu().numpy()) == rec["fnv_frames"][f], f"frame {f} (one at a time)"
    ctx.close()


def test_resident_4k_rubix_and_stripe_equal_oracle(bk):
    """the headline map (3840x2160 cube/panini) with the rubix tints on (fisheye.c:2416-2419), whole and as the 270-row stripe rank 3 of 8
    owns (bk_set_rows): frames byte-equal to the oracle's render_lensmap over the oracle's own 4K lensmap"""
    import torch
    import scripts as S
    key = ("cube", "panini", None, 3840, 2160)
    lm = O.lensmap(*key)
    W, H = lm.W, lm.H
    globe = O.lcg_globe(lm.ps, 6, 2)
    pal = O.palmap(O.synthetic_basepal())
    want_rubix = O.apply(lm.offsets, lm.tints, W, H, globe, np.zeros((H, W), np.uint8), W, 0, 0, True, pal)
    want_plain = O.apply(lm.offsets, lm.tints, W, H, globe, np.zeros((H, W), np.uint8))
    assert O.fnv(lm.offsets) == GOLD[key]["fnv_offsets"]
    for rows, rubix in (((0, H), True), ((810, 1080), False), ((810, 1080), True)):
        ctx = bk.Context()
        S.configure(ctx, *key[:3], (W, H))
        ctx.set_rows(*rows)
        ctx.build()
        for p in range(6):
            ctx.upload_plate(0, p, globe[p])
        out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.resident_begin(rubix, pal, idle_ms=500)
        info = ctx.resident_info()
        for _ in range(3):
            ctx.resident_wait(ctx.resident_submit(out.data_ptr(), W))
        ctx.resident_end()
        want = np.zeros((H, W), np.uint8)
        want[rows[0]:rows[1]] = (want_rubix if rubix else want_plain)[rows[0]:rows[1]]
        np.testing.assert_array_equal(out.cpu().numpy(), want, err_msg=f"rows {rows} rubix {rubix} {info}")
        ctx.close()


# ---- the drop-in calls through the resident kernel (bk_set_resident_apply) ------------------------------------------------------

@pytest.mark.parametrize("cfg", [("cube", "panini", None, 640, 480), ("cube", "hammer", None, 960, 540), ("cube", "stereographic", "f_fov 180", 322, 203)],
                         ids=lambda c: f"{c[1]}-{c[3]}x{c[4]}")
@pytest.mark.parametrize("reserve", [0, 1], ids=["plates-by-dma", "plates-by-kernel"])
def test_fresh_plates_every_frame_without_ending_the_session(bk, cfg, reserve):
    """what F_RenderView does (fisheye.c:764-803): every frame six freshly rendered plates (bk_upload_plate_async: re-tiled on the host,
    one DMA each - no kernel) and one bk_apply into a host frame with pitch and origin; 50 frames on ONE launch of the resident kernel,
    every frame the oracle's, the background of unmapped pixels untouched; then rubix is switched on (a new session), then off again.
    reserve = 1: a place per CU is left free and the plates are re-tiled by a KERNEL on whatever XCD takes it, beside the
    running resident kernel.  The globes here fit an XCD's L2 several times over and the SAME slot is rewritten every frame: that the
    workers see the new texels rests on their chunk loads' `nt` policy (no line of a globe is kept from frame to frame) - stated in
    DESIGN.md 3.2 as a dependency on the hardware; this is its permanent regression test, for both ways a slot gets rewritten."""
    lm = O.lensmap(*cfg)
    W, H = lm.W, lm.H
    pal = O.palmap(O.synthetic_basepal())
    ctx = make_ctx(bk, lm)
    ctx.set_lensmap(lm.offsets, lm.tints)
    ctx.set_resident_share(0, 1, reserve)
    ctx.set_resident_apply(True)
    pitch, x0, y0 = W + 8, 3, 2
    for i in range(50):
        globe = O.lcg_globe(lm.ps, 6, 100 + i)
        for p in range(6):
            (ctx.upload_plate_async if i % 2 else ctx.upload_plate)(0, p, globe[p])
        bg = background(H, pitch, 4)
        got = ctx.apply(bg.copy(), pitch=pitch, x0=x0, y0=y0)
        want = O.apply(lm.offsets, lm.tints, W, H, globe, bg.copy(), pitch, x0, y0, False, pal)
        np.testing.assert_array_equal(got, want, err_msg=f"frame {i}")
        info = ctx.resident_info()
        assert info["running"] and info["launches"] == 1, (i, info)
    globe = O.lcg_globe(lm.ps, 6, 7)
    for p in range(6):
        ctx.upload_plate_async(0, p, globe[p])
    for rubix, launches in ((True, 2), (True, 2), (False, 3)):
        bg = background(H, pitch, 4)
        got = ctx.apply(bg.copy(), pitch=pitch, x0=x0, y0=y0, rubix_on=rubix, pal=pal)
        np.testing.assert_array_equal(got, O.apply(lm.offsets, lm.tints, W, H, globe, bg.copy(), pitch, x0, y0, rubix, pal))
        assert ctx.resident_info()["launches"] == launches
    ctx.close()


def test_a_reserved_place_per_cu_lets_other_kernels_run_beside_the_resident_kernel(bk):
    """bk_set_resident_share(0, 1, 1): one workgroup place of every CU stays free - ANOTHER context's kernels (a build, synthetic plates, a
    batch apply on the null stream) run to completion while a chip-filling session (3840x2160 cube/panini) is up; without the reserve
    they queue until the resident kernel leaves (idle_ms is 20 s here).  The session is still its first launch afterwards and its
    frames are the reference's."""
    import torch
    import scripts as S
    key = ("cube", "panini", None, 3840, 2160)
    a = bk.Context()
    S.configure(a, *key[:3], (3840, 2160))
    a.build()
    for p in range(6):
        a.fill_plate_lcg(0, p, seed_frame=0)
    a.set_resident_share(0, 1, 1)
    out = torch.zeros((2160, 3840), dtype=torch.uint8, device="cuda")
    # the other context, built and warmed up BEFORE the session: hipFree is a device-wide synchronise in HIP, and a build or a first
    # apply frees scratch buffers - such calls do wait for a resident kernel whatever it leaves free (include/blinky_hip.h says so)
    b = bk.Context()
    S.configure(b, "cube", "hammer", None, (640, 480))
    b.build()
    b.fill_plate_lcg(0, 0, seed_frame=1)
    b.apply(np.zeros((480, 640), np.uint8))
    lmb = O.lensmap("cube", "hammer", None, 640, 480)
    t = torch.arange(1 << 20, device="cuda", dtype=torch.int32)
    a.synchronize()
    torch.cuda.synchronize()
    a.resident_begin(idle_ms=20000)
    a.resident_wait(a.resident_submit(out.data_ptr(), 3840))
    full = a.resident_info()
    t0 = time.time()
    for p in range(6):
        b.fill_plate_lcg(0, p, seed_frame=3)                         # kernels of another context, on the null stream
    got_b = b.apply(np.zeros((480, 640), np.uint8))                  # its warp + the copy back
    s_torch = int((t * 2 + 1).sum().item())                          # another library's kernels (PyTorch, null stream)
    dt = time.time() - t0
    np.testing.assert_array_equal(got_b, O.apply(lmb.offsets, lmb.tints, 640, 480, O.lcg_globe(lmb.ps, 6, 3), np.zeros((480, 640), np.uint8)))
    assert s_torch == (1 << 20) * ((1 << 20) - 1) + (1 << 20)
    assert dt < 5.0, f"the other context's kernels took {dt:.1f} s: they waited for the resident kernel ({full})"
    out.zero_()
    a.resident_wait(a.resident_submit(out.data_ptr(), 3840))
    info = a.resident_info()
    assert info["running"] and info["launches"] == 1, info
    a.resident_end()
    assert O.fnv(out.cpu().numpy()) == GOLD[key]["fnv_frame"]
    b.close()
    a.close()


def test_three_stripe_contexts_each_with_a_resident_kernel_on_one_gpu(bk):
    """bk_multi on a device named three times, bk_multi_set_resident_apply: the three stripe contexts split every XCD's CUs between them
    and each keeps its own resident kernel; fresh plates and a host frame per call, 12 frames, every one the oracle's"""
    import scripts as S
    lm = O.lensmap("cube", "hammer", None, 960, 540)
    W, H = lm.W, lm.H
    m = bk.Multi([0, 0, 0])
    m.load_globe(S.script("globes", "cube"), "cube")
    m.load_lens(S.script("lenses", "hammer"), "hammer")
    m.set_zoom(*S.zoom_args(m.ctx(0).lens_info().onload.decode()))
    m.resize(W, H)
    m.build()
    m.set_resident_apply(True)
    settled = None
    for i in range(16):
        globe = O.lcg_globe(lm.ps, 6, 40 + i)
        for p in range(6):
            m.upload_plate(0, p, globe[p])
        got = m.apply(np.zeros((H, W), np.uint8))
        np.testing.assert_array_equal(got, O.apply(lm.offsets, lm.tints, W, H, globe, np.zeros((H, W), np.uint8)), err_msg=f"frame {i}")
        if i == 3:
            # (while the sessions START they do interrupt one another: beginning one allocates and frees - hipFree is a device-wide
            #  synchronise - so an earlier one may have idled out and come back; from here on nothing is allocated any more)
            settled = [m.ctx(k).resident_info()["launches"] for k in range(3)]
    infos = [m.ctx(k).resident_info() for k in range(3)]
    assert all(x["running"] for x in infos) and [x["launches"] for x in infos] == settled, (settled, infos)
    m.close()


def test_a_second_session_takes_the_form_the_first_took(bk):
    """(r6) a map whose blocks the first session had to grow (4K gumby: 128x8 -> 128x16, three blocks per workgroup in registers and three
    fetched per frame) used to be grown AGAIN by the next session of the same lensmap - 128x32, eight blocks per workgroup on 256
    workgroups, 35 us per frame instead of 15.  The form is a property of the lensmap, not of how many sessions it has seen."""
    import torch
    import scripts as S
    W, H = 3840, 2160
    ctx = bk.Context()
    ctx.set_frames(2)
    S.configure(ctx, "cube", "gumby", None, (W, H))
    ctx.build()
    for f in range(2):
        for p in range(6):
            ctx.fill_plate_lcg(f, p, seed_frame=f)
    ref = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    ctx.apply_device(ref.data_ptr(), W, H * W, frame0=1, nframes=1)
    torch.cuda.synchronize()
    forms = []
    for session in range(3):
        out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.resident_begin(idle_ms=500)
        info = ctx.resident_info()
        ctx.resident_wait(ctx.resident_submit(out.data_ptr(), W, frame=1))
        ctx.resident_end()
        forms.append((info["workgroups"], info["blocks_in_registers"], info["block_h"]))
        assert torch.equal(out, ref), f"session {session}"
    assert forms[0] == forms[1] == forms[2], forms
    ctx.close()
