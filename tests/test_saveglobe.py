"""f_saveglobe (cmd_saveglobe + save_globe + WritePCXplate, fisheye.c:1120-1136, 1396-1484): the PCX plate files.
CPU: the oracle restatement against the goldens taken from the unmodified reference (tests/golden/saveglobe.json)
and, where oracle/_ref exists, against the reference itself.  GPU: the device-computed plate image
(bk_save_plate, incl. a globe with a globe_plate script) and the files the C host layer writes."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_ffi as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "saveglobe.json")))["files"]
HAND_C_GLOBES = ("cube", "trism")       # globes the oracle has hand transliterations of


@pytest.mark.parametrize("rec", [r for r in GOLD if r["globe"] in HAND_C_GLOBES],
                         ids=lambda r: f'{r["globe"]}-{r["W"]}x{r["H"]}-m{r["with_margins"]}-p{r["plate"]}')
def test_oracle_pcx_matches_reference_golden(rec):
    ps = min(rec["W"], rec["H"])
    plates = O.lcg_globe(ps, 6, rec["frame"])
    data = O.pcx_plate(rec["globe"], ps, rec["plate"], rec["with_margins"], plates[rec["plate"]], O.synthetic_basepal())
    assert data.size == rec["length"] and O.fnv(data) == rec["fnv"]
    assert data[0] == 0x0A and data[1] == 5 and data[3] == 8 and data[-769] == 0x0C      # PCX id, 256 colours, palette marker
    assert int(data[8]) + 256 * int(data[9]) == ps - 1


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref needs /root/reference (build container only)")
def test_oracle_pcx_equals_unmodified_reference():
    for globe, W, H, frame in (("cube", 97, 64, 5), ("trism", 120, 131, 0)):
        lm, _ = O.ref_run(globe, "panini", None, W, H, want_frame=False)
        ps = min(W, H)
        plates = O.lcg_globe(ps, 6, frame)
        for wm in (0, 1):
            for plate in range(lm.numplates):
                name, ref = O.ref_saveglobe("g", wm, frame, plate, ps)
                assert name == f"g{plate}.pcx"
                np.testing.assert_array_equal(O.pcx_plate(globe, ps, plate, wm, plates[plate], O.synthetic_basepal()), ref)


def _unpack_pcx(data, ps):
    """decode WritePCXplate's escape scheme: 0xC1 announces one literal byte >= 0xC0"""
    body = data[128:-769]
    out = np.empty(ps * ps, np.uint8)
    i = k = 0
    while k < ps * ps:
        b = body[i]
        if b == 0xC1:
            i += 1
            b = body[i]
        out[k] = b
        k += 1
        i += 1
    assert i == body.size
    return out.reshape(ps, ps)


@pytest.mark.gpu
@pytest.mark.parametrize("globe,W,H,frame", [("cube", 160, 120, 2), ("trism", 200, 150, 1), ("fast", 180, 96, 3)])
def test_device_plate_image_matches_reference(globe, W, H, frame):
    """bk_save_plate == the pixel loop of WritePCXplate: re-packed with the reference's escape scheme the image
    must hash to the golden file (fast.lua decides plate ownership with its globe_plate script)."""
    import blinky_amd
    import scripts as S
    ps = min(W, H)
    ctx = blinky_amd.Context()
    S.configure(ctx, globe, "hammer", None, (W, H))
    nplates = len([r for r in GOLD if r["globe"] == globe and r["W"] == W and r["with_margins"] == 0])
    for p in range(6):
        ctx.fill_plate_lcg(0, p, frame)
    pal = O.synthetic_basepal()
    for wm in (0, 1):
        for plate in range(nplates):
            rec = [r for r in GOLD if (r["globe"], r["W"], r["with_margins"], r["plate"]) == (globe, W, wm, plate)][0]
            img = ctx.save_plate(0, plate, wm)
            body = bytearray()
            for b in img.reshape(-1).tolist():
                if (b & 0xC0) == 0xC0:
                    body.append(0xC1)
                body.append(b)
            hdr = np.zeros(128, np.uint8)
            hdr[0:4] = (0x0A, 5, 1, 8)
            hdr[8], hdr[9], hdr[10], hdr[11] = (ps - 1) & 255, (ps - 1) >> 8, (ps - 1) & 255, (ps - 1) >> 8
            hdr[12], hdr[13], hdr[14], hdr[15] = ps & 255, ps >> 8, ps & 255, ps >> 8
            hdr[65], hdr[66], hdr[67], hdr[68] = 1, ps & 255, ps >> 8, 2
            data = np.concatenate([hdr, np.frombuffer(bytes(body), np.uint8), np.array([0x0C], np.uint8), pal])
            assert data.size == rec["length"] and O.fnv(data) == rec["fnv"], (globe, wm, plate)
            if wm:
                np.testing.assert_array_equal(img, O.lcg_globe(ps, 6, frame)[plate])
    ctx.close()


@pytest.mark.gpu
def test_f_saveglobe_through_the_c_host_layer(tmp_path):
    import blinky_amd  # noqa: F401  (loads torch's HIP runtime before libblinkyhip)
    from test_host_layer import HOSTLIB, build_hostlib, game_dir
    build_hostlib()
    h = C.CDLL(HOSTLIB)
    h.hosttest_console.restype = C.c_char_p
    assert h.hosttest_init(game_dir(tmp_path).encode()) == 1
    W, H, frame = 160, 120, 2
    ps = min(W, H)
    h.hosttest_cmd(b"f_globe cube")
    h.hosttest_cmd(b"f_lens hammer")              # full-sphere lens: all six plates are rendered every frame
    h.hosttest_resize(W, H, 0, 0, 0)
    order = (C.c_int * 6)(0, 1, 2, 3, 4, 5)
    out = np.zeros((h.hosttest_vidheight(), h.hosttest_rowbytes()), np.uint8)
    for wm, cmd in ((0, b"f_saveglobe shot"), (1, b"f_saveglobe shot 1")):
        h.hosttest_clear_files()
        h.hosttest_console_clear()
        h.hosttest_cmd(cmd)
        assert h.hosttest_frame(order, 6, frame, 0, out.ctypes.data_as(C.c_void_p)) == 6
        assert h.hosttest_num_files() == 6
        assert "Wrote shot5.pcx" in h.hosttest_console().decode()
        for plate in range(6):
            rec = [r for r in GOLD if (r["globe"], r["W"], r["with_margins"], r["plate"]) == ("cube", W, wm, plate)][0]
            buf = np.empty(ps * ps * 2 + 1000, np.uint8)
            name = C.create_string_buffer(64)
            n = h.hosttest_file(plate, name, buf.ctypes.data_as(C.c_void_p), buf.size)
            assert name.value.decode() == rec["name"] and n == rec["length"] and O.fnv(buf[:n]) == rec["fnv"]
            img = _unpack_pcx(buf[:n], ps)
            if wm:
                np.testing.assert_array_equal(img, O.lcg_globe(ps, 6, frame)[plate])
        # one frame later nothing more is written (globe.save.should was cleared, fisheye.c:1472)
        h.hosttest_frame(order, 6, frame, 0, out.ctypes.data_as(C.c_void_p))
        assert h.hosttest_num_files() == 6
    h.hosttest_cmd(b"f_saveglobe")
    assert "f_saveglobe <name> [full flag=0]" in h.hosttest_console().decode()
    h.hosttest_shutdown()
