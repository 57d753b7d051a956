#!/usr/bin/env python3
"""Will this lens / globe script run with libblinkyhip?  Loads it on a context without a device (no GPU needed), reports what the
host layer would see (map type, zoom limits, onload command), whether its per-pixel callbacks translate to GPU code - or which
construct does not (such a lens takes the HOST PATH: the library's interpreter evaluates its callbacks) - whether hiprtc compiles the result for gfx950, and whether the callbacks carry state from
pixel to pixel (bk_lens_carries_state: such a lens is built by one sequential scan on the host, as the reference builds every lens).

usage: tools/check_lens.py <lens.lua> [<globe.lua>] [--no-compile] [--preview out.png]

--preview (inverse-map lenses): builds the lensmap at 640x400 by running the generated code on the HOST (tests/hostemu: the very
translation unit the GPU gets, compiled by g++) and writes a picture of it - one colour per globe plate, shaded by the texel's position
inside the plate, black where the lens maps nothing - so the shape of a lens can be judged without a GPU.
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write_preview(ctx, path):
    import struct
    import zlib
    import numpy as np
    from hostemu import emu
    ctx.resize(640, 400)
    off, tin, flagged, err = emu.build_inverse(ctx)
    W, H, ps, r0, r1 = ctx.size()
    ref = emu.device_to_reference_layout(off, ps)                       # plate * ps * ps + py * ps + px, 0xFFFFFFFF = unmapped
    mapped = ref != 0xFFFFFFFF
    plate = np.where(mapped, ref // (ps * ps), 0)
    py = np.where(mapped, (ref % (ps * ps)) // ps, 0).astype(np.float64) / ps
    px = np.where(mapped, ref % ps, 0).astype(np.float64) / ps
    base = np.array([[230, 80, 80], [80, 200, 90], [90, 120, 240], [230, 200, 70], [200, 90, 220], [80, 210, 220]], np.float64)
    shade = 0.45 + 0.55 * (0.5 * px + 0.5 * py) + 0.12 * (((px * 8).astype(int) + (py * 8).astype(int)) & 1)
    rgb = np.clip(base[plate] * shade[:, None], 0, 255) * mapped[:, None]
    img = rgb.astype(np.uint8).reshape(r1 - r0, W, 3)
    raw = b"".join(b"\0" + img[y].tobytes() for y in range(img.shape[0]))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, img.shape[0], 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) +
                chunk(b"IEND", b""))
    print("  %d of %d pixels mapped, %d flagged for the host (libm-dependent), plates used: %s" % (
        int(mapped.sum()), mapped.size, len(flagged), sorted(set(plate[mapped].tolist()))))


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--preview"]
    if not args:
        print(__doc__)
        return 2
    import blinky_amd as bk
    import scripts as S
    ctx = bk.Context(bk.ffi.DEVICE_NONE)
    try:
        if len(args) > 1:
            ctx.load_globe(open(args[1]).read(), os.path.basename(args[1]))
        else:
            ctx.load_globe(S.script("globes", "cube"), "cube.lua")
        print("globe: %d plates" % len(ctx.globe()))
        ctx.load_lens(open(args[0]).read(), os.path.basename(args[0]))
    except bk.BlinkyError as e:
        print("does not load:", e)
        return 1
    if ctx.console():
        print("script output:\n" + ctx.console().rstrip())
    info = ctx.lens_info()
    kind = {bk.ffi.MAP_INVERSE: "inverse (lens_inverse per screen pixel)", bk.ffi.MAP_FORWARD: "forward (lens_forward per globe texel)"}.get(info.map_type, "none")
    print("map: %s; max_fov %d, max_vfov %d, lens size %g x %g, onload %r" % (kind, info.max_fov, info.max_vfov, info.lens_width, info.lens_height,
                                                                             info.onload.decode()))
    if info.map_type not in (bk.ffi.MAP_INVERSE, bk.ffi.MAP_FORWARD):
        print("no lens_inverse / lens_forward: nothing to build")
        return 1
    onload = info.onload.decode().split()
    if onload and onload[0] in S.ZOOM_CMD:
        ctx.set_zoom(S.ZOOM_CMD[onload[0]], int(float(onload[1])) if len(onload) > 1 else 0)
    else:
        ctx.set_zoom(S.ZOOM_CMD["f_fov"], 90)
    ctx.resize(640, 480)
    try:
        scale = ctx.calc_zoom()
        print("zoom at 640x480: scale %r" % (scale,))
    except bk.BlinkyError as e:
        print("zoom cannot be computed:", e)
    try:
        src = ctx.kernel_source(compile="--no-compile" not in sys.argv)
        print("callbacks translate to GPU code (%d lines)%s" % (src.count("\n"), "" if "--no-compile" in sys.argv else " and compile for gfx950"))
        loops = src.count("= bk_contract(")
        if loops:
            print("  %d assignment(s) inside self-correcting loops (one carried variable, smooth arithmetic) get the contraction-aware error bound" % loops)
        if src.count("bk_f_sincos("):
            print("  sin / cos pairs of one operand share an argument reduction in %d place(s)" % src.count("bk_f_sincos("))
    except bk.BlinkyError as e:
        if "GPU callback" not in str(e):
            print("callbacks do NOT translate:", e)
            return 1
        # (r6) a construct the emitter declines is not a refusal of the script: bk_build evaluates such callbacks with the library's own
        # interpreter on the host (bk_last_build_path says so after a build)
        try:
            carries, which = ctx.lens_carries_state()
        except bk.BlinkyError:
            carries, which = True, "(callbacks not analysable)"
        print("host path: the callbacks use a construct that does not become GPU code -", e)
        print("  bk_build evaluates them with the script interpreter on the host: %s - seconds instead of milliseconds at 4K, the reference's result" % (
            "ONE scan in the reference's order (state carried through '%s')" % which if carries else "on the worker pool (no state carried from call to call)"))
        return 0
    if "--preview" in sys.argv:
        out_path = sys.argv[sys.argv.index("--preview") + 1]
        if info.map_type != bk.ffi.MAP_INVERSE:
            print("--preview needs an inverse-map lens")
        else:
            write_preview(ctx, out_path)
            print("lensmap preview written to", out_path)
    carries, which = ctx.lens_carries_state()
    if carries:
        print("callbacks carry state from pixel to pixel through '%s': such a lens is built as ONE scan on the host, in the reference's"
              " order (bk_set_sequential_build mode 1, the default) - seconds instead of milliseconds at 4K; mode 0 builds it on the GPU,"
              " where every pixel starts from the value after load" % which)
    else:
        print("callbacks carry no state from pixel to pixel (scratch globals and keyed caches do not count): built on the GPU")
    return 0


if __name__ == "__main__":
    sys.exit(main())
