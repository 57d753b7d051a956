#!/usr/bin/env python3
"""ORACLE test infrastructure: a synthetic, self-made Quake game directory for the headless engines of
tests/test_engine_dropin.py.  The real game data (id1/pak0.pak) is not in the reference tree
(.MISSING_LARGE_BLOBS), so everything the engine insists on at start-up and for one playable map is
generated here from deterministic patterns:

  <out>/id1/gfx.wad             every lump the engine asks for by name (names collected from the engine
                                sources' Draw_PicFromWad / W_GetLumpName calls), 8x8 pictures
  <out>/id1/gfx/palette.lmp     pal[i] = (i * 37) mod 256          (SURVEY.md 8(d)'s synthetic palette)
  <out>/id1/gfx/colormap.lmp    64 light levels x 256 colours
  <out>/id1/gfx/conback.lmp, loading.lmp, pause.lmp
  <out>/id1/maps/box.bsp        BSP29: one cubic room, six differently textured and lit walls, the
                                player start in the middle
  <out>/id1/progs.dat           QuakeC program whose every function returns at once (globals and entity
                                fields laid out as NQ/progdefs-id1.h says)
  <out>/id1/progs/player.mdl    the one model the server wants precached for its client slots: a small pyramid
  <out>/id1/quake.rc            "stuffcmds" (the engine then runs the +commands of its command line)
  <out>/id1/gfx/pop.lmp         the engine only opens loose files in sub-directories for a "registered" game, which it
                                recognises by this file holding the 128 shorts of common/common.c's pop[] table
                                (COM_CheckRegistered, common.c:1181-1221): written from that table as the engine source has it
  <out>/lua-scripts/...         the lens and globe scripts of tests/golden/scripts.bundle

usage: mkgame.py <reference engine dir> <out dir>
"""
import os
import re
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))


def lcg_bytes(seed, n):
    out = bytearray(n)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s >> 24
    return bytes(out)


# ---- gfx.wad ---------------------------------------------------------------------------------------

def wad_names(engine):
    names = set()
    for rel in ("NQ/sbar.c", "common/draw.c", "common/screen.c"):
        text = open(os.path.join(engine, rel), encoding="latin-1").read()
        for m in re.finditer(r'(?:Draw_PicFromWad|W_GetLumpName)\s*\((?:[^"()]*,)?\s*(va\s*\()?\s*"([^"]+)"', text):
            name = m.group(2)
            if "%i" in name:
                names.update(name.replace("%i", str(i)) for i in range(10))
            else:
                names.add(name)
    return sorted(names)


def qpic(width, height, seed):
    return struct.pack("<ii", width, height) + lcg_bytes(seed, width * height)


def write_wad(path, engine):
    lumps = []
    for i, name in enumerate(wad_names(engine)):
        if name == "conchars":
            lumps.append((name, 68, lcg_bytes(7, 128 * 128)))             # raw 128x128 character sheet
        elif name == "backtile":
            lumps.append((name, 66, qpic(64, 64, 11)))
        elif name in ("sbar", "ibar", "scorebar"):
            lumps.append((name, 66, qpic(320, 24, 100 + i)))
        else:
            lumps.append((name, 66, qpic(8, 8, 100 + i)))
    body = b""
    table = b""
    pos = 12
    for name, typ, data in lumps:
        table += struct.pack("<iiibbh16s", pos + len(body), len(data), len(data), typ, 0, 0, name.encode())
        body += data
    with open(path, "wb") as f:
        f.write(b"WAD2" + struct.pack("<ii", len(lumps), 12 + len(body)) + body + table)


# ---- progs.dat -------------------------------------------------------------------------------------

def progdefs(engine):
    """slot offsets of globalvars_t and entvars_t, from the header qcc generated"""
    text = open(os.path.join(engine, "NQ", "progdefs-id1.h"), encoding="latin-1").read()
    structs = re.findall(r"typedef struct\s*\{(.*?)\}\s*(\w+);", text, re.S)
    out = {}
    for body, name in structs:
        ofs, fields = 0, {}
        for typ, ident, arr in re.findall(r"(\w+)\s+(\w+)(?:\[(\d+)\])?;", body):
            fields[ident] = ofs
            ofs += (3 if typ == "vec3_t" else 1) * (int(arr) if arr else 1)
        out[name] = (fields, ofs)
    crc = int(re.search(r"#define PROGHEADER_CRC (\d+)", text).group(1))
    return out["globalvars_t"], out["entvars_t"], crc


def write_progs(path, engine):
    (gfields, nglobals), (efields, nfields), crc = progdefs(engine)
    strings = bytearray(b"\0")

    def s(text):
        ofs = len(strings)
        strings.extend(text.encode() + b"\0")
        return ofs

    engine_funcs = ["main", "StartFrame", "PlayerPreThink", "PlayerPostThink", "ClientKill", "ClientConnect",
                    "PutClientInServer", "ClientDisconnect", "SetNewParms", "SetChangeParms"]
    file_name = s("box.qc")
    functions = [struct.pack("<iiiiiii8s", 0, 0, 0, 0, 0, 0, 0, b"")]               # function 0: none
    globals_ = [0] * (nglobals + 2)
    for i, name in enumerate(engine_funcs + ["info_player_start"]):
        # every one of these starts at statement 1, which is OP_DONE
        functions.append(struct.pack("<iiiiiii8s", 1, nglobals + 2, 0, 0, s(name), file_name, 0, b""))
        if name in gfields:
            globals_[gfields[name]] = i + 1
    # worldspawn precaches the one model the server insists on for client slots (SV_CreateBaseline, NQ/sv_main.c:1124):
    #   precache_model("progs/player.mdl");        builtin #20 (common/pr_cmds.c)
    OP_DONE, OP_STORE_S, OP_CALL1, OFS_PARM0 = 0, 33, 52, 4
    g_name, g_builtin = nglobals, nglobals + 1
    globals_[g_name] = s("progs/player.mdl")
    functions.append(struct.pack("<iiiiiii8s", -20, 0, 0, 0, s("precache_model"), file_name, 1, b"\1"))
    globals_[g_builtin] = len(functions) - 1
    functions.append(struct.pack("<iiiiiii8s", 2, nglobals + 2, 0, 0, s("worldspawn"), file_name, 0, b""))
    statements = [struct.pack("<Hhhh", OP_DONE, 0, 0, 0),                            # 0: error slot
                  struct.pack("<Hhhh", OP_DONE, 0, 0, 0),                            # 1: the empty function
                  struct.pack("<Hhhh", OP_STORE_S, g_name, OFS_PARM0, 0),            # 2: worldspawn
                  struct.pack("<Hhhh", OP_CALL1, g_builtin, 0, 0),
                  struct.pack("<Hhhh", OP_DONE, 0, 0, 0)]
    globaldefs = [struct.pack("<HHi", 0, 0, 0)]
    fielddefs = [struct.pack("<HHi", 0, 0, 0)]
    for name, typ in (("classname", 1), ("origin", 3), ("angles", 3), ("model", 1), ("spawnflags", 2)):
        fielddefs.append(struct.pack("<HHi", typ, efields[name], s(name)))
    nglobals += 2

    sections = [b"".join(statements), b"".join(globaldefs), b"".join(fielddefs), b"".join(functions), bytes(strings) + b"\0" * 8,
                struct.pack("<%di" % nglobals, *globals_)]
    counts = [len(statements), len(globaldefs), len(fielddefs), len(functions), len(strings), nglobals]
    header_size = 4 * 15
    header = struct.pack("<ii", 6, crc)
    pos = header_size
    for sec, n in zip(sections, counts):
        header += struct.pack("<ii", pos, n)
        pos += len(sec)
    header += struct.pack("<i", nfields)
    assert len(header) == header_size
    with open(path, "wb") as f:
        f.write(header + b"".join(sections))


# ---- progs/player.mdl ------------------------------------------------------------------------------

def write_mdl(path):
    """a four-sided pyramid: alias model version 6, one 8x8 skin, one frame (layout: include/modelgen.h)"""
    verts = [(0, 0, 255), (0, 0, 0), (255, 0, 0), (255, 255, 0), (0, 255, 0)]
    tris = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 1)]
    out = b"IDPO" + struct.pack("<i3f3ff3f", 6, 0.1, 0.1, 0.1, -12.0, -12.0, -12.0, 24.0, 0.0, 0.0, 0.0)
    out += struct.pack("<8if", 1, 8, 8, len(verts), len(tris), 1, 0, 0, 1.0)
    out += struct.pack("<i", 0) + lcg_bytes(77, 64)
    for i in range(len(verts)):
        out += struct.pack("<3i", 0, (i * 3) % 8, (i * 5) % 8)
    for t in tris:
        out += struct.pack("<4i", 1, *t)
    out += struct.pack("<i", 0) + struct.pack("<4B4B16s", 0, 0, 0, 0, 255, 255, 255, 0, b"stand")
    for v in verts:
        out += struct.pack("<4B", v[0], v[1], v[2], 0)
    open(path, "wb").write(out)


# ---- maps/box.bsp ----------------------------------------------------------------------------------

HALF = 120          # the room is [-120, 120]^3: texture extents 240 (the software renderer's limit is 256)


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def write_bsp(path):
    planes, faces, nodes, clipnodes, texinfos, verts, edges, surfedges = [], [], [], [], [], [], [(0, 0)], []
    lighting = bytearray()
    wall = 0
    for axis in range(3):
        for sign in (1, -1):
            normal = [0.0, 0.0, 0.0]
            normal[axis] = 1.0                                           # axial planes carry the positive normal (type = axis)
            planes.append(struct.pack("<4fi", normal[0], normal[1], normal[2], float(sign * HALF), axis))
            inward = [0, 0, 0]
            inward[axis] = -sign
            t1 = [0, 0, 0]
            t2 = [0, 0, 0]
            t1[(axis + 1) % 3] = 1
            t2[(axis + 2) % 3] = 1
            corners = []
            for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = [0, 0, 0]
                p[axis] = sign * HALF
                p[(axis + 1) % 3] = a * HALF
                p[(axis + 2) % 3] = b * HALF
                corners.append(tuple(p))
            # faces are wound clockwise seen from their front: the right-hand normal of the loop points OUT of the room
            e1 = [corners[1][i] - corners[0][i] for i in range(3)]
            e2 = [corners[2][i] - corners[1][i] for i in range(3)]
            n = cross(e1, e2)
            if sum(n[i] * inward[i] for i in range(3)) > 0:
                corners.reverse()
            first_edge = len(surfedges)
            base = len(verts)
            verts.extend(corners)
            for k in range(4):
                edges.append((base + k, base + (k + 1) % 4))
                surfedges.append(len(edges) - 1)
            texinfos.append(struct.pack("<8fii", t1[0], t1[1], t1[2], 0.0, t2[0], t2[1], t2[2], 0.0, wall, 0))
            side = 1 if sign > 0 else 0                                  # seen from the plane's back side where the normal points out
            samples = (2 * HALF // 16 + 1) ** 2
            light_ofs = len(lighting)
            lighting.extend(64 + (b >> 1) for b in lcg_bytes(900 + wall, samples))
            faces.append(struct.pack("<hhihh4Bi", wall, side, first_edge, 4, wall, 0, 255, 255, 255, light_ofs))
            inside, outside = (1, 0) if sign > 0 else (0, 1)
            children = [0, 0]
            children[outside] = -1                                        # leaf 0: solid
            children[inside] = wall + 1 if wall < 5 else -2               # next wall's node, finally leaf 1: the room
            nodes.append(struct.pack("<i2h3h3hHH", wall, children[0], children[1], -HALF, -HALF, -HALF, HALF, HALF, HALF, wall, 1))
            children[outside] = -2                                        # CONTENTS_SOLID
            children[inside] = wall + 1 if wall < 5 else -1               # CONTENTS_EMPTY
            clipnodes.append(struct.pack("<i2h", wall, children[0], children[1]))
            wall += 1

    # six 64x64 textures with their three mip levels
    miptex = []
    for t in range(6):
        data = b""
        for level in range(4):
            size = 64 >> level
            rows = bytearray(size * size)
            noise = lcg_bytes(500 + t * 4 + level, size * size)
            for y in range(size):
                for x in range(size):
                    checker = ((x * 8 // size) + (y * 8 // size)) & 1
                    rows[y * size + x] = (16 + 32 * t + (8 if checker else 0) + (noise[y * size + x] & 7)) & 0xFF
            data += bytes(rows)
        offsets = [40, 40 + 4096, 40 + 4096 + 1024, 40 + 4096 + 1024 + 256]
        miptex.append(struct.pack("<16sII4I", ("wall%d" % t).encode(), 64, 64, *offsets) + data)
    tex_header = struct.pack("<i", 6)
    pos = 4 + 4 * 6
    for m in miptex:
        tex_header += struct.pack("<i", pos)
        pos += len(m)
    textures = tex_header + b"".join(miptex)

    leafs = [struct.pack("<ii3h3hHH4B", -2, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0),
             struct.pack("<ii3h3hHH4B", -1, -1, -HALF, -HALF, -HALF, HALF, HALF, HALF, 0, 6, 0, 0, 0, 0)]
    marksurfaces = struct.pack("<6H", *range(6))
    models = struct.pack("<9f4i3i", -HALF - 1, -HALF - 1, -HALF - 1, HALF + 1, HALF + 1, HALF + 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 6)
    entities = (b'{\n"classname" "worldspawn"\n}\n{\n"classname" "info_player_start"\n"origin" "0 0 0"\n}\n\0')

    lumps = [entities, b"".join(planes), textures, b"".join(struct.pack("<3f", *v) for v in verts), b"",
             b"".join(nodes), b"".join(texinfos), b"".join(faces), bytes(lighting), b"".join(clipnodes), b"".join(leafs),
             marksurfaces, b"".join(struct.pack("<2H", *e) for e in edges), struct.pack("<%di" % len(surfedges), *surfedges), models]
    header = struct.pack("<i", 29)
    pos = 4 + 8 * 15
    body = b""
    for lump in lumps:
        header += struct.pack("<ii", pos + len(body), len(lump))
        body += lump + b"\0" * (-len(lump) % 4)
    with open(path, "wb") as f:
        f.write(header + body)


# ---- the rest --------------------------------------------------------------------------------------

def write_pop(path, engine):
    text = open(os.path.join(engine, "common", "common.c"), encoding="latin-1").read()
    table = re.search(r"unsigned short pop\[\]\s*=\s*\{(.*?)\};", text, re.S).group(1)
    values = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", table)]
    assert len(values) == 128
    open(path, "wb").write(struct.pack(">128H", *values))


def write_scripts(out):
    data = open(os.path.join(ROOT, "tests", "golden", "scripts.bundle"), "rb").read()
    pos = data.index(b"@@@ ")
    while pos < len(data):
        eol = data.index(b"\n", pos)
        _, name, size = data[pos:eol].decode().split()
        path = os.path.join(out, "lua-scripts", name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(data[eol + 1: eol + 1 + int(size)])
        pos = eol + 1 + int(size) + 1


def main():
    engine, out = sys.argv[1], sys.argv[2]
    id1 = os.path.join(out, "id1")
    os.makedirs(os.path.join(id1, "gfx"), exist_ok=True)
    os.makedirs(os.path.join(id1, "maps"), exist_ok=True)
    write_wad(os.path.join(id1, "gfx.wad"), engine)
    open(os.path.join(id1, "gfx", "palette.lmp"), "wb").write(bytes((i * 37) % 256 for i in range(768)))
    colormap = bytearray()
    for level in range(64):
        colormap.extend(((c & 0xF0) | max(0, min(15, (c & 15) + 8 - level // 4))) & 0xFF for c in range(256))
    open(os.path.join(id1, "gfx", "colormap.lmp"), "wb").write(bytes(colormap) + b"\x20")
    for name, (w, h) in (("conback", (320, 200)), ("loading", (144, 24)), ("pause", (128, 24))):
        open(os.path.join(id1, "gfx", name + ".lmp"), "wb").write(qpic(w, h, sum(name.encode())))
    write_pop(os.path.join(id1, "gfx", "pop.lmp"), engine)
    write_bsp(os.path.join(id1, "maps", "box.bsp"))
    write_progs(os.path.join(id1, "progs.dat"), engine)
    os.makedirs(os.path.join(id1, "progs"), exist_ok=True)
    write_mdl(os.path.join(id1, "progs", "player.mdl"))
    open(os.path.join(id1, "quake.rc"), "w").write("stuffcmds\n")
    write_scripts(out)


if __name__ == "__main__":
    main()
