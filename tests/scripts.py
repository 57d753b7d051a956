"""Access to the bundled lens / globe scripts (tests/golden/scripts.bundle) - test inputs."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = None


def _load():
    global _cache
    if _cache is None:
        data = open(os.path.join(_HERE, "golden", "scripts.bundle"), "rb").read()
        _cache = {}
        pos = data.index(b"@@@ ")
        while pos < len(data):
            eol = data.index(b"\n", pos)
            _, name, size = data[pos:eol].decode().split()
            body = data[eol + 1: eol + 1 + int(size)]
            _cache[name] = body.decode()
            pos = eol + 1 + int(size) + 1
    return _cache


def script(kind, name):
    """kind: 'lenses' or 'globes'"""
    return _load()[f"{kind}/{name}.lua"]


def names(kind):
    return sorted(k.split("/")[1][:-4] for k in _load() if k.startswith(kind + "/"))


LENSES = names("lenses")
GLOBES = names("globes")
ZOOM_CMD = {"f_fov": 1, "f_vfov": 2, "f_cover": 3, "f_contain": 4}


def zoom_args(cmd):
    """'f_fov 120' -> (zoom type, degrees) for Context.set_zoom; '' -> (0, 0)"""
    parts = cmd.split()
    if not parts:
        return 0, 0
    return ZOOM_CMD[parts[0]], int(float(parts[1])) if len(parts) > 1 else 0


def configure(ctx, globe, lens, zoom=None, size=None):
    """'f_globe G; f_lens L; <zoom or the lens' onload>' on a blinky_amd Context"""
    ctx.load_globe(script("globes", globe), globe + ".lua")
    ctx.load_lens(script("lenses", lens), lens + ".lua")
    info = ctx.lens_info()
    cmd = zoom if zoom else info.onload.decode()
    parts = cmd.split()
    if parts:
        ctx.set_zoom(ZOOM_CMD[parts[0]], int(float(parts[1])) if len(parts) > 1 else 0)
    else:
        ctx.set_zoom(0, 0)
    if size:
        ctx.resize(*size)
    return info
