#!/usr/bin/env python3
"""Bundle the reference's lens / globe scripts as TEST INPUT DATA (tests/golden/scripts.bundle).

BASELINE.json asks for parity "on identical Lua lens/globe scripts"; those scripts are game data
(game/lua-scripts, loaded at run time from <basedir>/lua-scripts by the reference and by this
library alike - the product ships none).  /root/reference does not exist on the GPU box, so the
inputs of the golden vectors travel as this one fixture file, verbatim, made by this script:
    python tests/golden/make_scripts_bundle.py
"""
import glob
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/game/lua-scripts"


def main():
    out = []
    for kind in ("globes", "lenses"):
        for path in sorted(glob.glob(os.path.join(SRC, kind, "*.lua"))):
            text = open(path, newline="").read()
            out.append(f"@@@ {kind}/{os.path.basename(path)} {len(text.encode())}\n{text}\n")
    with open(os.path.join(HERE, "scripts.bundle"), "w", newline="") as f:
        f.write("# lens / globe scripts of shaunlebron/blinky (game/lua-scripts), test inputs - see make_scripts_bundle.py\n")
        f.write("".join(out))
    print(f"{len(out)} scripts bundled")


if __name__ == "__main__":
    main()
