#!/usr/bin/env python3
"""Developer probe: bk_build of given lenses at a size, several times - kernel + host fix-up breakdown
(bk_debug_build_breakdown).  Also the target of the build-kernel counter passes (tools/pmc_build.sh).
usage: python tools/build_probe.py [--lenses panini,hammer,quincuncial] [--size 3840x2160] [--reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import blinky_amd  # noqa: E402
import scripts as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lenses", default="panini,hammer,quincuncial")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--host-entries", type=int, default=0)
    ap.add_argument("--no-host-module", action="store_true", help="do not wait for the compiled host module: the interpreter re-derives")
    args = ap.parse_args()
    W, H = [int(v) for v in args.size.split("x")]
    for lens in args.lenses.split(","):
        ctx = blinky_amd.Context()
        S.configure(ctx, "cube", lens, "f_fov 180" if lens == "panini" else None, (W, H))
        t0 = time.time()
        ctx.build()
        first = (time.time() - t0) * 1e3
        t0 = time.time()
        have_module = ctx.host_module_ready(wait=not args.no_host_module)       # the compiled host module of the fix-up (bk_set_host_compile)
        module_s = time.time() - t0
        walls, recs = [], []
        for _ in range(args.reps):
            t0 = time.time()
            ctx.build()
            walls.append((time.time() - t0) * 1e3)
            recs.append(ctx.build_breakdown())
        best = min(range(args.reps), key=lambda i: walls[i])
        b = recs[best]
        flagged, changed = ctx.last_build_fixups()
        print(f"{lens:14s} {W}x{H} first {first:8.1f} ms (host module {'ready after %.2f s' % module_s if have_module else 'none'}, used: {b['compiled_host_module']}) | wall best {walls[best]:7.3f} median {sorted(walls)[len(walls) // 2]:7.3f} ms | "
              f"device+fixup events {b['build_ms']:7.3f} ms, host re-evaluation {b['host_eval_ms']:7.3f} ms of it | flagged {flagged} changed {changed} "
              f"| kernel + read-back wall {b['kernel_wall_ms']:7.3f} ms, {b['retries']} retries | pool threads {b['pool_threads']}", flush=True)
        if args.host_entries:
            import numpy as np
            ids = np.random.default_rng(1).integers(0, W * H, args.host_entries).astype(np.uint32)
            ids.sort()
            ts = []
            for _ in range(3):
                t0 = time.time()
                ctx.host_entries(ids)
                ts.append((time.time() - t0) * 1e3)
            print(f"{'':14s} host re-evaluation of {args.host_entries} random pixels (bk_debug_host_entries, worker pool): best {min(ts):.3f} ms", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
