import sys, os, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_ffi as O, blinky_amd as bk
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lm = O.lensmap("cube", "hammer", None, 960, 540)
F = 8
globes = [O.lcg_globe(lm.ps, 6, f) for f in range(F)]
ctx = bk.Context(); ctx.set_frames(F); ctx.resize(lm.W, lm.H)
for f in range(F):
    for p in range(6): ctx.upload_plate(f, p, globes[f][p])
ctx.set_lensmap(lm.offsets, lm.tints)
ctx.set_ablation(flags)
want = [O.apply(lm.offsets, lm.tints, lm.W, lm.H, globes[f], np.zeros((lm.H, lm.W), np.uint8)) for f in range(F)]
bad = 0
for rnd in range(10):
    outs = [torch.zeros((lm.H, lm.W), dtype=torch.uint8, device="cuda") for _ in range(100)]
    torch.cuda.synchronize()
    ctx.resident_begin(idle_ms=2000)
    tickets = [ctx.resident_submit(outs[i].data_ptr(), lm.W, frame=(i * 3) % F) for i in range(100)]
    ctx.resident_wait(tickets[-1]); ctx.resident_end()
    for i in range(100):
        g = outs[i].cpu().numpy(); w = want[(i * 3) % F]
        if not np.array_equal(g, w):
            d = np.argwhere(g != w); bad += 1
            r0, r1, c0, c1 = d[:,0].min(), d[:,0].max() + 1, d[:,1].min(), d[:,1].max() + 1
            reg = g[r0:r1, c0:c1]
            same = [f for f in range(F) if np.array_equal(reg, want[f][r0:r1, c0:c1])]
            print(f"flags {flags} round {rnd} submission {i} (globe {(i * 3) % F}): {len(d)} px differ, rows {r0}..{r1-1} cols {c0}..{c1-1}; region all zero: {not reg.any()}; region equals the warp of globe(s) {same}")
print(f"flags {flags}: {bad} bad frames of 1000", ctx.resident_info())
