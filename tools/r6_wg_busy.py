"""(r6) per-workgroup busy time of the resident kernel (developer bit 8192) against what the plan gave each workgroup - chunks, lines, block count:
which of them the slowest workgroups have in common.  Developer probe; GPU box only.  usage: r6_wg_busy.py [lens ...]"""
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, bench, blinky_amd, scripts as S

W, H = 3840, 2160
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir(ROOT)
for lens in sys.argv[1:] or ["hammer", "quincuncial", "mercator", "panini"]:
    wl = bench.OneGpuWorkload(torch, blinky_amd, S, 0, "cube", lens, None, W, H, 1)
    blinky_amd.debug_set_option("print_model", 1)
    wl.ctx.set_ablation(8192)
    out = torch.zeros((8, H, W), dtype=torch.uint8, device="cuda")
    wl.ctx.resident_begin(idle_ms=200)
    wl.ctx.resident_wait(wl.ctx.resident_submit(out.data_ptr(), W, frame=0))
    frames = 800
    for b in range(frames // 8):
        last = wl.ctx.resident_submit_batch(out.data_ptr(), W, H * W, frame0=(b * 8) % wl.R, nframes=8)
    wl.ctx.resident_wait(last)
    wl.ctx.resident_end()
    blinky_amd.debug_set_option("print_model", 0)
    rows = []
    for ln in open("gpurun_out/res_wg_stats.txt"):
        p = ln.split()
        blocks = [tuple(int(v) for v in q.split(":")) for q in p[2:]]
        rows.append((int(p[0]), int(p[1]) / (frames + 1) / 100.0, len(blocks), sum(b[1] for b in blocks), sum(b[2] for b in blocks), max(b[1] for b in blocks)))
    a = np.array(rows, dtype=np.float64)
    busy = a[:, 1]
    def corr(x):
        return float(np.corrcoef(busy, x)[0, 1]) if x.std() > 0 else 0.0
    cu = (a[:, 0] // 8) % 32 + 32 * (a[:, 0] % 8)                       # the CU a workgroup runs on (XCD = b % 8, round the XCD's 32 CUs)
    cu_sum = {}
    for c, ch in zip(cu, a[:, 3]):
        cu_sum[c] = cu_sum.get(c, 0) + ch
    cu_load = np.array([cu_sum[c] for c in cu])
    cu_lines = {}
    for c, l in zip(cu, a[:, 4]):
        cu_lines[c] = cu_lines.get(c, 0) + l
    cu_l = np.array([cu_lines[c] for c in cu])
    print(f"RESULT {lens}: {len(a)} workers, busy us per frame: 5% {np.percentile(busy, 5):.2f} median {np.median(busy):.2f} 95% {np.percentile(busy, 95):.2f} max {busy.max():.2f} mean {busy.mean():.2f}; "
          f"correlation of busy with: blocks {corr(a[:, 2]):.2f}, chunks {corr(a[:, 3]):.2f}, lines {corr(a[:, 4]):.2f}, largest block {corr(a[:, 5]):.2f}, "
          f"its CU's chunks {corr(cu_load):.2f}, its CU's lines {corr(cu_l):.2f}, XCD {corr(a[:, 0] % 8):.2f}", flush=True)
    by_x = [busy[a[:, 0] % 8 == x].mean() for x in range(8)]
    print("RESULT    mean busy by XCD:", " ".join(f"{v:.2f}" for v in by_x), "| chunks per worker: min %d mean %d max %d; lines: min %d mean %d max %d" % (
        a[:, 3].min(), a[:, 3].mean(), a[:, 3].max(), a[:, 4].min(), a[:, 4].mean(), a[:, 4].max()))
    xs = int(np.argmax(by_x))
    sel = a[:, 0] % 8 == xs
    cu_of = ((a[:, 0] // 8) % 32).astype(int)
    slot_of = (a[:, 0] // 256).astype(int)
    print(f"RESULT    slowest XCD {xs}: mean busy by CU index (b/8 %% 32):", " ".join(f"{busy[sel & (cu_of == c)].mean():.1f}" if (sel & (cu_of == c)).any() else "-" for c in range(32)))
    print(f"RESULT    slowest XCD {xs}: mean busy by place (b / 256):", " ".join(f"{busy[sel & (slot_of == q)].mean():.2f}" if (sel & (slot_of == q)).any() else "-" for q in range(8)),
          "| another XCD by place:", " ".join(f"{busy[(a[:, 0] % 8 == (xs + 1) % 8) & (slot_of == q)].mean():.2f}" for q in range(8)))
    worst = np.argsort(-busy)[:8]
    print("RESULT    slowest:", "; ".join(f"wg {int(a[i, 0])} {busy[i]:.2f}us {int(a[i, 2])}blk {int(a[i, 3])}ch {int(a[i, 4])}ln" for i in worst))
    wl.close()
