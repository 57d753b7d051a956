#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (the default output of this ROCm 7.2 rocprofv3) as text:
   python tools/prof_summary.py <kernel-trace.db> [<pmc.db> ...] > profiles/<name>.txt
Kernel table: calls / total / avg / min / max duration (us).  PMC table: per kernel and counter,
calls / avg / min / max of the counter value (FETCH_SIZE and WRITE_SIZE are in KiB)."""
import sqlite3
import sys


def kernel_table(path):
    con = sqlite3.connect(path)
    print(f"== kernel trace: {path}")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
         "from kernels group by name order by sum(duration) desc")
    for name, n, tot, avg, mn, mx in con.execute(q):
        print(f"{n:6d} {tot:12.1f} {avg:10.2f} {mn:10.2f} {mx:10.2f}  {name[:110]}")
    # per launch-shape breakdown of the hot kernels (grid size tells batch launches from single-frame ones)
    print("-- by grid size (hot kernels)")
    q = ("select name, grid_x, grid_y, count(*), avg(duration)/1e3, min(duration)/1e3, vgpr_count, sgpr_count, lds_size from kernels "
         "where name like '%apply%' or name like '%bk_build%' or name like '%bk_forward%' or name like '%tile%' "
         "group by name, grid_x, grid_y order by name, grid_x, grid_y")
    for name, gx, gy, n, avg, mn, vg, sg, lds in con.execute(q):
        print(f"   grid {gx}x{gy} calls {n:4d} avg {avg:9.2f} us min {mn:9.2f} us vgpr {vg} sgpr {sg} lds {lds}  {name[:70]}")


def pmc_table(path):
    con = sqlite3.connect(path)
    print(f"== counters: {path}")
    q = ("select kernel_name, counter_name, grid_size, count(*), avg(value), min(value), max(value) from counters_collection "
         "group by kernel_name, counter_name, grid_size order by kernel_name, counter_name, grid_size")
    for k, c, g, n, avg, mn, mx in con.execute(q):
        print(f"{c:12s} grid {g:9d} calls {n:4d} avg {avg:14.2f} min {mn:14.2f} max {mx:14.2f}  {k[:90]}")


def pmc_sequence(path, like="%apply_%"):
    """per-dispatch counter values in launch order (for probes that step through configurations)"""
    con = sqlite3.connect(path)
    print(f"== per-dispatch counters: {path}")
    q = ("select dispatch_id, grid_size_x, grid_size_y, duration/1e3, counter_name, value from counters_collection "
         "where kernel_name like ? order by dispatch_id, counter_name")
    rows = {}
    for d, gx, gy, dur, c, v in con.execute(q, (like,)):
        rows.setdefault(d, [gx, gy, dur, {}])[3][c] = v
    for d in sorted(rows):
        gx, gy, dur, cs = rows[d]
        print(f"{d:5d} grid {gx}x{gy} {dur:9.2f} us  " + "  ".join(f"{k}={v:.0f}" for k, v in cs.items()))


def traffic_record(out_json, profile_name, bench_log, kt_db, fetch_db, write_db, rdreq_db):
    """profiles/apply_traffic.json: measured HBM bytes per launch of the bench's batch launches (the apply kernel with the
    largest grid), per MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB;
    on gfx950 FETCH_SIZE tallies the 128-byte TCC->EA read requests at 64 bytes, so reads are doubled - cross-checked here
    against the request-size counters of a third pass."""
    import hashlib
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def dominant(db, counter):
        con = sqlite3.connect(db)
        q = ("select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name = ? and "
             "kernel_name like '%apply_%' group by kernel_name, grid_size order by count(*) * grid_size desc")     # (not the largest grid alone: the block-height tuning times a few launches of other shapes)
        rows = list(con.execute(q, (counter,)))
        return rows[0] if rows else None
    f, w = dominant(fetch_db, "FETCH_SIZE"), dominant(write_db, "WRITE_SIZE")
    req = {c: dominant(rdreq_db, c) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
    con = sqlite3.connect(kt_db)
    krow = list(con.execute("select name, grid_x * grid_y, count(*), avg(duration)/1e3 from kernels where name like '%apply_%' "
                            "group by name, grid_x, grid_y order by count(*) * grid_x * grid_y desc"))[0]
    bench = None
    for line in open(bench_log):
        if line.startswith("{") and '"metric"' in line:
            bench = json.loads(line)
    h = hashlib.sha1()
    for rel in ("blinky_amd/csrc/bk_apply_coop.hip", "blinky_amd/csrc/bk_apply.hip", "blinky_amd/csrc/bk_build_params.h"):
        h.update(open(os.path.join(root, rel), "rb").read())
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = os.environ.get("BLINKY_COMMIT", "unknown (no .git on the GPU box; see the profile's commit in git log)")
    F, R = bench["config"]["frames_per_step"], bench["config"]["ring_globes"]
    rd128 = req["TCC_EA0_RDREQ_128B_sum"][3] if req["TCC_EA0_RDREQ_128B_sum"] else None
    rd64 = req["TCC_EA0_RDREQ_64B_sum"][3] if req["TCC_EA0_RDREQ_64B_sum"] else 0
    rd32 = req["TCC_EA0_RDREQ_32B_sum"][3] if req["TCC_EA0_RDREQ_32B_sum"] else 0
    rdall = req["TCC_EA0_RDREQ_sum"][3] if req["TCC_EA0_RDREQ_sum"] else None
    rec = {
        "workload": f"3840x2160 cube/panini x{F} ring{R}",
        "profile": profile_name, "commit": commit, "kernel_source_sha1_16": h.hexdigest()[:16],
        "kernel": f"{f[0]} (grid {f[1]} work-items, {f[2]} launches profiled)",
        "kernel_avg_us_in_kernel_trace": round(krow[3], 2),
        "FETCH_SIZE_KiB_per_launch": round(f[3], 2), "WRITE_SIZE_KiB_per_launch": round(w[3], 2),
        "hbm_bytes_per_launch": int(round((2 * f[3] + w[3]) * 1024)),
        "read_requests_per_launch": {"all": rdall, "32B": rd32, "64B": rd64, "128B": rd128},
        "read_bytes_by_request_size": int(round(rd128 * 128 + rd64 * 64 + rd32 * 32)) if rd128 is not None else None,
        "correction": "MI355X_MICROARCH.md (HBM): FETCH_SIZE on gfx950 counts the 128-byte TCC->EA read requests as 64 bytes -> reads "
                      "doubled (cross-check: read_bytes_by_request_size from the TCC_EA0_RDREQ_* pass); WRITE_SIZE as reported. "
                      "Separate --pmc passes, each with --kernel-trace only.",
        "algorithmic_bytes_per_launch": 6 * 3840 * 2160 * F,
    }
    json.dump(rec, open(out_json, "w"), indent=1)
    print("== traffic record:", json.dumps(rec))


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        traffic_record(*sys.argv[2:9])
        sys.exit(0)
    if sys.argv[1] == "--seq":
        for p in sys.argv[2:]:
            pmc_sequence(p)
        sys.exit(0)
    if sys.argv[1] == "--seq-like":           # --seq-like <pattern> <db>...
        for p in sys.argv[3:]:
            pmc_sequence(p, sys.argv[2])
        sys.exit(0)
    kernel_table(sys.argv[1])
    for p in sys.argv[2:]:
        pmc_table(p)
