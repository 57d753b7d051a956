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


if __name__ == "__main__":
    if sys.argv[1] == "--seq":
        for p in sys.argv[2:]:
            pmc_sequence(p)
        sys.exit(0)
    kernel_table(sys.argv[1])
    for p in sys.argv[2:]:
        pmc_table(p)
