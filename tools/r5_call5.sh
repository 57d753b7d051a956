cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
rm -rf /tmp/c4pmc; (cd $R && timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/c4pmc -o pmc -- python tools/r5_c4_pmc.py > /tmp/c4.log 2>&1)
db=$(find /tmp/c4pmc -name "*.db" | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for k, g, c, n, v, d in con.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration)/1e3 from counters_collection where kernel_name like '%apply_coop%' group by kernel_name, grid_size, counter_name order by grid_size"):
    print(f"C4PMC grid {g:9d} x{n:3d} {c:22s} avg {v:14.1f}  kernel {d:8.1f} us  {k[:50]}")
PY
done
