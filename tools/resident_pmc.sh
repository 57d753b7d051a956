#!/bin/bash
# Developer tool: HBM traffic of the resident apply per frame (separate --pmc passes, each with --kernel-trace only).
#   tools/resident_pmc.sh <outfile> <frames> <lens> [lens...]      (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); N=$2; shift 2
mkdir -p "$(dirname "$OUT")"
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
for LENS in "$@"; do
  # (PASSES="FETCH_SIZE|WRITE_SIZE" restricts the counter groups, '|' separated)
  IFS='|' read -r -a GROUPS_ <<< "${PASSES:-FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum|TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum}"
  for C in "${GROUPS_[@]}"; do
    rm -rf /tmp/rpmc
    timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/rpmc -o pmc -- python $R/tools/resident_traffic.py $LENS $N ${RES_ARGS} > /tmp/rpmc.log 2>&1
    grep RESIDENT /tmp/rpmc.log >> "$OUT"
    db=$(find /tmp/rpmc -name "*.db" | head -1)
    if [ -n "$db" ]; then python3 - "$db" $N >> "$OUT" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2])
for k, c, v, d in con.execute("select kernel_name, counter_name, sum(value), avg(duration)/1e3 from counters_collection where kernel_name like '%apply_resident%' group by kernel_name, counter_name"):
    unit = "KiB" if c.endswith("_SIZE") else ""
    print(f"   {c:24s} total {v:16.1f} per frame {v / n:14.2f} {unit}  kernel {d:.1f} us  {k[:60]}")
PY
    else echo "   no db for $C" >> "$OUT"; tail -3 /tmp/rpmc.log >> "$OUT"; fi
  done
done
