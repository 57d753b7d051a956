#!/bin/bash
# Profile `bench.py` the way DESIGN.md section 7 describes: one rocprofv3 --kernel-trace --stats run and
# separate --pmc passes (counters never combined with other trace domains), summarised as text, plus the traffic
# record bench.py reads back (profiles/apply_traffic.json: measured HBM bytes per launch of the dominant kernel,
# stamped with the workload, the commit and a hash of the kernel sources so that a stale record is detected).
#   tools/profile_bench.sh <outdir> <profile-name> [bench args...]        (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); NAME=$2; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprofv3 args...
    local name=$1; shift
    rm -rf /tmp/pb_$name
    timeout ${PROF_TIMEOUT:-300} rocprofv3 "$@" -d /tmp/pb_$name -o r -- python $R/bench.py --steps 20 --warmup 3 --repeats 3 --streams 1 --no-cpu-baseline --no-extra --no-live-traffic "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
    find /tmp/pb_$name -name "*.db" | head -1
}
BENCH_ARGS=("$@")
kt=$(run kt --kernel-trace --stats)
pf=$(run fetch --pmc FETCH_SIZE --kernel-trace)
pw=$(run write --pmc WRITE_SIZE --kernel-trace)
pr=$(run rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace)
pt=$(run tcp --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace)
{
    echo "# command: rocprofv3 <mode> -- python bench.py --steps 20 --warmup 3 --repeats 3 --streams 1 --no-cpu-baseline --no-extra --no-live-traffic ${BENCH_ARGS[*]}"
    echo "# (--streams 1: every launch on one stream, so that the trace shows the kernel alone; the default bench alternates two)"
    echo "# modes: --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCC_EA0_RDREQ_* | --pmc TCP_* (separate passes)"
    grep -h '"metric"' "$OUT/kt.log" | head -1
    python $R/tools/prof_summary.py "$kt" $pf $pw $pr $pt
} > "$OUT/summary.txt" 2>&1
python $R/tools/prof_summary.py --traffic "$OUT/apply_traffic.json" "$NAME" "$OUT/kt.log" "$kt" "$pf" "$pw" "$pr" >> "$OUT/summary.txt" 2>&1
