#!/bin/bash
# rocprofv3 passes over tools/apply_probe.py for one lens / size / batch (kernel trace, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes), summarised as text.   tools/profile_probe.sh <outdir> <lens> <W> <H> <frames>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); LENS=$2; W=$3; H=$4; F=$5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp BK_REPS=${BK_REPS:-6} BK_WGS=16
run() { local name=$1; shift; rm -rf /tmp/pp_$name; timeout ${PROF_TIMEOUT:-300} rocprofv3 "$@" -d /tmp/pp_$name -o r -- python $R/tools/apply_probe.py $LENS $W $H $F 2 > "$OUT/$name.log" 2>&1; find /tmp/pp_$name -name "*.db" | head -1; }
kt=$(run kt --kernel-trace --stats)
pf=$(run fetch --pmc FETCH_SIZE --kernel-trace)
pw=$(run write --pmc WRITE_SIZE --kernel-trace)
{
    echo "# command: rocprofv3 <mode> -- python tools/apply_probe.py $LENS $W $H $F 2    (BK_REPS=$BK_REPS)"
    grep -h "launch\|tile stats" "$OUT/kt.log"
    python $R/tools/prof_summary.py "$kt" $pf $pw
} > "$OUT/summary.txt" 2>&1
