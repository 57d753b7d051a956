#!/bin/bash
# Developer tool: hardware counters of the generated lensmap-build kernel (bk_build_inverse), one rocprofv3 --pmc pass per
# counter group (counters only ever together with --kernel-trace), summarised per kernel as text.
#   tools/pmc_build.sh <outdir> [lenses]        (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); LENSES=${2:-panini,hammer,quincuncial}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A G
G[valu]="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
G[busy]="SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
for g in valu busy; do
    rm -rf /tmp/pmcb_$g
    timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc ${G[$g]} --kernel-trace -d /tmp/pmcb_$g -o pmc -- python $R/tools/build_probe.py --lenses $LENSES --reps 2 > "$OUT/$g.log" 2>&1
    db=$(find /tmp/pmcb_$g -name "*.db" | head -1)
    if [ -n "$db" ]; then { python $R/tools/prof_summary.py --seq-like "%bk_build%" "$db"; python $R/tools/prof_summary.py "$db" "$db"; } 2>&1 | grep -E "bk_build|^==|grid" > "$OUT/$g.txt"; else echo "no db (rc/pass failed)" > "$OUT/$g.txt"; tail -5 "$OUT/$g.log" >> "$OUT/$g.txt"; fi
done
rm -rf /tmp/pmcb_kt
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pmcb_kt -o kt -- python $R/tools/build_probe.py --lenses $LENSES --reps 2 > "$OUT/kt.log" 2>&1
db=$(find /tmp/pmcb_kt -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" 2>&1 | grep -E "bk_build|^==" > "$OUT/kt.txt"
