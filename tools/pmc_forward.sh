#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r06_pmc_fwd}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A G
G[valu]="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
G[mem]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS"
G[tcc]="TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
for g in valu mem tcc; do
    rm -rf /tmp/pmcf_$g
    timeout 150 rocprofv3 --pmc ${G[$g]} --kernel-trace -d /tmp/pmcf_$g -o pmc -- python $R/tools/build_probe.py --lenses winkel2 --reps 2 > $OUT/$g.log 2>&1
    db=$(find /tmp/pmcf_$g -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/prof_summary.py --seq-like "%bk_forward%" "$db" 2>&1 | grep -E "bk_forward|^==|grid" > $OUT/$g.txt; else echo "no db" > $OUT/$g.txt; tail -5 $OUT/$g.log >> $OUT/$g.txt; fi
done
cat $OUT/valu.txt $OUT/mem.txt $OUT/tcc.txt
