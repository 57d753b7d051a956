#!/bin/bash
# Developer tool: collect hardware counters for the apply kernel, one rocprofv3 --pmc pass per
# counter group (counters only ever together with --kernel-trace), and summarise each pass as text.
#   tools/pmc_passes.sh <outdir> <lens> [groups...]      (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); LENS=${2:-panini}; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM"
G[sq2]="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
G[sq3]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
G[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
G[lat]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"
G[tccA]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
G[tccB]="TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"
G[tccC]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum TCC_REQ_sum"
G[tcpA]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
for g in "$@"; do
    rm -rf /tmp/pmc_$g
    timeout ${PMC_TIMEOUT:-100} rocprofv3 --pmc ${G[$g]} --kernel-trace -d /tmp/pmc_$g -o pmc -- python $R/tools/apply_probe.py $LENS 3840 2160 16 ${PROBE_VARIANT:-2} > "$OUT/$g.log" 2>&1
    db=$(find /tmp/pmc_$g -name "*.db" | head -1)
    if [ -n "$db" ]; then { if [ -n "$PMC_SEQ" ]; then python $R/tools/prof_summary.py --seq "$db"; else python $R/tools/prof_summary.py "$db" "$db"; fi; } 2>&1 | grep -E "apply_tiled|apply_coop|^==|grid" > "$OUT/$g.txt"; else echo "no db (rc/pass failed)" > "$OUT/$g.txt"; tail -5 "$OUT/$g.log" >> "$OUT/$g.txt"; fi
done
