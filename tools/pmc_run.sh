#!/bin/bash
# PMC passes over tools/kernel_times.py (eager steps): one rocprofv3 run per counter group (counters only, no other tracing
# besides --kernel-trace), CSVs under gpurun_out/<prefix>_<group>/.   usage: tools/pmc_run.sh <prefix> [kernel_times args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pfx=$1; shift
declare -A G
G[busy]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G[insts]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES"
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_TCC_WRITE_REQ"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
for g in ${PMC_GROUPS:-busy insts lds tcp}; do
  rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $R/gpurun_out/${pfx}_$g -o p -- python3 $R/tools/kernel_times.py --no-graph -n 8 "$@" > $R/gpurun_out/${pfx}_$g.log 2>&1 || { echo "pass $g failed"; tail -5 $R/gpurun_out/${pfx}_$g.log; exit 1; }
  echo "pass $g done"
done
