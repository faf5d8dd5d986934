#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HL_GP_DEFER=1
bash tools/trace_step.sh c47_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp > gpurun_out/c47.log 2>&1
head -50 gpurun_out/c47_cfg4_step_stats.txt | cut -c1-120
