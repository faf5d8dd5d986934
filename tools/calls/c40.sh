#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/trace_step.sh c40_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp > gpurun_out/c40.log 2>&1
head -50 gpurun_out/c40_cfg4_step_stats.txt | cut -c1-120
