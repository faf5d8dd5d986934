#!/bin/bash
# A/B: GP state update deferred onto the prior's stream (HL_GP_DEFER=1) together with the early fork and the look-ahead
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  for e in 0 1; do
    HL_GP_DEFER=$e timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also 2>gpurun_out/c41_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 defer=$e', round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" || { tail -5 gpurun_out/c41_err.log; exit 1; }
  done
done
HL_GP_DEFER=1 timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or GP or traj" > gpurun_out/c41_tests.log 2>&1
echo "exit $?" >> gpurun_out/c41_tests.log
tail -4 gpurun_out/c41_tests.log
