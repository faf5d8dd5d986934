#!/bin/bash
# A/B: GP prior's per-subject kernel forked behind the encoder (HL_GP_EARLY=1) vs behind the head kernel (=0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or GP or traj" > gpurun_out/c39_tests.log 2>&1
echo "exit $?" >> gpurun_out/c39_tests.log
tail -4 gpurun_out/c39_tests.log
grep -q "exit 0" gpurun_out/c39_tests.log || exit 1
for i in 1 2 3; do
  for e in 0 1; do
    HL_GP_EARLY=$e timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 early=$e', round(d['ms_per_step'],4))" || exit 1
  done
done
for e in 0 1; do
  HL_GP_EARLY=$e timeout -k 10 300 python bench.py --conv --kl gp --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convgp early=$e', round(d['ms_per_step'],4))" || exit 1
done
