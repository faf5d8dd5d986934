#!/bin/bash
mkdir -p gpurun_out
HL_GP_SPLIT_PREP=0 timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_trajectory_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "gp or config5 or GP or traj" > gpurun_out/c59_tests.log 2>&1
echo "split=0 exit $? $(grep -E 'passed|failed' gpurun_out/c59_tests.log | tail -1)"
