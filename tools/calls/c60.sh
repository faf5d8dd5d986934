#!/bin/bash
mkdir -p gpurun_out
r() { timeout -k 10 600 python -m pytest "$@" -x -q -m gpu > gpurun_out/c60_one.log 2>&1; echo "rc $? :: $* :: $(grep -E 'passed|failed' gpurun_out/c60_one.log | tail -1)"; }
r tests/test_gpu_parity.py -k "rccl"
r tests/test_trajectory_gpu.py tests/test_gpu_parity.py -k "traj or rccl"
r tests/test_gpu_configs.py tests/test_gpu_parity.py -k "nohid or rccl"
r tests/test_gpu_configs.py tests/test_gpu_parity.py -k "sharded_optimizer or rccl"
r tests/test_gpu_configs.py tests/test_gpu_parity.py -k "config5 or gp_prior or rccl"
