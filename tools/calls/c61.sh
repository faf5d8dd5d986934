#!/bin/bash
mkdir -p gpurun_out
r() { timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_trajectory_gpu.py tests/test_gpu_parity.py -x -q -m gpu "$@" > gpurun_out/c61_one.log 2>&1; echo "rc $? :: $* :: $(grep -E 'passed|failed' gpurun_out/c61_one.log | tail -1)"; }



r
