#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > gpurun_out/c37_conv.log 2>&1
echo "exit $?" >> gpurun_out/c37_conv.log
tail -30 gpurun_out/c37_conv.log
