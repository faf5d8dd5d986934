#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -s > gpurun_out/c38_conv.log 2>&1
echo "exit $?" >> gpurun_out/c38_conv.log
grep -n "conv_deep grads\|passed\|failed" gpurun_out/c38_conv.log
timeout -k 10 500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "conv_backward" > gpurun_out/c38_cfg.log 2>&1
echo "exit $?" >> gpurun_out/c38_cfg.log
tail -25 gpurun_out/c38_cfg.log
cat gpurun_out/parity_report.json 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print({k:v for k,v in d.items() if 'deep' in k})" || true
