#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "sharded_optimizer" -s > gpurun_out/c51_tests.log 2>&1
echo "exit $?" >> gpurun_out/c51_tests.log
grep "worst tensors\|passed\|failed" gpurun_out/c51_tests.log | cut -c1-1500
python -c "
import json; d=json.load(open('gpurun_out/parity_report_configs.json')); print({k:v for k,v in d.items() if k.startswith('sharded_vs')})"
