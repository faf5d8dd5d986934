#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "gp_captured_chain" > gpurun_out/c64_tests.log 2>&1
echo "exit $? $(grep -E 'passed|failed' gpurun_out/c64_tests.log | tail -1)"
grep -E "^E " gpurun_out/c64_tests.log | head -5 | cut -c1-400
python -c "
import json; d=json.load(open('gpurun_out/parity_report_configs.json')); print({k:v for k,v in d.items() if k.startswith('gp_chain_vs_eager')})"
