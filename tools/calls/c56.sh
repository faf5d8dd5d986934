#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "nohid" > gpurun_out/c56_tests.log 2>&1
echo "exit $?" >> gpurun_out/c56_tests.log
grep -E "passed|failed|FAILED|Error|assert " gpurun_out/c56_tests.log | cut -c1-300 | head -40
python -c "
import json; d=json.load(open('gpurun_out/parity_report_configs.json')); print({k:(v if len(str(v))<400 else str(v)[:400]) for k,v in d.items() if 'nohid' in k and 'mix' not in k})"
