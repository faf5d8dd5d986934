#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "conv_logvar_network_with_deeper" > gpurun_out/c54_tests.log 2>&1
echo "exit $?" >> gpurun_out/c54_tests.log
tail -25 gpurun_out/c54_tests.log | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/parity_report_configs.json')); print({k:v for k,v in d.items() if k.startswith('conv_logvar_deep')})"
