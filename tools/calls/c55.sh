#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "constructor_modes" > gpurun_out/c55_tests.log 2>&1
echo "exit $?" >> gpurun_out/c55_tests.log
tail -30 gpurun_out/c55_tests.log | cut -c1-400
python -c "
import json; d=json.load(open('gpurun_out/parity_report_configs.json')); print({k:v for k,v in d.items() if k.startswith('mix_nohid')})"
