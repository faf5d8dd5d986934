#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_gpu_configs.py -q -m gpu -k "conv" > gpurun_out/c57_tests.log 2>&1
echo "exit $?" >> gpurun_out/c57_tests.log
grep -E "passed|failed|FAILED|^E " gpurun_out/c57_tests.log | cut -c1-300 | head -20
grep -q "exit 0" gpurun_out/c57_tests.log || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --conv --no-cpu-baseline --no-also --no-in-step 2>gpurun_out/c57_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv', round(d['ms_per_step'],4), {k:v for k,v in d['roofline']['kernels_us_per_step'].items() if 'conv' in k})"
done
timeout -k 10 300 python bench.py --conv --kl gp --no-cpu-baseline --no-also --no-in-step 2>gpurun_out/c57_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convgp', round(d['ms_per_step'],4))"
