#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp_ or config5 or GP or conv_gp or prior" > gpurun_out/c63_tests.log 2>&1
echo "exit $? $(grep -E 'passed|failed' gpurun_out/c63_tests.log | tail -1)"
grep -q "passed" gpurun_out/c63_tests.log && ! grep -q "failed" gpurun_out/c63_tests.log || { grep -E "FAILED|^E " gpurun_out/c63_tests.log | head; exit 1; }
run() {
  env "$@" timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also --no-in-step > gpurun_out/c63_out.log 2> gpurun_out/c63_err.log
  echo "$* rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c63_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" 2>/dev/null)"
}
for i in 1 2 3; do
run HL_GP_FOLD_CLEAR=0
run HL_GP_FOLD_CLEAR=1
done
