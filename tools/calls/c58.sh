#!/bin/bash
mkdir -p gpurun_out
run() {
  env "$@" timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also --no-in-step > gpurun_out/c58_out.log 2> gpurun_out/c58_err.log
  echo "$* rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c58_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" 2>/dev/null)"
}
for i in 1 2 3; do
run HL_GP_SPLIT_PREP=0
run HL_GP_SPLIT_PREP=1
done
env HL_GP_SPLIT_PREP=0 timeout -k 10 300 python bench.py --conv --kl gp --no-cpu-baseline --no-also --no-in-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convgp split=0', round(d['ms_per_step'],4))"
env HL_GP_SPLIT_PREP=1 timeout -k 10 300 python bench.py --conv --kl gp --no-cpu-baseline --no-also --no-in-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convgp split=1', round(d['ms_per_step'],4))"
