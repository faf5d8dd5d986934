#!/bin/bash
mkdir -p gpurun_out
run() {
  env "$@" timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also --no-in-step > gpurun_out/c48_out.log 2> gpurun_out/c48_err.log
  echo "$* rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c48_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" 2>/dev/null)"
}
for i in 1 2 3; do
run HL_GP_EARLY=0
run HL_GP_EARLY=1 HL_GP_FORK_DIRECT=1
run HL_GP_EARLY=1 HL_GP_FORK_DIRECT=0
done
