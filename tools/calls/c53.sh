#!/bin/bash
mkdir -p gpurun_out
run() {
  env "$@" timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also --no-in-step > gpurun_out/c53_out.log 2> gpurun_out/c53_err.log
  echo "$* rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c53_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" 2>/dev/null) $(tail -2 gpurun_out/c53_err.log | cut -c1-150 | tr '\n' ' ')"
}
for i in 1 2; do
run HL_GP_GATE=0
run HL_GP_GATE=1
run HL_GP_GATE=2
done
run HL_GP_GATE=2 HL_GP_DEFER=1
