#!/bin/bash
mkdir -p gpurun_out
for ch in 2 4; do
  for e in 0 1; do
    HL_GP_DEFER=$e timeout -k 10 300 python bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also --no-in-step --chain $ch > gpurun_out/c43_out.log 2> gpurun_out/c43_err.log
    echo "chain $ch defer $e rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c43_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('final_nll_sum'))" 2>/dev/null)"
  done
done
