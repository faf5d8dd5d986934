#!/bin/bash
mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-also --no-in-step > gpurun_out/c50_out.log 2> gpurun_out/c50_err.log
  echo "$* rc $? $(python -c "import json; d=json.loads(open('gpurun_out/c50_out.log').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4))" 2>/dev/null)"
}
for i in 1 2; do
for ch in 8 16 32; do run --chain $ch; done
done
for ch in 2 8 16; do run --workload d4 --rows 50000 --batch 1024 --kl gp --chain $ch; done
