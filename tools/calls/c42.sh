#!/bin/bash
mkdir -p gpurun_out
HL_GP_DEFER=1 timeout -k 10 300 python -X faulthandler bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-cpu-baseline --no-also > gpurun_out/c42_out.log 2> gpurun_out/c42_err.log
echo "rc $?"
tail -40 gpurun_out/c42_err.log | cut -c1-200
