#!/bin/bash
mkdir -p gpurun_out
for v in plain early ahead early_ahead defer defer_early defer_ahead defer_early_ahead defer_early_viamain defer_early_own defer_ahead_own defer_early_ahead_own defer_early_ahead_own_viamain defer_early_prefork defer_early_ahead_prefork defer_early_ahead_own_prefork; do
  timeout -k 10 120 python tools/repro/capture_forkjoin.py $v > gpurun_out/c46_one.log 2>&1
  echo "$v rc $? $(grep OK gpurun_out/c46_one.log | cut -c1-100)"
done
