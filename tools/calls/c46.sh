#!/bin/bash
mkdir -p gpurun_out
for v in defer_early_viamain defer_ahead_kmain defer_ahead_konc defer_early_viamain_ahead_kmain defer_early_viamain_ahead_konc; do
  timeout -k 10 120 python tools/repro/capture_forkjoin.py $v > gpurun_out/c46_one.log 2>&1
  echo "$v rc $? $(grep OK gpurun_out/c46_one.log | cut -c1-100)"
done
