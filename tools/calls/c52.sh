#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/diag/conv_deep_paths.py deep > gpurun_out/c52.log 2>&1
echo "rc $?"; cut -c1-200 gpurun_out/c52.log | tail -45
