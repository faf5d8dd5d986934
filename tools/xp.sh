#!/bin/bash
# experiment driver: bench under each HLVAE_X value given, then a kernel trace of the last one
cd $GRAFT_REPO_ROOT
for x in "$@"; do
  HLVAE_X=$x python bench.py --no-cpu-baseline --steps 300 --warmup 30 > gpurun_out/xp_$x.log 2>&1 || exit 1
  echo "X=$x $(tail -1 gpurun_out/xp_$x.log | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"])')" >> gpurun_out/xp_summary.log
done
