#!/bin/bash
# experiment driver: bench under each value of the environment variable $1 given, summary in gpurun_out/xp_summary.log
cd $GRAFT_REPO_ROOT
var=$1; shift
for x in "$@"; do
  env $var=$x python bench.py --no-cpu-baseline --steps 400 --warmup 40 > gpurun_out/xp_$x.log 2>&1 || exit 1
  echo "$var=$x $(tail -1 gpurun_out/xp_$x.log | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"])')" >> gpurun_out/xp_summary.log
done
