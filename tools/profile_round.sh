#!/bin/bash
# One gpurun call: bench lines, rocprofv3 kernel stats and the PMC traffic passes of every BASELINE configuration at N = 1.
#   usage (on the GPU box): bash tools/profile_round.sh <tag> [cfg ...]       cfg in: cfg1 cfg2 cfg3 cfg4 conv convgp sharded
# Outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards (tools/pmc_traffic.py builds the
# traffic table from the *_fetch / *_write directories).
set -o pipefail
R=$GRAFT_REPO_ROOT
tag=$1; shift
cfgs=${@:-cfg1 cfg2 cfg3 cfg4}
declare -A F
F[cfg1]=""
F[cfg2]="--workload d4 --rows 100000 --batch 4096"
F[cfg3]="--workload tabular --rows 1000000 --batch 4096"
F[cfg4]="--workload d4 --rows 50000 --batch 1024 --kl gp"
F[conv]="--conv"
F[convgp]="--conv --kl gp"
F[sharded]="--sharded"
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
  fl=${F[$c]}
  echo "== $c: bench ($fl)"
  timeout -k 10 600 python3 $R/bench.py $fl --tag ${tag}_$c > $R/gpurun_out/${tag}_${c}_bench.json 2> $R/gpurun_out/${tag}_${c}_bench.log || { echo "bench $c failed"; tail -5 $R/gpurun_out/${tag}_${c}_bench.log; exit 1; }
  tail -c 400 $R/gpurun_out/${tag}_${c}_bench.json; echo
  echo "== $c: rocprofv3 --kernel-trace --stats"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_${c}_stats -o s -- python3 $R/bench.py $fl --no-cpu-baseline --no-also > $R/gpurun_out/${tag}_${c}_stats.log 2>&1 || { echo "stats $c failed"; tail -5 $R/gpurun_out/${tag}_${c}_stats.log; exit 1; }
  cp $(find $R/gpurun_out/${tag}_${c}_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_${c}_kernel_stats.csv
  if [ "$c" != "sharded" ]; then
    for g in FETCH_SIZE WRITE_SIZE; do
      echo "== $c: --pmc $g"
      timeout -k 10 400 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_${c}_$g -o p -- python3 $R/bench.py $fl --no-graph --no-cpu-baseline --no-also --no-in-step --steps 20 --warmup 5 > $R/gpurun_out/${tag}_${c}_$g.log 2>&1 || { echo "pmc $g $c failed"; tail -5 $R/gpurun_out/${tag}_${c}_$g.log; exit 1; }
      cp $(find $R/gpurun_out/${tag}_${c}_$g -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_${c}_$g.csv
    done
  fi
  rm -rf $R/gpurun_out/${tag}_${c}_stats $R/gpurun_out/${tag}_${c}_FETCH_SIZE $R/gpurun_out/${tag}_${c}_WRITE_SIZE
done
echo done
