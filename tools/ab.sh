#!/bin/bash
# A/B of environment switches on one box: bash tools/ab.sh "<bench flags>" VAR=val ... ("-" = no variable); two rounds, alternating
cd $GRAFT_REPO_ROOT
flags=$1; shift
for round in 1 2; do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then v=$(python bench.py --no-cpu-baseline --steps 400 --warmup 40 $flags 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
    else v=$(env $kv python bench.py --no-cpu-baseline --steps 400 --warmup 40 $flags 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])'); fi
    echo "round $round  $kv  $v"
  done
done
