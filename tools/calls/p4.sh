# final round-3 evidence, part B: bench lines (with cpu_baseline, also, traffic), rocprofv3 kernel stats, step timelines
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
declare -A F
F[cfg1]=""
F[cfg2]="--workload d4 --rows 100000 --batch 4096"
F[cfg3]="--workload tabular --rows 1000000 --batch 4096"
F[cfg4]="--workload d4 --rows 50000 --batch 1024 --kl gp"
F[conv]="--conv"
F[convgp]="--conv --kl gp"
F[sharded]="--sharded"
cd /tmp && export TMPDIR=/tmp
for c in ${@:-cfg1 cfg2 cfg3 cfg4 conv convgp sharded}; do
  fl=${F[$c]}
  extra="--no-also"; [ "$c" = "cfg1" ] && extra=""
  timeout -k 10 600 python3 $R/bench.py $fl $extra --tag r3_$c > $R/gpurun_out/r3_${c}_bench.json 2> $R/gpurun_out/r3_${c}_bench.log || { echo "bench $c failed"; tail -5 $R/gpurun_out/r3_${c}_bench.log; exit 1; }
  python3 $R/tools/calls/show.py $R/gpurun_out/r3_${c}_bench.json "$c" 2>/dev/null | cut -c1-260
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_${c}_stats -o s -- python3 $R/bench.py $fl --no-cpu-baseline --no-also > $R/gpurun_out/r3_${c}_stats.log 2>&1 || { echo "stats $c failed"; tail -5 $R/gpurun_out/r3_${c}_stats.log; exit 1; }
  cp $(find $R/gpurun_out/r3_${c}_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r3_${c}_kernel_stats.csv
  rm -rf $R/gpurun_out/r3_${c}_stats
done
cd $R
bash tools/trace_step.sh r3_cfg1 > /dev/null 2>&1; cp gpurun_out/r3_cfg1_step_stats.txt gpurun_out/r3_cfg1_step_timeline.txt; head -3 gpurun_out/r3_cfg1_step_timeline.txt | cut -c1-200
bash tools/trace_step.sh r3_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp > /dev/null 2>&1; cp gpurun_out/r3_cfg4_step_stats.txt gpurun_out/r3_cfg4_step_timeline.txt; head -3 gpurun_out/r3_cfg4_step_timeline.txt | cut -c1-200
