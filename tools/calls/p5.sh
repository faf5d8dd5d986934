# final round-3 evidence, part A (late): the whole GPU suite in ONE process (as the driver runs it), smoke(), PMC traffic passes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3g_suite.log 2>&1; echo "suite: $(tail -1 gpurun_out/r3g_suite.log)"
grep -q "passed" gpurun_out/r3g_suite.log && ! grep -q "failed" gpurun_out/r3g_suite.log || { tail -30 gpurun_out/r3g_suite.log; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
declare -A F
F[cfg1]=""
F[cfg2]="--workload d4 --rows 100000 --batch 4096"
F[cfg3]="--workload tabular --rows 1000000 --batch 4096"
F[cfg4]="--workload d4 --rows 50000 --batch 1024 --kl gp"
F[conv]="--conv"
F[convgp]="--conv --kl gp"
cd /tmp && export TMPDIR=/tmp
for c in cfg1 cfg2 cfg3 cfg4 conv convgp; do
  fl=${F[$c]}
  for g in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $R/gpurun_out/r3f_${c}_$g -o p -- python3 $R/bench.py $fl --no-graph --no-cpu-baseline --no-also --no-in-step --steps 20 --warmup 5 > $R/gpurun_out/r3f_${c}_$g.log 2>&1 || { echo "pmc $g $c failed"; tail -5 $R/gpurun_out/r3f_${c}_$g.log; exit 1; }
    cp $(find $R/gpurun_out/r3f_${c}_$g -name "*counter_collection.csv" | head -1) $R/gpurun_out/r3f_${c}_$g.csv
    rm -rf $R/gpurun_out/r3f_${c}_$g
  done
  echo "pmc $c done"
done
