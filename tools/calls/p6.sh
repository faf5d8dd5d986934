# flakiness check: the whole GPU suite in one process, twice more
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3h_suite_$i.log 2>&1; echo "suite $i: $(tail -1 gpurun_out/r3h_suite_$i.log)"
  grep -q "failed" gpurun_out/r3h_suite_$i.log && { grep -E "FAILED|^E " gpurun_out/r3h_suite_$i.log | head -10; }
done
true
