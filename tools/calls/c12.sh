cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub8.log 2>&1; grep "gemm_adam" gpurun_out/r3_ub8.log | grep -v "1-level\|rowmap 0" | head -30
for cfg in "a 0" "b 1" "a 0" "b 1"; do
  set -- $cfg
  HL_ADAM_WIDE=$2 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c12_$1.json 2> gpurun_out/r3_c12_$1.log || tail -5 gpurun_out/r3_c12_$1.log
  python tools/calls/show.py gpurun_out/r3_c12_$1.json "wide=$2"
done
