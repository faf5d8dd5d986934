cd $GRAFT_REPO_ROOT
echo "=== gp tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or shipped" > gpurun_out/r3_c18_tests.log 2>&1; tail -3 gpurun_out/r3_c18_tests.log
for cfg in "a 0 0 0" "b 0 1 0" "c 1 1 0" "d 1 1 1" "a 0 0 0" "b 0 1 0" "c 1 1 0" "d 1 1 1"; do
  set -- $cfg
  HL_GP_CHAIN=$2 HL_GP_SPLIT=$3 HL_GP_DEFER=$4 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c18_$1.json 2> gpurun_out/r3_c18_$1.log || tail -5 gpurun_out/r3_c18_$1.log
  python tools/calls/show.py gpurun_out/r3_c18_$1.json "gp chain=$2 split=$3 defer=$4" | head -1 | cut -c1-220
done
bash tools/trace_step.sh r3c_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -60
