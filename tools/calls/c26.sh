cd $GRAFT_REPO_ROOT
python tools/gp_chain_bench.py 120 2>&1 | tail -16
echo "=== gp tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or shipped" > gpurun_out/r3_c26_tests.log 2>&1; tail -3 gpurun_out/r3_c26_tests.log
for cfg in "a 1 2" "b 2 2" "c 2 1" "a 1 2" "b 2 2" "c 2 1"; do
  set -- $cfg
  HL_GP_CHAIN=$2 HL_GP_BALANCE=$3 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c26_$1.json 2> gpurun_out/r3_c26_$1.log || tail -5 gpurun_out/r3_c26_$1.log
  python tools/calls/show.py gpurun_out/r3_c26_$1.json "gp chain=$2 balance=$3" | head -1 | cut -c1-330
done
bash tools/trace_step.sh r3h_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -45
