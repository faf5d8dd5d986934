cd $GRAFT_REPO_ROOT
echo "=== gp tests (wave GJ)"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or shipped" > gpurun_out/r3_c23_tests.log 2>&1; tail -3 gpurun_out/r3_c23_tests.log
for cfg in "a 1 1" "b 2 1" "c 2 0" "a 1 1" "b 2 1" "c 2 0"; do
  set -- $cfg
  HL_GP_BALANCE=$2 HL_GP_LATE_JOIN=$3 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c23_$1.json 2> gpurun_out/r3_c23_$1.log || tail -5 gpurun_out/r3_c23_$1.log
  python tools/calls/show.py gpurun_out/r3_c23_$1.json "gp balance=$2 late_join=$3" | head -1 | cut -c1-330
done
bash tools/trace_step.sh r3g_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -45
echo "=== serial (alone) GP kernel times"
HL_GP_SERIAL=1 python bench.py --no-cpu-baseline --no-also --no-graph --no-in-step --workload d4 --rows 50000 --batch 1024 --kl gp --steps 50 --warmup 10 > gpurun_out/r3_c23_serial.json 2> gpurun_out/r3_c23_serial.log || tail -5 gpurun_out/r3_c23_serial.log
python tools/calls/show.py gpurun_out/r3_c23_serial.json "gp serial" | head -1 | cut -c1-900
