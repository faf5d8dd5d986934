cd $GRAFT_REPO_ROOT
echo "=== gp tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py tests/test_dp_gpu2.py -x -q -m gpu -k "gp or config5 or shipped or two_ranks" > gpurun_out/r3_c27_tests.log 2>&1; tail -3 gpurun_out/r3_c27_tests.log
for cfg in "a 0" "b 1" "a 0" "b 1"; do
  set -- $cfg
  HL_GP_AHEAD_SUBJECT=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c27_$1.json 2> gpurun_out/r3_c27_$1.log || tail -5 gpurun_out/r3_c27_$1.log
  python tools/calls/show.py gpurun_out/r3_c27_$1.json "gp ahead_subject=$2" | head -1 | cut -c1-330
done
bash tools/trace_step.sh r3i_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -45
