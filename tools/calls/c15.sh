cd $GRAFT_REPO_ROOT
echo "=== heads A/B"; for b in 512 4096; do HL_HEADS_CORE=1 python tools/heads_ab.py $b 2>&1 | tail -3; python tools/heads_ab.py $b 2>&1 | tail -3; done
echo "=== tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu > gpurun_out/r3_c15_tests.log 2>&1; tail -3 gpurun_out/r3_c15_tests.log
for cfg in "a 0" "b 1" "a 0" "b 1"; do
  set -- $cfg
  HL_GP_AHEAD=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c15_$1.json 2> gpurun_out/r3_c15_$1.log || tail -5 gpurun_out/r3_c15_$1.log
  python tools/calls/show.py gpurun_out/r3_c15_$1.json "gp ahead=$2"
done
