cd $GRAFT_REPO_ROOT
echo "=== gp tests (512-thread inverse)"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py -x -q -m gpu -k "gp or config5" > gpurun_out/r3_c16_tests.log 2>&1; tail -3 gpurun_out/r3_c16_tests.log
for cfg in "a 256" "b 512" "a 256" "b 512"; do
  set -- $cfg
  HL_GP_INV=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c16_$1.json 2> gpurun_out/r3_c16_$1.log || tail -5 gpurun_out/r3_c16_$1.log
  python tools/calls/show.py gpurun_out/r3_c16_$1.json "gp inv=$2"
done
bash tools/trace_step.sh r3b_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -60
