cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gp" > gpurun_out/r3_c31_tests.log 2>&1; tail -2 gpurun_out/r3_c31_tests.log
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "gp or config5" 2>&1 | tail -1
for cfg in "a 0" "b 1" "a 0" "b 1"; do
  set -- $cfg
  HL_GP_FGEMM=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c31_$1.json 2> gpurun_out/r3_c31_$1.log || tail -5 gpurun_out/r3_c31_$1.log
  python tools/calls/show.py gpurun_out/r3_c31_$1.json "gp fgemm=$2" | head -1 | cut -c1-260
done
HL_GP_SERIAL=1 HL_GP_FGEMM=1 python bench.py --no-cpu-baseline --no-also --no-graph --no-in-step --workload d4 --rows 50000 --batch 1024 --kl gp --steps 50 --warmup 10 > gpurun_out/r3_c31_serial.json 2> /dev/null; python tools/calls/show.py gpurun_out/r3_c31_serial.json "gp serial" | head -1 | cut -c1-500
