cd $GRAFT_REPO_ROOT
echo "=== parity"; python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3_c13_tests.log 2>&1; tail -3 gpurun_out/r3_c13_tests.log
echo "=== phases"; HL_HEADS_CORE=1 python tools/heads_phases.py 512 2>/dev/null | tail -14; python tools/heads_phases.py 512 2>/dev/null | tail -14; python tools/heads_phases.py 4096 2>/dev/null | tail -14
for cfg in "a 1" "b 2" "a 1" "b 2"; do
  set -- $cfg
  HL_HEADS_CORE=$2 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c13_$1.json 2> gpurun_out/r3_c13_$1.log || tail -5 gpurun_out/r3_c13_$1.log
  python tools/calls/show.py gpurun_out/r3_c13_$1.json "heads core=$2"
done
for cfg in "c 1" "d 2"; do
  set -- $cfg
  HL_HEADS_CORE=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 100000 --batch 4096 --steps 200 --warmup 20 > gpurun_out/r3_c13_$1.json 2> gpurun_out/r3_c13_$1.log || tail -5 gpurun_out/r3_c13_$1.log
  python tools/calls/show.py gpurun_out/r3_c13_$1.json "cfg2 heads core=$2"
done
