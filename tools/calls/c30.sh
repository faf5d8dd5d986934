cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3_c30_tests.log 2>&1; tail -2 gpurun_out/r3_c30_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-also --tag r3_cfg4 > $R/gpurun_out/r3_cfg4_bench.json 2> $R/gpurun_out/r3_cfg4_bench.log; python3 $R/tools/calls/show.py $R/gpurun_out/r3_cfg4_bench.json cfg4 | head -1 | cut -c1-200
python3 $R/bench.py --conv --kl gp --no-also --tag r3_convgp > $R/gpurun_out/r3_convgp_bench.json 2> $R/gpurun_out/r3_convgp_bench.log; python3 $R/tools/calls/show.py $R/gpurun_out/r3_convgp_bench.json convgp | head -1 | cut -c1-200
