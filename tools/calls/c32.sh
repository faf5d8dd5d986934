cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "gp or config5" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload d4 --rows 50000 --batch 1024 --kl gp --no-also --tag r3_cfg4 > $R/gpurun_out/r3_cfg4_bench.json 2> $R/gpurun_out/r3_cfg4_bench.log; python3 $R/tools/calls/show.py $R/gpurun_out/r3_cfg4_bench.json cfg4 | head -1 | cut -c1-700
python3 $R/bench.py --conv --kl gp --no-also --tag r3_convgp > $R/gpurun_out/r3_convgp_bench.json 2> $R/gpurun_out/r3_convgp_bench.log; python3 $R/tools/calls/show.py $R/gpurun_out/r3_convgp_bench.json convgp | head -1 | cut -c1-400
python3 $R/bench.py --tag r3_cfg1 > $R/gpurun_out/r3_cfg1_bench.json 2> $R/gpurun_out/r3_cfg1_bench.log; python3 $R/tools/calls/show.py $R/gpurun_out/r3_cfg1_bench.json cfg1 | cut -c1-300
