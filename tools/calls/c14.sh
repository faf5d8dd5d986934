cd $GRAFT_REPO_ROOT
echo "=== heads A/B"; for b in 512 4096; do HL_HEADS_CORE=1 python tools/heads_ab.py $b 2>/dev/null | tail -2; python tools/heads_ab.py $b 2>/dev/null | tail -2; done
echo "=== failing test x3"; for i in 1 2 3; do python -m pytest "tests/test_gpu_configs.py::test_sharded_optimizer_path_matches_fused_path" -x -q -m gpu 2>&1 | tail -2; done
echo "=== same with core 1"; HL_HEADS_CORE=1 python -m pytest "tests/test_gpu_configs.py::test_sharded_optimizer_path_matches_fused_path" -x -q -m gpu 2>&1 | tail -2
echo "=== gp tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py -x -q -m gpu -k "gp or config5 or GP" > gpurun_out/r3_c14_gp.log 2>&1; tail -3 gpurun_out/r3_c14_gp.log
python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c14_gp.json 2> gpurun_out/r3_c14_gpb.log || tail -5 gpurun_out/r3_c14_gpb.log
python tools/calls/show.py gpurun_out/r3_c14_gp.json "gp"
