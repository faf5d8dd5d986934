cd $GRAFT_REPO_ROOT
echo "=== gp tests"; python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -x -q -m gpu -k "gp or config5 or shipped" > gpurun_out/r3_c25_tests.log 2>&1; tail -3 gpurun_out/r3_c25_tests.log
python tools/gp_phases.py 2>&1 | grep -v "^{" | tail -22
for i in 1 2; do
python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c25_a.json 2> gpurun_out/r3_c25_a.log || tail -5 gpurun_out/r3_c25_a.log
python tools/calls/show.py gpurun_out/r3_c25_a.json "gp default" | head -1 | cut -c1-330
done
