cd $GRAFT_REPO_ROOT
python tools/gp_chain_bench.py 120 2>&1 | tail -7
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gp" 2>&1 | tail -2
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "gp or config5 or head_kernel_cores" 2>&1 | tail -2
for i in 1 2; do
python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c35_a.json 2> gpurun_out/r3_c35_a.log || tail -5 gpurun_out/r3_c35_a.log
python tools/calls/show.py gpurun_out/r3_c35_a.json "gp" | head -1 | cut -c1-420
done
