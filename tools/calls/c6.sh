# GPU call: where does the in-bench slowdown of the fused gradient + optimiser kernel come from? (ubench, both trees); one-side variant; GP fusion A/B
cd $GRAFT_REPO_ROOT
echo "=== ubench new tree"; bash tools/ubench/run.sh > gpurun_out/r3_ub4.log 2>&1; grep -v "old vs" gpurun_out/r3_ub4.log
echo "=== ubench old tree"; UB_OLD=1 bash tools/ubench/run.sh > gpurun_out/r3_ub4_old.log 2>&1; grep -v "old vs" gpurun_out/r3_ub4_old.log
echo "=== bench"
(cd .old && python bench.py --no-cpu-baseline --steps 400 --warmup 40 > ../gpurun_out/r3_c6_old.json 2> ../gpurun_out/r3_c6_old.log); python tools/calls/show.py gpurun_out/r3_c6_old.json "old tree"
python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c6_new.json 2> gpurun_out/r3_c6_new.log; python tools/calls/show.py gpurun_out/r3_c6_new.json "new dma"
HL_ONE_SIDE=1 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c6_one.json 2> gpurun_out/r3_c6_one.log; python tools/calls/show.py gpurun_out/r3_c6_one.json "new dma one-side"
echo "=== GP"
for v in 0 1 0 1; do
  HL_GP_FUSE=$v python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c6_gp$v.json 2> gpurun_out/r3_c6_gp$v.log || tail -5 gpurun_out/r3_c6_gp$v.log
  python tools/calls/show.py gpurun_out/r3_c6_gp$v.json "gp fuse=$v"
done
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gp_prior.py tests/test_dp_gpu2.py -m gpu -q -rf -k "gp or config5 or two_rank" > gpurun_out/r3_t6.log 2>&1; tail -6 gpurun_out/r3_t6.log
