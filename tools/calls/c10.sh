# GPU call: persistent kernel with 4 operand buffers; GP: early flush, kernel_matrix rewrite; GP tests
cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub7.log 2>&1; grep "gemm_adam" gpurun_out/r3_ub7.log | grep -v "1-level\|rowmap 0" | head -30
echo "=== bench cfg1"
for cfg in "a 0 0" "c 1 2" "b 1 1" "a 0 0"; do
  set -- $cfg
  env HL_ADAM_PERSIST=$2 HL_ONE_SIDE=$3 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c10_$1.json 2> gpurun_out/r3_c10_$1.log || tail -5 gpurun_out/r3_c10_$1.log
  python tools/calls/show.py gpurun_out/r3_c10_$1.json "persist=$2 one_side=$3"
done
echo "=== GP"
for v in 0 1 0 1; do
  HL_GP_FLUSH=$v python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c10_gp$v.json 2> gpurun_out/r3_c10_gp$v.log || tail -5 gpurun_out/r3_c10_gp$v.log
  python tools/calls/show.py gpurun_out/r3_c10_gp$v.json "gp flush=$v" | cut -c1-700
done
python bench.py --no-cpu-baseline --no-also --conv --kl gp > gpurun_out/r3_c10_convgp.json 2> gpurun_out/r3_c10_convgp.log; python tools/calls/show.py gpurun_out/r3_c10_convgp.json "conv+gp" | head -1 | cut -c1-100
echo "=== tests (gp)"
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py tests/test_dp_gpu2.py -m gpu -q -rf -k "gp or config5 or shipped or two_rank" > gpurun_out/r3_t10.log 2>&1; tail -6 gpurun_out/r3_t10.log
bash tools/trace_step.sh r3_gp2 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -62
