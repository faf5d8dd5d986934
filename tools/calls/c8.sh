# GPU call: merged optimiser launch, split-K 4, head-kernel phases, GP deferred state update, other configs
cd $GRAFT_REPO_ROOT
echo "=== bench cfg1"
for cfg in "a 0 -" "b 1 -" "c 2 -" "d 0 4" "e 2 4" "a 0 -" "c 2 -"; do
  set -- $cfg
  env HL_ONE_SIDE=$2 $( [ "$3" != "-" ] && echo HL_SPLITK=$3 ) python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c8_$1.json 2> gpurun_out/r3_c8_$1.log || tail -5 gpurun_out/r3_c8_$1.log
  python tools/calls/show.py gpurun_out/r3_c8_$1.json "one_side=$2 splitk=$3"
done
echo "=== heads phases"
HL_GEMM_CORE=nt python tools/heads_phases.py 512 2>&1 | tail -14
python tools/heads_phases.py 512 2>&1 | tail -14
echo "=== GP"
for cfg in "1 0" "1 1" "1 0" "1 1"; do
  set -- $cfg
  HL_GP_PREPARE=$1 HL_GP_DEFER=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c8_gp$1$2.json 2> gpurun_out/r3_c8_gp$1$2.log || tail -5 gpurun_out/r3_c8_gp$1$2.log
  python tools/calls/show.py gpurun_out/r3_c8_gp$1$2.json "gp prepare=$1 defer=$2"
done
echo "=== other configs"
python bench.py --no-cpu-baseline --no-also --workload d4 --rows 100000 --batch 4096 > gpurun_out/r3_c8_cfg2.json 2> gpurun_out/r3_c8_cfg2.log; python tools/calls/show.py gpurun_out/r3_c8_cfg2.json "cfg2"
python bench.py --no-cpu-baseline --no-also --workload tabular --rows 1000000 --batch 4096 > gpurun_out/r3_c8_cfg3.json 2> gpurun_out/r3_c8_cfg3.log; python tools/calls/show.py gpurun_out/r3_c8_cfg3.json "cfg3"
python bench.py --no-cpu-baseline --no-also --conv > gpurun_out/r3_c8_conv.json 2> gpurun_out/r3_c8_conv.log; python tools/calls/show.py gpurun_out/r3_c8_conv.json "conv"
python bench.py --no-cpu-baseline --no-also --conv --kl gp > gpurun_out/r3_c8_convgp.json 2> gpurun_out/r3_c8_convgp.log; python tools/calls/show.py gpurun_out/r3_c8_convgp.json "conv+gp"
python bench.py --no-cpu-baseline --no-also --sharded > gpurun_out/r3_c8_sharded.json 2> gpurun_out/r3_c8_sharded.log; python tools/calls/show.py gpurun_out/r3_c8_sharded.json "sharded"
echo "=== tests (gp)"
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_trajectory_gpu.py -m gpu -q -rf -k "gp or config5 or shipped" > gpurun_out/r3_t8.log 2>&1; tail -6 gpurun_out/r3_t8.log
