# GPU call: persistent wave-specialised optimiser kernel (ubench + bench), GP step timeline
cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub6.log 2>&1; grep "gemm_adam" gpurun_out/r3_ub6.log | grep -v "1-level" | head -40
echo "=== bench cfg1"
for cfg in "a 0 0" "b 1 1" "c 1 2" "a 0 0" "c 1 2" "d 0 1"; do
  set -- $cfg
  env HL_ADAM_PERSIST=$2 HL_ONE_SIDE=$3 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c9_$1.json 2> gpurun_out/r3_c9_$1.log || tail -5 gpurun_out/r3_c9_$1.log
  python tools/calls/show.py gpurun_out/r3_c9_$1.json "persist=$2 one_side=$3"
done
echo "=== GP timeline"
bash tools/trace_step.sh r3_gp --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -80
