# GPU call: micro-benchmark, A/B of the GEMM cores with / without stamps, GP inversion A/B, all GPU tests
cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub2.log 2>&1; cat gpurun_out/r3_ub2.log
echo "=== A/B core x stamps"
i=0
for cfg in "nt on" "dma on" "dma off" "nt off" "dma on" "dma off"; do
  set -- $cfg; c=$1; st=$2; i=$((i+1))
  fl=""; [ "$st" = "off" ] && fl="--no-in-step"
  HL_GEMM_CORE=$c python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 $fl > gpurun_out/r3_ab3_$i.json 2> gpurun_out/r3_ab3_$i.log || tail -20 gpurun_out/r3_ab3_$i.log
  python tools/calls/show.py gpurun_out/r3_ab3_$i.json "$c stamps=$st"
done
echo "=== cfg4 GP inversion A/B"
for v in unblocked blocked unblocked blocked; do
  HL_GP_SPD=$v python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_gp_$v.json 2> gpurun_out/r3_gp_$v.log || tail -20 gpurun_out/r3_gp_$v.log
  python tools/calls/show.py gpurun_out/r3_gp_$v.json "spd=$v"
done
echo "=== tests"; python -m pytest tests -m gpu -q > gpurun_out/r3_tests3.log 2>&1; tail -40 gpurun_out/r3_tests3.log
