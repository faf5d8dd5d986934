# GPU call: micro-benchmark of the dense kernels, A/B of the two GEMM cores in the full step, GPU tests
cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub1.log 2>&1; cat gpurun_out/r3_ub1.log
echo "=== A/B core"
for r in 1 2; do for c in nt dma; do
  HL_GEMM_CORE=$c python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_ab_${c}_$r.json 2> gpurun_out/r3_ab_${c}_$r.log || { tail -20 gpurun_out/r3_ab_${c}_$r.log; }
  python tools/calls/show.py gpurun_out/r3_ab_${c}_$r.json "$c $r"
done; done
echo "=== tests"; python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests2.log 2>&1; tail -15 gpurun_out/r3_tests2.log
