# GPU call: two-level completion tickets; stagger; one-side queue; GP prepare() under the forward pass; tests
cd $GRAFT_REPO_ROOT
echo "=== ubench new tree"; bash tools/ubench/run.sh > gpurun_out/r3_ub5.log 2>&1; grep "gemm_adam wy" gpurun_out/r3_ub5.log | grep -v "old vs"
echo "=== bench"
(cd .old && python bench.py --no-cpu-baseline --steps 400 --warmup 40 > ../gpurun_out/r3_c7_old.json 2> ../gpurun_out/r3_c7_old.log); python tools/calls/show.py gpurun_out/r3_c7_old.json "old tree"
for cfg in "a 1 0" "b 0 0" "c 1 1" "d 0 1" "a 1 0" "c 1 1"; do
  set -- $cfg
  env HL_ADAM_STAGGER=$2 $( [ "$3" = "1" ] && echo HL_ONE_SIDE=1 ) python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c7_$1.json 2> gpurun_out/r3_c7_$1.log || tail -5 gpurun_out/r3_c7_$1.log
  python tools/calls/show.py gpurun_out/r3_c7_$1.json "new stagger=$2 one_side=$3"
done
echo "=== GP"
for cfg in "0 0" "1 0" "1 1" "1 1"; do
  set -- $cfg
  HL_GP_FUSE=$1 HL_GP_PREPARE=$2 python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c7_gp$1$2.json 2> gpurun_out/r3_c7_gp$1$2.log || tail -5 gpurun_out/r3_c7_gp$1$2.log
  python tools/calls/show.py gpurun_out/r3_c7_gp$1$2.json "gp fuse=$1 prepare=$2"
done
echo "=== tests"
for f in tests/test_*.py; do
  b=$(basename $f .py)
  timeout -k 10 900 python -m pytest $f -m gpu -q -rf > gpurun_out/r3_t7_$b.log 2>&1
  echo "$b: $(tail -1 gpurun_out/r3_t7_$b.log)"
  grep -E "^FAILED|Segmentation|^E  " gpurun_out/r3_t7_$b.log | head -8
done
