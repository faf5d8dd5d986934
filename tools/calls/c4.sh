# GPU call: old tree (round-2 final) vs new tree on ONE box; GPU tests file by file
cd $GRAFT_REPO_ROOT
echo "=== old tree vs new"
for r in 1 2; do
  (cd .old && python bench.py --no-cpu-baseline --steps 400 --warmup 40 > ../gpurun_out/r3_c4_old_$r.json 2> ../gpurun_out/r3_c4_old_$r.log) || tail -5 gpurun_out/r3_c4_old_$r.log
  python tools/calls/show.py gpurun_out/r3_c4_old_$r.json "old tree"
  for c in nt dma; do
    HL_GEMM_CORE=$c python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c4_${c}_$r.json 2> gpurun_out/r3_c4_${c}_$r.log || tail -5 gpurun_out/r3_c4_${c}_$r.log
    python tools/calls/show.py gpurun_out/r3_c4_${c}_$r.json "new $c"
  done
done
HL_SLAB_NT=1 python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c4_slabnt.json 2> gpurun_out/r3_c4_slabnt.log; python tools/calls/show.py gpurun_out/r3_c4_slabnt.json "new dma slab-nt"
python bench.py --no-cpu-baseline --no-also --no-in-step --steps 400 --warmup 40 > gpurun_out/r3_c4_nostamp.json 2> gpurun_out/r3_c4_nostamp.log; python tools/calls/show.py gpurun_out/r3_c4_nostamp.json "new dma no stamps"
echo "=== tests, one process per file"
for f in tests/test_*.py; do
  b=$(basename $f .py)
  timeout -k 10 900 python -m pytest $f -m gpu -q -rf > gpurun_out/r3_t4_$b.log 2>&1
  echo "$b: $(tail -1 gpurun_out/r3_t4_$b.log)"
  grep -E "^FAILED|Segmentation|Error" gpurun_out/r3_t4_$b.log | head -8
done
