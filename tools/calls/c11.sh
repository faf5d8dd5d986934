cd $GRAFT_REPO_ROOT
python tools/capture_variance.py 10 2>&1 | tail -16
python bench.py --steps 100 --warmup 10 > gpurun_out/r3_c11_default.json 2> gpurun_out/r3_c11_default.log; python tools/calls/show.py gpurun_out/r3_c11_default.json "default (with cpu baseline + also)"; tail -c 1500 gpurun_out/r3_c11_default.json
echo; echo "=== all gpu tests"
for f in tests/test_*.py; do
  b=$(basename $f .py)
  timeout -k 10 900 python -m pytest $f -m gpu -q -rf > gpurun_out/r3_t11_$b.log 2>&1
  echo "$b: $(tail -1 gpurun_out/r3_t11_$b.log)"
  grep -E "^FAILED|Segmentation|^E  " gpurun_out/r3_t11_$b.log | head -8
done
