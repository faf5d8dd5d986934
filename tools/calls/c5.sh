# GPU call: tile shapes of the fused gradient + optimiser kernel (ubench), old tree vs new on one box, the tests that failed
cd $GRAFT_REPO_ROOT
echo "=== ubench"; bash tools/ubench/run.sh > gpurun_out/r3_ub3.log 2>&1; cat gpurun_out/r3_ub3.log
echo "=== old tree vs new"
for r in 1 2; do
  (cd .old && python bench.py --no-cpu-baseline --steps 400 --warmup 40 > ../gpurun_out/r3_c5_old_$r.json 2> ../gpurun_out/r3_c5_old_$r.log) || tail -5 gpurun_out/r3_c5_old_$r.log
  python tools/calls/show.py gpurun_out/r3_c5_old_$r.json "old tree"
  for c in nt dma; do
    HL_GEMM_CORE=$c python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c5_${c}_$r.json 2> gpurun_out/r3_c5_${c}_$r.log || tail -5 gpurun_out/r3_c5_${c}_$r.log
    python tools/calls/show.py gpurun_out/r3_c5_${c}_$r.json "new $c"
  done
done
for t in 1 2 3; do
  HL_ADAM_TILE=$t python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 > gpurun_out/r3_c5_tile$t.json 2> gpurun_out/r3_c5_tile$t.log || tail -5 gpurun_out/r3_c5_tile$t.log
  python tools/calls/show.py gpurun_out/r3_c5_tile$t.json "new dma adam-tile $t"
done
python bench.py --no-cpu-baseline --no-also --no-in-step --steps 400 --warmup 40 > gpurun_out/r3_c5_nostamp.json 2> gpurun_out/r3_c5_nostamp.log; python tools/calls/show.py gpurun_out/r3_c5_nostamp.json "new dma no stamps"
python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > gpurun_out/r3_c5_s20.json 2> gpurun_out/r3_c5_s20.log; python tools/calls/show.py gpurun_out/r3_c5_s20.json "new dma 20 steps"
echo "=== tests"
timeout -k 10 900 python -m pytest tests/test_trajectory_gpu.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -rf > gpurun_out/r3_t5.log 2>&1; tail -12 gpurun_out/r3_t5.log
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/parity_report_trajectory.json"))
    for k, v in j.items():
        print(k, {a: (b if not isinstance(b, (dict, list)) else (max(b.values()) if isinstance(b, dict) else max(b))) for a, b in v.items()})
        if "update_err" in v:
            print("   update_err:", {a: round(b, 4) for a, b in sorted(v["update_err"].items(), key=lambda kv: -kv[1])[:12]})
except Exception as e:
    print("no trajectory report", e)
PY
