cd $GRAFT_REPO_ROOT
python tools/gp_phases.py 2>&1 | grep -v "^{" | tail -30
for i in 1 2; do
python bench.py --no-cpu-baseline --no-also --workload d4 --rows 50000 --batch 1024 --kl gp --steps 200 --warmup 20 > gpurun_out/r3_c24_a.json 2> gpurun_out/r3_c24_a.log || tail -5 gpurun_out/r3_c24_a.log
python tools/calls/show.py gpurun_out/r3_c24_a.json "gp default" | head -1 | cut -c1-330
done
