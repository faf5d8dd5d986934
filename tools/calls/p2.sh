# profiles of round 3, part 2: configs[4], convolutional, conv + GP, sharded
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r3 cfg4 conv convgp sharded 2>&1 | tail -40
bash tools/trace_step.sh r3_cfg4 --workload d4 --rows 50000 --batch 1024 --kl gp 2>&1 | tail -5
cp gpurun_out/r3_cfg4_step_stats.txt gpurun_out/r3_cfg4_step_timeline.txt
