# profiles of round 3, part 1: configs[1..3]
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r3 cfg1 cfg2 cfg3 2>&1 | tail -30
bash tools/trace_step.sh r3_cfg1 2>&1 | tail -25
cp gpurun_out/r3_cfg1_step_stats.txt gpurun_out/r3_cfg1_step_timeline.txt
