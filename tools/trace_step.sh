#!/bin/bash
# kernel trace of the default bench + the timeline of one replayed step: bash tools/trace_step.sh <tag> [bench flags]
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-also --no-in-step --steps 200 --warmup 20 "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 1; }
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f k_mid_fwd_fused 150 > $R/gpurun_out/${tag}_timeline.txt
python3 $R/tools/step_stats.py $f > $R/gpurun_out/${tag}_step_stats.txt
rm -rf $R/gpurun_out/${tag}_trace
tail -2 $R/gpurun_out/${tag}_trace.log | cut -c1-200
cat $R/gpurun_out/${tag}_step_stats.txt
