#!/bin/bash
# build + run the stand-alone dense-kernel micro-benchmark on the GPU box:  bash tools/ubench/run.sh [args]
set -e
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-result $R/tools/ubench/ub_dense.hip -o /tmp/ub_dense
/tmp/ub_dense "$@"
