#!/bin/bash
# build + run the stand-alone dense-kernel micro-benchmark on the GPU box:  bash tools/ubench/run.sh [args]
# (UB_OLD=1: the same harness against the round-2 sources under .old/, when that copy exists)
set -e
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
F="-O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -Wno-unused-result"
if [ -n "$UB_OLD" ]; then
  sed "s#../../hl-vae_amd/csrc/dense.hip#$R/.old/hl-vae_amd/csrc/dense.hip#" $R/tools/ubench/ub_dense.hip > /tmp/ub_dense_old.hip
  /opt/rocm/bin/hipcc $F -DUB_OLD_TREE /tmp/ub_dense_old.hip -o /tmp/ub_dense_old
  /tmp/ub_dense_old "$@"
else
  /opt/rocm/bin/hipcc $F $R/tools/ubench/ub_dense.hip -o /tmp/ub_dense
  /tmp/ub_dense "$@"
fi
