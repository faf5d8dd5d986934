"""The NATIVE (RCCL) branch of the data-parallel step: async reduce_scatter_tensor / all_gather_into_tensor issued from the
trainer's side streams, their handles waited on from other streams, the in-place gather the first Linear's shadow aliases into,
and capture with capture_error_mode="thread_local".  Needs one GPU per rank: SKIPPED on the one-GPU boxes this repository is
developed on -- the branch is therefore UNVERIFIED on hardware so far (README.md, INTEGRATION.md say so); tests/test_dp_gpu2.py
and tests/test_bench_dp.py exercise the same step with gloo carrying the collectives, tests/test_dp_gloo.py the stream-ordering
code through fake Work handles."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (>= 2 devices)")
@pytest.mark.parametrize("kl", ["normal", "gp"])
def test_two_rank_rccl_step_matches_single_process(kl):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_gpu_worker.py"), kl, "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DPRESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(line[-1][len("DPRESULT "):])
    assert out["backend"] == "nccl"
    for a, b in zip(out["nll_dp"], out["nll_single"]):
        assert abs(a - b) <= 1e-5 * abs(b), out
    assert out["params"] < 2e-3 and out["replica_drift"] == 0.0 and out["shadow_vs_master"] == 0.0, out
    if kl == "normal":
        assert out["captured_ok"] is True, out
