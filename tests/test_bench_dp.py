"""The exact command shape the driver uses for the scaling curve, rehearsed with two ranks on ONE GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...

with HLVAE_BENCH_BACKEND=gloo (gloo moves the collectives, the HIP path computes).  Asserts: one JSON line, a finite NLL, the
sharded optimiser path, and -- with a capture failure forced on ONE rank -- that every rank ends up launching eagerly instead of
one of them waiting for a peer that never arrives (bench.py agrees on `ok` before and after the captures, outside any capture).
The RCCL transport itself (async reduce-scatter / all-gather on side streams, capture with thread_local error mode) needs a node
with >= 2 GPUs and is NOT exercised here: see tests/test_dp_nccl.py (skipped below two devices)."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra_env, steps=20, warmup=5):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", HLVAE_BENCH_BACKEND="gloo", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_scale_command_two_ranks_one_gpu():
    j, _ = _run({})
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 5 and j["scaling"] == "weak"
    assert j["value"] > 0 and math.isfinite(j["config"]["final_nll_sum"])
    assert "sharded" in j["config"]["optimizer"] and "world 2" in j["config"]["optimizer"]
    assert j["config"]["hip_graph"] is False                    # gloo's host-side collectives cannot be captured
    assert abs(j["value"] - 2 * 512 * 20 / (j["ms_per_step"] * 1e-3 * 20)) <= 1e-6 * j["value"]      # whole-job rows / max-over-ranks time


def test_forced_capture_failure_on_one_rank_falls_back_everywhere():
    # both ranks are told to try capturing; rank 1 is made to fail before its first capture.  Rank 0's own attempt fails too
    # under gloo (a host-synchronising collective inside a capture) or succeeds: either way the two agreements leave every rank
    # on the eager path and the run finishes.
    j, err = _run({"HLVAE_BENCH_TRY_CAPTURE": "1", "HLVAE_BENCH_FAIL_CAPTURE_RANK": "1"}, steps=10, warmup=2)
    assert j["config"]["hip_graph"] is False and "graph_note" in j["config"], j["config"]
    assert "launching eagerly" in err
    assert math.isfinite(j["config"]["final_nll_sum"]) and "sharded" in j["config"]["optimizer"]
