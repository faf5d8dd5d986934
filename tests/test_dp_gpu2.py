"""world_size = 2 data-parallel run of the HIP path on one GPU (gloo transports the collectives): the N > 1 code path of
hl-vae_amd/training.py (pipelined input stage with its statistics all-reduce, reduce-scatter of the two dense gradient slices,
sharded Adam, all-gather of the bf16 copies, shadows rebuilt from them) and the GP-prior exchange of elbo_functions.GPPriorHIP, checked against a single-process run of
the same global batch.  Tolerances: NLL 1e-5 rel (fp32 sums in a different order), GP bound 1e-7, parameters after two
Adam steps 2e-3 rel (Adam's first steps amplify rounding of tiny gradients), GP state 1e-6."""
import json
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


@pytest.mark.parametrize("kl", ["normal", "gp"])
def test_two_rank_step_matches_single_process(kl):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_gpu_worker.py"), kl]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DPRESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(line[-1][len("DPRESULT "):])
    for a, b in zip(out["nll_dp"], out["nll_single"]):
        assert abs(a - b) <= 1e-5 * abs(b), out
    tol = 1e-7 if kl == "gp" else 1e-5        # mu / log_var come out of the fp32 pipeline: rounding differs per partition
    for a, b in zip(out["kld_dp"], out["kld_single"]):
        assert abs(a - b) <= tol * abs(b) + 1e-9, out
    assert out["params"] < 2e-3, out
    assert out["replica_drift"] == 0.0 and out["shadow_vs_master"] == 0.0, out
    if kl == "gp":
        assert out["gp_theta"] < 1e-6 and out["gp_m"] < 1e-6 and out["gp_H"] < 1e-6, out
