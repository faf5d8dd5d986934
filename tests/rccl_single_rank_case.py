"""The body of tests/test_gpu_parity.py::test_data_parallel_code_path_single_rank_rccl, run as a process of its own (the test
spawns it): the data-parallel step on a ONE-rank RCCL group must reproduce the plain step bit for bit, and must be capturable
and replayable as a HIP graph.  A process of its own because the RCCL group's life cycle (and ROCm 7.2's replay of captured RCCL
nodes, which segfaulted behind eighty-odd other GPU tests in one process) should not depend on what ran before."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from tests_common import load_mix_case, rel_err   # noqa: E402


def main():
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.parallel import DataParallel
    from hlvae_amd.training import ELBOTrainer
    g, src, dims, state = load_mix_case(os.path.join(HERE, "golden"), "mix_trained")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=dev)
    try:
        data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
        eps = torch.tensor(g["eps"], device=dev).float()
        results = []
        for use_dp in (False, True):
            model = HLVAE(dims, src.types_info, src.n_variables, vy_init=[1.0, 0.5], conv=False, max_batch=128, materialize_samples=False)
            model.load_state_dict({k: v for k, v in state.items()})
            model = model.to(dev)
            tr = ELBOTrainer(model, P_total=40, kl="normal", max_batch=128, dp=DataParallel(dist.group.WORLD) if use_dp else None)
            for _ in range(2):
                tr.step(data, mask, 4, eps=eps)
            torch.cuda.synchronize()
            results.append((float(tr.scalars()["nll_sum"]), model._arena.clone()))
        assert results[0][0] == results[1][0], (results[0][0], results[1][0])
        # fp32 atomics in the small-gradient region may reorder between runs: allow rounding-level differences
        assert rel_err(results[1][1], results[0][1]) < 1e-5
        # the data-parallel step (statistics all-reduce, overlapped gradient all-reduces) is capturable in a HIP graph
        tr.capture("dp", data, mask, 4)
        before = model._arena.clone()
        tr.replay("dp")
        tr.replay("dp")
        torch.cuda.synchronize()
        assert np.isfinite(float(tr.scalars()["nll_sum"])) and not torch.equal(before, model._arena)
    finally:
        dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK")


if __name__ == "__main__":
    main()
