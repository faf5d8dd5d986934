"""Data-parallel path on CPU (gloo, world size 2): the exchange steps of hl-vae_amd/parallel.py
(statistics all-reduce, flat gradient-arena all-reduce) and whole-subject sharding reproduce the
single-process step.  The oracle is the compute provider here (there is no GPU in this test)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _partial_sums(data, mask, plan, chunks=16):
    """host restatement of the layout k_colstats writes: [chunks, 3, n_stat] masked sums (reals first, then pos)"""
    B = data.shape[0]
    cols = [(int(plan.sidx[d]) if plan.kind[d] == 0 else plan.n_real + int(plan.sidx[d]), int(plan.xoff[d]), d, int(plan.kind[d]))
            for d in range(plan.D) if plan.kind[d] in (0, 1)]
    out = torch.zeros(chunks, 3, max(len(cols), 1), dtype=torch.float64)
    rpc = (B + chunks - 1) // chunks
    for c in range(chunks):
        rows = slice(c * rpc, min(B, (c + 1) * rpc))
        for si, xo, d, kind in cols:
            m = mask[rows, d]
            x = data[rows, xo] * m
            if kind == 1:
                x = torch.log1p(x)
            out[c, 0, si] = m.sum()
            out[c, 1, si] = (x * m).sum()
            out[c, 2, si] = (x * x * m).sum()
    return out


def _stats_from_sums(sums, plan):
    t = sums.sum(0)
    n, s1, s2 = t[0], t[1], t[2]
    mean = s1 / n
    var = (s2 - 2 * mean * s1 + mean * mean * n) / n
    nr = plan.n_real
    pos_var = torch.clamp(var[nr:], 1e-6, 1e20)
    return [[mean[:nr], var[:nr]], [mean[nr:nr + plan.n_pos], pos_var[:plan.n_pos]]]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hlvae_oracle as orc
    from hlvae_amd import layout, synthetic
    from hlvae_amd.parallel import DataParallel
    from tests_common import MIX_SPEC
    torch.set_num_threads(1)
    src = synthetic.make_tabular(n_rows=48, T=6, seed=7, spec=MIX_SPEC)          # 8 subjects x 6 rows
    plan = layout.compile_plan(src.types_info, 5)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=5, std=0.2)
    dp = DataParallel(dist.group.WORLD)
    # one global batch of 8 subjects -> 4 per rank (whole subjects)
    batches = list(synthetic.subject_batches(src.labels, 4, rank=rank, world=world))
    assert len(batches) == 1
    rows = batches[0]
    assert len(np.unique(src.labels[rows, 2])) == 4
    data, mask = torch.tensor(src.data[rows]), torch.tensor(src.mask[rows])
    eps_all = torch.randn(48, 4, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    # (1) statistics: all-reduce of the per-chunk partial sums
    sums = _partial_sums(data, mask, plan)
    dp.allreduce_stats(sums)
    stats = _stats_from_sums(sums, plan)
    # (2) local gradients with GLOBAL statistics and the GLOBAL P_batch in the loss scale, then arena all-reduce
    P_total, P_batch = 40, 8
    names = [k for k in state if not k.startswith("hidden.")]

    def grads_for(rows_, stats_):
        st = {k: state[k].clone().requires_grad_(True) for k in names}
        for k in list(st):
            if k.startswith("d_layers."):
                st["hidden." + k[len("d_layers."):]] = st[k]
        om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
        out = om.forward(torch.tensor(src.data[rows_]), torch.tensor(src.mask[rows_]), eps_all[rows_], stats=stats_)
        loss = om.loss_function(out["log_p_x"]).sum() * P_total / P_batch + orc.standard_normal_kl(out["mu"], out["log_var"])
        loss.backward()
        return torch.cat([(st[k].grad if st[k].grad is not None else torch.zeros_like(st[k])).flatten() for k in names]), out

    arena, _ = grads_for(rows, stats)
    dp.allreduce_grads(arena)
    # reference: the single-process step on the whole global batch (its own statistics)
    all_rows = np.sort(np.concatenate([np.asarray(b) for r in range(world)
                                       for b in synthetic.subject_batches(src.labels, 4, rank=r, world=world)]))
    full, out_full = grads_for(all_rows, None)
    ok_stats = all(torch.allclose(a, b, rtol=1e-10, atol=1e-12) for a, b in
                   zip(stats[0] + stats[1], out_full["norm"][0] + out_full["norm"][1]))
    err = float((arena - full).norm() / full.norm())
    torch.save({"ok_stats": ok_stats, "err": err, "nrows": len(rows)}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_dp_world2_matches_single_process(tmp_path):
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert res["ok_stats"], "global statistics from all-reduced partial sums differ from the single-process ones"
        assert res["err"] < 1e-10, res
        assert res["nrows"] == 24
