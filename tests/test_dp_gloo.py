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


# ---- the sharded optimiser step (reduce-scatter -> Adam on the owned slices -> all-gather of the bf16 copy) ------------------
def _sharded_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hlvae_oracle as orc
    from hlvae_amd import layout, synthetic
    from hlvae_amd.datafeed import SubjectBatchSampler
    from hlvae_amd.parallel import DataParallel, ShardPlan, ShardedState
    from tests_common import MIX_SPEC
    torch.set_num_threads(1)
    src = synthetic.make_tabular(n_rows=96, T=6, seed=7, spec=MIX_SPEC)          # 16 subjects x 6 rows
    plan = layout.compile_plan(src.types_info, 5)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=5, std=0.2)
    dp = DataParallel(dist.group.WORLD)
    if os.environ.get("HLVAE_TEST_DEFERRED") == "1":
        # asynchronous semantics without RCCL: with async_op the collective happens only inside handle.wait(), and until then the
        # output buffer is poisoned -- a consumer that forgot to wait (or waited on the wrong handle) computes with NaN.  This is
        # what the native branch's Work handles mean; gloo's emulation in DataParallel is synchronous and would hide such a bug.
        class _Deferred:
            def __init__(self, fn):
                self.fn, self.done = fn, False

            def wait(self):
                if not self.done:
                    self.fn()
                    self.done = True
                return True

        class DeferredDP(DataParallel):
            def reduce_scatter(self, out, inp, async_op=False):
                base = DataParallel.reduce_scatter
                if not async_op:
                    return base(self, out, inp)
                snap = inp.clone()                       # (RCCL reads the input when the collective runs on its stream: here at issue)
                out.fill_(float("nan"))
                return _Deferred(lambda: base(self, out, snap))

            def all_gather(self, full, mine, async_op=False):
                base = DataParallel.all_gather
                if not async_op:
                    return base(self, full, mine)
                snap = mine.clone()
                r, n = self.rank, mine.numel()
                keep = full[r * n:(r + 1) * n].clone()
                full.fill_(float("nan"))
                full[r * n:(r + 1) * n] = keep
                return _Deferred(lambda: base(self, full, snap))

        dp = DeferredDP(dist.group.WORLD)
    # flat arena in the product's order: small tensors first, then the dense matrices with y_layer's weight LAST
    dense = ["d_layers.0.weight", "mean_layer.0.weight", "log_var_layer.0.weight", "VAE_encoder_common_layers.0.weight", "y_layer.0.weight"]
    names = [k for k in state if not k.startswith("hidden.") and k not in dense and k != "_disp_param"] + dense
    ru = lambda v, m: (v + m - 1) // m * m
    offs, o = {}, 0
    for k in names:
        if k == dense[0]:
            a0 = o
        offs[k] = o
        o = ru(o + state[k].numel(), 32)
    end, o_wy = ru(o, 64), offs["y_layer.0.weight"]
    splan = ShardPlan([(o_wy, end, 0x01), (a0, o_wy, 0x1e)], world, rank)
    sstate = ShardedState(dp, splan, "cpu", grad_dtype=torch.float64)

    def flat(get):
        v = torch.zeros(end + splan.pad, dtype=torch.float64)
        for k in names:
            t = get(k)
            if t is not None:
                v[offs[k]:offs[k] + t.numel()] = t.flatten()
        return v

    def grads_for(P_flat, rows_, stats_, eps_, scale):
        st = {k: P_flat[offs[k]:offs[k] + state[k].numel()].view(state[k].shape).clone().requires_grad_(True) for k in names}
        st["_disp_param"] = state["_disp_param"].clone()
        for k in list(st):
            if k.startswith("d_layers."):
                st["hidden." + k[len("d_layers."):]] = st[k]
        om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
        out = om.forward(torch.tensor(src.data[rows_]), torch.tensor(src.mask[rows_]), eps_, stats=stats_)
        (om.loss_function(out["log_p_x"]).sum() * scale + orc.standard_normal_kl(out["mu"], out["log_var"])).backward()
        return flat(lambda k: st[k].grad), out

    P = flat(lambda k: state[k].double())[:end].clone()                     # this rank's masters (stale outside its slices)
    P_ref = P.clone()                                                       # single-process run of the global batches
    m1, m2, r1, r2 = (torch.zeros(end, dtype=torch.float64) for _ in range(4))
    W = P.clone()                                                           # the weights the "kernels" compute with (bf16 copies)
    W_ref = P.clone()
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
    sampler = SubjectBatchSampler(src.labels[:, 2], 5, shuffle=True, seed=3, rank=rank, world=world)
    ref_sampler = SubjectBatchSampler(src.labels[:, 2], 5, shuffle=True, seed=3, min_last=world)     # the same global batches
    n_batches, tail = 0, None
    for it, (b, bref) in enumerate(zip(sampler.batches(), ref_sampler.batches())):
        assert b.P_batch == bref.P_batch and len(b.rows) > 0
        n_batches, tail = n_batches + 1, b.P_batch
        eps_all = torch.randn(96, 4, generator=torch.Generator().manual_seed(100 + it), dtype=torch.float64)
        data, mask = torch.tensor(src.data[b.rows]), torch.tensor(src.mask[b.rows])
        sums = _partial_sums(data, mask, plan)
        dp.allreduce_stats(sums)
        scale = 16.0 / b.P_batch
        G, _ = grads_for(W, b.rows, _stats_from_sums(sums, plan), eps_all[b.rows], scale)
        # --- the exchange + sharded update, as hl-vae_amd/training.py sequences it
        pend = [sstate.reduce_scatter_slice(0, G, async_op=True), sstate.reduce_scatter_slice(1, G, async_op=True)]
        dp.allreduce_(G[:a0])
        for k, h in enumerate(pend):
            h.wait()
            lo, hi = splan.slices[k].own(rank)
            g = sstate.gsh[k][:hi - lo]
            orc.adam_step([P[lo:hi]], [g], [m1[lo:hi]], [m2[lo:hi]], it + 1)
            sstate.own_copy_view(k)[:hi - lo] = P[lo:hi].to(torch.bfloat16)
        orc.adam_step([P[:a0]], [G[:a0]], [m1[:a0]], [m2[:a0]], it + 1)
        W[:a0] = P[:a0]
        # all-gathers issued LAST slice first and consumed in that order, as training.py does (the slice with the first Linear is
        # needed first by the next step)
        gath = [(k, sstate.all_gather_slice(k, async_op=True)) for k in range(len(splan.slices))[::-1]]
        for k, h in gath:
            h.wait()
            s_ = splan.slices[k]
            W[s_.lo:s_.hi] = sstate.pb[k][:s_.hi - s_.lo].to(torch.float64)
        # --- single process on the whole global batch
        Gr, _ = grads_for(W_ref, np.sort(bref.rows), None, eps_all[np.sort(bref.rows)], scale)
        orc.adam_step([P_ref], [Gr[:end]], [r1], [r2], it + 1)
        W_ref = P_ref.clone()
        W_ref[a0:] = bf(P_ref[a0:])
    sstate.sync_masters(P)
    chk = W.clone()
    dist.broadcast(chk, 0)
    mism = float(((W - W_ref).abs() > 1e-9 + 1e-2 * W_ref.abs()).double().mean())
    torch.save({"n_batches": n_batches, "tail": tail, "drift": float((chk - W).abs().max()),
                "masters": float((P - P_ref).norm() / P_ref.norm()), "weights_mismatch": mism,
                "pad": splan.pad, "owned": splan.owned()}, os.path.join(tmp, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,deferred", [(2, False), (4, False), (2, True)])
def test_sharded_optimizer_matches_single_process(tmp_path, world, deferred, monkeypatch):
    """three global batches of whole subjects (the last one larger: a 1-subject tail is folded into it so that every rank takes
    part in every exchange), each: statistics all-reduce, reduce-scatter of the two dense gradient slices, Adam on the owned
    slices, all-gather of the bf16 copies, replicated small region -- against one process stepping on the global batches."""
    port = 29500 + ((os.getpid() + 7 * world + 3 * deferred) % 500)
    monkeypatch.setenv("HLVAE_TEST_DEFERRED", "1" if deferred else "0")      # (deferred: collectives complete only in handle.wait())
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"s{r}.pt"))
        assert res["n_batches"] == 3 and res["tail"] == 6, res            # 16 subjects in batches of 5: 5, 5, 5 + 1
        assert res["drift"] == 0.0, res                                    # replicas compute with bit-identical weights
        assert res["masters"] < 1e-9, res                                  # fp32-master gather == the single-process parameters
        assert res["weights_mismatch"] < 2e-3, res                         # bf16 copies: identical up to rounding ties
        assert all(hi >= lo for lo, hi in res["owned"])
