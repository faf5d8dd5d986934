#!/usr/bin/env python3
"""HL-VAE ELBO training-step benchmark (BASELINE.json metric: samples(rows)/sec on Het-HealthMNIST-shaped
batches at 1/2/4/8 MI355X).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full pass of the hot path over one batch: normalise/pack -> encoder -> reparameterise ->
decoder + heads + log-likelihoods (+ backward) + imputed values -> per-variable reconstruction metrics -> KL ->
dense backward -> [RCCL reduce-scatter / sharded Adam / all-gather] -> Adam  (the training.py:70-137 sequence).

Default workload = BASELINE.json configs[1]: the 1k-sample synthetic D4 set (50 subjects x 20 rows, 324 real + 972
five-class categorical variables, 25 % missing), MLP [5184,[500],32,[500],5], batch 512 rows per GPU (25 whole subjects
+ 12 rows of a 26th), bf16 MFMA encoder/decoder with fp32 ELBO accumulation, inputs resident in HBM.  Weak scaling:
every rank holds its own data shard and processes its own 512-row batch; the global batch at N = 8 is 4096 rows
(configs[2]'s shape).  The other BASELINE configs at N = 1 (their lines are kept under profiles/):

    configs[2]  --workload d4 --rows 100000 --batch 4096            100k-sample D4, the whole global batch on one GPU
    configs[3]  --workload tabular --rows 1000000 --batch 4096      64 mixed-type features (16 real / 16 pos / 8 count / 16 cat5 /
                                                                    8 ordinal5, interleaved), 1 M rows
    configs[4]  --workload d4 --rows 50000 --batch 1024 --kl gp     GP-prior KL (L = 32, M = 120) alongside the HIP decoder

--rows and --batch are PER GPU.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import hlvae_amd                                  # noqa: E402
from hlvae_amd import synthetic                   # noqa: E402

T_SUBJECT = {"d4": 20, "tabular": 16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="d4", choices=["d4", "tabular"])
    ap.add_argument("--rows", type=int, default=None, help="rows of the synthetic data set PER GPU (default 1000)")
    ap.add_argument("--batch", type=int, default=512, help="rows per step PER GPU")
    ap.add_argument("--kl", default="normal", choices=["normal", "gp", "none"])
    ap.add_argument("--feed", default="compact", choices=["fp64", "compact"],
                    help="compact (default): the whole dataset resident in HBM at 5 B/entry, a batch = a vector of row indices, "
                         "the input stage gathers on the device (SURVEY 8(f).3; bit-identical packed inputs, tested); fp64: "
                         "the reference's expanded fp64 batch tensors resident in HBM (the drop-in HLVAE.forward call surface)")
    ap.add_argument("--prefetch", action="store_true", help="fp64 feed only: the next batch's input stage on a torch side stream")
    ap.add_argument("--conv", action="store_true",
                    help="convolutional encoder/decoder (conv_hivae = True, what config/hlvae_config_file.txt:51 selects) "
                         "instead of the MLP the north star names")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-feed-prefetch", dest="feed_prefetch", action="store_false",
                    help="compact feed: run each batch's input stage at the top of its own step instead of beside the previous "
                         "step's backward pass")
    ap.add_argument("--no-graph-chain", dest="graph_chain", action="store_false",
                    help="one HIP graph per step (default: the 4-batch ring is also captured as one graph of --chain steps)")
    ap.add_argument("--chain", type=int, default=8,
                    help="steps per HIP-graph launch (a multiple of the 4-batch ring): between two graph launches the executor "
                         "joins and re-forks its queues (~25 us on MI355X), inside a graph consecutive steps abut")
    ap.add_argument("--sharded", action="store_true",
                    help="N = 1: run the data-parallel optimiser path (flat sharded Adam -> bf16 copy -> shadows) instead of the "
                         "fused tile Adam, to price the code path the multi-GPU step uses")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the second measurement of the default run (BASELINE configs[4]: the reference's own training mode -- "
                         "GP-prior KL + natural gradient, batch 1024 -- timed by a child process and reported as `also`)")
    ap.add_argument("--no-in-step", action="store_true", help="skip the in-graph kernel stamps (roofline.in_step)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--tag", default=None, help="label copied into config.tag (profiles/ bookkeeping)")
    return ap.parse_args()


def make_source(a, rank):
    """the synthetic data set of this rank (its shard of the subjects): compact form above 2000 rows"""
    rows = a.rows if a.rows is not None else 1000
    T = T_SUBJECT[a.workload]
    n_subj = (rows + T - 1) // T
    expanded = rows <= 2000 or a.feed == "fp64"
    if expanded and rows > 20000:
        raise SystemExit(f"--feed fp64 needs the expanded fp64 matrices on the host: {rows} rows is too many, use --feed compact")
    if a.workload == "d4":
        return synthetic.make_d4(n_subjects=n_subj, T=T, seed=100 + rank, expanded=expanded), n_subj
    if a.conv:
        raise SystemExit("--conv views the variables as a 36 x 36 image: D4 workload only")
    return synthetic.make_tabular(n_rows=n_subj * T, T=T, seed=100 + rank, expanded=expanded), n_subj


def build_ring(src, batch, n_ring):
    """resident ring of batches: row windows spread over the data set (rows are sorted by subject, so a window is whole
    subjects plus the head of one more -- the reference's sampler semantics, utils.py:77-97)"""
    from hlvae_amd.datafeed import subject_index
    N = len(src)
    if batch > N:
        raise SystemExit(f"--batch {batch} exceeds the {N} rows of the data set (--rows)")
    out = []
    for i in range(n_ring):
        lo = (i * (N - batch) // max(n_ring - 1, 1)) if N > batch else 0
        rows = np.arange(lo, lo + batch)
        ids = src.labels[rows, src.id_covariate]
        out.append(dict(rows=rows, P_batch=int(np.unique(ids).size), groups=subject_index(ids)))
    return out


def host_cores():
    """host cores this process may actually use: the affinity mask capped by the cgroup CPU quota.  On the one-GPU boxes of
    this pool 256 hardware threads are visible but cpu.max is 16 cores' worth; 256 torch threads against that quota were
    measured at 31.6 s per oracle step against 0.07 s with 16 (gpurun_out/r2_b_default.json, round 2)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(src, dims, state, rows, P_total, P_batch, kl, steps, conv=False, gp_state=None, budget_s=45.0):
    """The oracle (CPU fp64 restatement, kind "port") executing the training.py:70-137 sequence -- forward, NLL, per-step
    metrics, KL (N(0, I) closed form or the GP prior + natural gradient), backward, Adam -- on ALL host cores of this box
    (3 warm-ups, >= 10 timed steps, median; fewer timed steps only if they would exceed the budget)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hlvae_oracle as orc
    import metrics_oracle as mo
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {cores} host cores usable by this process (affinity capped by the cgroup quota) = torch threads",
          file=sys.stderr, flush=True)
    st = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in state.items()
          if not k.startswith(("hidden.", "Decoder_Conv_layer."))}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
        if k.startswith("deconv_layer."):
            st["Decoder_Conv_layer." + k[len("deconv_layer."):]] = st[k]
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st, conv=conv)
    names = [k for k in st if not k.startswith(("hidden.", "Decoder_Conv_layer.")) and k != "_disp_param"]
    params = [st[k] for k in names]
    m1 = [torch.zeros_like(p) for p in params]
    m2 = [torch.zeros_like(p) for p in params]
    if hasattr(src, "expand_rows"):
        d_np, m_np = src.expand_rows(rows)
    else:
        d_np, m_np = src.data[rows], src.mask[rows]
    data, mask = torch.tensor(d_np), torch.tensor(m_np)
    labels = torch.tensor(src.labels[rows])
    g = torch.Generator().manual_seed(0)
    gpo = None
    if kl == "gp":
        import gp_oracle as gpo
        spec, kprm, gm_, gH_, z_, noise, N_total, id_cov = gp_state
        kprm = {k: v.clone().requires_grad_(True) for k, v in kprm.items()}
        z_ = z_.clone().requires_grad_(True)
        gp_leaves = list(kprm.values()) + [z_]
        gp_m1, gp_m2 = [torch.zeros_like(p) for p in gp_leaves], [torch.zeros_like(p) for p in gp_leaves]
    times, warm = [], 3

    def one(it):
        nonlocal gm_, gH_
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        eps = torch.randn(len(rows), dims[2], generator=g, dtype=torch.float64)
        out = om.forward(data, mask, eps)
        nll = om.loss_function(out["log_p_x"]).sum()
        mo.step_metrics([p.detach() for p in out["p_params"]], data, mask, src.types_info, st["_log_vy_pos"].detach(), conv=conv)
        loss = nll * P_total / P_batch
        if kl == "normal":
            loss = loss + orc.standard_normal_kl(out["mu"], out["log_var"])
        elif kl == "gp":
            for p in gp_leaves:
                p.grad = None
            kld, grad_m, grad_H = gpo.minibatch_kld_upper_bound_iter(spec, kprm, noise, dims[2], gm_, gH_, labels, out["mu"],
                                                                     out["log_var"], z_, P_total, P_batch, N_total, True, id_cov, 1e-6)
            loss = loss + kld.sum()
        loss.backward()
        live = [i for i, p in enumerate(params) if p.grad is not None]       # torch.optim.Adam skips grad-less params
        orc.adam_step([params[i] for i in live], [params[i].grad for i in live], [m1[i] for i in live],
                      [m2[i] for i in live], it + 1)
        if kl == "gp":
            orc.adam_step(gp_leaves, [p.grad for p in gp_leaves], gp_m1, gp_m2, it + 1)
            gm_, gH_ = gpo.natural_gradient_update(gm_, gH_, grad_m.detach(), grad_H.detach(), 0.01)
        return time.perf_counter() - t0

    t_first = one(0)
    n_timed = steps
    if t_first * (warm + steps) > budget_s:
        n_timed = max(3, int(budget_s / t_first) - warm)
    for it in range(1, warm):
        one(it)
    for it in range(n_timed):
        times.append(one(warm + it))
    med = float(np.median(times))
    return dict(value=len(rows) / med, unit="samples/s", cores=cores, kind="port",
                sample=f"{n_timed} timed steps after {warm} warm-ups of the same {len(rows)}-row batch, median {med:.3f} s/step, "
                       f"torch CPU fp64, {cores} threads")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # one rank per GPU.  Rehearsal on a box with fewer GPUs than ranks (HLVAE_BENCH_BACKEND=gloo): ranks share the devices
    backend = os.environ.get("HLVAE_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    from hlvae_amd.parallel import DataParallel
    dp = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        dp = DataParallel(dist.group.WORLD)
    elif a.sharded:
        dp = DataParallel.single()

    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset

    src, n_subj = make_source(a, rank)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(0)                                                   # identical initial weights on all ranks
    model = HLVAE(dims, src.types_info, src.n_variables, conv=a.conv, max_batch=a.batch, materialize_samples=False).to(dev)
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    P_total = n_subj * world
    kl = None if a.kl == "none" else a.kl
    gp = gp_state = None
    if kl == "gp":
        from hlvae_amd.elbo_functions import GPPriorHIP
        gp = GPPriorHIP.from_reference_config(model, src, P_total, dev, dp=dp if world > 1 else None)   # shipped kernels, M = 120
    # in-graph kernel stamps (roofline.in_step): the buffer is a kernel argument, so it exists before anything is captured; its slots
    # stay disarmed (zero) through the warm-up and the timed region
    from hlvae_amd import _lib as _hl, roofline
    stamps = None
    if not a.no_in_step:
        stamps = torch.zeros(_hl.load().hlvae_stamp_words(), dtype=torch.int64, device=dev)
        _hl.load().hlvae_stamp_buffer(_hl.ptr(stamps))
    trainer = ELBOTrainer(model, P_total=P_total, kl=kl, gp=gp, max_batch=a.batch, dp=dp, metrics=True)
    ring = build_ring(src, a.batch, 4)
    compact = a.feed == "compact"
    can_capture = world == 1 or backend == "nccl"                          # gloo's host-side collectives cannot be captured
    use_graph = not a.no_graph and can_capture
    feed_pf = False
    pipelined = a.prefetch and not compact
    if gp is not None and rank == 0 and not a.no_cpu_baseline:            # GP state at step 0 for the CPU baseline
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import gp_oracle as gpo
        spec = gpo.spec_from_config([2], [], [0], [{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                                   {"cont_covariate": 1, "cat_covariate": 4}], [], 2)
        kprm = {}
        for row, (which, t, f) in enumerate(gp.slot_names):
            kprm[f"{which}.{t}.scale" if f is None else f"{which}.{t}.{f}.ls"] = gp.prm[row].detach().cpu().clone()
        gp_state = (spec, kprm, gp.m.detach().cpu().clone(), gp.H.detach().cpu().clone(), gp.zt_list.detach().cpu().clone(),
                    torch.ones(dims[2], dtype=torch.float64), float(gp.N_total), 2)

    for b in ring:
        b["rows_dev"] = torch.tensor(b["rows"].astype(np.int32), device=dev)
        b["groups_dev"] = torch.tensor(b["groups"], device=dev) if kl == "gp" else None
    PB = [b["P_batch"] * world for b in ring]
    graph_note = None
    if compact:
        if hasattr(src, "raw"):
            dsd = CompactDataset.from_raw(src.raw, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
        else:
            dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
        # pipelined input stage (MLP): every step runs the statistics + pack kernels of the NEXT batch beside its backward pass
        # (they depend on the data only; with several ranks the statistics all-reduce rides on the same side stream) -- still
        # exactly one input stage per step inside the timed region, but off the critical path
        feed_pf = (use_graph or (not a.no_graph and os.environ.get("HLVAE_BENCH_TRY_CAPTURE") == "1")) and a.feed_prefetch and not a.conv
        try_graph = use_graph or (not a.no_graph and os.environ.get("HLVAE_BENCH_TRY_CAPTURE") == "1")     # (rehearsal: let gloo ranks try)

        def agree(ok):
            """every rank replays graphs, or none does (MIN over ranks; host-side, outside any capture)"""
            if world > 1:
                t = torch.tensor([int(ok)], device=dev, dtype=torch.int32)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
                ok = bool(int(t.item()))
            return ok

        if try_graph:
            R = [b["rows_dev"] for b in ring]
            Gr = [b["groups_dev"] for b in ring]
            nx = [R[(i + 1) % len(ring)] for i in range(len(ring))]
            # Phase 1 -- the eager warm-up steps (these execute the step's real collectives; an exception here is a genuine
            # failure of the run and is raised).  Phase 2 -- the ranks agree that all of them got here and can capture AT ALL
            # (a trivial capture with the trainer's capture mode).  Phase 3 -- the captures proper: a captured collective is
            # recorded, not executed, so a rank that throws in there leaves no peer blocked; every rank then reaches the
            # second agreement and all of them replay graphs or all launch eagerly.
            for _ in range(2):
                trainer.step_rows(dsd, R[0], PB[0], groups=Gr[0])
            torch.cuda.synchronize()
            ok, graph_note = 1, None
            try:
                if os.environ.get("HLVAE_BENCH_FAIL_CAPTURE_RANK") == str(rank):
                    raise RuntimeError("forced capture failure (HLVAE_BENCH_FAIL_CAPTURE_RANK)")
                probe = torch.cuda.CUDAGraph()
                pt = torch.zeros(8, device=dev)
                with torch.cuda.graph(probe, **trainer._capture_kw()):
                    pt.add_(1.0)
                probe.replay()
            except Exception as e:     # noqa: BLE001
                ok, graph_note = 0, f"capture probe failed: {type(e).__name__}: {e}"
            if not agree(ok):
                ok = 0
                graph_note = graph_note or "capture probe failed on another rank"
            if ok:
                try:
                    # Chains of consecutive steps as ONE graph each, for every ring position a region may start at and for the lengths
                    # a.chain, 4, 2: a timed region of any length is then whole chains plus at most one single step (an even chain
                    # double-buffers y_layer's shadows and has no executor join between its steps; a short driver run of 20 steps
                    # that fell back to single-step graphs at its ragged ends measured 0.145 instead of 0.137 ms/step).
                    # Capture order keeps the two input-buffer sets in phase: position o starts on set o % 2, an even chain ends on
                    # the set it began on, the single-step graph of position o moves on to the next.
                    chain_lens = sorted({l for l in (a.chain, 16, 8, 4, 2) if l <= a.chain and l % 2 == 0 and l >= 2}, reverse=True) if (a.graph_chain and feed_pf) else []
                    for o in range(len(ring)):
                        for l in chain_lens:
                            idx = [(o + j) % len(ring) for j in range(l)]
                            trainer.capture_rows(("chain", o, l), dsd, [R[k] for k in idx], [PB[k] for k in idx],
                                                 next_rows=[nx[k] for k in idx], groups=[Gr[k] for k in idx])
                        trainer.capture_rows(o, dsd, R[o], PB[o], next_rows=nx[o] if feed_pf else None, groups=Gr[o])
                    if a.graph_chain and not feed_pf:      # (no pipelined input stage: one chain from position 0, as before)
                        reps = max(1, a.chain // len(ring))
                        trainer.capture_rows("ring", dsd, R * reps, PB * reps, next_rows=None, groups=Gr * reps)
                except Exception as e:     # noqa: BLE001 -- a failed capture must not cost the whole measurement
                    ok, graph_note = 0, f"capture failed: {type(e).__name__}: {e}"
                    trainer.reset_after_failed_capture()
                if not agree(ok):
                    ok = 0
                    graph_note = graph_note or "capture failed on another rank"
            use_graph = bool(ok)
            if not ok:
                feed_pf = False
                print(f"[bench] rank {rank}: HIP-graph capture unavailable ({graph_note}); launching eagerly", file=sys.stderr, flush=True)
            elif feed_pf:
                trainer.prime_rows(dsd, R[0])
    else:
        for b in ring:
            b["data"] = torch.tensor(src.data[b["rows"]], dtype=torch.float64, device=dev)
            b["mask"] = torch.tensor(src.mask[b["rows"]], dtype=torch.float64, device=dev)
            b["labels"] = torch.tensor(src.labels[b["rows"]], dtype=torch.float64, device=dev)
        nxt = lambda i: (ring[(i + 1) % len(ring)]["data"], ring[(i + 1) % len(ring)]["mask"])
        if use_graph:
            for i, b in enumerate(ring):
                trainer.capture(i, b["data"], b["mask"], PB[i], train_x=b["labels"], prefetch=nxt(i) if pipelined else None)
            if pipelined:
                trainer.prime(ring[0]["data"], ring[0]["mask"])

    it = [0]                                   # the batch chain continues across warm-up and the timed region
    chain = use_graph and compact and a.graph_chain
    n_chain = len(ring) * max(1, a.chain // len(ring))
    chain_lens = sorted({l for l in (a.chain, 16, 8, 4, 2) if l <= a.chain and l % 2 == 0 and l >= 2}, reverse=True) if (chain and feed_pf) else []

    def run(n):
        left = n
        while left > 0:
            i = it[0]
            l = next((l for l in chain_lens if left >= l), 0)
            if l:                                                       # the longest chain that fits, from this ring position
                trainer.replay(("chain", i % len(ring), l))
                it[0] += l
                left -= l
                continue
            if chain and not feed_pf and i % len(ring) == 0 and left >= n_chain:
                trainer.replay("ring")
                it[0] += n_chain
                left -= n_chain
                continue
            it[0] += 1
            left -= 1
            b = ring[i % len(ring)]
            if use_graph:
                trainer.replay(i % len(ring))
            elif compact:
                trainer.step_rows(dsd, b["rows_dev"], PB[i % len(ring)], groups=b["groups_dev"])
            else:
                trainer.step(b["data"], b["mask"], PB[i % len(ring)], train_x=b["labels"], prefetch=nxt(i) if pipelined else None)

    run(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    value = world * a.batch * a.steps / dt
    nll_last = float(trainer.scalars()["nll_sum"])
    assert np.isfinite(nll_last), "non-finite NLL after the timed steps"
    if gp is not None:
        assert int(gp.fail.item()) == 0, "GP prior: a non-positive pivot in an SPD inversion during the timed steps"

    if rank == 0:
        print(f"[bench] {world} GPU(s): {value:.0f} samples/s, {1e3 * dt / a.steps:.4f} ms/step", file=sys.stderr, flush=True)
    # in-graph durations: the SAME captured step replayed a few more times with the stamp slots armed (single-step graphs: one
    # host read per step); every rank replays (the step contains collectives), rank 0 reports
    in_step = None
    if stamps is not None:
        in_step = roofline.measure_in_step(stamps, lambda: run(1), n=24)
    # every rank runs the eager per-kernel pass (the data-parallel step contains collectives); rank 0 reports
    roof = roofline.measure_dominant_kernel(trainer, ring[0], a.steps, ds=dsd if compact else None, P_batch=PB[0])
    if rank != 0:
        roof = None
    workload_key = f"{a.workload}_b{a.batch}" + ("_conv" if a.conv else "") + (f"_{a.kl}" if a.kl != "normal" else "")
    if roof is not None:
        roofline.add_in_step(roof, in_step, trainer.model, 1e3 * dt / a.steps)
        # HBM traffic from PMC counters is collected offline (separate rocprofv3 --pmc passes, tools/pmc_traffic.py); a file made
        # with another build of the library (ABI tag) or another kernel behind the label is refused
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")))
            k = pm.get(workload_key, {}).get("kernels", {})
            abi_ok = pm.get("abi") == _hl.load().hlvae_abi_version()
            if roof["kernel"] in k and abi_ok:
                roof["traffic"] = k[roof["kernel"]]["traffic_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r3_pmc_traffic.json:" + workload_key + ":" + k[roof["kernel"]].get("kernel", "")
            elif roof["kernel"] in k:
                roof["traffic_note"] = f"profiles/r3_pmc_traffic.json was collected with library ABI {pm.get('abi')}, this is {_hl.load().hlvae_abi_version()}: refused"
        except Exception:
            pass
    if rank == 0:
        print(f"[bench] roofline: {json.dumps(roof)}", file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(src, dims, state0, ring[0]["rows"], P_total, ring[0]["P_batch"], kl, a.cpu_steps, conv=a.conv, gp_state=gp_state)

    also = None
    default_run = (a.workload, a.batch, a.kl, a.conv, a.rows, a.sharded) == ("d4", 512, "normal", False, None, False)
    if rank == 0 and world == 1 and default_run and not a.no_also:
        # BASELINE configs[4] -- the reference's actual training mode (elbo_functions.py:196-285 + training.py:130-137: GP-prior KL and
        # natural gradient beside the decoder) -- measured in the same run by a child process (never exec from a GPU process)
        # ... and the configuration the reference SHIPS (config/hlvae_config_file.txt:22, 51: convolutional encoder / decoder + GP prior)
        import subprocess
        also = {}
        for key, flags in (("configs[4]", ["--workload", "d4", "--rows", "50000", "--batch", "1024", "--kl", "gp"]),
                           ("shipped_conv_gp", ["--conv", "--kl", "gp"])):
            cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-also"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                also[key] = ({k: j[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline")} |
                             {"workload": j["config"]["workload"], "kl": j["config"]["kl"], "final_nll_sum": j["config"]["final_nll_sum"],
                              "command": "bench.py " + " ".join(cmd[2:])})
            except Exception as e:     # noqa: BLE001 -- the headline line must not be lost to the extra leg
                also[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        T = T_SUBJECT[a.workload]
        if a.workload == "d4":
            what = (f"synthetic D4 Het-HealthMNIST set, {n_subj * T} rows/GPU ({n_subj} subjects x {T}; 324 real + 972 cat5, 25 pct missing), "
                    + ("convolutional encoder/decoder (conv 1-16-32 + 2592-500-32 | 32-500-2592 + deconv 32-16-5), " if a.conv
                       else "MLP [5184,[500],32,[500],5], "))
        else:
            what = (f"synthetic mixed-type tabular longitudinal set, {n_subj * T} rows/GPU, 64 features (16 real / 16 pos / 8 count / "
                    "16 cat5 / 8 ordinal5, interleaved, 25 pct missing), MLP [160,[500],32,[500],5], ")
        cfg_name = {("d4", 512, "normal"): "configs[1]", ("d4", 4096, "normal"): "configs[2] at N=1", ("tabular", 4096, "normal"): "configs[3] at N=1",
                    ("d4", 1024, "gp"): "configs[4]"}.get((a.workload, a.batch, a.kl), "custom")
        line = {
            "metric": "ELBO-steps/sec (samples/sec) on Het-HealthMNIST", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: " + what + f"batch {a.batch} rows/GPU, "
                                   + ("compact dataset (5 B/entry) resident in HBM, batches = row-index vectors" if compact
                                      else "fp64 inputs resident in HBM"),
                       "kl": a.kl, "hip_graph": use_graph, "steps_per_graph_launch": (n_chain if chain else 1),
                       "input_stage_prefetch": bool(pipelined or (compact and feed_pf)), "rows_per_step_per_gpu": a.batch,
                       "optimizer": "fused tile Adam" if dp is None else f"reduce-scatter + sharded Adam + bf16 all-gather (world {world})",
                       "final_nll_sum": nll_last, **({"graph_note": graph_note} if graph_note else {}), **({"tag": a.tag} if a.tag else {})},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if also is not None:
            line["also"] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
