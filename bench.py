#!/usr/bin/env python3
"""HL-VAE ELBO training-step benchmark (BASELINE.json metric: samples(rows)/sec on Het-HealthMNIST-shaped
batches at 1/2/4/8 MI355X).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full pass of the hot path over one batch: normalise/pack -> encoder -> reparameterise ->
decoder + heads + log-likelihoods (+ backward) + imputed values -> per-variable reconstruction metrics -> KL ->
dense backward -> [RCCL gradient all-reduce] -> Adam  (the training.py:70-137 sequence).  Workload at N=1 = BASELINE.json configs[1]: the 1k-sample synthetic D4 set (50 subjects x 20 rows,
324 real + 972 five-class categorical variables, 25 % missing), MLP [5184,[500],32,[500],5], batch 512 rows
(25 whole subjects + 12 rows of a 26th; whole-subject batching is the reference's sampler semantics),
bf16 MFMA encoder/decoder with fp32 ELBO accumulation.  Inputs are the reference's fp64 [B,X]/[B,D] batch
tensors, already resident in HBM.  Weak scaling: every rank processes its own 512-row batch.

Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--kl", default="normal", choices=["normal", "gp", "none"])
    ap.add_argument("--prefetch", action="store_true",
                    help="run the NEXT batch's input stage on a side stream inside each step (measured slower on MI355X: a forked "
                         "branch in the HIP graph costs more than the 31 us it hides; DESIGN.md section 5)")
    ap.add_argument("--feed", default="compact", choices=["fp64", "compact"],
                    help="compact (default): the whole dataset resident in HBM at 5 B/entry, a batch = a vector of row indices, "
                         "the input stage gathers on the device (SURVEY 8(f).3; bit-identical packed inputs, tested); fp64: "
                         "the reference's expanded fp64 batch tensors resident in HBM (the drop-in HLVAE.forward call surface)")
    ap.add_argument("--conv", action="store_true",
                    help="convolutional encoder/decoder (conv_hivae = True, what config/hlvae_config_file.txt:51 selects) "
                         "instead of the MLP the north star names")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-feed-prefetch", dest="feed_prefetch", action="store_false",
                    help="compact feed: run each batch's input stage at the top of its own step instead of beside the previous "
                         "step's backward pass")
    ap.add_argument("--no-graph-chain", dest="graph_chain", action="store_false",
                    help="one HIP graph per step (default: the 4-batch ring is also captured as one graph of 4 steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=6)
    return ap.parse_args()


def build_batches(src, batch, n_ring, dev):
    """resident ring of batches: consecutive row windows (rows are sorted by subject)"""
    N = len(src)
    out = []
    for i in range(n_ring):
        lo = (i * (N - batch) // max(n_ring - 1, 1)) if N > batch else 0
        rows = np.arange(lo, min(lo + batch, N))
        P_b = int(np.unique(src.labels[rows, src.id_covariate]).size)
        out.append(dict(data=torch.tensor(src.data[rows], dtype=torch.float64, device=dev),
                        mask=torch.tensor(src.mask[rows], dtype=torch.float64, device=dev),
                        labels=torch.tensor(src.labels[rows], dtype=torch.float64, device=dev), P_batch=P_b, rows=rows))
    return out


def cpu_baseline(src, dims, state, rows, P_total, P_batch, kl, steps, conv=False):
    """The oracle (CPU fp64 restatement, kind "port") executing the training.py:70-137 sequence:
    forward, NLL, per-step metrics, KL, backward, Adam -- on the host cores of this box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hlvae_oracle as orc
    import metrics_oracle as mo
    # a one-GPU box shares its host: 16 threads is this process's CPU share (more threads only thrash on
    # the many small fp64 ops of the step)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
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
    data, mask = torch.tensor(src.data[rows]), torch.tensor(src.mask[rows])
    g = torch.Generator().manual_seed(0)
    times = []
    for it in range(steps + 1):
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
        loss.backward()
        live = [i for i, p in enumerate(params) if p.grad is not None]       # torch.optim.Adam skips grad-less params
        orc.adam_step([params[i] for i in live], [params[i].grad for i in live], [m1[i] for i in live],
                      [m2[i] for i in live], it + 1)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    return dict(value=len(rows) / med, unit="samples/s", cores=cores, kind="port",
                sample=f"{steps} steps (after 1 warm-up) of the same {len(rows)}-row batch, median {med:.3f} s/step, torch CPU fp64")


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
    dp = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        from hlvae_amd.parallel import DataParallel
        dp = DataParallel(dist.group.WORLD)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer

    src = synthetic.make_d4(n_subjects=50, T=20, seed=100 + rank)          # BASELINE configs[1]: 1k-sample set
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(0)                                                   # identical initial weights on all ranks
    model = HLVAE(dims, src.types_info, src.n_variables, conv=a.conv, max_batch=a.batch, materialize_samples=False).to(dev)
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    P_total = 50 * world
    kl = None if a.kl == "none" else a.kl
    gp = None
    if kl == "gp":
        from hlvae_amd.elbo_functions import GPPriorHIP
        gp = GPPriorHIP.from_reference_config(model, src, P_total, dev, dp=dp)       # shipped kernels, M = 120 inducing points
    trainer = ELBOTrainer(model, P_total=P_total, kl=kl, gp=gp, max_batch=a.batch, dp=dp, metrics=True)
    ring = build_batches(src, a.batch, 4, dev)
    use_graph = not a.no_graph and world == 1
    # software pipeline of the input stage: while batch i trains, batch i+1 is normalised and packed on a side stream
    # (row A depends on the data only).  Every step still runs exactly one input stage inside the timed region.
    nxt = lambda i: (ring[(i + 1) % len(ring)]["data"], ring[(i + 1) % len(ring)]["mask"])
    pipelined = a.prefetch
    feed_pf = False
    compact = a.feed == "compact" and kl != "gp"       # the GP variant takes the batch's covariates as a tensor (fp64 feed)
    if compact:
        from hlvae_amd.datafeed import CompactDataset
        dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
        for b in ring:
            b["rows_dev"] = torch.tensor(b["rows"].astype(np.int32), device=dev)
        # pipelined input stage (MLP, single process): every step runs the statistics + pack kernels of the NEXT batch on
        # the side stream of its backward pass (they depend on the data only) -- still exactly one input stage per step
        # inside the timed region, but off the critical path
        feed_pf = use_graph and a.feed_prefetch and not a.conv
        if use_graph:
            R = [b["rows_dev"] for b in ring]
            PB = [b["P_batch"] * world for b in ring]
            nx = [R[(i + 1) % len(ring)] for i in range(len(ring))]
            for i in range(len(ring)):
                trainer.capture_rows(i, dsd, R[i], PB[i], next_rows=nx[i] if feed_pf else None)
            if a.graph_chain:      # the whole ring (4 consecutive steps, one per batch) as ONE graph; a replay = 4 steps
                trainer.capture_rows("ring", dsd, R, PB, next_rows=nx if feed_pf else None)
            if feed_pf:
                trainer.prime_rows(dsd, R[0])
    elif use_graph:
        for i, b in enumerate(ring):
            trainer.capture(i, b["data"], b["mask"], b["P_batch"] * world, train_x=b["labels"],
                            prefetch=nxt(i) if pipelined else None)
        if pipelined:
            trainer.prime(ring[0]["data"], ring[0]["mask"])

    it = [0]                                   # the batch chain continues across warm-up and the timed region

    chain = use_graph and compact and a.graph_chain

    def run(n):
        left = n
        while left > 0:
            i = it[0]
            if chain and i % len(ring) == 0 and left >= len(ring):      # 4 steps per launch; the ragged ends one by one
                trainer.replay("ring")
                it[0] += len(ring)
                left -= len(ring)
                continue
            it[0] += 1
            left -= 1
            b = ring[i % len(ring)]
            if use_graph:
                trainer.replay(i % len(ring))
            elif compact:
                trainer.step_rows(dsd, b["rows_dev"], b["P_batch"] * world)
            else:
                trainer.step(b["data"], b["mask"], b["P_batch"] * world, train_x=b["labels"],
                             prefetch=nxt(i) if pipelined else None)

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
    rows_per_step = sum(len(ring[i % len(ring)]["rows"]) for i in range(a.steps)) / a.steps
    value = world * rows_per_step * a.steps / dt
    nll_last = float(trainer.scalars()["nll_sum"])
    assert np.isfinite(nll_last), "non-finite NLL after the timed steps"

    if rank == 0:
        print(f"[bench] {world} GPU(s): {value:.0f} samples/s, {1e3 * dt / a.steps:.4f} ms/step", file=sys.stderr, flush=True)
    from hlvae_amd import roofline
    # every rank runs the eager per-kernel pass (the data-parallel step contains collectives); rank 0 reports
    roof = roofline.measure_dominant_kernel(trainer, ring[0], a.steps, ds=dsd if compact else None)
    if rank != 0:
        roof = None
    if roof is not None:      # HBM traffic from PMC counters is collected offline (separate rocprofv3 --pmc passes)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))["kernels"]
            if roof["kernel"] in pm and not a.conv and kl != "gp":
                roof["traffic"] = pm[roof["kernel"]]["traffic_bytes_per_launch"]
        except Exception:
            pass

    if rank == 0:
        print(f"[bench] roofline: {json.dumps(roof)}", file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(src, dims, state0, ring[0]["rows"], P_total, ring[0]["P_batch"], kl, a.cpu_steps, conv=a.conv)

    if rank == 0:
        line = {
            "metric": "ELBO-steps/sec (samples/sec) on Het-HealthMNIST", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 1k-sample D4 Het-HealthMNIST set (324 real + 972 cat5, 25 pct missing), "
                                   + ("convolutional encoder/decoder (conv 1-16-32 + 2592-500-32 | 32-500-2592 + deconv 32-16-5), "
                                      if a.conv else "MLP [5184,[500],32,[500],5], ")
                                   + f"batch {a.batch} rows/GPU, "
                                   + ("compact dataset (5 B/entry) resident in HBM, batches = row-index vectors" if compact
                                      else "fp64 inputs resident in HBM"),
                       "kl": a.kl, "hip_graph": use_graph, "steps_per_graph_launch": (len(ring) if chain else 1), "input_stage_prefetch": bool(pipelined or (compact and feed_pf)), "rows_per_step_per_gpu": rows_per_step,
                       "final_nll_sum": nll_last},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
