"""Two ranks on ONE GPU (gloo carries the collectives, both ranks compute on cuda:0): the data-parallel ELBO step of the
HIP path -- whole subjects per rank, statistics all-reduce, flat gradient all-reduce with the overlapped y_layer slice, and
(kl = gp) the packed [W | P1 | u | bound] exchange of the GP prior -- against the same steps in a single process on the
full batch.  Launched by tests/test_dp_gpu2.py through torch.distributed.run; rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-300))


def main():
    kl = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"      # "nccl" (= RCCL): one rank per GPU, needs >= world devices
    if backend == "nccl":
        lr_ = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr_)
        dev = torch.device("cuda", lr_)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    import hlvae_amd  # noqa: F401
    from hlvae_amd import synthetic
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.parallel import DataParallel
    from hlvae_amd.elbo_functions import GPPriorHIP
    from tests_common import MIX_SPEC

    src = synthetic.make_tabular(n_rows=96, T=6, seed=7, spec=MIX_SPEC)          # 16 subjects x 6 rows
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    subj = src.labels[:, src.id_covariate]
    ids = np.unique(subj)
    mine = np.isin(subj, ids[rank::world])
    rows = np.nonzero(mine)[0]
    P_total, P_batch = 80, len(ids)
    steps = 2
    eps = [torch.randn(96, dims[2], generator=torch.Generator().manual_seed(50 + i)) for i in range(steps)]
    t = lambda a, r=None: torch.tensor(a if r is None else a[r], dtype=torch.float64, device=dev)

    def build(dp):
        torch.manual_seed(0)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=128).to(dev)
        gp = None
        if kl == "gp":
            gp = GPPriorHIP(dims[2], t(src.labels), 10, src.id_covariate, N_total=480, seed=3, dp=dp)
        return model, gp, ELBOTrainer(model, P_total=P_total, kl=kl, gp=gp, max_batch=128, dp=dp)

    # single process, full batch
    model_s, gp_s, tr_s = build(None)
    nll_s, kld_s = [], []
    for i in range(steps):
        tr_s.step(t(src.data), t(src.mask), P_batch, eps=eps[i].to(dev), train_x=t(src.labels))
        nll_s.append(float(tr_s.scalars()["nll_sum"]))
        kld_s.append(float(gp_s.last_kld) if gp_s is not None else float(tr_s.scalars()["kl"]))
    # data parallel, this rank's subjects, fed from the device-resident compact dataset (what bench.py --gpus N runs)
    from hlvae_amd.datafeed import CompactDataset
    dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    from hlvae_amd.datafeed import subject_index
    rows_dev = torch.tensor(rows.astype(np.int32), device=dev)
    groups_dev = torch.tensor(subject_index(subj[rows]), device=dev)          # the sampler's subject structure (GP prior)
    dp = DataParallel(dist.group.WORLD)
    model_d, gp_d, tr_d = build(dp)
    nll_d, kld_d = [], []
    tr_d.prime_rows(dsd, rows_dev)          # pipelined input stage: every step packs the NEXT batch (statistics all-reduce included)
    for i in range(steps):
        tr_d.step_rows(dsd, rows_dev, P_batch, eps=eps[i][rows].to(dev), prefetch_rows=rows_dev, prepacked=True, groups=groups_dev)
        part = torch.stack([tr_d.scalars()["nll_sum"].double().reshape(()), tr_d.scalars()["kl"].double().reshape(())])
        dist.all_reduce(part)
        nll_d.append(float(part[0]))
        kld_d.append(float(gp_d.last_kld) if gp_d is not None else float(part[1]))
    torch.cuda.synchronize()
    captured = None
    if backend == "nccl" and kl == "normal":
        # the captured form of the same step (thread_local capture mode, async reduce-scatter / all-gather handles inside the
        # capture): two replays must leave the parameters finite and moving, and identical on every rank
        tr_d.capture_rows("c", dsd, [rows_dev, rows_dev], [P_batch, P_batch], next_rows=[rows_dev, rows_dev], groups=[groups_dev, groups_dev])
        tr_d.prime_rows(dsd, rows_dev)
        before = model_d._arena.clone()
        tr_d.replay("c")
        torch.cuda.synchronize()
        captured = bool(torch.isfinite(model_d._arena).all()) and not torch.equal(before, model_d._arena)
    model_d.state_dict()                    # collective: gathers the fp32 masters of the other rank's slices
    out = dict(kl=kl, world=world, rows=[int(len(rows))], nll_single=nll_s, nll_dp=nll_d, kld_single=kld_s, kld_dp=kld_d,
               params=rel(model_d._arena, model_s._arena))
    if gp_s is not None:
        out.update(gp_theta=rel(gp_d._theta, gp_s._theta), gp_m=rel(gp_d.m, gp_s.m), gp_H=rel(gp_d.H, gp_s.H))
    # replicas must agree exactly: the masters after the gather, and the bf16 shadows every rank computes with
    drift = 0.0
    for t in (model_d._arena, model_d._ws_t["wys"].float(), model_d._ws_t["w1s"].float(), model_d._ws_t["wyTs"].float(),
              model_d._ws_t["wmls"].float(), model_d._ws_t["wds"].float()):
        chk = t.clone()
        dist.broadcast(chk, 0)
        drift = max(drift, float((chk - t).abs().max()))
    out["replica_drift"] = drift
    out["backend"], out["captured_ok"] = backend, captured
    # the shadows ARE the bf16 rounding of the gathered masters
    d = model_d._dims
    wy = model_d.y_layer[0].weight.detach()[torch.as_tensor(model_d.kernel_wy_rows(), device=dev)]   # shadow rows: kernel's variable order
    out["shadow_vs_master"] = float((model_d._ws_t["wys"][:wy.shape[0], :wy.shape[1]].float() - wy.to(torch.bfloat16).float()).abs().max())
    if rank == 0:
        print("DPRESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
