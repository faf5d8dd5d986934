"""Training-TRAJECTORY parity (reference training.py:62-143 runs hundreds of epochs; the other parity tests compare one to three
steps).  The fused step on the GPU (bf16 MFMA operands, bf16 weight shadows, Adam in the weight-gradient epilogues, fp32 masters)
and the fp64 oracle run the SAME sequence -- same initial weights, same ring of four batches, same reparameterisation noise -- and
the curves are compared along the way:

  * BASELINE configs[1] (D4, MLP [5184,[500],32,[500],5], 512-row batches of a 1000-row set): 200 steps; the NLL within 1e-3
    relative at every 20th step, the parameter UPDATES (theta_200 - theta_0) per tensor against the oracle's;
  * the configuration the reference ships (convolutional model + GP prior, hidden 500, latent 32, M = 120 inducing points):
    50 steps; NLL and the GP bound along the way, the variational parameters m, H at the end.

What is compared at the end is the update, not the parameter (the initial weights are common and 20-100 x larger than 200 Adam
steps of 1e-3).  Bounds are about 2-3 x the values measured on MI355X (gpurun_out/parity_report_trajectory.json)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hlvae_amd                      # noqa: E402,F401
from hlvae_amd import synthetic       # noqa: E402
from tests_common import rel_err      # noqa: E402

REPORT = {}


def _report(key, **kv):
    REPORT.setdefault(key, {}).update(kv)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report_trajectory.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _oracle_state(state):
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items() if not k.startswith(("hidden.", "Decoder_Conv_layer."))}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
        if k.startswith("deconv_layer."):
            st["Decoder_Conv_layer." + k[len("deconv_layer."):]] = st[k]
    return st


def _ring(src, batch, n_ring=4):
    """bench.py's ring: row windows spread over the data set (rows are sorted by subject)"""
    N = len(src.labels)
    out = []
    for i in range(n_ring):
        lo = i * (N - batch) // max(n_ring - 1, 1)
        rows = np.arange(lo, lo + batch)
        out.append((rows, int(np.unique(src.labels[rows, src.id_covariate]).size)))
    return out


def _update_errors(model, state, names, params):
    """per tensor, for the updates d = theta_end - theta_0:  relative L2 error || d_gpu - d_ref || / || d_ref ||  and the cosine
    between d_gpu and d_ref"""
    sd = dict(model.named_parameters())
    errs, cos = {}, {}
    for k, p in zip(names, params):
        d_ref = (p.detach() - state[k].double()).numpy().ravel()
        if d_ref.size == 0 or np.linalg.norm(d_ref) == 0.0:
            continue
        d_gpu = (sd[k].detach().double().cpu() - state[k].double()).numpy().ravel()
        errs[k] = float(np.linalg.norm(d_gpu - d_ref) / np.linalg.norm(d_ref))
        cos[k] = float(d_gpu @ d_ref / max(np.linalg.norm(d_gpu) * np.linalg.norm(d_ref), 1e-300))
    return errs, cos


def test_configs1_trajectory_200_steps():
    import hlvae_oracle as orc
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.datafeed import CompactDataset
    from hlvae_amd.training import ELBOTrainer
    dev = torch.device("cuda:0")
    n_steps, B = 200, 512
    src = synthetic.make_d4(n_subjects=50, T=20, seed=100)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(0)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    P_total = 50
    ring = _ring(src, B)
    gen = torch.Generator().manual_seed(2024)
    eps = [torch.randn(B, dims[2], generator=gen) for _ in range(n_steps)]
    # ---- GPU: the fused step from the compact device-resident data set (what bench.py times)
    tr = ELBOTrainer(model, P_total=P_total, kl="normal", max_batch=B)
    dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows_dev = [torch.tensor(r.astype(np.int32), device=dev) for r, _ in ring]
    nll_gpu, kl_gpu = [], []
    for it in range(n_steps):
        r, pb = ring[it % 4]
        tr.step_rows(dsd, rows_dev[it % 4], pb, eps=eps[it].to(dev))
        if it % 20 == 19 or it == 0:
            sc = tr.scalars()
            nll_gpu.append(float(sc["nll_sum"]))
            kl_gpu.append(float(sc["kl"]))
    torch.cuda.synchronize()
    # ---- oracle: the same sequence in fp64
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    st = _oracle_state(state)
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    names = [k for k in st if not k.startswith("hidden.") and k != "_disp_param"]
    params = [st[k] for k in names]
    m1, m2 = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    batches = [(torch.tensor(src.data[r]), torch.tensor(src.mask[r])) for r, _ in ring]
    nll_ref, kl_ref = [], []
    for it in range(n_steps):
        for p in params:
            p.grad = None
        data, mask = batches[it % 4]
        out = om.forward(data, mask, eps[it].double())
        nll = om.loss_function(out["log_p_x"]).sum()
        kl = orc.standard_normal_kl(out["mu"], out["log_var"])
        (nll * P_total / ring[it % 4][1] + kl).backward()
        live = [i for i, p in enumerate(params) if p.grad is not None]       # torch.optim.Adam skips grad-less parameters (D4 has no pos variable)
        orc.adam_step([params[i] for i in live], [params[i].grad for i in live], [m1[i] for i in live], [m2[i] for i in live], it + 1)
        if it % 20 == 19 or it == 0:
            nll_ref.append(float(nll))
            kl_ref.append(float(kl))
    nll_rel = [abs(a - b) / abs(b) for a, b in zip(nll_gpu, nll_ref)]
    kl_rel = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(kl_gpu, kl_ref)]
    errs, cos = _update_errors(model, state, names, params)
    _report("configs1_200_steps", nll_rel_max=max(nll_rel), nll_rel=nll_rel, kl_rel_max=max(kl_rel), nll_first=nll_ref[0], nll_last=nll_ref[-1],
            update_err=errs, update_err_max=max(errs.values()), update_cos=cos, update_cos_min=min(cos.values()))
    assert nll_ref[-1] < 0.9 * nll_ref[0], "the oracle's NLL should have moved over 200 steps (otherwise the comparison is vacuous)"
    # Measured on MI355X (round 3): the NLL stays within 1.2e-3 of the fp64 curve at every 20th step (NLL 1.15e6 -> 6.5e5), the KL
    # within 2.7 %.  The UPDATES of the decoder-side tensors agree to 5 % (relative L2), those of the encoder-side matrices (first
    # Linear, mean / log-variance layers) only to 27-33 %: most of their entries see gradients of the size of the bf16 rounding of
    # the operands (dT, Xn), and Adam's normalisation moves an entry by ~lr per step whatever the gradient's magnitude, so a sign
    # decided by rounding is a full-size step -- the two runs random-walk apart on those entries while the loss does not care.
    # Bounds: ~2 x measured for the curves, and for every dense tensor a cosine >= 0.9 between the two updates.
    assert max(nll_rel) <= 2.5e-3, nll_rel
    assert max(kl_rel) <= 6e-2, kl_rel
    dec = {k: v for k, v in errs.items() if k.startswith(("y_layer", "d_layers", "obs_layer"))}
    assert max(dec.values()) <= 0.15, dec
    assert max(errs.values()) <= 0.6, errs
    dense = {k: v for k, v in cos.items() if k.endswith("weight") and not k.startswith("obs_layer")}
    assert min(dense.values()) >= 0.9, dense


def test_shipped_configuration_trajectory_50_steps():
    """conv + GP prior (config/hlvae_config_file.txt:22,51), hidden 500, latent 32, M = 120: 50 steps of 512 rows."""
    import gp_oracle as gpo
    import hlvae_oracle as orc
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.datafeed import CompactDataset, subject_index
    from hlvae_amd.elbo_functions import GPPriorHIP
    from hlvae_amd.training import ELBOTrainer
    dev = torch.device("cuda:0")
    n_steps, B = 50, 512
    src = synthetic.make_d4(n_subjects=50, T=20, seed=100)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(0)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=True, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    P_total, N_total = 50, 1000
    gp = GPPriorHIP.from_reference_config(model, src, P_total, dev)
    m0, H0, z0 = gp.m.detach().cpu().clone(), gp.H.detach().cpu().clone(), gp.zt_list.detach().cpu().clone()
    kprm = {}
    for row, (which, t, f) in enumerate(gp.slot_names):
        kprm[f"{which}.{t}.scale" if f is None else f"{which}.{t}.{f}.ls"] = gp.prm[row].detach().cpu().clone().requires_grad_(True)
    N_total = float(gp.N_total)
    ring = _ring(src, B)
    gen = torch.Generator().manual_seed(7)
    eps = [torch.randn(B, dims[2], generator=gen) for _ in range(n_steps)]
    tr = ELBOTrainer(model, P_total=P_total, kl="gp", gp=gp, max_batch=B)
    dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows_dev = [torch.tensor(r.astype(np.int32), device=dev) for r, _ in ring]
    groups_dev = [torch.tensor(subject_index(src.labels[r, src.id_covariate]), device=dev) for r, _ in ring]
    nll_gpu, kld_gpu = [], []
    for it in range(n_steps):
        r, pb = ring[it % 4]
        tr.step_rows(dsd, rows_dev[it % 4], pb, eps=eps[it].to(dev), groups=groups_dev[it % 4])
        if it % 10 == 9 or it == 0:
            nll_gpu.append(float(tr.scalars()["nll_sum"]))
            kld_gpu.append(float(gp.last_kld))
    torch.cuda.synchronize()
    assert int(gp.fail.item()) == 0
    # ---- oracle
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = gpo.spec_from_config([2], [], [0], [{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                               {"cont_covariate": 1, "cat_covariate": 4}], [], 2)
    st = _oracle_state(state)
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st, conv=True)
    names = [k for k in st if not k.startswith(("hidden.", "Decoder_Conv_layer.")) and k != "_disp_param"]
    params = [st[k] for k in names]
    m1, m2 = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    z_ = z0.clone().requires_grad_(True)
    gp_leaves = list(kprm.values()) + [z_]
    g1, g2 = [torch.zeros_like(p) for p in gp_leaves], [torch.zeros_like(p) for p in gp_leaves]
    gm_, gH_ = m0.clone(), H0.clone()
    noise = torch.ones(dims[2], dtype=torch.float64)
    batches = [(torch.tensor(src.data[r]), torch.tensor(src.mask[r]), torch.tensor(src.labels[r])) for r, _ in ring]
    nll_ref, kld_ref = [], []
    for it in range(n_steps):
        for p in params + gp_leaves:
            p.grad = None
        data, mask, labels = batches[it % 4]
        pb = ring[it % 4][1]
        out = om.forward(data, mask, eps[it].double())
        nll = om.loss_function(out["log_p_x"]).sum()
        kld, grad_m, grad_H = gpo.minibatch_kld_upper_bound_iter(spec, kprm, noise, dims[2], gm_, gH_, labels, out["mu"], out["log_var"], z_,
                                                                 P_total, pb, N_total, True, 2, 1e-6)
        (nll * P_total / pb + kld.sum()).backward()
        live = [i for i, p in enumerate(params) if p.grad is not None]
        orc.adam_step([params[i] for i in live], [params[i].grad for i in live], [m1[i] for i in live], [m2[i] for i in live], it + 1)
        orc.adam_step(gp_leaves, [p.grad for p in gp_leaves], g1, g2, it + 1)
        gm_, gH_ = gpo.natural_gradient_update(gm_, gH_, grad_m.detach(), grad_H.detach(), 0.01)
        if it % 10 == 9 or it == 0:
            nll_ref.append(float(nll))
            kld_ref.append(float(kld.sum()))
    nll_rel = [abs(a - b) / abs(b) for a, b in zip(nll_gpu, nll_ref)]
    kld_rel = [abs(a - b) / abs(b) for a, b in zip(kld_gpu, kld_ref)]
    errs, cos = _update_errors(model, state, names, params)
    _report("shipped_conv_gp_50_steps", nll_rel_max=max(nll_rel), nll_rel=nll_rel, kld_rel_max=max(kld_rel), kld_rel=kld_rel,
            gp_m=rel_err(gp.m, gm_), gp_H=rel_err(gp.H, gH_), gp_z=rel_err(gp.zt_list, z_.detach()), update_err=errs,
            update_err_max=max(errs.values()), update_cos=cos, update_cos_min=min(cos.values()))
    # (convolutional model: bf16 storage moves a few ReLU / max-pool gates, DESIGN.md section 4.3 -- the curves part by 1.0e-3 around
    #  step 20 and close again to 5e-5 by step 50; bound = 3 x measured)
    # measured (round 3): NLL 1.3e-3, GP bound 1.7e-2 (it is a function of the encoder's mu / log_var, which leave through bf16
    # products and whose weights random-walk as described above), m 2.8e-2, H 1.3e-4, inducing points 2e-7
    assert max(nll_rel) <= 3e-3, nll_rel
    assert max(kld_rel) <= 5e-2, kld_rel
    assert rel_err(gp.m, gm_) <= 8e-2 and rel_err(gp.H, gH_) <= 1e-3, (rel_err(gp.m, gm_), rel_err(gp.H, gH_))
    assert rel_err(gp.zt_list, z_.detach()) <= 1e-5
