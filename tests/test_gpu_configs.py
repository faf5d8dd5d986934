"""GPU parity at the sizes of BASELINE.json configs[2..4] (round-2 additions):

  * D4 at 1024 and 4096 rows -- above 2048 rows the fused middle kernels use 16-row tiles, the head kernel gives every XCD its
    own row blocks and y_layer's optimiser launch moves ahead of dU_splitk (csrc/heads.hip, csrc/cabi.hip):
    forward, every gradient tensor and one fused ELBOTrainer step from the compact feed against the fp64 oracle;
  * the full config-5 step: HIP decoder + GPPriorHIP on a 1024-row batch of 51 whole subjects x 20 rows + 4 rows of a
    52nd, against hlvae_oracle + gp_oracle;
  * GPPriorHIP loaded from the reference's own output tests/golden/gp_kl.npz (reference elbo_functions.py:196-285,
    training.py:130-137);
  * the convolutional model's BACKWARD pass at 512 and 1024 rows against the oracle.

Tolerances: ELBO / loss 1e-4 relative (north star); gradients per tensor 2.5e-2 relative L2 (bf16 operands of both backward
GEMMs set the floor: 1-2e-2 measured on the first encoder Linear); fp64 GP quantities 1e-7..1e-9; fp32 GP outputs 1e-5."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hlvae_amd                      # noqa: E402
from hlvae_amd import synthetic       # noqa: E402
from tests_common import max_abs_err, rel_err   # noqa: E402

ELBO_RTOL = 1e-4
GRAD_RTOL = 2.5e-2
REPORT = {}
GP_CFG = dict(cat=[2], bin=[], sqexp=[0], cat_int=[{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                                   {"cont_covariate": 1, "cat_covariate": 4}], bin_int=[])


def _report(key, **kv):
    REPORT.setdefault(key, {}).update({k: float(v) for k, v in kv.items()})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report_configs.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _oracle_state(state, conv=False):
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items()
          if not k.startswith(("hidden.", "Decoder_Conv_layer."))}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
        if k.startswith("deconv_layer."):
            st["Decoder_Conv_layer." + k[len("deconv_layer."):]] = st[k]
    return st


def _adam_reference(st):
    """names / leaves the reference's torch.optim.Adam would update (training.py:127-128)"""
    names = [k for k in st if not k.startswith(("hidden.", "Decoder_Conv_layer.")) and k != "_disp_param"]
    return names, [st[k] for k in names]


@pytest.mark.parametrize("B,hid_e,hid_d", [(1024, [500], [500]), (4096, [500], [500]), (1024, [500, 260], [132, 260, 500]),
                                           (1024, [], []), (512, [], [500]), (512, [500], [])],
                         ids=["1024", "4096", "1024-deep", "1024-nohid", "512-nohid_e", "512-nohid_d"])
def test_d4_large_batch_against_oracle(B, hid_e, hid_d):
    """BASELINE configs[2] (global batch 4096) and configs[4] (batch 1024) on the D4 layout, MLP [5184,[500],32,[500],5]; and the
    same layout with deeper trunks at model scale (two encoder layers, three decoder layers, widths that need padding:
    HLVAE.py:113, 125-137, 232-242), every layer's gradient and the fused optimiser step included."""
    import hlvae_oracle as orc
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    n_subj = (B + 19) // 20 + 1
    src = synthetic.make_d4(n_subjects=n_subj, T=20, seed=100 + B)
    rows = np.arange(B)
    P_batch = int(np.unique(src.labels[rows, 2]).size)
    P_total = 4 * n_subj
    scale = P_total / P_batch
    dims = [src.cov_dim_ext, hid_e, 32, hid_d, 5]
    torch.manual_seed(1234 + B)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(5))
    data = torch.tensor(src.data[rows], device=dev)
    mask = torch.tensor(src.mask[rows], device=dev)
    # ---- autograd surface: forward + every gradient
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    mu, lv, lpx = out[1], out[2], out[3]
    loss = scale * model.loss_function(lpx).sum() - 0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss.backward()
    torch.cuda.synchronize()
    st = _oracle_state(state)
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    ref = om.forward(torch.tensor(src.data[rows]), torch.tensor(src.mask[rows]), eps.double())
    nll_ref = om.loss_function(ref["log_p_x"]).sum()
    kl_ref = orc.standard_normal_kl(ref["mu"], ref["log_var"])
    ref_loss = scale * nll_ref + kl_ref
    ref_loss.backward()
    elbo, elbo_ref = float(lpx.double().sum()), float(ref["log_p_x"].sum())
    rel = abs(elbo - elbo_ref) / abs(elbo_ref)
    key = f"d4_b{B}" + ("_nohid" + "e" * (not hid_e) + "d" * (not hid_d) if not (hid_e and hid_d) else
                         "" if len(hid_e) + len(hid_d) == 2 else "_deep")
    _report(key, elbo_rel=rel, loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)),
            mu=max_abs_err(mu.cpu(), ref["mu"].detach()), lv=max_abs_err(lv.cpu(), ref["log_var"].detach()),
            lpx_max=max_abs_err(lpx.cpu(), ref["log_p_x"].detach()))
    assert rel <= ELBO_RTOL, rel
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss))
    e = np.abs(lpx.detach().double().cpu().numpy() - ref["log_p_x"].detach().numpy())
    assert np.all(e <= 3e-2 + 2e-2 * np.abs(ref["log_p_x"].detach().numpy()))
    e = np.abs(out[4].detach().double().cpu().numpy() - ref["log_p_x_missing"].detach().numpy())
    assert np.all(e <= 3e-2 + 2e-2 * np.abs(ref["log_p_x_missing"].detach().numpy()))
    sd = dict(model.named_parameters())
    n_checked, errs, deep = 0, {}, len(hid_e) + len(hid_d) > 2
    for k, p in sd.items():
        if p.grad is None or st[k].grad is None:
            assert k == "_disp_param" or p.numel() == 0, k
            continue
        err = rel_err(p.grad.double().cpu().numpy(), st[k].grad.numpy())
        _report(key + "_grads", **{k: err})
        errs[k] = err
        n_checked += 1
    for k, err in errs.items():
        # the first decoder Linear [h_d0][32]: its pre-activations are the smallest of the model (32 inputs), so bf16 rounding
        # flips the most ReLU gates there, each a whole term of the 1024-row sum (1.5e-2 with one hidden layer, 3.5e-2 below
        # two more bf16 layers)
        assert err < (5e-2 if deep and k == "d_layers.0.weight" else GRAD_RTOL), (k, err)
    assert n_checked >= 12 + 2 * (len(hid_e) + len(hid_d) - 2)
    # ---- the fused training step (no autograd) from the compact feed: same weights, same noise
    model2 = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False)
    model2.load_state_dict(state)
    model2 = model2.to(dev)
    tr = ELBOTrainer(model2, P_total=P_total, kl="normal", max_batch=B, metrics=True)
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows_dev = torch.tensor(rows.astype(np.int32), device=dev)
    tr.step_rows(ds, rows_dev, P_batch, eps=eps.to(dev))
    torch.cuda.synchronize()
    nll_gpu, kl_gpu = float(tr.scalars()["nll_sum"]), float(tr.scalars()["kl"])
    _report(key + "_step", nll_rel=abs(nll_gpu - float(nll_ref)) / abs(float(nll_ref)), kl_rel=abs(kl_gpu - float(kl_ref)) / abs(float(kl_ref)))
    assert abs(nll_gpu - float(nll_ref)) <= ELBO_RTOL * abs(float(nll_ref))
    assert abs(kl_gpu - float(kl_ref)) <= 2e-3 * abs(float(kl_ref))
    names, params = _adam_reference(st)
    live = [i for i, p in enumerate(params) if p.grad is not None]
    before = {names[i]: params[i].detach().clone() for i in live}
    orc.adam_step([params[i] for i in live], [params[i].grad for i in live], [torch.zeros_like(params[i]) for i in live],
                  [torch.zeros_like(params[i]) for i in live], 1)
    sd2 = dict(model2.named_parameters())
    for i in live:
        k = names[i]
        if params[i].numel() == 0:
            continue
        d_ref = (params[i].detach() - before[k]).numpy()
        d_gpu = (sd2[k].detach().double().cpu() - state[k].double()).numpy()
        bad = np.abs(d_gpu - d_ref) > 1e-3                    # first Adam step: +-lr wherever the gradient keeps its sign
        _report(key + "_step", **{"flip__" + k: bad.mean()})
        # per-tensor bound = 3 x the measured share: 4e-3 on the one-hidden-layer model; the decoder layers of the deeper trunk sit
        # below two more bf16 ReLU layers whose gates flip (1.6e-2 measured on d_layers.0.weight)
        deep_dec = len(hid_d) > 1 and k.startswith("d_layers.")
        assert bad.mean() <= (5e-2 if deep_dec else 1.2e-2), (k, bad.mean())
    # per-step metrics (row M) ran beside the backward pass: finite, inside [0, 1] for the discrete variables
    err = tr.err.cpu().numpy()
    assert np.isfinite(err).all() and err.min() >= 0.0
    disc = np.isin(model2.plan.kind, [3, 4])
    assert err[:, disc].max() <= 1.0


def _gp_config5_inputs(dev, seed=1):
    Ts = [20] * 51 + [4]
    rows = []
    for s_, T in enumerate(Ts):
        for t in range(T):
            sick = s_ % 2
            rows.append([float(t), float(t - 9) if sick else 0.0, float(s_), float(s_ % 2), float(sick), float((s_ // 2) % 2)])
    return torch.tensor(rows, dtype=torch.float64), len(Ts)


def _gp_oracle_params(gp):
    """hyper-parameter rows of a GPPriorHIP -> gp_oracle's parameter dict (leaves requiring grad)"""
    prm = {}
    for row, (which, t, f) in enumerate(gp.slot_names):
        key = f"{which}.{t}.scale" if f is None else f"{which}.{t}.{f}.ls"
        prm[key] = gp.prm[row].detach().cpu().clone().requires_grad_(True)
    return prm


def test_config5_full_step_against_oracles():
    """BASELINE configs[4]: GP-prior KL alongside the HIP decoder, batch 1024 = 51 subjects x 20 rows + 4 rows of a 52nd,
    L = 32, M = 120, the shipped kernel structure (config/hlvae_config_file.txt:41-45).  One fused ELBOTrainer step against
    hlvae_oracle (VAE half) + gp_oracle (reference elbo_functions.py:196-285, training.py:130-137)."""
    import gp_oracle as gpo
    import hlvae_oracle as orc
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.elbo_functions import GPPriorHIP
    from hlvae_amd.training import ELBOTrainer
    dev = _dev()
    src = synthetic.make_d4(n_subjects=52, T=20, seed=77)
    B = 1024
    rows = np.arange(B)                                     # 51 whole subjects + 4 rows of the 52nd
    P_batch, P_total, N_total = 52, 2500, 50000
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(99)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    labels = torch.tensor(src.labels[rows], device=dev)
    gp = GPPriorHIP(dims[2], torch.tensor(src.labels, device=dev), 120, 2, N_total=N_total, seed=3)
    with torch.no_grad():
        gp.prm.add_(0.2 * torch.randn(gp.prm.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev))
    m0, H0, z0 = gp.m.clone().cpu(), gp.H.clone().cpu(), gp.zt_list.detach().clone().cpu()
    kprm = _gp_oracle_params(gp)
    tr = ELBOTrainer(model, P_total=P_total, kl="gp", gp=gp, max_batch=B)
    eps = torch.randn(B, dims[2], generator=torch.Generator().manual_seed(9))
    data, mask = torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev)
    tr.step(data, mask, P_batch, eps=eps.to(dev), train_x=labels)
    torch.cuda.synchronize()
    assert int(gp.fail.item()) == 0
    nll_gpu, kld_gpu = float(tr.scalars()["nll_sum"]), float(gp.last_kld)
    mu_dev, lv_dev = model._ws_t["mu"][:B].double().cpu(), model._ws_t["lv"][:B].double().cpu()
    # ---- GP half, tight: the oracle evaluated at the encoder outputs the device produced (fp64 against fp64)
    spec = gpo.spec_from_config(GP_CFG["cat"], GP_CFG["bin"], GP_CFG["sqexp"], GP_CFG["cat_int"], GP_CFG["bin_int"], 2)
    mu_t, lv_t = mu_dev.clone().requires_grad_(True), lv_dev.clone().requires_grad_(True)
    z_t = z0.clone().requires_grad_(True)
    noise = torch.ones(dims[2], dtype=torch.float64)
    kld, gm, gH = gpo.minibatch_kld_upper_bound_iter(spec, kprm, noise, dims[2], m0, H0, torch.tensor(src.labels[rows]), mu_t, lv_t,
                                                     z_t, P_total, P_batch, N_total, True, 2, 1e-6)
    kld.sum().backward()
    _report("config5", kld_rel=abs(kld_gpu - float(kld)) / abs(float(kld)), grad_m=rel_err(gp._grad_m, gm.detach()),
            grad_H=rel_err(gp._grad_H, gH.detach()))
    # K0zz of 120 inducing points drawn from 1040 covariate rows with a 1e-6 jitter has a condition number ~1e8: Gauss-Jordan
    # (device) and Cholesky (oracle) inverses agree to ~1e-7, and so does everything built from iK (measured 5e-8 / 6e-7)
    assert abs(kld_gpu - float(kld)) <= 5e-7 * abs(float(kld)), (kld_gpu, float(kld))
    assert rel_err(gp._grad_m, gm.detach()) < 5e-6 and rel_err(gp._grad_H, gH.detach()) < 5e-6
    m_ref, H_ref = gpo.natural_gradient_update(m0, H0, gm.detach(), gH.detach(), 0.01)
    _report("config5", m_new=rel_err(gp.m, m_ref), H_new=rel_err(gp.H, H_ref))
    assert rel_err(gp.m, m_ref) < 5e-6 and rel_err(gp.H, H_ref) < 5e-6
    # Adam's first update of the GP parameters is -lr * sign(g) wherever g != 0 (HLVAE_main.py:277-278)
    dz = (gp.zt_list.detach().cpu() - z0)
    big = z_t.grad.abs() > 1e-3 * z_t.grad.abs().max()
    assert (torch.sign(dz[big]) == -torch.sign(z_t.grad[big])).double().mean() > 0.999
    for row, (which, t, f) in enumerate(gp.slot_names):
        key = f"{which}.{t}.scale" if f is None else f"{which}.{t}.{f}.ls"
        g_ref = kprm[key].grad
        d = gp.prm[row].detach().cpu() - kprm[key].detach()
        nz = g_ref.abs() > 1e-4                              # |step| = lr |g| / (|g| + 1e-8)
        assert torch.equal(torch.sign(d[nz]), -torch.sign(g_ref[nz])), key
        if bool(nz.any()):
            assert float((d[nz].abs() - 1e-3).abs().max()) < 2e-7, key
    # ---- VAE half + coupling: oracle forward with the same weights and noise
    st = _oracle_state(state)
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    out = om.forward(torch.tensor(src.data[rows]), torch.tensor(src.mask[rows]), eps.double())
    nll = om.loss_function(out["log_p_x"]).sum()
    kld2, _, _ = gpo.minibatch_kld_upper_bound_iter(spec, {k: v.detach() for k, v in kprm.items()}, noise, dims[2], m0, H0,
                                                    torch.tensor(src.labels[rows]), out["mu"], out["log_var"], z0, P_total, P_batch,
                                                    N_total, True, 2, 1e-6)
    (nll * P_total / P_batch + kld2.sum()).backward()
    _report("config5", nll_rel=abs(nll_gpu - float(nll)) / abs(float(nll)), kld_e2e_rel=abs(kld_gpu - float(kld2)) / abs(float(kld2)))
    assert abs(nll_gpu - float(nll)) <= ELBO_RTOL * abs(float(nll))
    assert abs(kld_gpu - float(kld2)) <= 1e-3 * abs(float(kld2))           # bf16 encoder outputs enter the bound
    # the encoder feels the GP gradient through mu / log_var: direction of the first Adam update of the mean / log-var layers
    sd = dict(model.named_parameters())
    for k in ("mean_layer.0.weight", "log_var_layer.0.weight", "VAE_encoder_common_layers.0.weight", "y_layer.0.weight"):
        delta = sd[k].detach().double().cpu() - state[k].double()
        gref = st[k].grad
        big = gref.abs() > 0.05 * gref.abs().max()
        agree = float((torch.sign(delta[big]) == -torch.sign(gref[big])).double().mean())
        _report("config5", **{"sign__" + k: agree})
        assert agree > 0.98, (k, agree)


def test_gp_prior_hip_against_reference_fixture(golden_dir):
    """GPPriorHIP straight against tests/golden/gp_kl.npz -- the output of the reference's own
    minibatch_KLD_upper_bound_iter (elbo_functions.py:196-285) and natural-gradient step (training.py:130-137): bound,
    grad_m, grad_H, gradients w.r.t. mu / log-variance / inducing points / every hyper-parameter, m_new, H_new."""
    from hlvae_amd.elbo_functions import GPPriorHIP
    g = np.load(os.path.join(golden_dir, "gp_kl.npz"))
    dev = _dev()
    P, P_b, N, eps, idc, lr = (float(v) for v in g["scalars"])
    x = torch.tensor(g["x"], device=dev)
    L, M = g["mu"].shape[1], g["z"].shape[1]
    gp = GPPriorHIP(L, x, M, int(idc), N_total=N, eps=eps, natural_gradient_lr=lr)
    with torch.no_grad():
        gp.zt_list.copy_(torch.tensor(g["z"], device=dev))
        for row, (which, t, f) in enumerate(gp.slot_names):
            key = f"kp__{which}.{t}.scale" if f is None else f"kp__{which}.{t}.{f}.ls"
            gp.prm[row].copy_(torch.tensor(g[key], device=dev))
        gp.m.copy_(torch.tensor(g["m"], device=dev))
        gp.H.copy_(torch.tensor(g["H"], device=dev))
        gp.noise.copy_(torch.tensor(g["noise"], device=dev))
    assert len(gp.slot_names) == sum(1 for k in g.files if k.startswith("kp__"))
    mu = torch.tensor(g["mu"], device=dev).float()
    lv = torch.tensor(g["log_v"], device=dev).float()
    # the fixture's mu / log_v are fp64; the product takes the VAE's fp32 outputs: feed the fp32-rounded values to BOTH sides
    # for the tight comparison of everything but d_mu / d_log_v ... the reference saw the unrounded ones, so compare the bound
    # at fp32-input accuracy (1e-6) and the derivative outputs (fp32) at 1e-5
    g_mu, g_lv = gp.kl_and_grads(mu, lv, x, P, P_b)
    torch.cuda.synchronize()
    assert int(gp.fail.item()) == 0
    res = dict(kld=rel_err(gp.last_kld, g["kld"]), grad_m=rel_err(gp._grad_m, g["grad_m"]), grad_H=rel_err(gp._grad_H, g["grad_H"]),
               d_mu=rel_err(g_mu, g["d_mu"]), d_log_v=rel_err(g_lv, g["d_log_v"]), d_z=rel_err(gp.zt_list.grad, g["d_z"]))
    for row, (which, t, f) in enumerate(gp.slot_names):
        key = f"{which}.{t}.scale" if f is None else f"{which}.{t}.{f}.ls"
        res["kg__" + key] = rel_err(gp.prm.grad[row], g["kg__" + key])
    _report("gp_kl_fixture", **res)
    # measured on MI355X: kld 1.4e-9, grad_m 4.5e-9, grad_H 3e-16, d_mu / d_log_v 3e-8 (fp32 outputs of fp32-rounded inputs),
    # d_z 1.5e-8, hyper-parameter gradients <= 1e-8
    assert res["kld"] < 1e-8 and res["grad_m"] < 5e-8 and res["grad_H"] < 1e-10
    assert res["d_mu"] < 3e-7 and res["d_log_v"] < 3e-7 and res["d_z"] < 1e-7
    for k, v in res.items():
        if k.startswith("kg__"):
            assert v < 1e-7, (k, v)
    gp.optimizer_step()
    torch.cuda.synchronize()
    assert int(gp.fail.item()) == 0
    _report("gp_kl_fixture", m_new=rel_err(gp.m, g["m_new"]), H_new=rel_err(gp.H, g["H_new"]))
    assert rel_err(gp.m, g["m_new"]) < 1e-9 and rel_err(gp.H, g["H_new"]) < 1e-10      # measured 7e-12 / 5e-16


@pytest.mark.parametrize("B,hid_e,hid_d", [(512, [500], [500]), (1024, [500], [500]), (512, [500, 132], [260, 500]), (512, [], [])],
                         ids=["512", "1024", "512-deep", "512-nohid"])
def test_conv_backward_against_oracle(B, hid_e, hid_d):
    """convolutional model (what config/hlvae_config_file.txt:51 selects) at 512 and 1024 rows, hidden 500, latent 32 -- and
    (round 3) with two hidden layers either side of the convolutional stages: ELBO, loss and EVERY gradient tensor against
    the fp64 oracle on the same weights and noise (the head kernel runs in its `ysrc` mode: the tile of y_grouped comes from
    the second transposed convolution)."""
    import hlvae_oracle as orc
    from hlvae_amd.HLVAE import HLVAE
    dev = _dev()
    src = synthetic.make_d4(n_subjects=(B + 19) // 20 + 1, T=20, seed=100)
    dims = [src.cov_dim_ext, hid_e, 32, hid_d, 5]
    torch.manual_seed(3)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=True, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(1))
    data, mask = torch.tensor(src.data[:B]), torch.tensor(src.mask[:B])
    scale = 7.5
    out = model(data.to(dev), mask.to(dev), None, src.types_info, eps=eps.to(dev))
    mu, lv, lpx = out[1], out[2], out[3]
    loss = scale * model.loss_function(lpx).sum() - 0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss.backward()
    torch.cuda.synchronize()
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
        if k.startswith("Decoder_Conv_layer."):
            st[k] = st["deconv_layer." + k[len("Decoder_Conv_layer."):]]
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st, conv=True)
    ref = om.forward(data, mask, eps.double())
    ref_loss = scale * om.loss_function(ref["log_p_x"]).sum() + orc.standard_normal_kl(ref["mu"], ref["log_var"])
    ref_loss.backward()
    elbo, elbo_ref = float(lpx.double().sum()), float(ref["log_p_x"].sum())
    key = f"conv_b{B}" + ("_deep" if len(hid_e) > 1 else "_nohid" if not hid_e else "")
    _report(key, elbo_rel=abs(elbo - elbo_ref) / abs(elbo_ref), loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    assert abs(elbo - elbo_ref) <= ELBO_RTOL * abs(elbo_ref), (elbo, elbo_ref)
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss))
    sd = dict(model.named_parameters())
    errs = {}
    for k, p in sd.items():
        if k.startswith(("hidden.", "Decoder_Conv_layer.")) or p.grad is None or st[k].grad is None or p.numel() == 0:
            continue
        errs[k] = rel_err(p.grad.double().cpu().numpy(), st[k].grad.numpy())
    _report(key + "_grads", **errs)
    assert len(errs) >= 20, sorted(errs)
    for k, e in errs.items():
        # d_layers.0.weight = dU^T z sums B signed terms per element with heavy cancellation, and dU has passed two transposed
        # convolutions in bf16 with ReLU gates: 3.4e-2 at 512 rows, 2.6e-2 at 1024 (falls as 1 / sqrt(B): rounding noise, not
        # logic; the MLP path shows 1.4e-2 / 0.7e-2 at 1024 / 4096 rows).  Everything else sits inside the MLP path's bound.
        tol = 5e-2 if k == "d_layers.0.weight" else GRAD_RTOL
        assert e < tol, (k, e)


def test_conv_logvar_network_with_deeper_trunks_against_oracle():
    """the two round-3 constructor combinations TOGETHER (conv=True, logvar_network=True, two hidden layers per side), 512 rows:
    ELBO, loss and every gradient tensor against the fp64 oracle (pinned to the reference on each combination separately:
    d4_conv_logvar_small, d4_conv_deep_small)."""
    import hlvae_oracle as orc
    from hlvae_amd import layout
    from hlvae_amd.HLVAE import HLVAE
    dev = _dev()
    B = 512
    src = synthetic.make_d4(n_subjects=B // 20 + 2, T=20, seed=101)
    info = layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True)
    dims = [src.cov_dim_ext, [300, 132], 32, [260, 300], 5]
    torch.manual_seed(7)
    model = HLVAE(dims, info, src.n_variables, conv=True, logvar_network=True, max_batch=B, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(2))
    data, mask = torch.tensor(src.data[:B]), torch.tensor(src.mask[:B])
    scale = 7.5
    out = model(data.to(dev), mask.to(dev), None, info, eps=eps.to(dev))
    mu, lv, lpx = out[1], out[2], out[3]
    loss = scale * model.loss_function(lpx).sum() - 0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss.backward()
    torch.cuda.synchronize()
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
        if k.startswith("Decoder_Conv_layer."):
            st[k] = st["deconv_layer." + k[len("Decoder_Conv_layer."):]]
    om = orc.OracleHLVAE(dims, info, src.n_variables, st, conv=True)
    ref = om.forward(data, mask, eps.double())
    ref_loss = scale * om.loss_function(ref["log_p_x"]).sum() + orc.standard_normal_kl(ref["mu"], ref["log_var"])
    ref_loss.backward()
    elbo, elbo_ref = float(lpx.detach().double().sum()), float(ref["log_p_x"].sum())
    _report("conv_logvar_deep_b512", elbo_rel=abs(elbo - elbo_ref) / abs(elbo_ref),
            loss_rel=abs(float(loss.detach()) - float(ref_loss)) / abs(float(ref_loss)))
    assert abs(elbo - elbo_ref) <= ELBO_RTOL * abs(elbo_ref), (elbo, elbo_ref)
    assert abs(float(loss.detach()) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss))
    sd = dict(model.named_parameters())
    errs = {}
    for k, p in sd.items():
        if k.startswith(("hidden.", "Decoder_Conv_layer.")) or p.grad is None or st[k].grad is None or p.numel() == 0:
            continue
        errs[k] = rel_err(p.grad.double().cpu().numpy(), st[k].grad.numpy())
    _report("conv_logvar_deep_b512_grads", **errs)
    assert len(errs) >= 24, sorted(errs)
    for k, e in errs.items():
        # d_layers.0.weight = dU^T z: B signed terms per element with heavy cancellation, dU behind two transposed convolutions AND a
        # second decoder layer in bf16 (test_conv_backward_against_oracle: 3.4e-2 with one layer, 4.9e-2 with two; here 5.3e-2,
        # every other tensor <= 1.5e-2)
        assert e < (7.5e-2 if k == "d_layers.0.weight" else GRAD_RTOL), (k, e)


@pytest.mark.parametrize("hid_e,hid_d,conv", [([500], [500], False), ([500, 132], [260, 500], False), ([500], [500], True),
                                              ([500, 132], [260, 500], True), ([], [], False), ([], [], True)],
                         ids=["mlp", "deep", "conv", "conv-deep", "nohid", "conv-nohid"])
def test_sharded_optimizer_path_matches_fused_path(hid_e, hid_d, conv):
    """The data-parallel optimiser path at world size 1 (flat Adam on the whole dense region -> bf16 copy -> shadows rebuilt
    from it; hl-vae_amd/parallel.py with the collectives skipped) against the fused tile Adam of the single-process step: D4,
    512 rows, three steps with the same noise -- same arithmetic, so the parameters agree to fp32 atomics' reordering and the
    shadows bit for bit."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.parallel import DataParallel
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    src = synthetic.make_d4(n_subjects=30, T=20, seed=11)
    dims = [src.cov_dim_ext, hid_e, 32, hid_d, 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows = [torch.tensor(np.arange(i * 40, i * 40 + 512).astype(np.int32), device=dev) for i in range(2)]
    eps = [torch.randn(512, 32, generator=torch.Generator().manual_seed(40 + i)).to(dev) for i in range(3)]
    res, named = [], []
    for dp in (None, DataParallel.single()):
        torch.manual_seed(5)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=conv, max_batch=512, materialize_samples=False).to(dev)
        tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=512, dp=dp, metrics=True)
        nll = []
        if not conv:
            tr.prime_rows(ds, rows[0])
        for i in range(3):
            if conv:        # (the convolutional input stage reads the representation layer's weights: it cannot run a step ahead)
                tr.step_rows(ds, rows[i % 2], 26, eps=eps[i])
            else:
                tr.step_rows(ds, rows[i % 2], 26, eps=eps[i], prefetch_rows=rows[(i + 1) % 2], prepacked=True)
            nll.append(float(tr.scalars()["nll_sum"]))
        model.state_dict()              # (data-parallel path: finishes y_layer's all-gather + shadow rebuild left running)
        torch.cuda.synchronize()
        assert int(tr.opt.step_count[0]) == 3
        named.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        res.append((nll, model._arena.clone(), {k: model._ws_t[k].clone() for k in model._ws_t
                                                if k in ("wys", "wyTs", "w1s", "w1Ts", "wds", "wdTs", "wmls", "wmlTs") or k.endswith(("_w", "_wT"))}))
    (nll_a, P_a, sh_a), (nll_b, P_b, sh_b) = res
    assert rel_err(np.array(nll_b), np.array(nll_a)) < 1e-6, (nll_a, nll_b)
    # conv: the two paths fold the convolutions' per-workgroup partial gradient rows in different launches (k_conv_grad_finish before
    # the tile Adam / before the flat Adam) and the feature gradient's atomics reorder; Adam's normalisation turns the last-bit noise
    # of near-zero gradients into visible steps in the first updates (measured up to 9e-6 of the largest parameter after three steps; the NLL of all three steps agrees to 1e-10)
    p_tol, flip_tol = (2e-4, 5e-2) if conv else (1e-6, 1e-4)       # (conv, measured over five runs: 1.3e-7 ... 8.6e-6 and 1e-5 ... 7.9e-3)
    per = {k: float((named[0][k].double() - named[1][k].double()).abs().max() / (named[0][k].double().abs().max() + 1e-30))
           for k in named[0] if named[0][k].numel()}
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:6]
    flips = {k: float((sh_a[k] != sh_b[k]).float().mean()) for k in sh_a}
    print("sharded-vs-fused worst tensors", worst, "shadow flips", flips)
    _report("sharded_vs_fused_" + ("conv" if conv else "mlp") + ("_deep" if len(hid_e) > 1 else "_nohid" if not hid_e else ""), params=rel_err(P_b, P_a),
            nll=rel_err(np.array(nll_b), np.array(nll_a)), shadow_flips=max(float((sh_a[k] != sh_b[k]).float().mean()) for k in sh_a))
    assert rel_err(P_b, P_a) < p_tol
    for k in sh_a:
        # (at most ONE bf16 ulp: 2^-7 of the element at the top of its binade)
        assert float((sh_a[k].float() - sh_b[k].float()).abs().max()) <= 2.0 ** -7 * float(sh_a[k].float().abs().max()), k
        assert float((sh_a[k] != sh_b[k]).float().mean()) < flip_tol, k       # a last-bit difference of a master may flip a rounding


def test_vy_fixed_parameters_stay_put_under_the_fused_optimiser():
    """vy_fixed = True (reference HLVAE.py:209-216: _log_vy_real / _log_vy_pos without requires_grad): torch.optim.Adam skips
    them; the fused optimiser step must too, while every other parameter trains."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    dev = _dev()
    src = synthetic.make_tabular(n_rows=96, T=8, seed=3)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    model = HLVAE(dims, src.types_info, src.n_variables, vy_fixed=True, conv=False, max_batch=128, materialize_samples=False).to(dev)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    tr = ELBOTrainer(model, P_total=12, kl="normal", max_batch=128)
    data, mask = torch.tensor(src.data, device=dev), torch.tensor(src.mask, device=dev)
    for _ in range(3):
        tr.step(data, mask, 12)
    torch.cuda.synchronize()
    after = dict(model.named_parameters())
    assert torch.equal(after["_log_vy_real"], before["_log_vy_real"]) and torch.equal(after["_log_vy_pos"], before["_log_vy_pos"])
    for k in ("obs_layer.0.weight", "y_layer.0.bias", "d_layers.0.bias", "y_layer.0.weight"):
        assert not torch.equal(after[k], before[k]), k
    d = model._dims
    assert float(tr.opt.m1[int(d.frozen_lo):int(d.frozen_hi)].abs().max()) == 0.0
    assert float(model._grad_arena[int(d.frozen_lo):int(d.frozen_hi)].abs().max()) == 0.0      # cleared for the next step's atomics


@pytest.mark.parametrize("name", ["mix_logvar", "mix_deep", "mix_logvar_deep", "mix_nohid_e", "mix_nohid_d", "mix_nohid"])
def test_constructor_modes_against_reference_fixture(golden_dir, name):
    """The two constructor modes the shipped configuration does not use, against outputs of the reference itself
    (tests/golden/make_golden.py: case_mix_logvar_deep): logvar_network=True -- the variance of every real / pos ENTRY comes
    from a second head (HLVAE.py:25-51; loglik.py:45-47, 105) -- and two hidden layers per side (HLVAE.py:113, 125-137,
    232-242).  Forward, loss, every gradient, p_params in the reference's shapes, state_dict keys, fused training steps."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from tests_common import load_mode_case
    g, src, dims, state, info, lvn = load_mode_case(golden_dir, name)
    dev = _dev()
    model = HLVAE(dims, info, src.n_variables, vy_init=[1.0, 0.5], logvar_network=lvn, conv=False, max_batch=128,
                  materialize_samples=False)
    assert set(model.state_dict().keys()) == set(state.keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(state[k].shape), k
    model.load_state_dict(state)
    model = model.to(dev)
    data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    eps = torch.tensor(g["eps"], device=dev)
    p_samples, mu, lv, lpx, lpm, p_params, q_samples, q_params = model(data, mask, None, info, eps=eps)
    torch.cuda.synchronize()
    e_lpx = np.abs(lpx.detach().double().cpu().numpy() - g["log_p_x"])
    elbo, elbo_ref = float(lpx.double().sum()), float(g["log_p_x"].sum())
    _report(name, mu=max_abs_err(mu.cpu(), g["mu"]), lv=max_abs_err(lv.cpu(), g["log_var"]), lpx_max=e_lpx.max(),
            elbo_rel=abs(elbo - elbo_ref) / abs(elbo_ref))
    assert max_abs_err(mu.cpu(), g["mu"]) < 2e-2 and max_abs_err(lv.cpu(), g["log_var"]) < 2e-2
    assert np.all(e_lpx <= 3e-2 + 2e-2 * np.abs(g["log_p_x"]))
    assert abs(elbo - elbo_ref) <= ELBO_RTOL * abs(elbo_ref)
    for i, p in enumerate(p_params["x"]):
        ref = g[f"p_params_{i}"]
        if isinstance(p, list):                       # [est_mean, est_var] of a real / pos block under logvar_network
            assert len(p) == 2 and tuple(p[0].shape) == tuple(p[1].shape)
            p = torch.cat(p, 1)
        assert tuple(p.shape) == ref.shape, i
        assert max_abs_err(p.cpu(), ref) <= 3e-2 + 2e-2 * np.abs(ref).max(), i
    loss = float(g["nll_scale"][0]) * model.loss_function(lpx).sum() - 0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"][0])) <= ELBO_RTOL * abs(float(g["loss"][0]))
    # row T (HLVAE.py:455-475): deterministic encode, decode(mean_qz) -- the decoder entered at the latent (its own trunk launch)
    qs, qp, ps, pp, lpt, lpmt = model.get_test_samples(data, mask, None)
    assert max_abs_err(qp["z"][0].cpu(), g["test_mu"]) < 2e-2
    assert np.all(np.abs(lpt.detach().double().cpu().numpy() - g["test_log_p_x"]) <= 3e-2 + 2e-2 * np.abs(g["test_log_p_x"]))
    sd = dict(model.named_parameters())
    errs = {}
    for k in g.files:
        if k.startswith("grad__"):
            pn = k[len("grad__"):]
            if pn.startswith("hidden."):
                continue
            assert sd[pn].grad is not None, pn
            errs[pn] = rel_err(sd[pn].grad.double().cpu().numpy(), g[k])
    _report(name + "_grads", **errs)
    for k, e in errs.items():
        assert e < GRAD_RTOL, (k, e)
    assert len(errs) >= 15
    # fused training steps run and reduce the loss of a fixed batch
    tr = ELBOTrainer(model, P_total=12, kl="normal", max_batch=128, metrics=True)
    nll = []
    for _ in range(25):
        tr.step(data, mask, 4)
        nll.append(float(tr.scalars()["nll_sum"]))
    assert np.isfinite(nll).all() and nll[-1] < nll[0], nll
    # dims without hidden layers: the kernels' identity "layers" are not parameters and must come out of training untouched
    torch.cuda.synchronize()
    L = int(dims[2])
    eye = torch.eye(L, device=dev)
    if model._id_enc:
        wmu, wlv, bmu, blv = model._id_enc
        assert torch.equal(wmu, torch.cat([eye, torch.zeros_like(eye)], 1)) and torch.equal(wlv, torch.cat([torch.zeros_like(eye), eye], 1))
        assert float(bmu.abs().max()) == 0.0 and float(blv.abs().max()) == 0.0
    if model._id_dec:
        assert torch.equal(model._id_dec[0], eye) and float(model._id_dec[1].abs().max()) == 0.0
    moved = {k: not torch.equal(v.detach().cpu().double(), state[k].double()) for k, v in model.state_dict().items() if v.numel()}
    assert all(moved[k] for k in ("mean_layer.0.weight", "log_var_layer.0.weight", "mean_layer.0.bias", "y_layer.0.weight")), moved


@pytest.mark.parametrize("workload", ["d4", "tabular"])
def test_step_is_deterministic(workload):
    """Identical steps (learning rate 0, the same noise) must leave bit-identical dY (both layouts), log-likelihoods and dense
    gradients: nothing on that path uses atomics (except the split-K weight gradients of small models).  (Regression: one form of the real / pos head compiled into a kernel that
    occasionally lost one 16-lane group's store of one dY column -- 15 wrong cells in 3.3 M, invisible to every tolerance.)"""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    if workload == "d4":
        src, B = synthetic.make_d4(n_subjects=30, T=20, seed=11), 512
    else:
        src, B = synthetic.make_tabular(n_rows=1024, T=16, seed=5), 1024
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows = torch.tensor(np.arange(B).astype(np.int32), device=dev)
    eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(40)).to(dev)
    torch.manual_seed(5)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False).to(dev)
    tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=B, lr=0.0, metrics=True)
    d = model._dims
    snaps = []
    for _ in range(4):
        tr.step_rows(ds, rows, 26, eps=eps)
        torch.cuda.synchronize()
        t = model._ws_t
        snaps.append([t["dy"].clone(), t["dyT"].clone(), t["log_p_x"].clone(), t["xhat"].clone(),
                      t["G"][int(d.atomic_region):int(d.arena_size)].clone()])
    for a, b in zip(snaps[1:-1], snaps[2:]):
        for name, x, y in zip(("dy", "dyT", "log_p_x", "xhat", "dense gradients"), a, b):
            if name == "dense gradients" and workload == "tabular":
                # 64 features: few output tiles, the weight-gradient GEMMs slice the batch axis and add with fp32 atomics
                assert rel_err(x, y) < 1e-6, name
            else:
                assert torch.equal(x, y), name


@pytest.mark.gpu
def test_grouped_variable_order_matches_own_order():
    """The kernel-facing variable order (grouped by kind, HLVAE.kernel_var_order) is invisible from outside: per-variable
    outputs, the ELBO and every gradient agree with the same model run in the variables' own order, and the bf16 shadow of
    y_layer's weight holds the master's rows in the kernel's order."""
    import torch
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd import synthetic
    dev = torch.device("cuda:0")
    src = synthetic.make_tabular(n_rows=640, T=16, seed=5, expanded=True)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    data = torch.tensor(src.data[:512], device=dev)
    mask = torch.tensor(src.mask[:512], device=dev)
    eps = torch.randn(512, 32, generator=torch.Generator().manual_seed(40)).to(dev)
    outs = []
    for grouped in (False, True):
        torch.manual_seed(3)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, group_variables=grouped).to(dev)
        assert (model.kernel_var_order() is not None) == grouped
        _, mu, lv, lpx, lpm, p_params, _, _ = model(data, mask, None, src.types_info, eps=eps)
        nll = -lpx.sum()
        nll.backward()
        torch.cuda.synchronize()
        t = model._ws_t
        outs.append(dict(log_p_x=lpx.detach().clone(), xhat=t["xhat"].clone(), nll=float(nll),
                         pp=[q.detach().clone() for q in _flatten(p_params["x"])],
                         grads={n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        if grouped:
            rows = torch.as_tensor(model.kernel_wy_rows(), device=dev)
            wy = model.y_layer[0].weight.detach()[rows]
            assert torch.equal(t["wys"][:wy.shape[0], :wy.shape[1]].float(), wy.to(torch.bfloat16).float())
            assert torch.equal(t["wyTs"][:wy.shape[1], :wy.shape[0]].float(), wy.to(torch.bfloat16).float().t())
        model._release_device_state()
    a, b = outs
    # same bf16 operands, same fp32 accumulation order per output cell: the per-variable results are bitwise equal
    assert torch.equal(a["log_p_x"], b["log_p_x"])
    assert torch.equal(a["xhat"], b["xhat"])
    for x, y in zip(a["pp"], b["pp"]):
        assert torch.equal(x, y)
    assert set(a["grads"]) == set(b["grads"])
    worst = 0.0
    for n in a["grads"]:
        ga, gb = a["grads"][n], b["grads"][n]
        err = float((ga - gb).abs().max() / (ga.abs().max() + 1e-30))
        worst = max(worst, err)
        # y_layer and the heads see identical dY; below them dU = dY Wy sums its 320 columns in another order before the bf16
        # rounding of dU (measured 4e-4 worst)
        assert err < (1e-6 if n.startswith(("y_layer", "obs", "_log_vy", "_disp")) else 2e-3), (n, err)
    print("grouped vs own order: worst gradient difference", worst)


def _flatten(x):
    if isinstance(x, (list, tuple)):
        for y in x:
            yield from _flatten(y)
    else:
        yield x


@pytest.mark.gpu
def test_fused_optimiser_epilogue_matches_separate_launches():
    """hlvae_backward_adam applies the optimiser step in the epilogue of the weight-gradient GEMMs (csrc/dense.hip: k_gemm_adam; D4,
    512 rows).  Three ways through six identical steps must leave the same parameters, optimiser state and shadows:
    (a) eager steps, fused epilogue, shadows updated in place;  (b) eager steps with the separate gradient + k_adam_tiled launches
    (HL_NO_FUSED_ADAM);  (c) a captured chain of two pipelined steps replayed twice, y_layer's shadows double-buffered."""
    import os
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    from hlvae_amd import _lib
    dev = _dev()
    src = synthetic.make_d4(n_subjects=30, T=20, seed=11)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    R = [torch.tensor(np.arange(i * 40, i * 40 + 512).astype(np.int32), device=dev) for i in range(2)]

    def run(mode):
        torch.manual_seed(5)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
        tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=512, metrics=True)
        model._ensure_device_state(512)
        assert _lib.load().hlvae_backward_adam_fused(model._plan_handle, 512) == (0 if mode == "separate" else 1)
        if mode == "graph":
            tr.capture_rows("c", ds, R, [26, 26], next_rows=[R[1], R[0]])       # (runs two warm-up steps on R[0] first)
            tr.prime_rows(ds, R[0])
            for _ in range(2):
                tr.replay("c")
        else:
            for r in (R[0], R[0], R[0], R[1], R[0], R[1]):
                tr.step_rows(ds, r, 26)
        torch.cuda.synchronize()
        assert int(tr.opt.step_count[0]) == 6 and int(tr.opt.step_count[1]) == 0
        t = model._ws_t
        rows = torch.as_tensor(model.kernel_wy_rows(), device=dev)
        wy = model.y_layer[0].weight.detach()[rows].to(torch.bfloat16)
        assert torch.equal(t["wys"][:wy.shape[0], :wy.shape[1]], wy), mode                 # shadows = bf16 of the masters,
        assert torch.equal(t["wyTs"][:wy.shape[1], :wy.shape[0]], wy.t()), mode             # in the kernel's row order
        w1 = model.VAE_encoder_common_layers[0].weight.detach().to(torch.bfloat16)
        assert torch.equal(t["w1s"][:w1.shape[0], :w1.shape[1]], w1), mode
        wm = model.mean_layer[0].weight.detach().to(torch.bfloat16)
        assert torch.equal(t["wmls"][:wm.shape[0], :wm.shape[1]], wm) and torch.equal(t["wmlTs"][:wm.shape[1], :wm.shape[0]], wm.t()), mode
        out = (model._arena.clone(), tr.opt.m1.clone(), tr.opt.m2.clone(), float(tr.scalars()["nll_sum"]))
        model._release_device_state()
        return out

    a = run("fused")
    os.environ["HL_NO_FUSED_ADAM"] = "1"
    try:
        b = run("separate")
    finally:
        del os.environ["HL_NO_FUSED_ADAM"]
    c = run("graph")
    for name, other in (("separate launches", b), ("captured chain", c)):
        assert abs(a[3] - other[3]) <= 1e-6 * abs(a[3]), name
        # same arithmetic per element; the first moments follow the gradients linearly and so carry the fp32 reordering of the
        # head gradients' atomics and of the two GEMM tile shapes (measured 9e-6), the parameters move by ~lr per step either way
        for x, y, what, tol in zip(a[:3], other[:3], ("parameters", "m", "v"), (1e-6, 5e-5, 5e-5)):
            assert rel_err(x, y) < tol, (name, what, rel_err(x, y))


@pytest.mark.gpu
@pytest.mark.parametrize("hid", [[500], []], ids=["hidden500", "nohid"])
def test_gp_captured_chain_matches_eager_steps(hid):
    """GP prior + fused optimiser inside a captured chain of two pipelined steps (y_layer's shadows double-buffered, the prior's
    chains on streams of their own) against the same six steps launched eagerly: VAE parameters, GP hyper-parameters, inducing
    points, m and H.  Also for a model without hidden layers (the general optimiser path inside the captured chain)."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset, subject_index
    from hlvae_amd.elbo_functions import GPPriorHIP
    dev = _dev()
    src = synthetic.make_d4(n_subjects=40, T=20, seed=21)
    dims = [src.cov_dim_ext, hid, 32, hid, 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    win = [np.arange(0, 256), np.arange(300, 556)]
    R = [torch.tensor(w.astype(np.int32), device=dev) for w in win]
    G = [torch.tensor(subject_index(src.labels[w, src.id_covariate]), device=dev) for w in win]
    PB = [int(np.unique(src.labels[w, src.id_covariate]).size) for w in win]

    def run(graph):
        torch.manual_seed(7)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=256, materialize_samples=False).to(dev)
        gp = GPPriorHIP.from_reference_config(model, src, 40, dev)
        tr = ELBOTrainer(model, P_total=40, kl="gp", gp=gp, max_batch=256, metrics=True)
        if graph:
            tr.capture_rows("c", ds, R, PB, next_rows=[R[1], R[0]], groups=G)       # (two warm-up steps on R[0] first)
            tr.prime_rows(ds, R[0])
            for _ in range(2):
                tr.replay("c")
        else:
            for i in (0, 0, 0, 1, 0, 1):
                tr.step_rows(ds, R[i], PB[i], groups=G[i])
        torch.cuda.synchronize()
        gp.check()
        out = dict(arena=model._arena.clone(), theta=gp._theta.clone(), m=gp.m.clone(), H=gp.H.clone(), kld=float(gp.last_kld))
        model._release_device_state()
        return out

    a, b = run(False), run(True)
    errs = {k: rel_err(a[k], b[k]) for k in ("arena", "theta", "m", "H")}
    errs["kld"] = abs(a["kld"] - b["kld"]) / abs(a["kld"])
    _report("gp_chain_vs_eager" + ("" if hid else "_nohid"), **errs)
    # fp32 / fp64 atomics (head gradients, the prior's per-subject sums) reorder between runs, and Adam turns a sign flip of a tiny
    # gradient into a +-lr step: measured kld 7e-6, parameters 2.7e-4 of the largest one, m 3e-6, H 5e-8, hyper-parameters 1e-9
    assert errs["kld"] < 1e-4 and errs["arena"] < 2e-3 and errs["theta"] < 1e-6 and errs["m"] < 1e-4 and errs["H"] < 1e-5, errs


@pytest.mark.gpu
def test_narrow_model_paths_match_the_general_ones():
    """64-feature model (X = 104, D y = 320), 1024 rows: the narrow-model forms of round 2 -- first encoder Linear and dY Wy inside
    the fused middle kernels, y_layer's gradient in the grouped launch, one optimiser launch, gradients cleared by the middle
    kernel -- against the general launch sequence (HL_NO_MID_DIRECT: separate split-K GEMMs): forward outputs of one step and the
    parameters after four steps."""
    import os
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    src = synthetic.make_tabular(n_rows=2048, T=16, seed=9)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    R = [torch.tensor(np.arange(i * 512, i * 512 + 1024).astype(np.int32), device=dev) for i in range(2)]
    eps = [torch.randn(1024, 32, generator=torch.Generator().manual_seed(70 + i)).to(dev) for i in range(4)]

    def run():
        torch.manual_seed(5)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=1024, materialize_samples=False).to(dev)
        tr = ELBOTrainer(model, P_total=128, kl="normal", max_batch=1024, metrics=True)
        tr.step_rows(ds, R[0], 64, eps=eps[0])
        torch.cuda.synchronize()
        t = model._ws_t
        first = dict(mu=t["mu"][:1024].clone(), lpx=t["log_p_x"][:1024].clone(), nll=float(tr.scalars()["nll_sum"]))
        for i in range(1, 4):
            tr.step_rows(ds, R[i % 2], 64, eps=eps[i])
        torch.cuda.synchronize()
        out = (first, model._arena.clone(), tr.opt.m1.clone(), tr.err.clone())
        model._release_device_state()
        return out

    a = run()
    os.environ["HL_NO_MID_DIRECT"] = "1"
    try:
        b = run()
    finally:
        del os.environ["HL_NO_MID_DIRECT"]
    # the same products with another split of the K axis (fp32 slabs summed vs one MFMA chain): last-bit differences before the bf16
    # rounding of T and dU, which a few of their elements turn into one bf16 ulp
    assert rel_err(a[0]["mu"], b[0]["mu"]) < 2e-3
    assert rel_err(a[0]["lpx"], b[0]["lpx"]) < 2e-3
    assert abs(a[0]["nll"] - b[0]["nll"]) <= 1e-5 * abs(b[0]["nll"])
    assert rel_err(a[1], b[1]) < 2e-3, rel_err(a[1], b[1])         # parameters after four steps (Adam: +-lr per sign flip of a tiny gradient)
    assert torch.isfinite(a[3]).all()


@pytest.mark.gpu
def test_workspace_grows_at_the_top_of_the_step_not_inside_it():
    """The sampler folds a short tail into the batch before it, so a prefetched batch larger than max_batch is a normal event.
    The workspace is sized for it at the TOP of the step that prefetches it (growing it between the forward and the backward pass
    handed the backward pass freshly zeroed buffers); the noise stream continues across the re-allocation.  Same trajectory as a
    trainer whose workspace was large enough from the start."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.datafeed import CompactDataset
    from hlvae_amd.training import ELBOTrainer
    from tests_common import MIX_SPEC
    dev = _dev()
    src = synthetic.make_tabular(n_rows=320, T=8, seed=19, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    dsd = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    small = torch.arange(0, 96, dtype=torch.int32, device=dev)
    large = torch.arange(96, 296, dtype=torch.int32, device=dev)          # 200 rows: beyond a 128-row workspace
    eps = [torch.randn(n, dims[2], generator=torch.Generator().manual_seed(5 + i)).to(dev) for i, n in enumerate((96, 200, 96))]

    def run(max_batch):
        torch.manual_seed(4)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=max_batch, materialize_samples=False).to(dev)
        tr = ELBOTrainer(model, P_total=40, kl="normal", max_batch=max_batch)
        with torch.no_grad():
            model._ws_t["rng"][0] = 12345                       # same Philox seed in both runs
        out = []
        tr.prime_rows(dsd, small)
        tr.step_rows(dsd, small, 12, eps=eps[0], prefetch_rows=large, prepacked=True)      # grows here (max_batch 128), before the forward
        out.append(float(tr.scalars()["nll_sum"]))
        tr.step_rows(dsd, large, 25, eps=eps[1], prefetch_rows=small, prepacked=True)
        out.append(float(tr.scalars()["nll_sum"]))
        tr.step_rows(dsd, small, 12, eps=eps[2], prepacked=True)
        out.append(float(tr.scalars()["nll_sum"]))
        torch.cuda.synchronize()
        return out, model._arena.detach().clone(), int(model._ws.Bp_max), int(model._ws_t["rng"][0].item())

    a, b = run(128), run(256)
    assert a[2] >= 256 and b[2] == 256 and a[3] == b[3] == 12345
    for x, y in zip(a[0], b[0]):
        assert abs(x - y) <= 1e-6 * abs(y), (a[0], b[0])
    assert rel_err(a[1], b[1]) < 1e-6
    # a batch that was NOT announced at the top of the step cannot be squeezed in afterwards
    torch.manual_seed(4)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=128, materialize_samples=False).to(dev)
    model._ensure_device_state(96)
    with pytest.raises(RuntimeError, match="does not fit the workspace"):
        model._require_capacity(200)


@pytest.mark.gpu
def test_head_kernel_cores_are_bit_identical():
    """The head kernel's GEMM phase exists in two LDS-DMA forms (csrc/gemm_dma.h): [A; B] through LDS (HL_HEADS_CORE=1) and, the default
    for K a multiple of 256, A fragments loaded straight to registers by inline assembly with hand-counted s_waitcnt.  A stage
    register copied by the compiler before its wait would hand stale data on -- so the two forms must leave bit-identical dY (both
    layouts), log-likelihoods and x_hat (tools/heads_ab.py, one process per form: the choice is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for core in ("1", "2"):
        env = dict(os.environ, HL_HEADS_CORE=core)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "heads_ab.py"), "512"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l.split(": ", 1)[1] for l in r.stdout.splitlines() if l.startswith("core=")]
        assert len(lines) == 3, r.stdout[-2000:]
        outs.append(lines)
    assert outs[0] == outs[1], (outs[0][-1], outs[1][-1])
