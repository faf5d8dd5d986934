"""The CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Tolerance: 1e-12 relative -- both are float64 evaluations of the
same formulas; only summation order differs."""
import os

import numpy as np
import pytest
import torch

import hlvae_oracle as orc
import gp_oracle as gpo
import metrics_oracle as mo
from hlvae_amd import synthetic
from tests_common import MIX_SPEC, load_mix_case, rel_err

T64 = torch.float64
TOL = 1e-11


@pytest.mark.parametrize("name", ["mix_init", "mix_trained"])
def test_forward_backward_mix(golden_dir, name):
    g, src, dims, state = load_mix_case(golden_dir, name)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    # the reference registers the decoder trunk twice (hidden / d_layers share tensors, HLVAE.py:232-242)
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
    model = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    data, mask = torch.tensor(g["data"]), torch.tensor(g["mask"])
    out = model.forward(data, mask, torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "z", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    for i, p in enumerate(out["p_params"]):
        assert rel_err(p.detach().numpy().reshape(g[f"p_params_{i}"].shape), g[f"p_params_{i}"]) < TOL
    nll = model.loss_function(out["log_p_x"])
    loss = float(g["nll_scale"][0]) * nll.sum() + orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    checked = 0
    for k in g.files:
        if k.startswith("grad__"):
            name_ = k[len("grad__"):]
            assert rel_err(st[name_].grad.numpy(), g[k]) < 1e-9, name_
            checked += 1
    assert checked >= 15
    # row T
    ts = model.test_samples(data, mask)
    assert rel_err(ts["mu"].numpy(), g["test_mu"]) < TOL
    assert rel_err(ts["log_p_x"].numpy(), g["test_log_p_x"]) < TOL
    assert rel_err(ts["log_p_x_missing"].numpy(), g["test_log_p_x_missing"]) < TOL
    # row M
    xh, e_obs, e_mis, e_all = mo.step_metrics([p.detach() for p in out["p_params"]], data, mask, src.types_info,
                                              st["_log_vy_pos"].detach())
    assert rel_err(xh.numpy(), g["x_hat_mean"]) < TOL
    assert rel_err(e_obs.numpy(), g["err_observed"]) < TOL
    assert rel_err(e_mis.numpy(), g["err_missing"]) < TOL


def test_d4_small(golden_dir):
    g = np.load(os.path.join(golden_dir, "d4_small.npz"))
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    d = src.data[:8]
    assert np.allclose([d.sum(), (d * np.arange(d.shape[1])).sum()], g["data_argsum"]), "generator drifted"
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=3, std=0.05)
    chk = float(sum(v.double().abs().sum() for v in state.values()))
    assert abs(chk - g["state_checksum"][0]) < 1e-9 * chk, "weight generator drifted"
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
    model = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    out = model.forward(torch.tensor(d), torch.tensor(g["mask"]), torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    loss = float(g["nll_scale"][0]) * model.loss_function(out["log_p_x"]).sum() + \
        orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    for k in g.files:
        if k.startswith("grad__"):
            assert rel_err(st[k[6:]].grad.numpy(), g[k]) < 1e-9, k
    assert rel_err(st["y_layer.0.weight"].grad[:40].numpy(), g["grad_slice__y_layer.0.weight"]) < 1e-9
    assert rel_err(st["VAE_encoder_common_layers.0.weight"].grad[:, :64].numpy(),
                   g["grad_slice__VAE_encoder_common_layers.0.weight"]) < 1e-9


def test_d4_conv_logvar_small(golden_dir):
    """conv=True with logvar_network=True (round 3): the oracle against the reference run -- sigmoid on the mean half of a real
    variable's head only, per-entry variance from the head's second output"""
    from hlvae_amd import layout
    g = np.load(os.path.join(golden_dir, "d4_conv_logvar_small.npz"))
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    d = src.data[:8]
    assert np.allclose([d.sum(), (d * np.arange(d.shape[1])).sum()], g["data_argsum"]), "generator drifted"
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    info = layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True)
    assert np.array_equal(np.asarray(info["param_indexes"]), g["param_indexes"])
    state = orc.init_state(dims, info, src.n_variables, seed=19, std=0.05, conv=True, logvar_network=True)
    chk = float(sum(v.double().abs().sum() for v in state.values()))
    assert abs(chk - g["state_checksum"][0]) < 1e-9 * chk, "weight generator drifted"
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
        if k.startswith("Decoder_Conv_layer."):
            st[k] = st["deconv_layer." + k[len("Decoder_Conv_layer."):]]
    model = orc.OracleHLVAE(dims, info, src.n_variables, st, conv=True)
    out = model.forward(torch.tensor(d), torch.tensor(g["mask"]), torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    loss = float(g["nll_scale"][0]) * model.loss_function(out["log_p_x"]).sum() + \
        orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    n = 0
    for k in g.files:
        if k.startswith("grad__"):
            assert rel_err(st[k[6:]].grad.numpy(), g[k]) < 1e-9, k
            n += 1
    assert n >= 20
    assert rel_err(st["y_layer.0.weight"].grad[:40].numpy(), g["grad_slice__y_layer.0.weight"]) < 1e-9


def test_d4_conv_deep_small(golden_dir):
    """conv=True with two hidden layers per side (round 3): the oracle against the reference run"""
    g = np.load(os.path.join(golden_dir, "d4_conv_deep_small.npz"))
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    d = src.data[:8]
    assert np.allclose([d.sum(), (d * np.arange(d.shape[1])).sum()], g["data_argsum"]), "generator drifted"
    dims = [src.cov_dim_ext, [int(v) for v in g["hid_e"]], 8, [int(v) for v in g["hid_d"]], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=23, std=0.05, conv=True)
    chk = float(sum(v.double().abs().sum() for v in state.values()))
    assert abs(chk - g["state_checksum"][0]) < 1e-9 * chk, "weight generator drifted"
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
        if k.startswith("Decoder_Conv_layer."):
            st[k] = st["deconv_layer." + k[len("Decoder_Conv_layer."):]]
    model = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st, conv=True)
    out = model.forward(torch.tensor(d), torch.tensor(g["mask"]), torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    loss = float(g["nll_scale"][0]) * model.loss_function(out["log_p_x"]).sum() + \
        orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    n = 0
    for k in g.files:
        if k.startswith("grad__"):
            assert rel_err(st[k[6:]].grad.numpy(), g[k]) < 1e-9, k
            n += 1
    assert n >= 24
    assert rel_err(st["y_layer.0.weight"].grad[:40].numpy(), g["grad_slice__y_layer.0.weight"]) < 1e-9
    assert rel_err(st["VAE_encoder_common_layers.0.weight"].grad[:, :64].numpy(),
                   g["grad_slice__VAE_encoder_common_layers.0.weight"]) < 1e-9


def test_d4_conv_small(golden_dir):
    """convolutional front/back end: the oracle against the reference run with conv=True"""
    g = np.load(os.path.join(golden_dir, "d4_conv_small.npz"))
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    d = src.data[:8]
    assert np.allclose([d.sum(), (d * np.arange(d.shape[1])).sum()], g["data_argsum"]), "generator drifted"
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=13, std=0.05, conv=True)
    chk = float(sum(v.double().abs().sum() for v in state.values()))
    assert abs(chk - g["state_checksum"][0]) < 1e-9 * chk, "weight generator drifted"
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
        if k.startswith("Decoder_Conv_layer."):
            st[k] = st["deconv_layer." + k[len("Decoder_Conv_layer."):]]
    model = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st, conv=True)
    out = model.forward(torch.tensor(d), torch.tensor(g["mask"]), torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    loss = float(g["nll_scale"][0]) * model.loss_function(out["log_p_x"]).sum() + \
        orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    n = 0
    for k in g.files:
        if k.startswith("grad__"):
            assert rel_err(st[k[6:]].grad.numpy(), g[k]) < 1e-9, k
            n += 1
    assert n >= 20
    assert rel_err(st["y_layer.0.weight"].grad[:40].numpy(), g["grad_slice__y_layer.0.weight"]) < 1e-9
    assert rel_err(st["VAE_encoder_common_layers.0.weight"].grad[:, :64].numpy(),
                   g["grad_slice__VAE_encoder_common_layers.0.weight"]) < 1e-9
    ts = model.test_samples(torch.tensor(d), torch.tensor(g["mask"]))
    assert rel_err(ts["mu"].numpy(), g["test_mu"]) < TOL and rel_err(ts["log_p_x"].numpy(), g["test_log_p_x"]) < TOL
    # row M under conv (read_functions.py:366-369)
    xh, e_obs, e_mis, e_all = mo.step_metrics([p.detach() for p in out["p_params"]], torch.tensor(d), torch.tensor(g["mask"]),
                                              src.types_info, st["_log_vy_pos"].detach(), conv=True)
    assert rel_err(xh.numpy(), g["x_hat_mean"]) < TOL
    assert rel_err(e_obs.numpy(), g["err_observed"]) < 1e-7 and rel_err(e_mis.numpy(), g["err_missing"]) < 1e-7


def test_gp_kl(golden_dir):
    g = np.load(os.path.join(golden_dir, "gp_kl.npz"))
    P, P_b, N, eps, idc, lr = g["scalars"]
    L = g["mu"].shape[1]
    spec = gpo.spec_from_config([2], [], [0], [{"cont_covariate": 0, "cat_covariate": 2},
                                               {"cont_covariate": 0, "cat_covariate": 3},
                                               {"cont_covariate": 1, "cat_covariate": 4}], [], int(idc))
    prm = {k[4:]: torch.tensor(g[k]).requires_grad_(True) for k in g.files if k.startswith("kp__")}
    assert set(prm) == set(gpo.init_kernel_params(spec, L))
    mu = torch.tensor(g["mu"]).requires_grad_(True)
    lv = torch.tensor(g["log_v"]).requires_grad_(True)
    z = torch.tensor(g["z"]).requires_grad_(True)
    m, H = torch.tensor(g["m"]), torch.tensor(g["H"])
    kld, gm, gH = gpo.minibatch_kld_upper_bound_iter(spec, prm, torch.tensor(g["noise"]), L, m, H, torch.tensor(g["x"]),
                                                     mu, lv, z, P, P_b, N, True, int(idc), float(eps))
    assert rel_err(kld.detach().numpy(), g["kld"]) < TOL
    assert rel_err(gm.detach().numpy(), g["grad_m"]) < 1e-9
    assert rel_err(gH.detach().numpy(), g["grad_H"]) < 1e-9
    kld.sum().backward()
    assert rel_err(mu.grad.numpy(), g["d_mu"]) < 1e-9
    assert rel_err(lv.grad.numpy(), g["d_log_v"]) < 1e-9
    assert rel_err(z.grad.numpy(), g["d_z"]) < 1e-8
    for k, p in prm.items():
        assert rel_err(p.grad.numpy(), g["kg__" + k]) < 1e-8, k
    m_new, H_new = gpo.natural_gradient_update(m, H, gm.detach(), gH.detach(), float(lr))
    assert rel_err(m_new.numpy(), g["m_new"]) < 1e-9
    assert rel_err(H_new.numpy(), g["H_new"]) < 1e-9


def test_gp_predict(golden_dir):
    """GP posterior prediction (reference utils.py:99-191, run with torch.solve mapped onto torch.linalg.solve)"""
    g = np.load(os.path.join(golden_dir, "gp_predict.npz"))
    eps, idc = float(g["scalars"][0]), int(g["scalars"][1])
    L = g["mu"].shape[1]
    spec = gpo.spec_from_config([2], [], [0], [{"cont_covariate": 0, "cat_covariate": 2},
                                               {"cont_covariate": 0, "cat_covariate": 3},
                                               {"cont_covariate": 1, "cat_covariate": 4}], [], idc)
    prm = {k[4:]: torch.tensor(g[k]) for k in g.files if k.startswith("kp__")}
    Zp = gpo.batch_predict_varying_T(spec, prm, torch.tensor(g["noise"]), L, torch.tensor(g["x"]), torch.tensor(g["test_x"]),
                                     torch.tensor(g["mu"]), torch.tensor(g["z"]), idc, eps)
    assert rel_err(Zp.numpy(), g["Z_pred"]) < 1e-9


@pytest.mark.parametrize("name", ["mix_logvar", "mix_deep", "mix_logvar_deep", "mix_nohid_e", "mix_nohid_d", "mix_nohid"])
def test_constructor_modes(golden_dir, name):
    """logvar_network=True (HLVAE.py:25-51; loglik.py:45-47, 105) and two hidden layers per side (HLVAE.py:113, 125-137, 232-242):
    forward, every gradient, get_test_samples and the per-step metrics against the reference's own outputs."""
    from tests_common import load_mode_case
    g, src, dims, state, info, lvn = load_mode_case(golden_dir, name)
    st = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    for k in list(st):
        if k.startswith("hidden."):
            st[k] = st["d_layers." + k[len("hidden."):]]
    model = orc.OracleHLVAE(dims, info, src.n_variables, st)
    data, mask = torch.tensor(g["data"]), torch.tensor(g["mask"])
    out = model.forward(data, mask, torch.tensor(g["eps"]))
    for k in ("mu", "log_var", "z", "log_p_x", "log_p_x_missing"):
        assert rel_err(out[k].detach().numpy(), g[k]) < TOL, k
    for i, p in enumerate(out["p_params"]):
        p = torch.cat(p, 1) if isinstance(p, list) else p
        assert rel_err(p.detach().numpy().reshape(g[f"p_params_{i}"].shape), g[f"p_params_{i}"]) < TOL, i
    loss = float(g["nll_scale"][0]) * model.loss_function(out["log_p_x"]).sum() + orc.standard_normal_kl(out["mu"], out["log_var"])
    assert rel_err(loss.detach().numpy(), g["loss"][0]) < TOL
    loss.backward()
    checked = 0
    for k in g.files:
        if k.startswith("grad__"):
            name_ = k[len("grad__"):]
            assert rel_err(st[name_].grad.numpy(), g[k]) < 1e-9, name_
            checked += 1
    assert checked >= (17 if lvn else 15) + (4 if "deep" in name else 0)
    ts = model.test_samples(data, mask)
    assert rel_err(ts["mu"].numpy(), g["test_mu"]) < TOL and rel_err(ts["log_p_x"].numpy(), g["test_log_p_x"]) < TOL
    det = [([q.detach() for q in p] if isinstance(p, list) else p.detach()) for p in out["p_params"]]
    xh, e_obs, e_mis, e_all = mo.step_metrics(det, data, mask, info, None if lvn else st["_log_vy_pos"].detach())
    assert rel_err(xh.numpy(), g["x_hat_mean"]) < TOL
    assert rel_err(e_obs.numpy(), g["err_observed"]) < TOL and rel_err(e_mis.numpy(), g["err_missing"]) < TOL
