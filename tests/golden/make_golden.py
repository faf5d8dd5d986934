#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

For every case the inputs come from this repo's deterministic generators, the weights from
oracle/hlvae_oracle.init_state (loaded into the reference model through load_state_dict), and the
expected outputs from the reference's own code:

  HLVAE.forward / loss_function / backward        (reference HLVAE.py:364-379)
  HLVAE.get_test_samples                          (HLVAE.py:455-475)
  read_functions.read_data (types_info layout)    (HL_VAE/read_functions.py:13-203)
  read_functions metrics                          (HL_VAE/read_functions.py:206-412)
  elbo_functions.minibatch_KLD_upper_bound_iter   (elbo_functions.py:196-285), GP_model.py kernels
  natural-gradient update                         (training.py:130-137, cholesky -> linalg.cholesky)

Only DATA is stored (npz arrays): inputs and expected outputs.
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, REF)

import hlvae_amd                                    # noqa: E402  (repo package: generators + layout)
from hlvae_amd import layout, synthetic             # noqa: E402
import hlvae_oracle as orc                          # noqa: E402
import gp_oracle as gpo                             # noqa: E402

# torch >= 2 removed torch.cholesky; the reference calls it (elbo_functions.py:225-251).
if not hasattr(torch, "cholesky") or True:
    torch.cholesky = torch.linalg.cholesky


def load_ref(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


ref_hlvae = load_ref("ref_HLVAE", "HLVAE.py")
ref_rf = load_ref("ref_read_functions", "HL_VAE/read_functions.py")
ref_elbo = load_ref("ref_elbo_functions", "elbo_functions.py")
ref_gp = load_ref("ref_GP_model", "GP_model.py")
# torch >= 2 removed torch.solve(B, A) -> (X, LU); the reference calls it (utils.py:162,169)
torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)      # torch 2.10 keeps a stub that only raises
ref_utils = load_ref("ref_utils", "utils.py")

T64 = torch.float64


def np64(t):
    return t.detach().cpu().numpy().astype(np.float64)


MIX_SPEC = [("real", 1), ("cat", 3), ("pos", 1), ("ordinal", 4), ("count", 1), ("cat", 5), ("real", 1),
            ("ordinal", 5), ("pos", 1), ("cat", 5), ("count", 1), ("cat", 3), ("real", 1), ("ordinal", 4)]


def run_reference_model(src, rows, dims, state, seed, nll_scale, conv=False, logvar_network=False):
    """reference forward + loss + backward; returns dict of arrays."""
    info = dict(src.types_info, conv=conv)
    if logvar_network:      # parameter layout with (mean, log-variance) slots for real / pos (read_functions.py:162-183)
        info = dict(layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True), conv=conv)
        for t in info["types_dict"]:
            t["dim"], t["nclass"] = int(t["dim"]), int(t["nclass"])
    model = ref_hlvae.HLVAE(dims, info, src.n_variables, vy_init=[1.0, 0.5], logvar_network=logvar_network,
                            conv=conv).to(T64)
    missing = model.load_state_dict(state, strict=True)
    model = model.double()
    data = torch.tensor(src.data[rows], dtype=T64)
    mask = torch.tensor(src.mask[rows], dtype=T64)
    pmask = torch.tensor((info["param_miss_mask"] if logvar_network else src.param_mask)[rows], dtype=T64)
    B = data.shape[0]
    torch.manual_seed(seed)
    eps = torch.randn(B, dims[2], dtype=T64)
    torch.manual_seed(seed)
    p_samples, mu, lv, log_p_x, log_p_x_missing, p_params, q_samples, q_params = model(data, mask, pmask, info)
    assert torch.equal(q_samples["z"], mu + eps * torch.exp(0.5 * lv)), "eps recovery failed"
    nll = model.loss_function(log_p_x)
    kl = -0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss = nll_scale * torch.sum(nll) + kl
    model.zero_grad()
    loss.backward()
    out = dict(data=src.data[rows], mask=src.mask[rows], eps=np64(eps), mu=np64(mu), log_var=np64(lv),
               z=np64(q_samples["z"]), log_p_x=np64(log_p_x), log_p_x_missing=np64(log_p_x_missing),
               nll=np64(nll), loss=np64(loss.reshape(1)), nll_scale=np.array([nll_scale]))
    for i, p in enumerate(p_params["x"]):
        out[f"p_params_{i}"] = np64(p if not isinstance(p, list) else torch.cat(p, 1))
    log_vy = [getattr(model, "_log_vy_real", None), getattr(model, "_log_vy_pos", None)]
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    # metrics (training.py:84-95)
    full = ref_rf.p_params_concatenation_by_key([p_params], info, B, data.device, "x")
    dtr = ref_rf.discrete_variables_transformation(data, info)
    xh, xmode = ref_rf.statistics(full, info, data.device, False, log_vy)
    e_obs, e_mis, _ = ref_rf.error_computation(dtr, xh, info, mask, dim=0)
    out.update(p_params_full=np64(full), x_transformed=np64(dtr), x_hat_mean=np64(xh), x_hat_mode=np64(xmode),
               err_observed=np64(e_obs), err_missing=np64(e_mis))
    # get_test_samples (HLVAE.py:455-475)
    qs, qp, ps, pp, lpt, lpmt = model.get_test_samples(data, mask, pmask)
    out.update(test_mu=np64(qp["z"][0]), test_log_p_x=np64(lpt), test_log_p_x_missing=np64(lpmt))
    for i, p in enumerate(pp["x"]):
        out[f"test_p_params_{i}"] = np64(p if not isinstance(p, list) else torch.cat(p, 1))
    return out, grads


def state_checksum(state):
    return np.array([float(sum((v.double().abs().sum() for v in state.values())))])


def case_mix(name, seed_state, std, nll_scale):
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=seed_state, std=std)
    if std > 0.1:   # "trained-like": spread the free variances and the ordinal thresholds too
        g = torch.Generator().manual_seed(99)
        state["_log_vy_real"] = state["_log_vy_real"] + torch.randn(state["_log_vy_real"].shape, generator=g, dtype=T64)
        state["_log_vy_pos"] = state["_log_vy_pos"] + torch.randn(state["_log_vy_pos"].shape, generator=g, dtype=T64)
        for k in state:
            if k.endswith("weight_thresholds"):
                state[k] = state[k] + 0.5 * torch.randn(state[k].shape, generator=g, dtype=T64)
    out, grads = run_reference_model(src, np.arange(24), dims, state, seed=11, nll_scale=nll_scale)
    for k, g in grads.items():
        out["grad__" + k] = np64(g)
    for k, v in state.items():
        out["state__" + k] = np64(v)
    out["dims"] = np.array([dims[0], dims[1][0], dims[2], dims[3][0], dims[4]])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", out["loss"])


def case_mix_logvar_deep():
    """logvar_network=True (HLVAE.py:25-51, loglik.py:45-47, 105: the variance of every real / pos ENTRY comes from a second head)
    together with TWO hidden layers per side (HLVAE.py:125-137, 232-242; h_dim_d is reversed, :113): the two modes of the
    reference's constructor that the shipped configuration does not use."""
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    for name, hid_e, hid_d, lvn in (("mix_logvar", [16], [16], True), ("mix_deep", [24, 16], [12, 20], False),
                                    ("mix_logvar_deep", [24, 16], [12, 20], True)):
        dims = [src.cov_dim_ext, hid_e, 4, hid_d, 5]
        info = src.types_info
        if lvn:
            info = layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True)
        state = orc.init_state(dims, info, src.n_variables, seed=17, std=0.2, logvar_network=lvn)
        out, grads = run_reference_model(src, np.arange(24), dims, state, seed=13, nll_scale=3.0, logvar_network=lvn)
        for k, g in grads.items():
            out["grad__" + k] = np64(g)
        for k, v in state.items():
            out["state__" + k] = np64(v)
        out["hid_e"], out["hid_d"], out["logvar_network"] = np.array(hid_e), np.array(hid_d), np.array([int(lvn)])
        out["param_indexes"] = np.asarray(info["param_indexes"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", out["loss"])


def case_mix_no_hidden():
    """dims without hidden layers (HLVAE.py:128, 233: h_dim = [] -- mean / log-var heads on the input, y_layer on the latent): one side,
    the other, both.  Same data, seeds and file format as case_mix_logvar_deep."""
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    for name, hid_e, hid_d in (("mix_nohid_e", [], [16]), ("mix_nohid_d", [16], []), ("mix_nohid", [], [])):
        dims = [src.cov_dim_ext, hid_e, 4, hid_d, 5]
        info = src.types_info
        state = orc.init_state(dims, info, src.n_variables, seed=17, std=0.2)
        out, grads = run_reference_model(src, np.arange(24), dims, state, seed=13, nll_scale=3.0)
        for k, g in grads.items():
            out["grad__" + k] = np64(g)
        for k, v in state.items():
            out["state__" + k] = np64(v)
        out["hid_e"], out["hid_d"], out["logvar_network"] = np.array(hid_e, dtype=np.int64), np.array(hid_d, dtype=np.int64), np.array([0])
        out["param_indexes"] = np.asarray(info["param_indexes"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss", out["loss"])


def case_d4():
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=3, std=0.05)
    out, grads = run_reference_model(src, np.arange(8), dims, state, seed=21, nll_scale=2.5)
    # data/mask are regenerated by recipe in the test; store the compact raw form to pin the generator
    out["data_argsum"] = np.array([out["data"].sum(), (out["data"] * np.arange(out["data"].shape[1])).sum()])
    del out["data"]
    out["state_checksum"] = state_checksum(state)
    for k in ("mean_layer.0.weight", "mean_layer.0.bias", "log_var_layer.0.bias", "d_layers.0.bias", "_log_vy_real",
              "obs_layer.0.bias", "obs_layer.1.bias_mean", "obs_layer.1.weight_mean", "VAE_encoder_common_layers.0.bias"):
        out["grad__" + k] = np64(grads[k])
    out["grad_slice__y_layer.0.weight"] = np64(grads["y_layer.0.weight"][:40])
    out["grad_slice__VAE_encoder_common_layers.0.weight"] = np64(grads["VAE_encoder_common_layers.0.weight"][:, :64])
    out["grad_slice__obs_layer.0.weight"] = np64(grads["obs_layer.0.weight"][:50])
    out["grad_norms"] = np.array([float(grads[k].norm()) for k in sorted(grads)])
    out["grad_names"] = np.array(sorted(grads))
    for k in list(out):
        if k.startswith("p_params") or k.startswith("test_p_params") or k in ("p_params_full",):
            out[k] = out[k][:, :200] if out[k].ndim == 2 else out[k][:, :40]
    out["dims"] = np.array([dims[0], 32, 8, 32, 5])
    np.savez_compressed(os.path.join(HERE, "d4_small.npz"), **out)
    print("d4_small loss", out["loss"])


def case_d4_conv():
    """the convolutional front/back end the shipped configuration selects (config/hlvae_config_file.txt:51; HLVAE.py:139-152,
    253-259, 293-308, 338-341, 428-430; HL_VAE/utils.py:99-102; loglik.py:36-41)"""
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=13, std=0.05, conv=True)
    out, grads = run_reference_model(src, np.arange(8), dims, state, seed=23, nll_scale=2.5, conv=True)
    out["data_argsum"] = np.array([out["data"].sum(), (out["data"] * np.arange(out["data"].shape[1])).sum()])
    del out["data"]
    out["state_checksum"] = state_checksum(state)
    small = ("mean_layer.0.weight", "mean_layer.0.bias", "log_var_layer.0.bias", "d_layers.0.bias", "_log_vy_real",
             "obs_layer.0.bias", "obs_layer.1.bias_mean", "obs_layer.1.weight_mean", "VAE_encoder_common_layers.0.bias",
             "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "deconv_layer.0.weight", "deconv_layer.0.bias",
             "deconv_layer.2.weight", "deconv_layer.2.bias", "representation_layer.0.weight", "representation_layer.0.bias",
             "y_layer.0.bias")
    for k in small:
        out["grad__" + k] = np64(grads[k])
    out["grad_slice__y_layer.0.weight"] = np64(grads["y_layer.0.weight"][:40])
    out["grad_slice__VAE_encoder_common_layers.0.weight"] = np64(grads["VAE_encoder_common_layers.0.weight"][:, :64])
    out["grad_slice__obs_layer.0.weight"] = np64(grads["obs_layer.0.weight"][:50])
    out["grad_norms"] = np.array([float(grads[k].norm()) for k in sorted(grads)])
    out["grad_names"] = np.array(sorted(grads))
    for k in list(out):
        if k.startswith("p_params") or k.startswith("test_p_params") or k in ("p_params_full",):
            out[k] = out[k][:, :200] if out[k].ndim == 2 else out[k][:, :40]
    out["dims"] = np.array([dims[0], 32, 8, 32, 5])
    np.savez_compressed(os.path.join(HERE, "d4_conv_small.npz"), **out)
    print("d4_conv_small loss", out["loss"])


def case_d4_conv_logvar():
    """conv=True together with logvar_network=True (round 3): the sigmoid of the convolutional decoder goes on the MEAN half of a real
    variable's head output only (HLVAE.py:428-430: obs_output[:, :cov_dim]), the log-variance half is the head's second output
    (HLVAE.py:42-51, loglik.py:45-47)."""
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    info = layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True)
    state = orc.init_state(dims, info, src.n_variables, seed=19, std=0.05, conv=True, logvar_network=True)
    out, grads = run_reference_model(src, np.arange(8), dims, state, seed=29, nll_scale=2.5, conv=True, logvar_network=True)
    out["data_argsum"] = np.array([out["data"].sum(), (out["data"] * np.arange(out["data"].shape[1])).sum()])
    del out["data"]
    out["state_checksum"] = state_checksum(state)
    small = ("mean_layer.0.weight", "mean_layer.0.bias", "log_var_layer.0.bias", "d_layers.0.bias", "obs_layer.0.bias",
             "obs_layer.1.bias_mean", "obs_layer.1.weight_mean", "obs_layer.1.bias_logvar", "obs_layer.1.weight_logvar",
             "VAE_encoder_common_layers.0.bias", "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "deconv_layer.0.weight",
             "deconv_layer.0.bias", "deconv_layer.2.weight", "deconv_layer.2.bias", "representation_layer.0.weight",
             "representation_layer.0.bias", "y_layer.0.bias")
    for k in small:
        out["grad__" + k] = np64(grads[k])
    out["grad_slice__y_layer.0.weight"] = np64(grads["y_layer.0.weight"][:40])
    out["grad_slice__VAE_encoder_common_layers.0.weight"] = np64(grads["VAE_encoder_common_layers.0.weight"][:, :64])
    out["grad_slice__obs_layer.0.weight"] = np64(grads["obs_layer.0.weight"][:50])
    for k in list(out):
        if k.startswith("p_params") or k.startswith("test_p_params") or k in ("p_params_full",):
            out[k] = out[k][:, :200] if out[k].ndim == 2 else out[k][:, :40]
    out["param_indexes"] = np.asarray(info["param_indexes"])
    np.savez_compressed(os.path.join(HERE, "d4_conv_logvar_small.npz"), **out)
    print("d4_conv_logvar_small loss", out["loss"])


def case_d4_conv_deep():
    """conv=True with two hidden layers per side (round 3; HLVAE.py:125-137 / 156-165 under conv, 232-242; h_dim_d reversed, :113): the
    convolutional features feed the first extra encoder layer, the last decoder layer feeds y_layer's [2592]-wide Linear."""
    src = synthetic.make_d4(n_subjects=2, T=4, seed=5)
    hid_e, hid_d = [40, 32], [24, 48]
    dims = [src.cov_dim_ext, hid_e, 8, hid_d, 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=23, std=0.05, conv=True)
    out, grads = run_reference_model(src, np.arange(8), dims, state, seed=31, nll_scale=2.5, conv=True)
    out["data_argsum"] = np.array([out["data"].sum(), (out["data"] * np.arange(out["data"].shape[1])).sum()])
    del out["data"]
    out["state_checksum"] = state_checksum(state)
    small = ("mean_layer.0.weight", "mean_layer.0.bias", "log_var_layer.0.bias", "d_layers.0.bias", "d_layers.0.weight", "d_layers.2.bias",
             "d_layers.2.weight", "_log_vy_real", "obs_layer.0.bias", "obs_layer.1.bias_mean", "obs_layer.1.weight_mean",
             "VAE_encoder_common_layers.0.bias", "VAE_encoder_common_layers.2.bias", "VAE_encoder_common_layers.2.weight", "conv1.weight",
             "conv1.bias", "conv2.weight", "conv2.bias", "deconv_layer.0.weight", "deconv_layer.0.bias", "deconv_layer.2.weight",
             "deconv_layer.2.bias", "representation_layer.0.weight", "representation_layer.0.bias", "y_layer.0.bias")
    for k in small:
        out["grad__" + k] = np64(grads[k])
    out["grad_slice__y_layer.0.weight"] = np64(grads["y_layer.0.weight"][:40])
    out["grad_slice__VAE_encoder_common_layers.0.weight"] = np64(grads["VAE_encoder_common_layers.0.weight"][:, :64])
    for k in list(out):
        if k.startswith("p_params") or k.startswith("test_p_params") or k in ("p_params_full",):
            out[k] = out[k][:, :200] if out[k].ndim == 2 else out[k][:, :40]
    out["hid_e"], out["hid_d"] = np.array(hid_e), np.array(hid_d)
    np.savez_compressed(os.path.join(HERE, "d4_conv_deep_small.npz"), **out)
    print("d4_conv_deep_small loss", out["loss"])


def case_types_info():
    """reference read_data on CSV files written from the mix spec (layout pin)."""
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        # raw file: class indices for cat/ordinal, values otherwise (read_data re-encodes them)
        raw = np.zeros((24, len(MIX_SPEC)))
        off = 0
        for j, (t, k) in enumerate(MIX_SPEC):
            w = k if t in ("cat", "ordinal") else 1
            blk = src.data[:, off:off + w]
            raw[:, j] = blk.argmax(1) if t == "cat" else (blk.sum(1) - 1 if t == "ordinal" else blk[:, 0])
            if t == "count":
                raw[:, j] -= 1          # read_data shifts +1 only when the column min is 0 (:103-105)
                raw[0, j] = 0
            if t in ("cat", "ordinal"):
                raw[:k, j] = np.arange(k)    # make every level appear so np.unique keeps the codes
            off += w
        np.savetxt(os.path.join(d, "data.csv"), raw, delimiter=",", fmt="%.10g")
        np.savetxt(os.path.join(d, "mask.csv"), src.mask.astype(int), delimiter=",", fmt="%d")
        with open(os.path.join(d, "types.csv"), "w") as f:
            f.write("type,dim,nclass\n")
            for t, k in MIX_SPEC:
                f.write(f"{t},1,{k if t in ('cat', 'ordinal') else 1}\n")
        data, info, miss, true_miss, n, nv = ref_rf.read_data(os.path.join(d, "data.csv"), os.path.join(d, "mask.csv"),
                                                              os.path.join(d, "none.csv"), os.path.join(d, "types.csv"), None)
    np.savez_compressed(os.path.join(HERE, "types_info_mix.npz"), raw=raw, mask=src.mask, data=data,
                        set_of_types=np.array(["%s:%s" % t for t in info["set_of_types"]]),
                        data_types_indexes=info["data_types_indexes"], exp_types_indexes=info["exp_types_indexes"],
                        param_indexes=info["param_indexes"], param_miss_mask=info["param_miss_mask"])
    print("types_info", info["set_of_types"])


class _Evaluated:
    def __init__(self, t):
        self.t = t

    def evaluate(self):
        return self.t


class _LazyAdapter(torch.nn.Module):
    """gpytorch kernels return lazy tensors with .evaluate(); GP_model.py kernels return tensors."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, a, b):
        return _Evaluated(self.k(a, b))


class _Noise:
    class _NC:
        pass

    def __init__(self, noise):
        self.noise_covar = self._NC()
        self.noise_covar.noise = noise

    def eval(self):
        return self


def case_gp():
    torch.manual_seed(0)
    L, M, Q, idc = 4, 10, 6, 2
    Ts = [3, 6, 4, 5, 6]
    rows = []
    for s, T in enumerate(Ts):
        sick = s % 2
        for t in range(T):
            rows.append([float(t), float(t - 2) if sick else 0.0, float(s + 10), float(s % 2), float(sick), float((s // 2) % 2)])
    x = torch.tensor(rows, dtype=T64)
    perm = torch.randperm(x.shape[0])                # rows of a subject need not be contiguous
    x = x[perm]
    B = x.shape[0]
    cfg = dict(cat_kernel=[2], bin_kernel=[], sqexp_kernel=[0],
               cat_int_kernel=[{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                               {"cont_covariate": 1, "cat_covariate": 4}], bin_int_kernel=[])
    k0, k1 = ref_gp.generate_kernel_batched(L, cfg["cat_kernel"], cfg["bin_kernel"], cfg["sqexp_kernel"],
                                            cfg["cat_int_kernel"], cfg["bin_int_kernel"], [], idc)
    k0, k1 = k0.double(), k1.double()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():            # de-symmetrise the hyper-parameters over latent dims
        for p in list(k0.parameters()) + list(k1.parameters()):
            p.add_(0.3 * torch.randn(p.shape, generator=g, dtype=T64))
    mu = torch.randn(B, L, generator=g, dtype=T64).requires_grad_(True)
    log_v = (0.5 * torch.randn(B, L, generator=g, dtype=T64) - 1.0).requires_grad_(True)
    z = x[torch.randperm(B, generator=g)[:M]].clone()[None].repeat(L, 1, 1)
    z = (z + 0.05 * torch.randn(z.shape, generator=g, dtype=T64)).requires_grad_(True)
    m = torch.randn(L, M, 1, generator=g, dtype=T64)
    Hh = torch.randn(L, M, M, generator=g, dtype=T64) / 10
    H = Hh @ Hh.transpose(-1, -2) + 0.05 * torch.eye(M, dtype=T64)
    noise = torch.ones(L, 1, dtype=T64)              # constrain_scales: noise fixed to 1 (HLVAE_main.py:211-213)
    P, P_b, N, eps = 40, len(Ts), 777, 1e-6
    kld, grad_m, grad_H = ref_elbo.minibatch_KLD_upper_bound_iter(
        _LazyAdapter(k0), _LazyAdapter(k1), _Noise(noise), L, m, H, x, mu, log_v, z, P, P_b, N, True, idc, eps)
    kld.sum().backward()
    # natural-gradient step, training.py:130-137
    lr = 0.01
    LH = torch.linalg.cholesky(H)
    eye = torch.eye(M, dtype=T64)
    iH = torch.cholesky_solve(eye, LH)
    iH_new = iH + lr * (grad_H + grad_H.transpose(-1, -2))
    H_new = torch.cholesky_solve(eye, torch.linalg.cholesky(iH_new)).detach()
    m_new = torch.matmul(H_new, torch.matmul(iH, m) - lr * (grad_m - 2 * torch.matmul(grad_H, m))).detach()
    out = dict(x=np64(x), mu=np64(mu), log_v=np64(log_v), z=np64(z), m=np64(m), H=np64(H), noise=np64(noise.flatten()),
               scalars=np.array([P, P_b, N, eps, idc, lr]), kld=np64(kld), grad_m=np64(grad_m), grad_H=np64(grad_H),
               d_mu=np64(mu.grad), d_log_v=np64(log_v.grad), d_z=np64(z.grad), m_new=np64(m_new), H_new=np64(H_new))
    # raw hyper-parameters in the oracle's naming (same term order as generate_kernel_batched)
    for name, add in (("k0", k0), ("k1", k1)):
        for ti, sk in enumerate(add.kernels):
            out[f"kp__{name}.{ti}.scale"] = np64(sk._log_scale)
            out[f"kg__{name}.{ti}.scale"] = np64(sk._log_scale.grad)
            fac = []

            def walk(k):
                if isinstance(k, ref_gp.ProductKernel):
                    walk(k.k1); walk(k.k2)
                else:
                    fac.append(k)
            walk(sk.kernel)
            for fi, f in enumerate(fac):
                if isinstance(f, ref_gp.RbfKernel):
                    out[f"kp__{name}.{ti}.{fi}.ls"] = np64(f._log_lengthscale)
                    out[f"kg__{name}.{ti}.{fi}.ls"] = np64(f._log_lengthscale.grad)
    np.savez_compressed(os.path.join(HERE, "gp_kl.npz"), **out)
    print("gp_kl", out["kld"])


def case_gp_predict():
    """GP posterior prediction of the latent (reference utils.py:99-191, batch_predict_varying_T) run with the GP_model.py
    kernels: known subjects at new times plus rows of an unseen subject."""
    torch.manual_seed(1)
    L, M, Q, idc = 4, 10, 6, 2
    Ts = [3, 6, 4, 5, 6]
    rows = []
    for s, T in enumerate(Ts):
        sick = s % 2
        for t in range(T):
            rows.append([float(t), float(t - 2) if sick else 0.0, float(s + 10), float(s % 2), float(sick), float((s // 2) % 2)])
    x = torch.tensor(rows, dtype=T64)
    x = x[torch.randperm(x.shape[0])]
    trows = []
    for s in (1, 3, 4):                                    # known subjects, later time points
        sick = s % 2
        for t in (7, 8, 9):
            trows.append([float(t), float(t - 2) if sick else 0.0, float(s + 10), float(s % 2), float(sick), float((s // 2) % 2)])
    for t in range(3):                                     # an unseen subject
        trows.append([float(t), 0.0, 99.0, 1.0, 0.0, 1.0])
    tx = torch.tensor(trows, dtype=T64)
    cfg = dict(cat_kernel=[2], bin_kernel=[], sqexp_kernel=[0],
               cat_int_kernel=[{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                               {"cont_covariate": 1, "cat_covariate": 4}], bin_int_kernel=[])
    k0, k1 = ref_gp.generate_kernel_batched(L, cfg["cat_kernel"], cfg["bin_kernel"], cfg["sqexp_kernel"],
                                            cfg["cat_int_kernel"], cfg["bin_int_kernel"], [], idc)
    k0, k1 = k0.double(), k1.double()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for p in list(k0.parameters()) + list(k1.parameters()):
            p.add_(0.3 * torch.randn(p.shape, generator=g, dtype=T64))
    mu = torch.randn(x.shape[0], L, generator=g, dtype=T64)
    z = x[torch.randperm(x.shape[0], generator=g)[:M]].clone()[None].repeat(L, 1, 1)
    z = z + 0.05 * torch.randn(z.shape, generator=g, dtype=T64)
    noise = torch.ones(L, 1, dtype=T64)
    with torch.no_grad():
        Zp = ref_utils.batch_predict_varying_T(L, _LazyAdapter(k0), _LazyAdapter(k1), _Noise(noise), x, tx, mu, z, idc, 1e-6)
    out = dict(x=np64(x), test_x=np64(tx), mu=np64(mu), z=np64(z), noise=np64(noise.flatten()), Z_pred=np64(Zp),
               scalars=np.array([1e-6, idc]))
    for name, add in (("k0", k0), ("k1", k1)):
        for ti, sk in enumerate(add.kernels):
            out[f"kp__{name}.{ti}.scale"] = np64(sk._log_scale)
            fac = []

            def walk(k):
                if isinstance(k, ref_gp.ProductKernel):
                    walk(k.k1); walk(k.k2)
                else:
                    fac.append(k)
            walk(sk.kernel)
            for fi, f in enumerate(fac):
                if isinstance(f, ref_gp.RbfKernel):
                    out[f"kp__{name}.{ti}.{fi}.ls"] = np64(f._log_lengthscale)
    np.savez_compressed(os.path.join(HERE, "gp_predict.npz"), **out)
    print("gp_predict", out["Z_pred"][:2])


if __name__ == "__main__":
    torch.set_num_threads(4)
    case_types_info()
    case_mix("mix_init", seed_state=1, std=0.05, nll_scale=1.7)
    case_mix("mix_trained", seed_state=2, std=0.3, nll_scale=0.4)
    case_d4()
    case_d4_conv()
    case_gp()
    case_gp_predict()
    case_mix_logvar_deep()
    case_d4_conv_logvar()
    case_d4_conv_deep()
    case_mix_no_hidden()
