"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-step "imputed values" metrics
(SURVEY.md section 8(a) row M): reference training.py:84-101 calling
HL_VAE/read_functions.py:206-218 (p_params_concatenation_by_key), :221-235
(discrete_variables_transformation), :268-339 (statistics) and :342-412 (error_computation).
float64, MLP path (types_info['conv'] False), real/pos/count/cat/ordinal only.
Pinned by tests/golden (fixtures produced by the reference's own read_functions).
"""
from __future__ import annotations

import torch

from hlvae_oracle import block_indices

DT = torch.float64


def params_by_key(p_params, blocks, B, Theta):
    """read_functions.py:206-218: per-block params -> [B, Theta]."""
    out = torch.zeros(B, Theta, dtype=DT)
    for b, p in zip(blocks, p_params):
        if isinstance(p, (list, tuple)):                                 # [mean, var] of a real / pos block under logvar_network (:211-213)
            p = torch.cat(list(p), 1)
        out[:, b["par"]] = p.reshape(B, -1)
    return out


def discrete_transform(data, blocks, D):
    """read_functions.py:221-235: one-hot -> argmax, thermometer -> sum-1, others copied."""
    B = data.shape[0]
    out = torch.zeros(B, D, dtype=DT)
    for b in blocks:
        x = data[:, b["exp"]]
        if b["type"] == "cat":
            out[:, b["var"]] = torch.argmax(x.reshape(B, -1, b["K"]), 2).to(DT)
        elif b["type"] == "ordinal":
            out[:, b["var"]] = (x.reshape(B, -1, b["K"]).sum(2) - 1).to(DT)
        else:
            out[:, b["var"]] = x
    return out


def statistics(params_full, blocks, D, log_vy_pos):
    """read_functions.py:268-302: per-type mean and mode of the fitted likelihoods."""
    B = params_full.shape[0]
    mean = torch.zeros(B, D, dtype=DT)
    mode = torch.zeros(B, D, dtype=DT)
    for b in blocks:
        p = params_full[:, b["par"]]
        n = len(b["var"])
        if b["type"] == "real":
            mean[:, b["var"]] = p[:, :n]                                  # :276-277 (indx = the first sz columns)
            mode[:, b["var"]] = p[:, :n]
        elif b["type"] == "pos":
            # :283-286: exp(log_vy[1]) when the variance is a free parameter, otherwise (logvar_network: log_vy[1] is None, the
            # exp raises) the est_var columns of the parameter block
            var = torch.exp(log_vy_pos) if log_vy_pos is not None else p[:, n:]
            mean[:, b["var"]] = torch.exp(p[:, :n] + 0.5 * var) - 1.0     # :288
            mode[:, b["var"]] = torch.exp(p[:, :n] - var) - 1.0           # :290
        elif b["type"] == "count":
            mean[:, b["var"]] = p
            mode[:, b["var"]] = torch.floor(p)
        else:
            am = torch.argmax(p.reshape(B, -1, b["K"]), 2).to(DT)
            mean[:, b["var"]] = am
            mode[:, b["var"]] = am
    return mean, mode


def error_computation(x_train, x_hat, blocks, mask, conv=False):
    """read_functions.py:342-386 with true_miss_mask = ones, dim 0 (conv: :366-369).
    Returns per-variable (error_observed, error_missing, error_all) BEFORE the per-type sqrt
    bookkeeping, and the same after it (:388-393)."""
    err = torch.zeros_like(x_train)
    for b in blocks:
        xt, xh = x_train[:, b["var"]], x_hat[:, b["var"]]
        if b["type"] == "cat":
            e = (xt != xh).to(DT)
        elif b["type"] == "ordinal":
            e = torch.abs(xt - xh) / b["K"]
        elif conv:
            xt = xt / 255                                                 # :367
            if b["type"] in ("pos", "count"):
                xh = xh / 255                                             # :368-369
            e = (xh - xt) ** 2
        else:
            # get_norm_terms (HL_VAE/utils.py:216-225) fills a torch.empty(sz) -> FLOAT32 vector, and
            # read_functions.py:373 squares it in float32 before the float64 division.
            nt = (xt.max(0).values - xt.min(0).values).to(torch.float32)
            nt = torch.where(nt == 0, torch.ones_like(nt), nt)            # :372
            e = (xh - xt) ** 2 / (nt ** 2).to(DT)
        err[:, b["var"]] = e
    ones = torch.ones_like(mask)
    known_missing = ones * (1 - mask)
    ms = mask.sum(0); ms = torch.where(ms == 0, torch.ones_like(ms), ms)
    mm = known_missing.sum(0); mm = torch.where(mm == 0, torch.ones_like(mm), mm)
    ks = ones.sum(0)
    e_obs = (err * mask).sum(0) / ms
    e_mis = (err * known_missing).sum(0) / mm
    e_all = err.sum(0) / ks
    is_disc = torch.zeros(x_train.shape[1], dtype=torch.bool)
    for b in blocks:
        if b["type"] in ("cat", "ordinal"):
            is_disc[b["var"]] = True
    fin = lambda e: torch.where(is_disc, e, torch.sqrt(e))                # :390-393
    return fin(e_obs), fin(e_mis), fin(e_all)


def step_metrics(p_params, data, mask, types_info, log_vy_pos, conv=False):
    """training.py:84-101 up to error_computation; returns (x_hat_mean, e_obs, e_mis, e_all)."""
    blocks = block_indices(types_info)
    B, D = mask.shape
    full = params_by_key(p_params, blocks, B, len(types_info["param_indexes"]))
    xt = discrete_transform(data, blocks, D)
    xh, _ = statistics(full, blocks, D, log_vy_pos)
    e_obs, e_mis, e_all = error_computation(xt, xh, blocks, mask, conv=conv)
    return xh, e_obs, e_mis, e_all
