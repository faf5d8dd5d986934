#!/usr/bin/env python3
"""Probes of the REFERENCE itself (run in the build container only; needs /root/reference) for the modes this build scopes
out, so that DESIGN.md section 7 rests on what the reference actually does, not on a reading of it:

  * 'beta' variables: HLVAE.forward + the per-step metrics of training.py:84-101 (p_params_concatenation_by_key, statistics)
    with 1, 2, 3 beta variables, logvar_network False / True;
  * logvar_network=True without beta variables;
  * more than one hidden layer per side.

Prints one line per probe.  Nothing is stored: these are behaviours (exceptions), not vectors."""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


H = load_ref("ref_HLVAE", "HLVAE.py")
RF = load_ref("ref_rf", "HL_VAE/read_functions.py")


def types_info(td, logvar):
    tt = [((d["type"], str(d["dim"])) if d["type"] == "beta" else (d["type"], str(d["nclass"]))) for d in td]
    sot = sorted(set(tt))
    idx, eidx, pidx = [], [], []
    for d, t in zip(td, tt):
        i = sot.index(t)
        idx.append(i)
        K = int(d["nclass"]) if d["type"] in ("cat", "ordinal") else 1
        eidx += [i] * K
        sz = K if d["type"] in ("cat", "ordinal") else (1 if d["type"] in ("count", "beta") or not logvar else 2)   # read_functions.py:162-171
        pidx += [i] * sz
    return dict(types_dict=td, set_of_types=sot, data_types_indexes=np.array(idx, float), exp_types_indexes=np.array(eidx, float),
                param_indexes=np.array(pidx, float), beta_ranges=[[0, 10 + 1e-3] for d in td if d["type"] == "beta"], conv=False,
                use_ranges=False, conv_range=False)


def probe(label, td, logvar=False, hidden=(8,)):
    info = types_info(td, logvar)
    X, Th, D, B = len(info["exp_types_indexes"]), len(info["param_indexes"]), len(td), 6
    torch.manual_seed(0)
    data = torch.rand(B, X, dtype=torch.float64) * 9 + 0.5
    mask = torch.ones(B, D, dtype=torch.float64)
    stage = "construct"
    try:
        m = H.HLVAE([X, list(hidden), 2, list(hidden), 5], info, D, vy_init=[1., .5], logvar_network=logvar, conv=False).double()
        stage = "forward"
        out = m(data, mask, torch.ones(B, Th, dtype=torch.float64), info)
        stage = "backward"
        m.loss_function(out[3]).sum().backward()
        stage = "metrics: p_params_concatenation_by_key"
        full = RF.p_params_concatenation_by_key([out[5]], info, B, data.device, "x")
        stage = "metrics: statistics"
        RF.statistics(full, info, data.device, False, [m._log_vy_real, m._log_vy_pos])
        print(f"{label:58s} runs (forward, backward, per-step metrics)")
    except Exception as e:      # noqa: BLE001
        print(f"{label:58s} FAILS at {stage}: {type(e).__name__}: {str(e)[:110]}")


real = {"type": "real", "dim": 1, "nclass": 1}
pos = {"type": "pos", "dim": 1, "nclass": 1}
cat = {"type": "cat", "dim": 1, "nclass": 3}
beta = {"type": "beta", "dim": 1, "nclass": 1}
c = lambda *a: [dict(x) for x in a]
probe("1 beta variable, logvar_network=False", c(real, beta))
probe("2 beta variables, logvar_network=False", c(real, beta, beta))
probe("3 beta variables, logvar_network=False", c(real, beta, beta, beta))
probe("2 beta variables, logvar_network=True", c(real, beta, beta), logvar=True)
probe("pos + beta variables (pos_dim counts both)", c(pos, pos, beta, beta))
probe("real + pos + cat, logvar_network=True", c(real, pos, cat, real), logvar=True)
probe("real + pos + cat, logvar_network=False, hidden [8]", c(real, pos, cat, real))
probe("real + pos + cat, hidden [8, 6]", c(real, pos, cat, real), hidden=(8, 6))
