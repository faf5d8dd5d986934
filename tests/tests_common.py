"""Shared helpers for the test-suite (fixture loading, error norms)."""
import os

import numpy as np
import torch

from hlvae_amd import synthetic

MIX_SPEC = [("real", 1), ("cat", 3), ("pos", 1), ("ordinal", 4), ("count", 1), ("cat", 5), ("real", 1),
            ("ordinal", 5), ("pos", 1), ("cat", 5), ("count", 1), ("cat", 3), ("real", 1), ("ordinal", 4)]


def _np(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().double().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b):
    a, b = _np(a), _np(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def max_abs_err(a, b):
    return float(np.max(np.abs(_np(a) - _np(b))))


def load_mix_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    assert np.array_equal(src.data, g["data"]), "synthetic generator drifted from the fixture inputs"
    d = [int(v) for v in g["dims"]]
    dims = [d[0], [d[1]], d[2], [d[3]], d[4]]
    state = {k[len("state__"):]: torch.tensor(g[k]) for k in g.files if k.startswith("state__")}
    return g, src, dims, state


def load_mode_case(golden_dir, name):
    """fixtures of the reference's non-default constructor modes (mix_logvar, mix_deep, mix_logvar_deep): logvar_network=True
    and / or two hidden layers per side.  Returns (g, src, dims, state, types_info, logvar_network)."""
    from hlvae_amd import layout
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    assert np.array_equal(src.data, g["data"]), "synthetic generator drifted from the fixture inputs"
    lvn = bool(int(g["logvar_network"][0]))
    info = src.types_info
    if lvn:
        info = layout.build_types_info(src.types_info["types_dict"], miss_mask=src.mask, logvar_network=True)
        for t in info["types_dict"]:
            t["dim"], t["nclass"] = int(t["dim"]), int(t["nclass"])
    assert np.array_equal(np.asarray(info["param_indexes"]), g["param_indexes"])
    dims = [src.cov_dim_ext, [int(v) for v in g["hid_e"]], 4, [int(v) for v in g["hid_d"]], 5]
    state = {k[len("state__"):]: torch.tensor(g[k]) for k in g.files if k.startswith("state__")}
    return g, src, dims, state, info, lvn
