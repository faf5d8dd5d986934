"""Column plan ("layout compiler") for heterogeneous variables.

The reference describes the heterogeneous column layout with a ``types_info``
dict built while reading the CSV files (reference HL_VAE/read_functions.py:142-198)
and then walks it with boolean-mask gathers for every type block in every step
(HL_VAE/utils.py:94-141, HLVAE.py:387-412, HLVAE.py:422-452).

Here the same information is compiled ONCE into a static per-variable table that
the HIP kernels index directly:

    kind[d]   0 real, 1 pos, 2 count, 3 cat, 4 ordinal
    ncls[d]   K for cat/ordinal, 1 otherwise
    xoff[d]   first column of variable d in the expanded data matrix  [B, X]
    poff[d]   first column of variable d in the parameter matrix      [B, Theta]
    blk[d]    type-block id  (index into ``set_of_types`` / ``obs_layer``)
    bidx[d]   index of variable d inside its type block (row of the head weights)
    sidx[d]   index into the real / pos batch-statistic vectors (-1 otherwise)

The head parameters of variable d (``obs_layer[blk[d]]`` row ``bidx[d]``) stay in the
reference's tensor shapes inside one flat fp32 parameter arena; their arena offsets
are appended to this table by the model (HLVAE.py in this package).

``build_types_info`` produces the reference's dict (same keys, same dtypes,
same lexicographic block ordering, read_functions.py:145-146) from a plain list
of per-variable type records so that no CSV round trip is needed.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

KIND_REAL, KIND_POS, KIND_COUNT, KIND_CAT, KIND_ORDINAL = 0, 1, 2, 3, 4
KIND_OF = {"real": KIND_REAL, "pos": KIND_POS, "count": KIND_COUNT,
           "cat": KIND_CAT, "ordinal": KIND_ORDINAL}
KIND_NAME = {v: k for k, v in KIND_OF.items()}

Y_DIM_MAX = 8      # per-variable homogeneous representation width supported by the head kernel
NCLS_MAX = 32      # largest class count supported by the head kernel


def make_types_dict(spec: Sequence[Tuple[str, int]]) -> List[Dict[str, str]]:
    """[(type, nclass), ...] -> list of dicts as csv.DictReader would give
    (all values are strings, reference read_functions.py:15-17)."""
    out = []
    for t, k in spec:
        if t not in KIND_OF:
            raise ValueError(f"unsupported variable type {t!r} (supported: {sorted(KIND_OF)})")
        out.append({"type": t, "dim": "1", "nclass": str(int(k) if t in ("cat", "ordinal") else 1)})
    return out


def build_types_info(types_dict: Sequence[Dict[str, str]], miss_mask: np.ndarray | None = None,
                     logvar_network: bool = False, conv: bool = False) -> dict:
    """Re-statement of the index bookkeeping of read_functions.read_data
    (reference HL_VAE/read_functions.py:142-198) without the CSV parsing.

    Returns the reference's ``types_info`` dict.  ``miss_mask`` ([N, D], 1 = observed)
    is only needed for ``param_miss_mask``.
    """
    types_dict = [dict(t) for t in types_dict]
    D = len(types_dict)
    type_tuple = [(t["type"], str(t["nclass"])) for t in types_dict]
    # sorted set of (type, nclass-STRING): lexicographic, so ('cat','10') < ('cat','5')
    set_of_types = sorted(set(type_tuple))
    sizes = []
    for t in types_dict:
        k = int(t["nclass"])
        sizes.append(k if t["type"] in ("cat", "ordinal") else 1)
    X = int(np.sum(sizes))
    # parameter slots per variable (read_functions.py:162-171): real / pos carry (mean, log-variance) under logvar_network
    psizes = [2 if (logvar_network and t["type"] in ("real", "pos")) else sz for t, sz in zip(types_dict, sizes)]
    types_indexes = np.zeros(D)
    exp_types_indexes = np.zeros(X)
    param_indexes = np.zeros(int(np.sum(psizes)))
    pos = ppos = 0
    for i, t in enumerate(types_dict):
        tid = set_of_types.index((t["type"], str(t["nclass"])))
        types_indexes[i] = tid
        exp_types_indexes[pos:pos + sizes[i]] = tid
        param_indexes[ppos:ppos + psizes[i]] = tid
        pos += sizes[i]
        ppos += psizes[i]
    info = {
        "types_dict": types_dict,
        "set_of_types": set_of_types,
        "data_types_indexes": types_indexes,
        "exp_types_indexes": exp_types_indexes,
        "param_indexes": param_indexes,
        "beta_ranges": [],
        "conv": bool(conv),
        "use_ranges": False,
        "conv_range": False,
    }
    if miss_mask is not None:
        mm = np.asarray(miss_mask, dtype=np.float64)
        pm = expand_mask(mm, psizes)
        if logvar_network:      # read_functions.py:179-183: a real / pos block's slots are [all means | all log-variances]
            for i, tpl in enumerate(set_of_types):
                if tpl[0] in ("real", "pos"):
                    blk = mm[:, types_indexes == i]
                    pm[:, param_indexes == i] = np.concatenate([blk, blk], 1)
        info["param_miss_mask"] = pm
    return info


def expand_mask(mask: np.ndarray, sizes: Sequence[int]) -> np.ndarray:
    """[N, D] observation mask -> [N, Theta] parameter mask: each variable's bit is
    repeated over its parameter slots (reference read_functions.py:173-176)."""
    return np.repeat(mask, np.asarray(sizes, dtype=np.int64), axis=1)


@dataclass
class ColumnPlan:
    """Static per-variable table derived from ``types_info`` (see module docstring)."""
    D: int
    X: int
    Theta: int
    y_dim: int
    set_of_types: List[Tuple[str, str]]
    kind: np.ndarray
    ncls: np.ndarray
    xoff: np.ndarray
    poff: np.ndarray
    blk: np.ndarray
    bidx: np.ndarray
    sidx: np.ndarray
    n_real: int
    n_pos: int
    poff2: np.ndarray = None        # column of the log-variance slot of a real / pos variable under logvar_network, else -1
    logvar_network: bool = False
    blocks: List[dict] = field(default_factory=list)   # per type-block summary





def compile_plan(types_info: dict, y_dim: int) -> ColumnPlan:
    """types_info -> ColumnPlan.  Checks everything the kernels assume."""
    if y_dim < 1 or y_dim > Y_DIM_MAX:
        raise ValueError(f"y_dim={y_dim} outside supported range 1..{Y_DIM_MAX}")
    td = types_info["types_dict"]
    sot = [tuple(map(str, t)) for t in types_info["set_of_types"]]
    D = len(td)
    dti = np.asarray(types_info["data_types_indexes"]).astype(np.int64)
    eti = np.asarray(types_info["exp_types_indexes"]).astype(np.int64)
    pti = np.asarray(types_info["param_indexes"]).astype(np.int64)
    kind = np.zeros(D, np.int32)
    ncls = np.ones(D, np.int32)
    xoff = np.zeros(D, np.int32)
    poff = np.zeros(D, np.int32)
    blk = np.zeros(D, np.int32)
    bidx = np.zeros(D, np.int32)
    sidx = -np.ones(D, np.int32)
    counters = [0] * len(sot)
    x = 0
    n_real = n_pos = 0
    n_cont = sum(1 for t in td if t["type"] in ("real", "pos"))
    if len(pti) not in (len(eti), len(eti) + n_cont):
        raise ValueError("types_info['param_indexes'] has %d slots: expected %d, or %d with logvar_network" % (len(pti), len(eti), len(eti) + n_cont))
    logvar = len(pti) != len(eti)
    # columns of every block inside the parameter matrix [B, Theta], in order (theta of a block is scattered into them in order:
    # HLVAE.py:448; a real / pos block under logvar_network is [means | log-variances], HLVAE.py:50)
    pcols = [np.nonzero(pti == b)[0] for b in range(len(sot))]
    n_in_block = [int(np.sum(dti == b)) for b in range(len(sot))]
    poff2 = -np.ones(D, np.int32)
    for d, t in enumerate(td):
        ty = t["type"]
        if ty not in KIND_OF:
            raise ValueError(f"variable {d}: type {ty!r} is not on the MLP hot path "
                             f"(supported: {sorted(KIND_OF)})")
        K = int(t["nclass"]) if ty in ("cat", "ordinal") else 1
        if ty in ("cat", "ordinal") and not (2 <= K <= NCLS_MAX):
            raise ValueError(f"variable {d}: nclass={K} outside supported range 2..{NCLS_MAX}")
        b = sot.index((ty, str(t["nclass"])))
        if dti[d] != b:
            raise ValueError("types_info['data_types_indexes'] inconsistent with set_of_types")
        kind[d], ncls[d], xoff[d], blk[d] = KIND_OF[ty], K, x, b
        if not np.all(eti[x:x + K] == b):
            raise ValueError("types_info index vectors inconsistent at variable %d" % d)
        bidx[d] = counters[b]
        counters[b] += 1
        if ty in ("cat", "ordinal"):
            poff[d] = pcols[b][bidx[d] * K]
            if not np.all(pcols[b][bidx[d] * K:(bidx[d] + 1) * K] == poff[d] + np.arange(K)):
                raise ValueError("types_info['param_indexes'] inconsistent at variable %d" % d)
        else:
            poff[d] = pcols[b][bidx[d]]
            if logvar and ty in ("real", "pos"):
                poff2[d] = pcols[b][n_in_block[b] + bidx[d]]
        if ty == "real":
            sidx[d] = n_real
            n_real += 1
        elif ty == "pos":
            sidx[d] = n_pos
            n_pos += 1
        x += K
    if x != len(eti):
        raise ValueError("expanded width mismatch: %d vs %d" % (x, len(eti)))
    blocks = []
    for b, (ty, k) in enumerate(sot):
        sel = np.nonzero(blk == b)[0]
        blocks.append({"type": ty, "nclass": int(k), "n_vars": int(len(sel)), "vars": sel})
    return ColumnPlan(D=D, X=x, Theta=len(pti), y_dim=y_dim, set_of_types=sot, kind=kind, ncls=ncls,
                      xoff=xoff, poff=poff, blk=blk, bidx=bidx, sidx=sidx,
                      n_real=n_real, n_pos=n_pos, blocks=blocks, poff2=poff2, logvar_network=logvar)
