"""Synthetic Heterogeneous-HealthMNIST-shaped inputs.

The reference ships no data: its generator needs MNIST JPEGs from an external URL
(reference Heterogeneous_Health_MNIST_generate.py:142-148) and the CSV splits named in
config/hlvae_config_file.txt are absent.  This module synthesises inputs with the SAME
shapes, encodings and label schema so the hot path sees what it would see in production:

* 36x36 = 1296 variables per row (generate.py:109,189); D4 layout: top-left quadrant
  ``region_1`` real-valued pixels 0..255, the other three quadrants quantised to five
  levels {25,75,125,175,225} (generate.py:58-66,120-135,190-197) -> 5-class categorical,
  one-hot expanded exactly as read_functions.read_data does (read_functions.py:67-82).
* i.i.d. Bernoulli(observed = 0.75) observation mask (generate.py:29,74,117-118).
* T = 20 rows per subject, time_age = 0..19, disease_time = -9..10 for sick subjects and
  NaN -> 0 otherwise (generate.py:105-106,183-188; dataset_def.py:84), label columns in the
  order [time_age, disease_time, subject, gender, disease, location] (dataset_def.py:46-47).

Pixel values are NOT MNIST digits: each subject is a pair of anisotropic Gaussian blobs
rotated by the same angle schedule the generator applies to the digit
(5 deg baseline, +45*sigmoid(t) when sick, N(0,2) jitter) and shifted diagonally by t/10.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .layout import build_types_info, make_types_dict

IMG = 36
LEVELS = np.array([25, 75, 125, 175, 225])


def region_index_sets() -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Quadrant index sets of the flattened 36x36 image (generate.py:120-135)."""
    idx = np.arange(IMG * IMG).reshape(IMG, IMG)
    return (idx[:18, :18].ravel(), idx[:18, 18:].ravel(), idx[18:, :18].ravel(), idx[18:, 18:].ravel())


def d4_type_spec() -> List[Tuple[str, int]]:
    r1, _, _, _ = region_index_sets()
    spec = [("cat", 5)] * (IMG * IMG)
    for i in r1:
        spec[i] = ("real", 1)
    return spec


def tabular_type_spec(n_real=16, n_pos=16, n_count=8, n_cat=16, n_ord=8, K=5, interleave=True):
    """BASELINE.json config 4: 64 mixed-type features.  K is not fixed by BASELINE; K=5 is used."""
    spec = ([("real", 1)] * n_real + [("pos", 1)] * n_pos + [("count", 1)] * n_count
            + [("cat", K)] * n_cat + [("ordinal", K)] * n_ord)
    if interleave:   # deterministic interleave so type blocks are NOT contiguous in the file order
        order = np.argsort((np.arange(len(spec)) * 37) % len(spec), kind="stable")
        spec = [spec[i] for i in order]
    return spec


@dataclass
class HetBatchSource:
    """Host-side container of a whole synthetic data set in the reference's array formats
    (the outputs of read_functions.read_data + the label frame of dataset_def.py)."""
    data: np.ndarray          # [N, X] float64, expanded (one-hot / thermometer), NaN-free
    mask: np.ndarray          # [N, D] float64, 1 = observed
    param_mask: np.ndarray    # [N, Theta] float64
    labels: np.ndarray        # [N, Q] float64
    types_info: dict
    n_variables: int
    cov_dim_ext: int
    id_covariate: int = 2

    def __len__(self):
        return self.data.shape[0]

    @property
    def n_subjects(self):
        return int(np.unique(self.labels[:, self.id_covariate]).size)


def _render_subjects(P: int, T: int, rng: np.random.Generator):
    """[P*T, 1296] integer pixel rows 0..255 and the label table."""
    t_age = np.arange(T, dtype=np.float64)
    t_dis = np.arange(-9, -9 + T, dtype=np.float64)
    sick = rng.binomial(1, 0.5, P)
    loc = rng.binomial(1, 0.5, P)
    gender = (np.arange(P) >= P // 2).astype(np.int64)       # first half "3", second half "6"
    rot = rng.normal(0.0, 2.0, (P, T))
    rot = rot + np.where(sick[:, None] == 1, 45.0 / (1.0 + np.exp(-t_dis))[None, :], 5.0)
    ang = np.deg2rad(rot)                                    # [P, T]
    yy, xx = np.meshgrid(np.arange(IMG, dtype=np.float64), np.arange(IMG, dtype=np.float64), indexing="ij")
    c = (IMG - 1) / 2.0
    shift = (t_age / 10.0)[None, :, None, None]
    # every random draw happens here, for all subjects at once (the rendering below is chunked over subjects to bound the
    # memory of the 100 k-row sets: 8 arrays of [chunk, T, 36, 36] instead of [P, T, 36, 36])
    strokes = []
    for b in range(2):                                       # two strokes per subject
        cx = rng.uniform(-7, 7, (P, 1, 1, 1)) + (gender[:, None, None, None] * 2 - 1) * (3 - 6 * b)
        cy = rng.uniform(-7, 7, (P, 1, 1, 1))
        sx = rng.uniform(2.5, 9.0, (P, 1, 1, 1))
        sy = rng.uniform(2.0, 5.0, (P, 1, 1, 1))
        strokes.append((cx, cy, sx, sy))
    img = np.zeros((P, T, IMG, IMG))
    for lo in range(0, P, 256):
        hi = min(P, lo + 256)
        X0 = xx[None, None] - c - shift
        Y0 = yy[None, None] - c - shift
        ca, sa = np.cos(ang[lo:hi])[:, :, None, None], np.sin(ang[lo:hi])[:, :, None, None]
        XR = ca * X0 + sa * Y0
        YR = -sa * X0 + ca * Y0
        part = np.zeros((hi - lo, T, IMG, IMG))
        for cx, cy, sx, sy in strokes:
            part += np.exp(-0.5 * (((XR - cx[lo:hi]) / sx[lo:hi]) ** 2 + ((YR - cy[lo:hi]) / sy[lo:hi]) ** 2))
        img[lo:hi] = np.clip(np.rint(255.0 * part / part.max(axis=(2, 3), keepdims=True)), 0, 255)
    pixels = img.reshape(P * T, IMG * IMG)
    subj = np.repeat(np.arange(P), T).astype(np.float64)
    labels = np.stack([
        np.tile(t_age, P),
        np.where(np.repeat(sick, T) == 1, np.tile(t_dis, P), 0.0),   # NaN -> 0 (dataset_def.py:84)
        subj,
        np.repeat(gender, T).astype(np.float64),
        np.repeat(sick, T).astype(np.float64),
        np.repeat(loc, T).astype(np.float64),
    ], axis=1)
    return pixels, labels


def quantise5(px: np.ndarray) -> np.ndarray:
    """pixel 0..255 -> class index 0..4 (generate.py:58-66 maps to {25,...,225})."""
    return np.minimum(px // 50, 4).astype(np.int64)


def expand(raw: np.ndarray, spec: List[Tuple[str, int]]) -> np.ndarray:
    """[N, D] raw values (class indices for cat/ordinal) -> [N, X] expanded matrix:
    one-hot for cat (read_functions.py:77-81), thermometer for ordinal (:94-99)."""
    N = raw.shape[0]
    widths = [k if t in ("cat", "ordinal") else 1 for t, k in spec]
    off = np.concatenate([[0], np.cumsum(widths)])
    out = np.zeros((N, off[-1]))
    rows = np.arange(N)
    for d, (t, k) in enumerate(spec):
        if t == "cat":
            out[rows, off[d] + raw[:, d].astype(np.int64)] = 1.0
        elif t == "ordinal":
            cls = raw[:, d].astype(np.int64)
            out[:, off[d]:off[d] + k] = (np.arange(k)[None, :] <= cls[:, None]).astype(np.float64)
        else:
            out[:, off[d]] = raw[:, d]
    return out


def make_d4(n_subjects: int = 50, T: int = 20, missing: float = 0.25, seed: int = 100, expanded: bool = True):
    """D4 Het-HealthMNIST-shaped set: n_subjects*T rows x 1296 variables (324 real + 972 cat5).
    expanded = False: a ``CompactSource`` (one number per variable, no one-hot matrix) for the 50 k / 100 k-row sets."""
    rng = np.random.default_rng(seed)
    px, labels = _render_subjects(n_subjects, T, rng)
    spec = d4_type_spec()
    raw = px.copy()
    r1, r2, r3, r4 = region_index_sets()
    for r in (r2, r3, r4):
        raw[:, r] = quantise5(px[:, r].astype(np.int64))
    mask = (rng.random(raw.shape) >= missing).astype(np.float64)
    return _finish(raw, mask, labels, spec) if expanded else _finish_compact(raw, mask, labels, spec)


def make_tabular(n_rows: int = 4096, T: int = 16, missing: float = 0.25, seed: int = 100,
                 spec: List[Tuple[str, int]] | None = None, expanded: bool = True):
    """BASELINE.json config 4 mix: real~N(mu_d, s_d), pos~LogNormal, count~Poisson(+1 shift,
    read_functions.py:103-105), cat/ordinal~Uniform{0..K-1}.  Rows grouped in subjects of T."""
    rng = np.random.default_rng(seed)
    spec = tabular_type_spec() if spec is None else spec
    D = len(spec)
    raw = np.zeros((n_rows, D))
    for d, (t, k) in enumerate(spec):
        if t == "real":
            raw[:, d] = rng.normal(rng.uniform(-3, 3), rng.uniform(0.5, 4.0), n_rows)
        elif t == "pos":
            raw[:, d] = np.exp(rng.normal(rng.uniform(0, 2), rng.uniform(0.3, 1.0), n_rows))
        elif t == "count":
            raw[:, d] = rng.poisson(rng.uniform(1, 8), n_rows) + 1.0
        else:
            raw[:, d] = rng.integers(0, k, n_rows)
    mask = (rng.random(raw.shape) >= missing).astype(np.float64)
    P = (n_rows + T - 1) // T
    subj = np.repeat(np.arange(P), T)[:n_rows].astype(np.float64)
    t_age = np.tile(np.arange(T, dtype=np.float64), P)[:n_rows]
    sick = rng.binomial(1, 0.5, P)
    labels = np.stack([t_age, np.where(np.repeat(sick, T)[:n_rows] == 1, t_age - T // 2, 0.0), subj,
                       np.repeat(rng.binomial(1, 0.5, P), T)[:n_rows].astype(np.float64),
                       np.repeat(sick, T)[:n_rows].astype(np.float64),
                       np.repeat(rng.binomial(1, 0.5, P), T)[:n_rows].astype(np.float64)], axis=1)
    return _finish(raw, mask, labels, spec) if expanded else _finish_compact(raw, mask, labels, spec)


@dataclass
class CompactSource:
    """a synthetic data set in the compact form of datafeed.CompactDataset (raw value | class index | level - 1 per variable)"""
    raw: np.ndarray           # [N, D] float32
    mask: np.ndarray          # [N, D] uint8, 1 = observed
    labels: np.ndarray        # [N, Q] float64
    types_info: dict
    n_variables: int
    cov_dim_ext: int
    id_covariate: int = 2

    def __len__(self):
        return self.raw.shape[0]

    @property
    def n_subjects(self):
        return int(np.unique(self.labels[:, self.id_covariate]).size)

    def expand_rows(self, rows, spec=None):
        """(data fp64 [B, X], mask fp64 [B, D]) of a few rows in the reference's expanded form (CPU baseline, parity checks)"""
        td = self.types_info["types_dict"]
        spec = [(t["type"], int(t["nclass"])) for t in td]
        return expand(self.raw[rows].astype(np.float64), spec), self.mask[rows].astype(np.float64)


def _finish_compact(raw, mask, labels, spec) -> CompactSource:
    info = build_types_info(make_types_dict(spec), miss_mask=None)
    for t in info["types_dict"]:
        t["dim"] = int(t["dim"])
        t["nclass"] = int(t["nclass"])
    X = int(sum(k if t in ("cat", "ordinal") else 1 for t, k in spec))
    return CompactSource(raw=raw.astype(np.float32), mask=(mask != 0).astype(np.uint8), labels=labels, types_info=info,
                         n_variables=len(spec), cov_dim_ext=X)


def _finish(raw, mask, labels, spec) -> HetBatchSource:
    types_dict = make_types_dict(spec)
    info = build_types_info(types_dict, miss_mask=mask)
    for t in info["types_dict"]:           # dataset_def.py:31-33 casts these to int
        t["dim"] = int(t["dim"])
        t["nclass"] = int(t["nclass"])
    data = expand(raw, spec)
    return HetBatchSource(data=data, mask=mask, param_mask=info["param_miss_mask"], labels=labels,
                          types_info=info, n_variables=len(spec), cov_dim_ext=data.shape[1])


def subject_batches(labels: np.ndarray, subjects_per_batch: int, id_covariate: int = 2,
                    rng: np.random.Generator | None = None, rank: int = 0, world: int = 1):
    """Row-index batches made of WHOLE subjects (the semantics of the reference's
    VaryingLengthSubjectSampler / VaryingLengthBatchSampler, utils.py:53-97), optionally
    sharded over data-parallel ranks: every rank gets ``subjects_per_batch`` subjects of
    each global batch of ``world * subjects_per_batch`` subjects."""
    ids = labels[:, id_covariate]
    uniq, first = np.unique(ids, return_index=True)
    uniq = uniq[np.argsort(first)]
    order = np.arange(len(uniq))
    if rng is not None:
        rng.shuffle(order)
    srt = np.argsort(ids, kind="stable")
    su, st = np.unique(ids[srt], return_index=True)
    rows_of = dict(zip(su, np.split(srt, st[1:])))
    gsz = subjects_per_batch * world
    for g in range(0, len(order), gsz):
        grp = order[g:g + gsz]
        if len(grp) < world:             # every rank takes part in every global batch (the step contains collectives)
            break
        mine = grp[rank::world]          # round-robin: a short last batch still gives every rank a subject
        yield np.concatenate([rows_of[uniq[s]] for s in mine])
