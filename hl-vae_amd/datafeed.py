"""Data feed of the ELBO step.

The reference keeps the expanded fp64 matrices in pandas frames, builds every batch row by row on the host
(dataset_def.py:67-92, ``iloc`` per row: 0.25 s per 400 rows, which would cap a GPU at ~1.6 k rows/s) and draws whole
subjects per batch with Python samplers written against an older torch (utils.py:36-97).  Here:

* ``CompactDataset``: the ``read_data`` output (HL_VAE/read_functions.py:13-203: one-hot / thermometer expanded fp64,
  NaN -> 0, observation mask) folded into 5 bytes per entry -- ``values`` fp32 [N, D] (raw value | class index | level - 1),
  ``mask`` u8 [N, D] -- plus the covariates ``labels`` [N, Q]; a binary columnar cache on disk (one ``.npy`` per column
  block, memory-mapped on load) and ONE resident copy in HBM (D4, 100 k rows: 0.65 GB of the 288 GB).
* ``SubjectBatchSampler``: whole subjects per batch, shuffled per epoch, consecutive rows of a subject kept together
  (the semantics of VaryingLengthSubjectSampler + VaryingLengthBatchSampler, utils.py:53-97), sharded over ranks by
  subject for data parallelism.  A batch is a vector of row indices; the gather happens on the device inside the input
  stage (csrc/feed.hip), so a training step moves 4 bytes per row across PCIe.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from .layout import KIND_CAT, KIND_ORDINAL, ColumnPlan, compile_plan


@dataclass
class CompactDataset:
    values: np.ndarray           # fp32 [N, D]
    mask: np.ndarray             # u8   [N, D]
    labels: np.ndarray           # fp64 [N, Q]
    types_info: dict
    id_covariate: int = 2

    # ---- construction ------------------------------------------------------------------------------------------
    @classmethod
    def from_expanded(cls, data: np.ndarray, mask: np.ndarray, labels: np.ndarray, types_info: dict, id_covariate: int = 2,
                      y_dim: int = 5) -> "CompactDataset":
        """data [N, X]: the reference's expanded matrix (read_functions.py:67-105); mask [N, D] 0/1."""
        plan = compile_plan(types_info, y_dim)
        N = data.shape[0]
        vals = np.empty((N, plan.D), dtype=np.float32)
        for d in range(plan.D):
            xo, K, kind = int(plan.xoff[d]), int(plan.ncls[d]), int(plan.kind[d])
            if kind == KIND_CAT:                               # argmax of the one-hot row, -1 when it is all zero
                blk = data[:, xo:xo + K]
                vals[:, d] = np.where(blk.max(1) > 0, blk.argmax(1), -1)
            elif kind == KIND_ORDINAL:                         # thermometer -> level - 1 (loglik.py:172)
                vals[:, d] = data[:, xo:xo + K].astype(np.int64).sum(1) - 1
            else:
                vals[:, d] = data[:, xo]
        return cls(vals, np.ascontiguousarray(mask != 0).astype(np.uint8), np.asarray(labels, dtype=np.float64), types_info,
                   id_covariate)

    @classmethod
    def from_raw(cls, raw: np.ndarray, mask: np.ndarray, labels: np.ndarray, types_info: dict, id_covariate: int = 2) -> "CompactDataset":
        """raw [N, D]: one number per variable -- the value (real / pos / count), the class index (cat, -1 = none) or
        level - 1 (ordinal): what ``from_expanded`` folds the one-hot / thermometer columns into.  For data sets that are
        never materialised in the expanded fp64 form (100 k rows of D4 = 4.1 GB of mostly zeros)."""
        return cls(np.ascontiguousarray(raw, dtype=np.float32), np.ascontiguousarray(np.asarray(mask) != 0).astype(np.uint8),
                   np.asarray(labels, dtype=np.float64), types_info, id_covariate)

    def __len__(self):
        return self.values.shape[0]

    def plan(self, y_dim: int = 5) -> ColumnPlan:
        return compile_plan(self.types_info, y_dim)

    def expand(self, rows, y_dim: int = 5) -> Tuple[np.ndarray, np.ndarray]:
        """(data fp64 [B, X], mask fp64 [B, D]) of the given rows in the reference's expanded form."""
        plan = self.plan(y_dim)
        rows = np.asarray(rows)
        out = np.zeros((len(rows), plan.X), dtype=np.float64)
        v = self.values[rows]
        for d in range(plan.D):
            xo, K, kind = int(plan.xoff[d]), int(plan.ncls[d]), int(plan.kind[d])
            if kind == KIND_CAT:
                c = v[:, d].astype(np.int64)
                ok = c >= 0
                out[np.nonzero(ok)[0], xo + c[ok]] = 1.0
            elif kind == KIND_ORDINAL:
                c = v[:, d].astype(np.int64)
                out[:, xo:xo + K] = (np.arange(K)[None, :] <= c[:, None]).astype(np.float64)
            else:
                out[:, xo] = v[:, d]
        return out, self.mask[rows].astype(np.float64)

    # ---- binary columnar cache -----------------------------------------------------------------------------------
    def save(self, path: str):
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "values.npy"), self.values)
        np.save(os.path.join(path, "mask.npy"), self.mask)
        np.save(os.path.join(path, "labels.npy"), self.labels)
        ti = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in self.types_info.items()
              if k in ("types_dict", "set_of_types", "data_types_indexes", "exp_types_indexes", "param_indexes", "conv")}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump({"types_info": ti, "id_covariate": self.id_covariate}, f)

    @classmethod
    def load(cls, path: str, mmap: bool = True) -> "CompactDataset":
        mode = "r" if mmap else None
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        ti = meta["types_info"]
        ti["set_of_types"] = [tuple(t) for t in ti["set_of_types"]]
        for k in ("data_types_indexes", "exp_types_indexes", "param_indexes"):
            ti[k] = np.asarray(ti[k])
        return cls(np.load(os.path.join(path, "values.npy"), mmap_mode=mode), np.load(os.path.join(path, "mask.npy"), mmap_mode=mode),
                   np.load(os.path.join(path, "labels.npy"), mmap_mode=mode), ti, meta["id_covariate"])

    def to(self, device) -> "DeviceDataset":
        return DeviceDataset(torch.as_tensor(np.ascontiguousarray(self.values), device=device),
                             torch.as_tensor(np.ascontiguousarray(self.mask), device=device),
                             torch.as_tensor(np.ascontiguousarray(self.labels), device=device), self)


@dataclass
class DeviceDataset:
    values: torch.Tensor         # fp32 [N, D], resident in HBM
    mask: torch.Tensor           # u8   [N, D]
    labels: torch.Tensor         # fp64 [N, Q]
    host: CompactDataset


def subject_index(ids: np.ndarray) -> np.ndarray:
    """Subject structure of a batch for the GP prior (reference elbo_functions.py:242-252 walks ``torch.unique`` of the id
    covariate): int32 [S, Tmax], row s = positions INSIDE THE BATCH of the rows of the s-th subject (sorted by id, as
    torch.unique returns them), -1 = padding.  Host arithmetic on the sampler's side: no device round trip per step."""
    ids = np.asarray(ids)
    order = np.argsort(ids, kind="stable")
    uniq, start, count = np.unique(ids[order], return_index=True, return_counts=True)
    idx = -np.ones((len(uniq), int(count.max())), dtype=np.int32)
    for s, (a, c) in enumerate(zip(start, count)):
        idx[s, :c] = order[a:a + c]
    return idx


@dataclass
class Batch:
    """one training batch as the sampler hands it over: dataset rows, subjects of the GLOBAL batch, subject structure"""
    rows: np.ndarray             # int32 [B] dataset rows of THIS rank
    P_batch: int                 # subjects in the global batch (loss scale P / P_batch, training.py:121-122)
    groups: np.ndarray           # int32 [S, Tmax] (subject_index of this rank's rows)

    def to(self, device):
        return (torch.as_tensor(self.rows, device=device), self.P_batch, torch.as_tensor(self.groups, device=device))


class SubjectBatchSampler:
    """Batches of WHOLE subjects (the GP prior needs each subject's T x T block, elbo_functions.py:243-252).

    Every epoch: subjects in random order, the rows of a subject consecutive and in dataset order, ``subjects_per_batch``
    subjects per batch, last batch smaller (utils.py:53-97).  Data parallel: rank r of ``world`` takes subjects
    r, r + world, ... of every global batch; ``P_batch`` is the number of subjects of the GLOBAL batch (the loss scale
    P / P_batch, training.py:121-122).  Every rank must take part in every global batch (the step contains collectives): a
    last batch with fewer subjects than ranks is folded into the batch before it, identically on every rank."""

    def __init__(self, subject_ids: np.ndarray, subjects_per_batch: int, shuffle: bool = True, seed: int = 0, rank: int = 0,
                 world: int = 1, min_last: Optional[int] = None):
        ids = np.asarray(subject_ids)
        self.min_last = world if min_last is None else min_last      # smallest last batch that stays a batch of its own
        if subjects_per_batch < world:
            raise ValueError(f"subjects_per_batch={subjects_per_batch} < world={world}: a rank would get no subject")
        order = np.argsort(ids, kind="stable")                  # one sort instead of one scan per subject
        uniq, start = np.unique(ids[order], return_index=True)
        rows_sorted = np.split(order, start[1:])
        first = np.array([r[0] for r in rows_sorted])           # subjects in order of first appearance (utils.py:62-64)
        by_first = np.argsort(first, kind="stable")
        self.subjects = uniq[by_first]
        self.rows_of = [rows_sorted[i] for i in by_first]
        self.subject_ids = ids
        self.P = len(self.subjects)
        if self.P < world:
            raise ValueError(f"{self.P} subjects for {world} ranks")
        self.subjects_per_batch, self.shuffle, self.rank, self.world = subjects_per_batch, shuffle, rank, world
        self.rng = np.random.default_rng(seed)                  # same seed on every rank: identical global batches

    def _bounds(self):
        lo = list(range(0, self.P, self.subjects_per_batch))
        hi = lo[1:] + [self.P]
        if len(lo) > 1 and hi[-1] - lo[-1] < self.min_last:     # short tail: one larger last batch instead
            lo.pop()
            hi.pop(-2)
        return list(zip(lo, hi))

    def __len__(self):
        return len(self._bounds())

    @property
    def max_rows(self) -> int:
        """upper bound of the rows any batch of THIS rank can have (whatever the shuffle): the largest number of subjects a batch
        gives this rank (a folded tail included) times the longest subjects -- what ELBOTrainer(max_batch=...) must cover"""
        n_sub = max(-(-(hi - lo) // self.world) for lo, hi in self._bounds())
        lens = sorted((len(r) for r in self.rows_of), reverse=True)
        return int(sum(lens[:n_sub]))

    def batches(self) -> Iterator[Batch]:
        r = np.arange(self.P)
        if self.shuffle:
            self.rng.shuffle(r)
        for lo, hi in self._bounds():
            batch = r[lo:hi]
            mine = batch[self.rank::self.world]
            rows = np.concatenate([self.rows_of[s] for s in mine]).astype(np.int32)
            yield Batch(rows, len(batch), subject_index(self.subject_ids[rows]))

    def __iter__(self) -> Iterator[Tuple[np.ndarray, int]]:
        for b in self.batches():
            yield b.rows, b.P_batch
