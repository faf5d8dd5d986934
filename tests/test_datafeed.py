"""Data feed, host side: the compact dataset round-trips the reference's expanded form, the
binary cache round-trips the dataset, the subject sampler keeps the semantics of the reference's
VaryingLengthSubjectSampler + VaryingLengthBatchSampler (utils.py:53-97) and shards by subject."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hlvae_amd                                    # noqa: E402
from hlvae_amd import synthetic                     # noqa: E402
from hlvae_amd.datafeed import CompactDataset, SubjectBatchSampler   # noqa: E402
from tests_common import MIX_SPEC                   # noqa: E402


def test_compact_roundtrip_and_cache(tmp_path):
    src = synthetic.make_tabular(n_rows=60, T=6, seed=3, spec=MIX_SPEC)
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate)
    assert ds.values.dtype == np.float32 and ds.mask.dtype == np.uint8 and ds.values.shape == (60, src.n_variables)
    rows = np.array([5, 0, 59, 17, 17])
    data, mask = ds.expand(rows)
    # discrete columns exactly; continuous ones to fp32 (the device kernels read the reference's fp64 as float as well)
    assert np.array_equal(mask, src.mask[rows])
    assert np.allclose(data, src.data[rows], rtol=1e-6, atol=0)
    plan = ds.plan()
    disc = np.isin(plan.kind, [3, 4])
    for d in np.nonzero(disc)[0]:
        xo, K = int(plan.xoff[d]), int(plan.ncls[d])
        assert np.array_equal(data[:, xo:xo + K], src.data[rows][:, xo:xo + K])
    ds.save(str(tmp_path / "cache"))
    ds2 = CompactDataset.load(str(tmp_path / "cache"))
    assert np.array_equal(ds2.values, ds.values) and np.array_equal(ds2.mask, ds.mask) and np.array_equal(ds2.labels, ds.labels)
    assert ds2.plan().X == plan.X and list(ds2.types_info["set_of_types"]) == list(ds.types_info["set_of_types"])
    assert isinstance(ds2.values, np.memmap)


def test_subject_sampler_semantics_and_sharding():
    # subjects with different numbers of rows, ids not sorted
    ids = np.array([7] * 3 + [2] * 5 + [9] * 1 + [4] * 4 + [1] * 2 + [8] * 6 + [3] * 2)
    s = SubjectBatchSampler(ids, subjects_per_batch=3, shuffle=True, seed=1)
    assert len(s) == 3 and s.P == 7
    seen = []
    for rows, P_b in s:
        subj = ids[rows]
        # whole subjects, their rows consecutive and in dataset order
        change = np.nonzero(np.diff(subj) != 0)[0] + 1
        groups = np.split(rows, change)
        assert len(groups) == P_b <= 3
        for gr in groups:
            sid = ids[gr[0]]
            assert np.array_equal(gr, np.nonzero(ids == sid)[0])
        seen += list(np.unique(subj))
    assert sorted(seen) == sorted(np.unique(ids)) and len(seen) == 7        # every subject exactly once per epoch
    # a second epoch is shuffled differently
    e1 = [tuple(r) for r, _ in s]
    e2 = [tuple(r) for r, _ in s]
    assert e1 != e2
    # no shuffle: order of first appearance, like the reference with the shuffle removed (utils.py:62-64)
    s0 = SubjectBatchSampler(ids, 3, shuffle=False)
    first = next(iter(s0))[0]
    assert list(ids[first]) == [7] * 3 + [2] * 5 + [9]
    # data parallel: two ranks cover each global batch disjointly, by whole subjects, same P_batch
    a = SubjectBatchSampler(ids, 4, shuffle=True, seed=5, rank=0, world=2)
    b = SubjectBatchSampler(ids, 4, shuffle=True, seed=5, rank=1, world=2)
    for (ra, pa), (rb, pb) in zip(a, b):
        assert pa == pb
        sa, sb = set(ids[ra]), set(ids[rb])
        assert not (sa & sb) and len(sa) + len(sb) == pa


def test_sampler_max_rows_covers_every_batch():
    """ELBOTrainer(max_batch=sampler.max_rows) must never need a larger workspace: the bound holds for every shuffle, every rank,
    ragged subject lengths and a folded tail."""
    from hlvae_amd.datafeed import SubjectBatchSampler
    rng = np.random.default_rng(3)
    lens = rng.integers(3, 21, size=23)                       # 23 subjects of 3..20 rows
    ids = np.repeat(np.arange(23), lens)
    for world in (1, 2, 4):
        for rank in range(world):
            sm = SubjectBatchSampler(ids, subjects_per_batch=8, shuffle=True, seed=5, rank=rank, world=world)
            seen = 0
            for _ in range(6):                                # six epochs, six shuffles
                for b in sm.batches():
                    seen = max(seen, len(b.rows))
                    assert len(b.rows) <= sm.max_rows
            assert seen > 0 and sm.max_rows <= int(np.sort(lens)[::-1][:-(-(8 + 7) // world)].sum())
