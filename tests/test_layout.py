"""Column plan / types_info bookkeeping against the reference's read_data output
(fixture types_info_mix.npz, produced by reference HL_VAE/read_functions.py:13-203)."""
import os

import numpy as np
import pytest

from hlvae_amd import layout, synthetic
from tests_common import MIX_SPEC


def test_types_info_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "types_info_mix.npz"))
    src = synthetic.make_tabular(n_rows=24, T=6, seed=7, spec=MIX_SPEC)
    info = src.types_info
    assert ["%s:%s" % t for t in info["set_of_types"]] == list(g["set_of_types"])
    for k in ("data_types_indexes", "exp_types_indexes", "param_indexes", "param_miss_mask"):
        assert np.array_equal(np.asarray(info[k]), g[k]), k
    # the expanded data matrix (one-hot / thermometer / +1 count shift) as read_data builds it
    data = g["data"].copy()
    ours = synthetic.expand(_raw_from_fixture(g["raw"]), MIX_SPEC)
    assert np.allclose(ours, data, rtol=1e-9, atol=0)      # CSV text round trip (%.10g) on real columns
    disc = np.isin(np.asarray(info["exp_types_indexes"]), [0, 1, 3, 4])
    assert np.array_equal(ours[:, disc], data[:, disc])


def _raw_from_fixture(raw):
    raw = raw.copy()
    for j, (t, k) in enumerate(MIX_SPEC):
        if t == "count":
            raw[:, j] += 1           # read_functions.py:103-105
    return raw


def test_lexicographic_block_order():
    spec = [("cat", 10), ("cat", 5), ("real", 1), ("ordinal", 3)]
    info = layout.build_types_info(layout.make_types_dict(spec))
    assert info["set_of_types"] == [("cat", "10"), ("cat", "5"), ("ordinal", "3"), ("real", "1")]
    plan = layout.compile_plan(info, 5)
    assert plan.X == 10 + 5 + 1 + 3 and list(plan.xoff) == [0, 10, 15, 16]
    assert list(plan.blk) == [0, 1, 3, 2] and list(plan.sidx) == [-1, -1, 0, -1]


def test_d4_plan():
    src = synthetic.make_d4(n_subjects=1, T=2, seed=1)
    plan = layout.compile_plan(src.types_info, 5)
    assert (plan.D, plan.X, plan.n_real) == (1296, 5184, 324)
    assert src.types_info["set_of_types"] == [("cat", "5"), ("real", "1")]
    assert src.data.shape == (2, 5184) and src.param_mask.shape == (2, 5184)
    # every categorical variable is one-hot
    cat = plan.kind == layout.KIND_CAT
    sums = np.add.reduceat(src.data, plan.xoff, axis=1)
    assert np.all(sums[:, cat] == 1)


def test_plan_rejects_unsupported():
    with pytest.raises(ValueError):
        layout.make_types_dict([("beta", 1)])
    info = layout.build_types_info(layout.make_types_dict([("cat", 40)]))
    with pytest.raises(ValueError):
        layout.compile_plan(info, 5)


def test_subject_batches_shard_whole_subjects():
    src = synthetic.make_tabular(n_rows=96, T=6, seed=3, spec=MIX_SPEC)
    all_rows = []
    for rank in range(2):
        for rows in synthetic.subject_batches(src.labels, 4, rank=rank, world=2):
            subj = np.unique(src.labels[rows, 2])
            assert len(subj) <= 4
            for s in subj:      # whole subjects only
                assert np.sum(src.labels[rows, 2] == s) == np.sum(src.labels[:, 2] == s)
            all_rows.append(rows)
    cat = np.concatenate(all_rows)
    assert len(cat) == 96 and len(np.unique(cat)) == 96
