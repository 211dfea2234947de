"""SURVEY 8f N2 -- the array data plane: split flags and private item order from C (el_host_*), held-out CSR without dicts.
Everything here runs without a GPU (the el_host_* entry points are plain CPU code inside libelliot_hip.so)."""
import math

import numpy as np
import pytest

from elliot_amd.dataset import dataset as D
from elliot_amd.evaluation.evaluator import Evaluator


def _py_split(users, mode, param, seed):
    """The reference's loop (base_splitter.py:256-281) with the interpreter's own legacy shuffle."""
    users = np.asarray(users)
    rs = np.random.RandomState(seed)
    flags = np.zeros(users.shape[0], dtype=np.int8)
    order = np.argsort(users, kind="stable")
    su = users[order]
    bounds = np.flatnonzero(np.concatenate([[True], su[1:] != su[:-1], [True]]))
    for a, b in zip(bounds[:-1], bounds[1:]):
        n = b - a
        ntrain = int(math.floor(n * (1 - param))) if mode == 0 else n - int(param)
        lst = [0] * ntrain + [1] * (n - ntrain)
        rs.shuffle(lst)
        flags[order[a:b]] = lst
    return flags


@pytest.mark.parametrize("n_users,n_rows,ratio,seed", [(1, 1, 0.2, 42), (7, 40, 0.2, 42), (300, 20000, 0.2, 42), (2500, 150000, 0.35, 7),
                                                       (50, 3000, 0.0, 1), (50, 3000, 1.0, 1)])
def test_split_flags_equal_the_legacy_numpy_stream(n_users, n_rows, ratio, seed):
    rs = np.random.RandomState(n_rows)
    users = rs.randint(0, n_users, size=n_rows) * 3 + 11
    assert np.array_equal(D.random_subsampling(users, ratio, seed), _py_split(users, 0, ratio, seed))


def test_leave_n_out_flags():
    rs = np.random.RandomState(5)
    users = rs.randint(0, 400, size=30000)
    got = D.leave_n_out(users, 2, seed=42)
    assert np.array_equal(got, _py_split(users, 1, 2, 42))
    assert (np.bincount(users, weights=got) == 2).all()


def test_split_flags_empty_and_single_rows():
    assert D.random_subsampling(np.zeros(0, dtype=np.int64), 0.2).shape == (0,)
    assert D.random_subsampling(np.array([5, 6, 7]), 0.2).tolist() == [1, 1, 1]      # floor(1 * 0.8) = 0 train rows (reference behaviour)


@pytest.mark.parametrize("n,hi", [(0, 10), (1, 1), (9, 8), (100, 50), (1000, 5000), (6000, 1 << 20), (70000, 100000), (260000, 10 ** 7),
                                  (50000, 1 << 45)])
def test_private_item_order_is_the_cpython_set_order(n, hi):
    rs = np.random.RandomState(n + 1)
    keys = rs.randint(0, hi, size=n).astype(np.int64)
    assert D.pyset_order(keys).tolist() == list({int(k) for k in keys.tolist()})


def test_pyset_order_sequential_and_clustered_ids():
    for keys in (np.arange(200000), np.arange(0, 3_000_000, 17), np.repeat(np.arange(5000), 3)[::-1].copy(),
                 (np.arange(40000) * 1024) % 1_000_003):
        assert D.pyset_order(keys).tolist() == list({int(k) for k in keys.tolist()})


def test_pyset_order_other_key_types_use_the_interpreter():
    assert D.pyset_order(np.array([-3, 5, -3, 7])).tolist() == list({k for k in [-3, 5, -3, 7]})
    assert sorted(D.pyset_order(np.array(["b", "a", "b"])).tolist()) == ["a", "b"]


def _triples(rs, n_users, n_items, n):
    return rs.randint(0, n_users, size=n) * 2 + 1, rs.randint(0, n_items, size=n) * 5, rs.randint(1, 6, size=n).astype(np.float64)


def test_dataset_private_ids_follow_the_reference_construction():
    """users: first appearance in train (dataset.py:248); items: set order of the user-major item stream (:202)."""
    rs = np.random.RandomState(3)
    tu, ti, tr = _triples(rs, 300, 800, 20000)
    ds = D.DataSet(D.default_config(), (tu, ti, tr), _triples(rs, 300, 800, 500))
    users = list(dict.fromkeys(tu.tolist()))
    train_dict = {u: {} for u in users}
    for u, i, r in zip(tu.tolist(), ti.tolist(), tr.tolist()):
        train_dict[u][i] = r
    items = list({k for a in train_dict.values() for k in a.keys()})
    assert ds.users == users and ds.items == items
    assert ds.transactions == sum(len(a) for a in train_dict.values())


def test_split_csr_equals_the_dict_route():
    rs = np.random.RandomState(8)
    tu, ti, tr = _triples(rs, 200, 300, 9000)
    # held-out rows: known and unknown users / items, duplicated pairs (the last rating wins in a dict)
    eu, ei, er = _triples(rs, 230, 340, 2500)
    eu = np.concatenate([eu, eu[:40]])
    ei = np.concatenate([ei, ei[:40]])
    er = np.concatenate([er, er[:40] + 1.0])
    ds = D.DataSet(D.default_config(), (tu, ti, tr), (eu, ei, er), (eu[::3], ei[::3], er[::3]))
    for validation in (False, True):
        indptr, cols, vals = ds.split_csr(validation)
        split = ds.val_dict if validation else ds.test_dict

        class _Ops:                                   # the dict route of the evaluator, captured instead of uploaded
            @staticmethod
            def DeviceTestSet(ip, c, v, device):
                return ip, c, v
        ip2, c2, v2 = Evaluator._split_to_csr(_Ops, ds, split, None)
        assert np.array_equal(indptr, ip2)
        for u in range(ds.num_users):
            a, b = slice(indptr[u], indptr[u + 1]), slice(ip2[u], ip2[u + 1])
            known = cols[a] < ds.num_items
            assert np.array_equal(cols[a][known], c2[b][c2[b] < ds.num_items])
            assert np.array_equal(vals[a][known], v2[b][c2[b] < ds.num_items])
            assert sorted(vals[a][~known].tolist()) == sorted(v2[b][c2[b] >= ds.num_items].tolist())
            assert len(set(cols[a].tolist())) == cols[a].shape[0]


def test_evaluator_builds_no_dicts_until_asked():
    rs = np.random.RandomState(2)
    ds = D.DataSet(D.default_config(), _triples(rs, 50, 80, 2000), _triples(rs, 50, 80, 300))
    ev = Evaluator(ds, None)
    assert ev._dict_splits is None and "test" not in ds._cache
    recs = {u: [(i, 1.0) for i in list(ds.items)[:10]] for u in ds.users}
    out = ev.eval(({}, recs))
    assert ev._dict_splits is not None and out[10]["test_results"]["nDCG"] >= 0.0


# ---- evaluation with sampled negatives (negative_sampling/negative_sampling.py) ---------------------------------------
import os                                     # noqa: E402
import random                                 # noqa: E402

import scipy.sparse as sp                     # noqa: E402

from elliot_amd.dataset import negative_sampling as NS     # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "negative_sampling_ref.npz")


@pytest.mark.parametrize("tag,num", [("A", 99), ("A", 5), ("A", 300), ("B", 99)])
def test_negative_sampling_equals_the_reference_sampler(tag, num, tmp_path):
    """Fixture = the reference's own NegativeSampler.process_sampling run twice in a row (validation, then test: the order of
    NegativeSampler.sample) from random.seed(42), oracle/gen_golden.py::gen_negative_sampling; pool and selection-set branches
    of random.sample."""
    g = np.load(GOLD)
    U, I = g[f"{tag}_shape"].tolist()
    train = sp.csr_matrix((np.ones(g[f"{tag}_train_indices"].shape[0], dtype=np.int8), g[f"{tag}_train_indices"], g[f"{tag}_train_indptr"]),
                          shape=(U, I))
    tu, ti = g[f"{tag}_test_users"], g[f"{tag}_test_items"]
    known = (tu < U) & (ti < I)
    test = sp.csr_matrix((np.ones(int(known.sum()), dtype=np.int8), (tu[known], ti[known])), shape=(U, I))
    excl = (train + test).astype(bool)
    rng = random.Random(42)
    for name in ("first", "second"):                       # one stream across both calls
        neg = NS.sample_by_random_uniform(excl, num, rng)
        assert neg.shape == (U, num)
        assert np.array_equal(np.sort(neg, axis=1).reshape(-1), g[f"{tag}_n{num}_{name}_indices"])
        assert not excl[np.repeat(np.arange(U), num), neg.reshape(-1)].any()
    # the negatives file as the reference leaves it (written by the second call), byte for byte
    from types import SimpleNamespace
    NS._write_negatives(str(tmp_path / "n.tsv"), SimpleNamespace(private_users={u: u for u in range(U)}, items=list(range(I))), neg)
    assert (tmp_path / "n.tsv").read_bytes() == g[f"{tag}_n{num}_file"].tobytes()
    # the Python module's state moved exactly as far as the reference's two calls moved it
    ref = random.Random(42)
    for _ in range(2):
        for u in range(U):
            ref.sample(range(I - excl[u].nnz), num)
    assert rng.getstate() == ref.getstate()


def test_negative_sampling_attach_builds_candidate_csrs_and_the_file(tmp_path):
    g = np.load(GOLD)
    U, I = g["A_shape"].tolist()
    ip, ix = g["A_train_indptr"], g["A_train_indices"]
    tr_u = np.repeat(np.arange(U), np.diff(ip))
    # public ids == private ids: fix both orders
    ds = D.DataSet(D.default_config(), (tr_u, ix.astype(np.int64), np.ones(ix.shape[0])),
                   (g["A_test_users"], g["A_test_items"], np.ones(g["A_test_users"].shape[0])),
                   public_users=np.arange(U), public_items=np.arange(I))
    path = tmp_path / "neg.tsv"
    NS.attach(ds, {"strategy": "random", "num_items": 99, "file_path": str(path)})
    # no validation split: ONE sampling call, = the fixture's first call; candidates = negatives + the known test items
    cip, cix = ds.test_cand_csr
    tip, tcols, _ = ds.split_csr(False)
    neg = g["A_n99_first_indices"].reshape(U, 99)
    for u in range(U):
        own = tcols[tip[u]:tip[u + 1]]
        assert sorted(set(neg[u].tolist()) | set(own[own < I].tolist())) == cix[cip[u]:cip[u + 1]].tolist()
    assert not hasattr(ds, "val_cand_csr")
    lines = path.read_text().splitlines()
    assert len(lines) == U and lines[3].split("\t")[0] == "(3,)" and [int(x) for x in lines[3].split("\t")[1:]] == neg[3].tolist()
    # strategy "fixed" reads the same file back
    ds2 = D.DataSet(D.default_config(), (tr_u, ix.astype(np.int64), np.ones(ix.shape[0])),
                    (g["A_test_users"], g["A_test_items"], np.ones(g["A_test_users"].shape[0])),
                    public_users=np.arange(U), public_items=np.arange(I))
    NS.attach(ds2, {"strategy": "fixed", "files": [str(path)]})
    assert np.array_equal(ds2.test_cand_csr[0], cip) and np.array_equal(ds2.test_cand_csr[1], cix)


def test_negative_sampling_more_negatives_than_candidates_raises_like_random_sample():
    excl = sp.csr_matrix(np.ones((2, 10), dtype=bool))
    excl[1, 3] = False
    excl.eliminate_zeros()
    with pytest.raises(ValueError):
        NS.sample_by_random_uniform(excl, 2, random.Random(42))


# ---- property tests (hypothesis): the C loops against the interpreter's own set / shuffle / sample on arbitrary inputs ---------
from hypothesis import given, settings, strategies as hst       # noqa: E402


@settings(max_examples=120, deadline=None)
@given(hst.lists(hst.integers(min_value=0, max_value=(1 << 61) - 2), max_size=400))
def test_pyset_order_property(keys):
    assert D.pyset_order(np.array(keys, dtype=np.int64)).tolist() == list({k for k in keys})


@settings(max_examples=60, deadline=None)
@given(hst.lists(hst.integers(min_value=0, max_value=2000), max_size=3000))
def test_pyset_order_property_dense_small_ids(keys):
    """many duplicates and collisions of the low bits: resizes at 5 / 19 / 76 / ... entries, linear probes, perturbed probes"""
    assert D.pyset_order(np.array(keys, dtype=np.int64)).tolist() == list({k for k in keys})


@settings(max_examples=60, deadline=None)
@given(hst.lists(hst.integers(min_value=0, max_value=40), min_size=1, max_size=600), hst.floats(min_value=0.0, max_value=1.0),
       hst.integers(min_value=0, max_value=2 ** 32 - 1), hst.integers(min_value=1, max_value=3))
def test_split_flags_property(users, ratio, seed, folds):
    users = np.array(users)
    got = D.random_subsampling(users, ratio, seed, folds=folds)
    got = got[None] if folds == 1 else got
    rs = np.random.RandomState(seed)
    order = np.argsort(users, kind="stable")
    su = users[order]
    bounds = np.flatnonzero(np.concatenate([[True], su[1:] != su[:-1], [True]]))
    for f in range(folds):
        exp = np.zeros(users.shape[0], dtype=np.int8)
        for a, b in zip(bounds[:-1], bounds[1:]):
            n = b - a
            ntrain = int(math.floor(n * (1 - ratio)))
            lst = [0] * ntrain + [1] * (n - ntrain)
            rs.shuffle(lst)
            exp[order[a:b]] = lst
        assert np.array_equal(got[f], exp)


@settings(max_examples=40, deadline=None)
@given(hst.integers(min_value=1, max_value=12), hst.integers(min_value=30, max_value=1500), hst.integers(min_value=0, max_value=25),
       hst.integers(min_value=0, max_value=2 ** 31 - 1))
def test_negative_sampling_property(n_users, n_items, num, seed):
    """el_host_negative_sample == random.sample(range(n_candidates), num) mapped onto the ascending candidates, user after user on
    one stream -- both branches of random.sample (pool: n <= setsize; selection set otherwise)."""
    rs = np.random.RandomState(seed % (2 ** 31))
    dense = rs.uniform(size=(n_users, n_items)) < 0.3
    excl = sp.csr_matrix(dense)
    rng = random.Random(seed)
    ref = random.Random(seed)
    if any(n_items - int(dense[u].sum()) < num for u in range(n_users)):
        with pytest.raises(ValueError):
            NS.sample_by_random_uniform(excl, num, rng)
        return
    got = NS.sample_by_random_uniform(excl, num, rng)
    for u in range(n_users):
        cand = np.flatnonzero(~dense[u])
        assert got[u].tolist() == cand[ref.sample(range(cand.shape[0]), num)].tolist()
    assert rng.getstate() == ref.getstate()


def test_negative_sampling_stream_runs_on_over_the_data_objects_of_a_run():
    """The reference seeds the `random` MODULE once (negative_sampling.py:16); the data objects of a run (test folds x validation
    folds, run.py loops over all of them) draw their negatives one after the other from that stream.  Two data objects with one
    shared random.Random(42): the second one's negatives are the continuation of the stream -- what a single generator drawing
    for both exclusion matrices in turn produces --, not a replay of the first object's draws."""
    g = np.load(GOLD)
    U, I = g["A_shape"].tolist()
    ip, ix = g["A_train_indptr"], g["A_train_indices"]
    tr_u = np.repeat(np.arange(U), np.diff(ip))
    objs = []
    for drop in (0, 1):                                    # two "folds": the second loses every 7th train row
        keep = np.ones(ix.shape[0], bool)
        if drop:
            keep[::7] = False
            keep[ip[:-1]] = True                           # (every user keeps a train item)
        objs.append(D.DataSet(D.default_config(), (tr_u[keep], ix[keep].astype(np.int64), np.ones(int(keep.sum()))),
                              (g["A_test_users"], g["A_test_items"], np.ones(g["A_test_users"].shape[0])),
                              public_users=np.arange(U), public_items=np.arange(I)))
    rng = random.Random(42)
    for ds in objs:
        NS.attach(ds, {"strategy": "random", "num_items": 20}, rng=rng)
    ref = random.Random(42)
    for ds in objs:
        excl = (ds.sp_i_train.astype(np.int8) + NS._known_split_csr(ds, False)).astype(bool)
        neg = NS.sample_by_random_uniform(excl, 20, ref)
        cip, cix = ds.test_cand_csr
        for u in range(0, U, 17):
            held = NS._known_split_csr(ds, False)[u].indices
            assert set(cix[cip[u]:cip[u + 1]].tolist()) == set(neg[u].tolist()) | set(held.tolist())
    assert rng.getstate() == ref.getstate()
    # a fresh generator per object (the round-2 behaviour) gives the second object other negatives
    again = D.DataSet(D.default_config(), objs[1]._train_triples, objs[1]._test_triples, public_users=np.arange(U), public_items=np.arange(I))
    NS.attach(again, {"strategy": "random", "num_items": 20})
    assert not np.array_equal(again.test_cand_csr[1], objs[1].test_cand_csr[1])
