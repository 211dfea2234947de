"""Evaluation with sampled negatives (the `negative_sampling:` block of the YAML): candidate sets in CSR form.

Mirrors elliot/negative_sampling/negative_sampling.py:22-121 and its use in elliot/dataset/dataset.py:221-243:
  strategy "random", num_items N   per user N items drawn without replacement from the items in neither train nor test
                                   (`random.sample(range(n_candidates), N)` on the `random` module's stream seeded with 42 at
                                   import, candidates ascending), written to `file_path` as `(user,)<TAB>item<TAB>...`;
  strategy "fixed", files [test, validation]   the same file format read back.
  candidate set of a split = its negatives + its own held-out items (`val_mask` / `test_mask`).
The reference builds `candidate_negatives = ((i_test + i_train).astype('bool') != True)` -- a U x I matrix with U*I - nnz
stored entries -- and dense masks; here the draws run in C on the exclusion CSR (`el_host_negative_sample`, the same stream)
and the result stays CSR (`DataSet.val_cand_csr` / `test_cand_csr`, what recommender/masks.py uploads).
Quirk kept: the reference samples the VALIDATION negatives against the TEST items too (negative_sampling.py:29-31 passes `test`).
"""
import ctypes
import math
import random
from ast import literal_eval

import numpy as np
import scipy.sparse as sp


def _known_split_csr(data, validation):
    """Held-out items the model has a row for, as a bool CSR [U, I] (dataset.py:to_bool_sparse semantics for known ids)."""
    ip, cols, _ = data.split_csr(validation)
    keep = cols < data.num_items
    rows = np.repeat(np.arange(data.num_users, dtype=np.int64), np.diff(ip))[keep]
    m = sp.csr_matrix((np.ones(int(keep.sum()), dtype=np.int8), (rows, cols[keep].astype(np.int64))),
                      shape=(data.num_users, data.num_items))
    m.sum_duplicates()
    m.sort_indices()
    return m


def sample_by_random_uniform(excl, num_items, rng):
    """negative_sampling.py:95-105 on the exclusion CSR `excl` (train + test, bool) with the stream of `rng`
    (a random.Random): returns an int32 [U, num_items] array of private item ids in sample order."""
    from .. import _lib
    excl = excl.tocsr()
    excl.sort_indices()
    U, I = excl.shape
    setsize = 21
    if num_items > 5:
        setsize += 4 ** math.ceil(math.log(num_items * 3, 4))          # random.sample's own expression
    version, state, gauss = rng.getstate()
    st = np.array(state, dtype=np.uint32)
    out = np.empty((U, num_items), dtype=np.int32)
    indptr = np.ascontiguousarray(excl.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(excl.indices, dtype=np.int32)
    try:
        _lib.check(_lib.load().el_host_negative_sample(indptr.ctypes.data, indices.ctypes.data, U, I, int(num_items), int(setsize),
                                                       st.ctypes.data, out.ctypes.data), "el_host_negative_sample")
    except _lib.ElliotHipError as e:
        raise ValueError("Sample larger than population or is negative") from e       # what random.sample raises
    rng.setstate((version, tuple(int(x) for x in st), gauss))
    return out


def _write_negatives(path, data, neg):
    """negative_sampling.py:66-76: one line per user, columns in ascending private id."""
    pu, pi = data.private_users, np.asarray(data.items, dtype=object)
    with open(path, "w") as f:
        for u in range(neg.shape[0]):
            f.write(str((pu[u],)) + "\t" + "\t".join(map(str, pi[np.sort(neg[u])].tolist())) + "\n")


def read_from_files(data, path):
    """negative_sampling.py:107-121."""
    rows, cols = [], []
    pub_u, pub_i = data.public_users, data.public_items
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            u = pub_u[int(literal_eval(parts[0])[0])]
            its = {pub_i[int(i)] for i in parts[1:] if int(i) in pub_i}
            rows.extend([u] * len(its))
            cols.extend(its)
    m = sp.csr_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(data.num_users, data.num_items))
    m.sum_duplicates()
    return m


def attach(data, ns, seed=42, rng=None):
    """Equivalent of dataset.py:221-243: sets `data.test_cand_csr` (and `val_cand_csr` when the data set has a validation
    split) = (indptr int64, indices int32) of negatives + held-out items.  `ns` = the `negative_sampling` namespace / dict.
    rng: the run's `random.Random` -- the reference seeds the `random` MODULE once, at the import of negative_sampling.py (:16),
    and every data object of a run (test folds x validation folds) continues that one stream; a caller with several data objects
    creates ONE random.Random(42) and hands it to every attach() (run.load_data_objects does).  None = a fresh stream (one object)."""
    get = (lambda k, d=None: ns.get(k, d)) if isinstance(ns, dict) else (lambda k, d=None: getattr(ns, k, d))
    strategy = get("strategy")
    test_pos = _known_split_csr(data, False)
    has_val = data.split_csr(True) is not None
    rng = rng if rng is not None else random.Random(seed)
    train = data.sp_i_train.astype(np.int8)

    def negatives(validation):
        if strategy == "random":
            num = get("num_items")
            if num is None:
                raise Exception("Number of negative items option is missing")
            if not str(num).isdigit():
                raise Exception("Number of negative items value not recognized")
            neg = sample_by_random_uniform((train + test_pos).astype(bool), int(num), rng)
            if get("file_path"):
                _write_negatives(get("file_path"), data, neg)
            rows = np.repeat(np.arange(data.num_users, dtype=np.int64), neg.shape[1])
            return sp.csr_matrix((np.ones(neg.size, dtype=np.int8), (rows, neg.reshape(-1).astype(np.int64))),
                                 shape=(data.num_users, data.num_items))
        if strategy == "fixed":
            files = get("files")
            files = files if isinstance(files, list) else [files]
            return read_from_files(data, files[1] if validation else files[0])
        raise Exception("Missing strategy")

    def as_csr(m):
        m = m.tocsr()
        m.sum_duplicates()
        m.sort_indices()
        return np.ascontiguousarray(m.indptr, dtype=np.int64), np.ascontiguousarray(m.indices, dtype=np.int32)

    val_neg = negatives(True) if has_val else None
    test_neg = negatives(False)
    data.test_cand_csr = as_csr(test_neg + test_pos)
    if has_val:
        data.val_cand_csr = as_csr(val_neg + _known_split_csr(data, True))
    data._elliot_amd_masks = None
    return data
