"""Data objects the latent-factor plugins consume.

`DataSet` exposes the attributes the reference's DataSet gives its models (elliot/dataset/dataset.py:199-245):
num_users/num_items/transactions, private_/public_ id maps, sp_i_train (CSR fp32), test_dict/val_dict, config,
get_test()/get_validation() -- built from index arrays in vectorised NumPy instead of per-user pandas filters
(`dataframe_to_dict`, dataset.py:247-255, is O(U*T)).  The dict-of-dict views (`train_dict`, `i_train_dict`) and
the dense `allunrated_mask` are materialised lazily and only on request: the kernels never need them, and the
stand-alone evaluator's device path takes the held-out splits as CSR arrays (`split_csr`) without the dicts.

The two loops NumPy cannot vectorise run in C (include/elliot_hip.h, "host-side data plane"; no GPU involved):
the per-user shuffles of the splitter (`el_host_split_flags`) and the CPython-set iteration order that defines the
reference's private item ids (`el_host_pyset_order`) -- both bit-identical to the reference at any size
(tests/test_host_dataplane.py).

`load_tsv_dataset` covers the slice of the reference loader the hello-world experiment uses
(config_files/sample_hello_world.yml:3-9): a `user<TAB>item<TAB>rating[<TAB>timestamp]` file plus per-user
random subsampling (splitter/base_splitter.py:256-274, seeded np.random legacy stream, users in groupby order).
"""
import math
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp


class DataSet:
    def __init__(self, config, train, test, val=None, public_users=None, public_items=None):
        """train/test/val: (user, item, rating) triples of *public* ids as three aligned arrays.

        Private ids follow the reference: users in first-appearance order of the train rows (dataset.py:248),
        items in the iteration order of the Python set of train items (dataset.py:202) unless `public_items`
        fixes the order (synthetic data whose ids are already dense)."""
        self.config = config
        tu, ti, tr = (np.asarray(x) for x in train)
        if public_users is None:
            public_users = self._first_appearance(tu)
        if public_items is None:
            # user-major order of appearance, as `{k for a in train_dict.values() for k in a}` inserts them (dataset.py:202)
            public_items = pyset_order(self._items_in_dict_order(tu, ti, public_users))
        self.users = list(np.asarray(public_users).tolist())
        self.items = list(np.asarray(public_items).tolist())
        self.num_users, self.num_items = len(self.users), len(self.items)
        self._pub_u = np.asarray(public_users)
        self._pub_i = np.asarray(public_items)
        self.private_users = dict(enumerate(self.users))
        self.public_users = {v: k for k, v in self.private_users.items()}
        self.private_items = dict(enumerate(self.items))
        self.public_items = {v: k for k, v in self.private_items.items()}

        pu = self._to_private(tu, self._pub_u)
        pi = self._to_private(ti, self._pub_i)
        m = sp.csr_matrix((np.asarray(tr, dtype=np.float32), (pu, pi)), shape=(self.num_users, self.num_items))
        m.sum_duplicates()
        m.sort_indices()
        self.sp_i_train_ratings = m
        ones = m.copy()
        ones.data[:] = 1.0
        self.sp_i_train = ones
        self.transactions = int(m.nnz)
        self._train_triples, self._test_triples, self._val_triples = (tu, ti, tr), test, val
        self._cache = {}

    @staticmethod
    def _items_in_dict_order(tu, ti, public_users):
        public_users = np.asarray(public_users)
        rank = DataSet._to_private(tu, public_users)
        return ti[DataSet._group_stable(rank, public_users.shape[0])]

    @staticmethod
    def _to_private(pub, table, missing=None):
        """Position of every element of `pub` in `table` (distinct values); `missing` = what an absent value maps to (None:
        the caller guarantees membership).  Integer ids inside a modest range go through a lookup table (one gather);
        otherwise a binary search in the sorted table (searchsorted with a `sorter=` argument dereferences it at every
        comparison: 10 s per 2e7 queries, measured)."""
        pub, table = np.asarray(pub), np.asarray(table)
        if table.shape[0] == 0:
            return np.full(pub.shape[0], -1 if missing is None else missing, dtype=np.int64)
        if table.dtype.kind in "iu" and pub.dtype.kind in "iu":
            lo, hi = int(table.min()), int(table.max())
            if hi - lo < 8 * table.shape[0] + (1 << 24):
                lut = np.full(hi - lo + 1, -1 if missing is None else missing, dtype=np.int64)
                lut[table - lo] = np.arange(table.shape[0], dtype=np.int64)
                if missing is None:
                    return lut[pub - lo]
                inside = (pub >= lo) & (pub <= hi)
                out = np.full(pub.shape[0], missing, dtype=np.int64)
                out[inside] = lut[pub[inside] - lo]
                return out
        sorter = np.argsort(table, kind="stable")
        st = table[sorter]
        loc = np.searchsorted(st, pub)
        if missing is None:
            return sorter[loc]
        loc = np.minimum(loc, st.shape[0] - 1)
        return np.where(st[loc] == pub, sorter[loc], missing)

    @staticmethod
    def _first_appearance(values):
        """Distinct values in order of first appearance (`list(data['userId'].unique())`, dataset.py:248)."""
        values = np.asarray(values)
        if values.dtype.kind in "iu" and values.shape[0]:
            lo, hi = int(values.min()), int(values.max())
            if hi - lo < 8 * values.shape[0] + (1 << 24):
                first = np.full(hi - lo + 1, -1, dtype=np.int64)
                first[values[::-1] - lo] = np.arange(values.shape[0] - 1, -1, -1, dtype=np.int64)   # repeated index: the last write wins
                present = np.flatnonzero(first >= 0)
                return values[np.sort(first[present])]
        _, first = np.unique(values, return_index=True)
        return values[np.sort(first)]

    @staticmethod
    def _group_stable(rank, n_groups):
        """Permutation that brings equal ranks together, groups ascending, file order inside a group: a counting sort
        (scipy's coo -> csr conversion), O(T) instead of a comparison sort of the whole interaction list."""
        n = rank.shape[0]
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        m = sp.coo_matrix((np.ones(n, dtype=np.int8), (rank, np.arange(n, dtype=np.int64))), shape=(n_groups, n))
        return sp.csr_matrix(m).indices.astype(np.int64, copy=False)

    # -- lazily materialised dict views ----------------------------------------------------------------
    def _dict_of(self, triples, restrict_users=True):
        u, i, r = (np.asarray(x) for x in triples)
        out = {uu: {} for uu in self.users} if restrict_users else {}
        for uu, ii, rr in zip(u.tolist(), i.tolist(), r.tolist()):
            if uu in out or not restrict_users:
                out.setdefault(uu, {})[ii] = rr
        return out

    @property
    def train_dict(self):
        if "train" not in self._cache:
            self._cache["train"] = self._dict_of(self._train_triples)
        return self._cache["train"]

    @property
    def i_train_dict(self):
        if "itrain" not in self._cache:
            pu, pi = self.public_users, self.public_items
            self._cache["itrain"] = {pu[u]: {pi[i]: v for i, v in items.items()} for u, items in self.train_dict.items()}
        return self._cache["itrain"]

    @property
    def test_dict(self):
        if "test" not in self._cache:
            self._cache["test"] = self._dict_of(self._test_triples)
        return self._cache["test"]

    @property
    def val_dict(self):
        if self._val_triples is None:
            raise AttributeError("val_dict")
        if "val" not in self._cache:
            self._cache["val"] = self._dict_of(self._val_triples)
        return self._cache["val"]

    @property
    def allunrated_mask(self):
        """dataset.py:245 -- only for small data / compatibility; the kernels use the CSR."""
        if self.num_users * self.num_items > 2_000_000_000:
            raise MemoryError("dense allunrated_mask refused at this size; use sp_i_train")
        return self.sp_i_train.toarray() == 0

    def split_csr(self, validation=False):
        """The held-out split as CSR arrays in PRIVATE ids (indptr int64 [U+1], cols int32 sorted per row, ratings fp32):
        what `Evaluator._split_to_csr` derives from the dicts, built from the triples in vectorised NumPy.  Users the model
        has no row for are dropped, items it has no row for get ids >= num_items (never recommended, still relevant); a
        (user, item) pair listed twice keeps its LAST rating, as dict(zip(items, ratings)) does (dataset.py:252)."""
        triples = self._val_triples if validation else self._test_triples
        if triples is None:
            return None
        u, i, r = (np.asarray(x) for x in triples)
        pu = self._to_private(u, self._pub_u, missing=-1)
        ok = pu >= 0
        pu, i, r = pu[ok], i[ok], np.asarray(r, dtype=np.float32)[ok]
        pi = self._to_private(i, self._pub_i, missing=-1)
        known = pi >= 0
        if not known.all():
            # items the model has no row for: ids num_items, num_items + 1, ... (any numbering does: they are never
            # recommended and only count as relevant items of their user)
            vals, inv = np.unique(i[~known], return_inverse=True)
            pi[~known] = self.num_items + inv
        key = pu * (int(pi.max(initial=0)) + 1) + pi
        order = np.argsort(key, kind="stable")
        key, pu, pi, r = key[order], pu[order], pi[order], r[order]
        last = np.concatenate([key[1:] != key[:-1], [True]]) if key.shape[0] else np.zeros(0, dtype=bool)
        pu, pi, r = pu[last], pi[last], r[last]
        indptr = np.zeros(self.num_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(pu, minlength=self.num_users), out=indptr[1:])
        return indptr, pi.astype(np.int32), r

    def get_test(self):
        return self.test_dict

    def get_validation(self):
        return self.val_dict if self._val_triples is not None else None


# ---------------------------------------------------------------------------------------------------
def np_seed_state(seed):
    """The legacy np.random state right after np.random.seed(seed): MT19937 init_genrand, 624 key words + position 624."""
    st = np.empty(625, dtype=np.uint32)
    x = int(seed) & 0xFFFFFFFF
    for i in range(624):
        st[i] = x
        x = (1812433253 * (x ^ (x >> 30)) + i + 1) & 0xFFFFFFFF
    st[624] = 624
    return st


def _split_flags(users, mode, param, seed, folds=1, state=None):
    """state: a np_seed_state() array carried by the caller -- the draws continue that stream and leave it advanced (the levels of
    a train / validation / test hierarchy share ONE stream, base_splitter.py:73,86-98); None = a fresh stream seeded with `seed`."""
    from .. import _lib
    users = np.asarray(users)
    lo = int(users.min()) if users.shape[0] and users.dtype.kind in "iu" else 0
    if users.shape[0] and users.dtype.kind in "iu" and int(users.max()) - lo < 8 * users.shape[0] + (1 << 24):
        # groupby order = ascending id: counting sort instead of a comparison sort of the whole file
        cnt = np.bincount(users - lo)
        seg = np.ascontiguousarray(cnt[cnt > 0], dtype=np.int64)
        rank = (np.cumsum(cnt > 0) - 1)[users - lo]
        order = DataSet._group_stable(rank, seg.shape[0])
    else:
        order = np.argsort(users, kind="stable")
        su = users[order]
        bounds = np.flatnonzero(np.concatenate([[True], su[1:] != su[:-1], [True]])) if su.shape[0] else np.zeros(1, dtype=np.int64)
        seg = np.ascontiguousarray(np.diff(bounds), dtype=np.int64)
    sorted_flags = np.empty((folds, users.shape[0]), dtype=np.int8)
    if state is None:
        state = np_seed_state(seed)
    assert state.dtype == np.uint32 and state.shape == (625,) and state.flags.c_contiguous
    _lib.check(_lib.load().el_host_split_flags_state(seg.ctypes.data, seg.shape[0], mode, float(param), state.ctypes.data, int(folds),
                                                     sorted_flags.ctypes.data), "el_host_split_flags_state")
    flags = np.empty((folds, users.shape[0]), dtype=np.int8)
    flags[:, order] = sorted_flags
    return flags


def random_subsampling(users, ratio, seed=42, folds=1):
    """Per-user train/test flags as splitter/base_splitter.py:256-274 draws them: np.random.seed(seed) once
    (process_splitting :73), then for every user in groupby (= sorted id) order a list of floor(n(1-r)) zeros and
    the rest ones is shuffled with the legacy np.random.shuffle and laid over the user's rows in file order.  The shuffles
    (a sequential MT19937 stream) run in C: `el_host_split_flags`.  folds > 1: [folds, rows] flags, fold after fold on one stream."""
    f = _split_flags(users, 0, ratio, seed, folds)
    return f[0] if folds == 1 else f


def leave_n_out(users, n=1, seed=42, folds=1):
    """base_splitter.py:276-294 (random leave-n-out): n held-out rows per user, same stream discipline."""
    f = _split_flags(users, 1, n, seed, folds)
    return f[0] if folds == 1 else f


def pyset_order(keys):
    """`list(set_built_by_inserting(keys))` -- the reference's private item order (dataset.py:202, :211-214).  Non-negative
    integer ids take the C emulation of the CPython set (`el_host_pyset_order`, any size); anything else (string ids, negative
    ids) goes through the interpreter's own set."""
    keys = np.asarray(keys)
    if keys.dtype.kind in "iu" and (keys.shape[0] == 0 or (int(keys.min()) >= 0 and int(keys.max()) < (1 << 61) - 1)):
        from .. import _lib
        k64 = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty(k64.shape[0], dtype=np.int64)
        import ctypes
        n_out = ctypes.c_int64(0)
        _lib.check(_lib.load().el_host_pyset_order(k64.ctypes.data, k64.shape[0], out.ctypes.data, ctypes.byref(n_out)),
                   "el_host_pyset_order")
        return out[:n_out.value].astype(keys.dtype, copy=False)
    return np.array(list({k for k in keys.tolist()}))


def load_tsv_dataset(config, path, test_ratio=0.2, seed=42):
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None)
    df = df.iloc[:, :3]
    df.columns = ["userId", "itemId", "rating"]
    flags = random_subsampling(df["userId"].values, test_ratio, seed)
    tr, te = df[flags == 0], df[flags == 1]
    return DataSet(config, (tr["userId"].values, tr["itemId"].values, tr["rating"].values),
                   (te["userId"].values, te["itemId"].values, te["rating"].values))


def default_config(top_k=10, cutoffs=None, simple_metrics=("nDCG",), out_dir="./results", config_test=False,
                   relevance_threshold=0):
    """The base-namespace fields the plugin layer reads: top_k, evaluation.*, output paths."""
    ev = SimpleNamespace(simple_metrics=list(simple_metrics), relevance_threshold=relevance_threshold, paired_ttest=False,
                         wilcoxon_test=False, complex_metrics=[])
    if cutoffs is not None:
        ev.cutoffs = list(cutoffs)
    return SimpleNamespace(top_k=top_k, evaluation=ev, config_test=config_test,
                           path_output_rec_result=f"{out_dir}/recs", path_output_rec_weight=f"{out_dir}/weights",
                           path_output_rec_performance=f"{out_dir}/performance")
