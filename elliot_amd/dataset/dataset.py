"""Data objects the latent-factor plugins consume.

`DataSet` exposes the attributes the reference's DataSet gives its models (elliot/dataset/dataset.py:199-245):
num_users/num_items/transactions, private_/public_ id maps, sp_i_train (CSR fp32), test_dict/val_dict, config,
get_test()/get_validation() -- built from index arrays in vectorised NumPy instead of per-user pandas filters
(`dataframe_to_dict`, dataset.py:247-255, is O(U*T)).  The dict-of-dict views (`train_dict`, `i_train_dict`) and
the dense `allunrated_mask` are materialised lazily and only on request: the kernels never need them.

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
            _, first = np.unique(tu, return_index=True)
            public_users = tu[np.sort(first)]
        if public_items is None:
            if ti.shape[0] <= 5_000_000:
                # user-major order of appearance, as `{k for a in train_dict.values() for k in a}` inserts them
                public_items = np.array(list({int(k) for k in self._items_in_dict_order(tu, ti, public_users)}))
            else:
                public_items = np.unique(ti)
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
        pos = {u: n for n, u in enumerate(np.asarray(public_users).tolist())}
        rank = np.fromiter((pos[u] for u in tu.tolist()), dtype=np.int64, count=tu.shape[0])
        return ti[np.argsort(rank, kind="stable")].tolist()

    @staticmethod
    def _to_private(pub, table):
        sorter = np.argsort(table, kind="stable")
        loc = np.searchsorted(table, pub, sorter=sorter)
        return sorter[loc]

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

    def get_test(self):
        return self.test_dict

    def get_validation(self):
        return self.val_dict if self._val_triples is not None else None


# ---------------------------------------------------------------------------------------------------
def random_subsampling(users, ratio, seed=42):
    """Per-user train/test flags as splitter/base_splitter.py:256-274 draws them: np.random.seed(seed) once
    (process_splitting :73), then for every user in groupby (= sorted id) order a list of floor(n(1-r)) zeros and
    the rest ones is shuffled with the legacy np.random.shuffle and laid over the user's rows in file order."""
    users = np.asarray(users)
    rs = np.random.RandomState(seed)
    flags = np.zeros(users.shape[0], dtype=np.int8)
    order = np.argsort(users, kind="stable")
    su = users[order]
    bounds = np.flatnonzero(np.concatenate([[True], su[1:] != su[:-1], [True]]))
    for a, b in zip(bounds[:-1], bounds[1:]):
        n = b - a
        ntrain = int(math.floor(n * (1 - ratio)))
        lst = [0] * ntrain + [1] * (n - ntrain)
        rs.shuffle(lst)
        flags[order[a:b]] = lst
    return flags


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
