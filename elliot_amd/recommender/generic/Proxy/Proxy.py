"""ProxyRecommender (YAML key `external.ProxyRecommender`): evaluates a recommendation file written earlier -- by this
backend, by the reference, or by anything that writes Elliot's `user<TAB>item<TAB>score` lines (utils/write.py:35-44).

Contract of elliot/recommender/generic/Proxy/Proxy.py:10-75: parameters `path` and optional `name` (default: the file's
base name); train() = read the file + one evaluate(); per user the rows are ordered by score, descending, rows of equal
score keeping their file order (:74, Python's stable sort); get_single_recommendation keeps the rows at positions < k that
the candidate mask allows (:53-65: the position cut comes BEFORE the filter).  Users that do not occur in the file get no
entry.  The file is parsed into flat arrays once (no per-user DataFrame groups, no dense [U, I] mask);
this class needs no GPU."""
import ntpath

import numpy as np
import pandas as pd

from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin


class ProxyRecommender(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [param("name", "name", ""), param("path", "path", "")]
        self.autoset_params()
        if not self._name:
            self._name = ntpath.basename(self._path).rsplit(".", 1)[0]

    @property
    def name(self):
        return self._name

    def train(self):
        print("Reading recommendations")
        self._table = self.read_recommendations(self._path)
        print("Evaluating recommendations")
        self.evaluate()

    def _device_metrics(self):
        return False                                    # one pass over a file: the host evaluator is the consumer

    def read_recommendations(self, path):
        """-> dict of flat arrays sorted by (user, score desc, file order): public ids `user`, `item`, `score`, the
        position `rank` of every row inside its user's list, and the private ids (-1 = unknown to this dataset)."""
        df = pd.read_csv(path, sep="\t", header=None, names=["userId", "itemId", "prediction", "timestamp"])
        user, item, score = df["userId"].to_numpy(), df["itemId"].to_numpy(), df["prediction"].to_numpy(dtype=np.float64)
        order = np.lexsort((np.arange(len(user)), -score, user))          # last key first: user, then score desc, then line
        user, item, score = user[order], item[order], score[order]
        starts = np.r_[0, np.flatnonzero(user[1:] != user[:-1]) + 1] if len(user) else np.zeros(0, np.int64)
        seg = np.zeros(len(user), np.int64)
        seg[starts] = 1
        rank = np.arange(len(user)) - starts[np.cumsum(seg) - 1] if len(user) else seg
        pu, pi = self._data.public_users, self._data.public_items
        return {"user": user, "item": item, "score": score, "rank": rank,
                "u": np.fromiter((pu.get(x, -1) for x in user.tolist()), np.int64, len(user)),
                "i": np.fromiter((pi.get(x, -1) for x in item.tolist()), np.int64, len(item))}

    def get_recommendations(self, top_k):
        return self.process_protocol(top_k)

    def get_candidate_mask(self, validation=False):
        """("excl", train CSR) | ("cand", candidate CSR), as RecMixin does, but host-side scipy matrices."""
        import scipy.sparse as sp
        if not self._negative_sampling:
            return ("excl", self._data.sp_i_train.tocsr())
        which = "val" if validation else "test"
        if hasattr(self._data, f"{which}_cand_csr"):
            ip, ix = getattr(self._data, f"{which}_cand_csr")
            return ("cand", sp.csr_matrix((np.ones(len(ix), np.int8), ix, ip), shape=self._data.sp_i_train.shape))
        return ("cand", sp.csr_matrix(np.asarray(getattr(self._data, f"{which}_mask"), dtype=bool)))

    def get_single_recommendation(self, mask, k, *args):
        t = self._table
        kind, csr = mask
        n_items = csr.shape[1]
        known = (t["u"] >= 0) & (t["i"] >= 0)
        coo = csr.tocoo()
        in_csr = np.zeros(len(known), bool)
        in_csr[known] = np.isin(t["u"][known] * n_items + t["i"][known], coo.row.astype(np.int64) * n_items + coo.col)
        allowed = known & (~in_csr if kind == "excl" else in_csr)
        keep = (t["rank"] < k) & allowed
        recs = {u: [] for u in np.unique(t["user"]).tolist()}             # a user whose rows are all filtered keeps []
        for u, i, s in zip(t["user"][keep].tolist(), t["item"][keep].tolist(), t["score"][keep].tolist()):
            recs[u].append((i, s))
        return recs
