"""Stand-alone evaluator with the interface of elliot/evaluation/evaluator.py:37-162 for the accuracy
metrics a latent-factor experiment normally asks for (nDCG, Precision, Recall, HR, MAP, MRR, F1).

Definitions follow the reference:
  relevance          test items with rating >= relevance_threshold            relevance.py:87-96
  nDCG gain          2**(rating - threshold + 1) - 1, discount ln2/ln(rank+2) relevance.py:49-55,71-82
  IDCG               the user's gains sorted descending, first min(n, cutoff)  ndcg.py:68-79
  Precision / Recall hits / cutoff ; hits / #relevant                         precision.py:66, recall.py:66
  HR, MAP, MRR, F1   hit_rate.py:66, map.py:69-80, mrr.py:63-70, f1.py:56-68
  averaging          over users that have recommendations AND >= 1 relevant test item (ndcg.py:124-125)
Computation is vectorised over an [n_users, k] item matrix instead of per-user Python loops.  The other
metric families of the reference (coverage, diversity, novelty, bias, fairness) are out of scope here; inside
an Elliot process the genuine Evaluator is used instead (recommender/_compat.py).
"""
import math
from types import SimpleNamespace

import numpy as np

SUPPORTED = ("nDCG", "Precision", "Recall", "HR", "MAP", "MRR", "F1")


def _canon(name):
    for s in SUPPORTED:
        if s.lower() == name.lower():
            return s
    raise Exception(f"Metric {name} is not available in the stand-alone evaluator (supported: {SUPPORTED})")


class _Split:
    """One held-out split ({public_user: {public_item: rating}}) in array form."""

    def __init__(self, test, threshold):
        self.users = [u for u, items in test.items() if any(r >= threshold for r in items.values())]
        self.row = {u: n for n, u in enumerate(self.users)}
        self.rel = [{i: r for i, r in test[u].items() if r >= threshold} for u in self.users]
        self.threshold = threshold


class Evaluator:
    def __init__(self, data, params):
        cfg = data.config
        self._data, self._params = data, params
        self._k = getattr(cfg.evaluation, "cutoffs", [cfg.top_k])
        self._k = self._k if isinstance(self._k, list) else [self._k]
        if any(np.array(self._k) > cfg.top_k):
            raise Exception("Cutoff values must be smaller than recommendation list length (top_k)")
        self._rel_threshold = getattr(cfg.evaluation, "relevance_threshold", 0)
        self._metrics = [_canon(m) for m in cfg.evaluation.simple_metrics]
        self._dict_splits = None          # the dict-based form is only built when eval() is handed recommendation dicts
        self._needed_recommendations = cfg.top_k

    def _splits(self):
        if self._dict_splits is None:
            data = self._data
            val = data.get_validation() if hasattr(data, "get_validation") else None
            self._dict_splits = (_Split(data.get_test(), self._rel_threshold), _Split(val, self._rel_threshold) if val else None)
        return self._dict_splits

    @property
    def _test(self):
        return self._splits()[0]

    @property
    def _val(self):
        return self._splits()[1]

    def get_needed_recommendations(self):
        return self._needed_recommendations

    # ---- device path: metrics straight from the [users, k] index tensors --------------------------
    supports_device = True

    def device_sets(self, data, device):
        """Held-out splits as device CSRs in PRIVATE ids (rows = the model's user order).  Test items the model has no
        row for get ids >= num_items: never recommended, but they count as relevant items like in the reference."""
        if getattr(self, "_dev_sets", None) is None:
            from .. import ops
            if hasattr(data, "split_csr"):                 # array data plane: no dicts on the way to the device
                test = data.split_csr(False)
                val = data.split_csr(True)
                self._dev_sets = {"test": ops.DeviceTestSet(*test, device),
                                  "val": ops.DeviceTestSet(*val, device) if val is not None and val[1].shape[0] else None}
                return self._dev_sets
            self._dev_sets = {"test": self._split_to_csr(ops, data, self._test_dict_raw(data, False), device)}
            raw_val = self._test_dict_raw(data, True)
            self._dev_sets["val"] = self._split_to_csr(ops, data, raw_val, device) if raw_val else None
        return self._dev_sets

    @staticmethod
    def _test_dict_raw(data, validation):
        if validation:
            return data.get_validation() if hasattr(data, "get_validation") else None
        return data.get_test()

    @staticmethod
    def _split_to_csr(ops, data, split, device):
        pu, pi = data.public_users, data.public_items
        U, I = data.num_users, data.num_items
        rows = [[] for _ in range(U)]
        unknown = {}
        for u, items in split.items():
            if u not in pu:
                continue
            r = rows[pu[u]]
            for it, rating in items.items():
                col = pi.get(it)
                if col is None:
                    col = unknown.setdefault(it, I + len(unknown))
                r.append((col, rating))
        indptr = np.zeros(U + 1, dtype=np.int64)
        cols, vals = [], []
        for u, r in enumerate(rows):
            r.sort()
            indptr[u + 1] = indptr[u] + len(r)
            cols.extend(c for c, _ in r)
            vals.extend(v for _, v in r)
        return ops.DeviceTestSet(indptr, np.asarray(cols, dtype=np.int32), np.asarray(vals, dtype=np.float32), device)

    def eval_device(self, ctx, data, blocks):
        """blocks: iterable of (first_private_user, idx_val, idx_test) with [n, k] int32 device tensors (idx_val may be
        the same tensor).  Returns the same structure as eval()."""
        from .. import ops
        import torch
        sets = self.device_sets(data, ctx.device)
        acc = {(sp, c): torch.zeros(8, dtype=torch.float64, device=ctx.device)
               for sp in ("val", "test") if sets[sp] is not None for c in self._k}
        for first, idx_val, idx_test in blocks:
            for (sp, c), sums in acc.items():
                ops.rec_metrics(ctx, idx_val if sp == "val" else idx_test, sets[sp], self._rel_threshold, c, u_start=first, sums=sums)
        host = {key: v.cpu().numpy() for key, v in acc.items()}

        def means(sp, c):
            s = host[(sp, c)]
            if s[7] == 0:
                return {}
            return {m: float(s[ops.METRIC_NAMES.index(m)] / s[7]) for m in self._metrics}

        res = {}
        for c in self._k:
            test = means("test", c)
            val = means("val", c) if sets["val"] is not None else None
            if not val:
                val = test
            res[c] = {"val_results": val, "val_statistical_results": {}, "test_results": test, "test_statistical_results": {}}
        return res

    # ---------------------------------------------------------------------------------------------
    def _eval_split(self, recs, split, cutoff):
        """recs: {public_user: [(public_item, score), ...]}"""
        users = [u for u in recs if u in split.row]
        if not users:
            return {}
        n = len(users)
        hits = np.zeros((n, cutoff), dtype=bool)
        gains = np.zeros((n, cutoff), dtype=np.float64)
        nrel = np.zeros(n, dtype=np.float64)
        idcg = np.zeros(n, dtype=np.float64)
        disc = np.array([math.log(2) / math.log(r + 2) for r in range(cutoff)])
        thr = split.threshold
        for r, u in enumerate(users):
            rel = split.rel[split.row[u]]
            nrel[r] = len(rel)
            g = sorted((2 ** (s - thr + 1) - 1 for s in rel.values()), reverse=True)[:cutoff]
            idcg[r] = float(np.dot(g, disc[:len(g)]))
            for c, (item, _) in enumerate(recs[u][:cutoff]):
                s = rel.get(item)
                if s is not None:
                    hits[r, c] = True
                    gains[r, c] = 2 ** (s - thr + 1) - 1
        nh = hits.sum(1)
        out = {}
        for m in self._metrics:
            if m == "nDCG":
                dcg = gains @ disc
                v = np.where(dcg > 0, dcg / np.where(idcg > 0, idcg, 1.0), 0.0)
            elif m == "Precision":
                v = nh / cutoff
            elif m == "Recall":
                v = nh / nrel
            elif m == "HR":
                v = (nh > 0).astype(np.float64)
            elif m == "MAP":
                v = (np.cumsum(hits, 1) / np.arange(1, cutoff + 1)).mean(1)
            elif m == "MRR":
                first = np.argmax(hits, 1)
                v = np.where(nh > 0, 1.0 / (first + 1), 0.0)
            elif m == "F1":
                p, rc = nh / cutoff, nh / nrel
                v = np.where(p + rc > 0, 2 * p * rc / np.where(p + rc > 0, p + rc, 1.0), 0.0)
            out[m] = float(np.average(v))
        return out

    def eval(self, recommendations):
        """recommendations = (recs_val, recs_test); returns {cutoff: {"val_results", "test_results", ...}}
        exactly as evaluator.py:79-92 (a missing validation split mirrors the test results)."""
        res = {}
        for k in self._k:
            val = self._eval_split(recommendations[0], self._val, k) if self._val else None
            test = self._eval_split(recommendations[1], self._test, k)
            if not val:
                val = test
            res[k] = {"val_results": val, "val_statistical_results": {}, "test_results": test,
                      "test_statistical_results": {}}
        return res
