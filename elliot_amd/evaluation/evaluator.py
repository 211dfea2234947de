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
        self._test = _Split(data.get_test(), self._rel_threshold)
        val = data.get_validation() if hasattr(data, "get_validation") else None
        self._val = _Split(val, self._rel_threshold) if val else None
        self._needed_recommendations = cfg.top_k

    def get_needed_recommendations(self):
        return self._needed_recommendations

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
