"""Epoch sampler of NeuMF -- counterpart of elliot/recommender/neural/NeuMF/custom_sampler.py:14-48.

An epoch is the SET of all positives (label 1) plus `m` uniformly drawn non-positive items per positive (label 0,
set-deduplicated), shuffled with `random.sample`, then cut into batches (:27-48).  With the default m = 0 there are
no negatives at all (neural_matrix_factorization.py:67).  This is index bookkeeping, done on the host exactly like the
reference does (same Python set / random.sample / np.random.randint calls, seeds 42), then shipped to the device once
per epoch; for interaction counts where Python sets are impractical a vectorised path draws the same distribution.
"""
import random

import numpy as np
import torch

from .... import ops


class Sampler:
    PY_LIMIT = 3_000_000

    def __init__(self, indexed_ratings, m, ctx=None, sp_i_train=None):
        np.random.seed(42)                                  # :16
        random.seed(42)                                     # :17
        self.ctx = ctx                                      # (resolved in step(): the epoch bookkeeping itself is host work)
        self._m = int(m)
        self._indexed_ratings = indexed_ratings
        self._csr = sp_i_train.tocsr() if sp_i_train is not None else None
        if indexed_ratings is not None:
            self._nitems = len({k for a in indexed_ratings.values() for k in a.keys()})
            self._ui_dict = {u: list(set(indexed_ratings[u])) for u in indexed_ratings}
            self._n_pos = sum(len(v) for v in self._ui_dict.values())
        else:
            self._nitems = self._csr.shape[1]
            self._n_pos = int(self._csr.nnz)

    def _epoch_python(self):
        r_int = np.random.randint
        n_items, ui_dict = self._nitems, self._ui_dict
        pos = {(u, i, 1) for u, items in ui_dict.items() for i in items}       # :31
        neg = set()
        for u, i, _ in pos:                                                    # :34-41
            ui = ui_dict[u]
            for _ in range(self._m):
                j = r_int(n_items)
                while j in ui:
                    j = r_int(n_items)
                neg.add((u, j, 0))
        samples = list(pos)
        samples.extend(list(neg))
        samples = random.sample(samples, len(samples))                         # :45
        return np.asarray(samples, dtype=np.int64)

    def _epoch_vectorised(self):
        m = self._csr
        u = np.repeat(np.arange(m.shape[0], dtype=np.int64), np.diff(m.indptr))
        i = m.indices.astype(np.int64)
        rows = [np.stack([u, i, np.ones_like(u)], 1)]
        if self._m > 0:
            nu = np.repeat(u, self._m)
            nj = np.random.randint(0, self._nitems, nu.shape[0])
            key = m.indptr[nu]  # membership test through a dense hash of (u, j)
            present = np.asarray(m[nu, nj]).reshape(-1) > 0
            while present.any():
                nj[present] = np.random.randint(0, self._nitems, int(present.sum()))
                present[present] = np.asarray(m[nu[present], nj[present]]).reshape(-1) > 0
            neg = np.unique(np.stack([nu, nj, np.zeros_like(nu)], 1), axis=0)
            rows.append(neg)
        s = np.concatenate(rows)
        return s[np.random.permutation(s.shape[0])]

    def step(self, batch_size: int):
        use_py = self._indexed_ratings is not None and self._n_pos * (1 + self._m) <= self.PY_LIMIT
        s = self._epoch_python() if use_py else self._epoch_vectorised()
        if self.ctx is None:
            self.ctx = ops.get_context(0)
        d = self.ctx.device
        u = torch.from_numpy(np.ascontiguousarray(s[:, 0], dtype=np.int32)).to(d)
        i = torch.from_numpy(np.ascontiguousarray(s[:, 1], dtype=np.int32)).to(d)
        y = torch.from_numpy(np.ascontiguousarray(s[:, 2], dtype=np.float32)).to(d)
        for start in range(0, s.shape[0], batch_size):
            e = min(start + batch_size, s.shape[0])
            yield u[start:e], i[start:e], y[start:e]
