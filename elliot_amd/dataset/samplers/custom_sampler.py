"""BPR triplet sampler -- device-side counterpart of elliot/dataset/samplers/custom_sampler.py:14-46.

Same constructor argument and the same `step(events, batch_size)` generator contract (one batch of
(user, pos, neg) per iteration, `events` triplets per epoch), but the batch is produced by the Philox kernel
`el_bpr_sample` and stays in HBM: three int32 device tensors of shape [B] instead of three int64 host arrays of
shape [B, 1].  The reference's distribution is kept (u uniform over USERS, i uniform over pos(u), j uniform over
items rejected while in pos(u), :32-41); its MT19937 bit stream is not (oracle/sampler.py restates that one).
"""
import numpy as np
import scipy.sparse as sp

from ... import ops


class Sampler:
    def __init__(self, indexed_ratings, ctx=None, seed=42, n_items=None, replay=False):
        """indexed_ratings: the reference's {private_user: {private_item: rating}} dict (dataset.py:216-217)
        or a scipy CSR train matrix (`data.sp_i_train`).

        replay=True (needs the dict): emit the reference's EXACT triplet stream -- MT19937 seeded 42, per-user lists in
        `list(set(...))` order (custom_sampler.py:15,21) -- through el_bpr_sample_mt19937 instead of the Philox
        sampler.  Meant for seed-exact comparisons with the reference; the Philox path is the fast one."""
        self.ctx = ctx or ops.get_context(0)
        self._replay = None
        if replay and sp.issparse(indexed_ratings):
            raise ValueError("replay=True needs the reference's i_train_dict (its set order defines the stream)")
        if sp.issparse(indexed_ratings):
            m = indexed_ratings.tocsr()
            m.sort_indices()
            indptr, indices, n_items = m.indptr, m.indices, m.shape[1]
        else:
            users = list(indexed_ratings.keys())
            rows = [np.sort(np.fromiter(indexed_ratings[u].keys(), dtype=np.int32)) for u in users]
            indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
            indices = np.concatenate(rows) if rows else np.zeros(0, np.int32)
            if n_items is None:
                n_items = len({int(k) for r in rows for k in r})      # custom_sampler.py:19-20
        self._nusers = indptr.shape[0] - 1
        self._nitems = int(n_items)
        self.pos = ops.DeviceCSR(indptr, indices, self._nitems, self.ctx.device)
        self.seed = seed                                              # custom_sampler.py:15 seeds 42
        self._drawn = 0
        if replay:
            lists = [list(set(indexed_ratings[u])) for u in indexed_ratings]      # custom_sampler.py:21
            self._replay = ops.MtReplaySampler(self.ctx, lists, self.pos, seed=42)

    @property
    def philox(self):
        """True when batches come from the counter-based device sampler (an epoch can then be drawn inside the library)."""
        return self._replay is None

    def advance(self, n):
        """Account for n samples drawn on the caller's behalf (BprmfDeviceState.train_loop); returns the old offset."""
        first, self._drawn = self._drawn, self._drawn + int(n)
        return first

    def step(self, events: int, batch_size: int):
        for start in range(0, events, batch_size):
            n = min(start + batch_size, events) - start
            if self._replay is not None:
                yield self._replay.sample(n)
                continue
            u, i, j = ops.bpr_sample(self.ctx, self.pos, n, seed=self.seed, first_sample=self._drawn)
            self._drawn += n
            yield u, i, j
