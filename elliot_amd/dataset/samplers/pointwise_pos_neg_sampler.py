"""Point-wise (user, item, label) sampler of GMF / MF / PMF / FunkSVD / LogisticMF -- device counterpart of
elliot/dataset/samplers/pointwise_pos_neg_sampler.py:14-50: u uniform over users, a fair coin picks a positive of u
(label 1) or a uniformly drawn non-positive item (label 0).  Same `step(events, batch_size)` contract; batches are
(int32 [B], int32 [B], float32 [B]) device tensors.

Two streams:
  philox (default)  `el_pointwise_sample`: counter-based, drawn in HBM -- the reference's distribution, not its bits;
  replay            the reference's EXACT stream: it interleaves NumPy's legacy MT19937 (`np.random.randint`) with Python's
                    (`random.getrandbits(1)`), both seeded 42 (:16-17), per-user lists in `list(set(...))` order (:23), one
                    sample at a time with data-dependent rejection (:33-46) -- inherently sequential index bookkeeping, so it
                    is replayed on the host with private `RandomState(42)` / `random.Random(42)` generators (the same
                    sequences as the reference's global ones) and shipped to the device per batch.  For seed-exact
                    comparisons with the reference; pinned by tests/golden/pointwise_sampler_ref.npz."""
import random

import numpy as np
import scipy.sparse as sp
import torch

from ... import ops
from .custom_sampler import Sampler as _BprSampler


def replay_stream(indexed_ratings, events, state=None):
    """`events` samples of pointwise_pos_neg_sampler.Sampler.step's stream (:33-46) as int64 arrays (u, i, b).
    state: (RandomState, random.Random, ui_dict, n_items) from an earlier call to continue the stream (next epoch)."""
    if state is None:
        n_items = len({k for a in indexed_ratings.values() for k in a.keys()})          # :21-22
        ui_dict = {u: list(set(indexed_ratings[u])) for u in indexed_ratings}            # :23
        state = (np.random.RandomState(42), random.Random(42), ui_dict, n_items)         # :16-17
    rs, pr, ui_dict, n_items = state
    n_users = len(ui_dict)
    r_int, bit = rs.randint, pr.getrandbits
    out = np.empty((events, 3), np.int64)

    def sample(depth=0):
        u = r_int(n_users)
        ui = ui_dict[u]
        lui = len(ui)
        if lui == n_items:                                   # :37-38 (the recursive result is discarded by the reference)
            if depth > 64:
                raise RuntimeError("a user interacted with every item: the reference's sampler does not terminate here")
            sample(depth + 1)
        b = bit(1)                                           # :39
        if b:
            i = ui[r_int(lui)]                               # :41
        else:
            i = r_int(n_items)                               # :43
            while i in ui:                                   # :44-45 (list scan in the reference; same draws)
                i = r_int(n_items)
        return u, i, b

    for t in range(events):
        out[t] = sample()
    return out[:, 0], out[:, 1], out[:, 2], state


class Sampler(_BprSampler):
    def __init__(self, indexed_ratings, ctx=None, seed=42, n_items=None, replay=False):
        if replay and sp.issparse(indexed_ratings):
            raise ValueError("replay=True needs the reference's i_train_dict (its set order defines the stream)")
        super().__init__(indexed_ratings, ctx=ctx, seed=seed, n_items=n_items, replay=False)
        self._pw_replay = bool(replay)
        self._ratings = indexed_ratings if replay else None
        self._pw_state = None

    @property
    def philox(self):
        return not self._pw_replay

    def step(self, events: int, batch_size: int):
        d = self.ctx.device
        for start in range(0, events, batch_size):
            n = min(start + batch_size, events) - start
            if self._pw_replay:
                u, i, b, self._pw_state = replay_stream(self._ratings, n, self._pw_state)
                yield (torch.from_numpy(u.astype(np.int32)).to(d), torch.from_numpy(i.astype(np.int32)).to(d),
                       torch.from_numpy(b.astype(np.float32)).to(d))
                continue
            u, i, y = ops.pointwise_sample(self.ctx, self.pos, n, seed=self.seed, first_sample=self._drawn)
            self._drawn += n
            yield u, i, y
