"""Point-wise (user, item, label) sampler of GMF -- device counterpart of
elliot/dataset/samplers/pointwise_pos_neg_sampler.py:14-50: u uniform over users, a fair coin picks a positive of u
(label 1) or a uniformly drawn non-positive item (label 0).  Same `step(events, batch_size)` contract; batches are
(int32 [B], int32 [B], float32 [B]) device tensors produced by `el_pointwise_sample` (Philox stream; the reference
interleaves NumPy's and Python's MT19937 streams, :16-17,33-46)."""
import scipy.sparse as sp

from ... import ops
from .custom_sampler import Sampler as _BprSampler


class Sampler(_BprSampler):
    def step(self, events: int, batch_size: int):
        for start in range(0, events, batch_size):
            n = min(start + batch_size, events) - start
            u, i, y = ops.pointwise_sample(self.ctx, self.pos, n, seed=self.seed, first_sample=self._drawn)
            self._drawn += n
            yield u, i, y
