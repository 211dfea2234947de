"""User-batch sampler of the autoencoders -- counterpart of elliot/dataset/samplers/sparse_sampler.py:13-25.

The reference shuffles the users with `random.sample(range(users), users)` (Python `random`, seeded 42 in the
constructor, :16) and yields `train[rows].toarray()`: dense fp32 [B, I] blocks built on the host every step.
Here the same permutation is drawn (same call, same seed => same user order) but a batch is just the int32
user ids on the device; the kernels read the rows from the device-resident CSR.
"""
import random

import numpy as np
import torch

from ... import ops


class Sampler:
    def __init__(self, sp_i_train, ctx=None):
        random.seed(42)                                   # sparse_sampler.py:16
        self.ctx = ctx or ops.get_context(0)
        m = sp_i_train.tocsr()
        m.sort_indices()
        self.train = ops.DeviceCSR(m.indptr, m.indices, m.shape[1], self.ctx.device)

    def step(self, users: int, batch_size: int):
        shuffled = random.sample(range(users), users)     # :21
        order = torch.from_numpy(np.asarray(shuffled, dtype=np.int32)).to(self.ctx.device)
        for start in range(0, users, batch_size):
            yield order[start:min(start + batch_size, users)]
