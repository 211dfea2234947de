"""Sampler of MF2020 -- counterpart of elliot/recommender/latent_factor_models/MF2020/custom_sampler_rendle.py:15-85.

One epoch = every train interaction once with label 1 plus `m` items drawn uniformly from the catalogue with label 0 (no check against
the user's positives, duplicates allowed: :66-71), all of it shuffled once (:81-82), yielded in batches.  The draws replay the
reference's generators: NumPy's global stream (seeded in __init__, :17) for the negatives, Python's `random` (:18) for the permutation
-- the same [n (1 + m), 3] matrix, row for row (tests/golden/mf2020_ref.npz)."""
import random

import numpy as np


class Sampler:
    def __init__(self, indexed_ratings, m, sparse_matrix, seed):
        self._rs = np.random.RandomState(seed)
        self._py = random.Random(seed)
        self._sparse = sparse_matrix
        self._m = int(m)
        self._nonzero = self._sparse.nonzero()
        self._nitems = len({int(c) for c in self._nonzero[1]})          # :25 the distinct train items
        self._num_pos_examples = len(self._nonzero[0])

    def step(self, batch_size):
        n, m = self._num_pos_examples, self._m
        mat = np.empty((n * (1 + m), 3), np.int32)
        rows, cols = self._nonzero
        mat[::1 + m, 0], mat[::1 + m, 1], mat[::1 + m, 2] = rows, cols, 1
        if m:
            # the reference draws r_int(num_items) once per negative, positive after positive (:70): one array call of the legacy
            # generator consumes the same stream in the same order (masked rejection over successive 32-bit outputs, per element)
            neg = self._rs.randint(self._nitems, size=n * m).astype(np.int32).reshape(n, m)
            for k in range(m):
                mat[1 + k::1 + m, 0], mat[1 + k::1 + m, 1], mat[1 + k::1 + m, 2] = rows, neg[:, k], 0
        perm = self._py.sample(range(mat.shape[0]), mat.shape[0])
        mat = mat[perm]
        for start in range(0, mat.shape[0], batch_size):
            yield mat[start:min(start + batch_size, mat.shape[0])]
