"""NumPy (fp64) restatement of the reference's MF2020 model.  TEST INFRASTRUCTURE.

Follows elliot/recommender/latent_factor_models/MF2020/MF_model.py:
  initialize          :37-56    np.random.seed(seed); biases zero; N(0, 0.1) user factors, then item factors
  train_step          :80-113   per (user, item, rating), in order: prediction with the global bias; the two stable branches of the
                                logistic loss; five in-place updates -- `uf_` / `if_` are VIEWS of the rows (:84-85), so the item row's
                                update (:104) sees the user row :103 has just modified; the biases are scalar copies (old values)
  prepare_predictions :115-116
and MF2020/custom_sampler_rendle.py:32-85 (every positive once + m uniform negatives each, one random.sample shuffle).
Pinned against the reference's own MFModel / Sampler in oracle/gen_golden.py (tests/golden/mf2020_ref.npz).
"""
import numpy as np


def initialize(n_users, n_items, factors, seed, loc=0.0, scale=0.1):
    rs = np.random.RandomState(seed)
    P = rs.normal(loc=loc, scale=scale, size=(n_users, factors))
    Q = rs.normal(loc=loc, scale=scale, size=(n_items, factors))
    return P, Q, np.zeros(n_users), np.zeros(n_items), 0.0


def train_step(P, Q, bu, bi, gb, batch, lr, reg):
    """:80-113 in place on P, Q, bu, bi; returns (sum_of_loss, new global bias)."""
    total = 0.0
    for user, item, rating in batch:
        user, item = int(user), int(item)
        uf, itf = P[user].copy(), Q[item].copy()
        ub, ib = bu[user], bi[item]
        pred = gb + ub + ib + np.dot(uf, itf)
        if pred > 0:
            opm = 1.0 + np.exp(-pred)
            sig = 1.0 / opm
            loss = np.log(opm) + (1.0 - rating) * pred
        else:
            ep = np.exp(pred)
            sig = ep / (1.0 + ep)
            loss = -rating * pred + np.log(1.0 + ep)
        grad = rating - sig
        ufn = uf + lr * (grad * itf - reg * uf)
        P[user] = ufn
        Q[item] = itf + lr * (grad * ufn - reg * itf)        # (the view: updated user row)
        bu[user] = ub + lr * (grad - reg * ub)
        bi[item] = ib + lr * (grad - reg * ib)
        gb = gb + lr * (grad - reg * gb)
        total += loss
    return total, gb


def prepare_predictions(P, Q, bu, bi, gb):
    return bu[:, None] + (gb + bi + P @ Q.T)


def rendle_epoch(sp_i_train, m, seed):
    """custom_sampler_rendle.Sampler (:16-31) + one step() epoch (:32-85): the shuffled [n (1 + m), 3] training matrix.  Both global
    generators are seeded in __init__ (:17-18): NumPy's draws the negatives, Python's `random` the final permutation."""
    import random
    rs = np.random.RandomState(seed)
    pyr = random.Random(seed)
    rows, cols = sp_i_train.nonzero()
    n_items = len({int(c) for c in cols})                      # :25 the number of DISTINCT train items
    out = np.empty((len(rows) * (1 + m), 3), np.int32)
    k = 0
    for u, i in zip(rows, cols):
        out[k] = (u, i, 1)
        k += 1
        for _ in range(m):
            out[k] = (u, rs.randint(n_items), 0)
            k += 1
    perm = pyr.sample(range(out.shape[0]), out.shape[0])
    return out[perm]
