"""NumPy (fp64) restatement of the BPRMF per-sample SGD model.  TEST INFRASTRUCTURE.

Follows elliot/recommender/latent_factor_models/BPRMF/BPRMF_model.py:
  initialize           :40-56   (np.random.seed(seed) then N(0, 0.1) user factors, item factors)
  update_factors       :91-117  (five in-place updates; item rows see the UPDATED user row because
                                 `user_factors` is a view that :109 overwrites -- SURVEY 7.3-4)
  get_user_predictions :70-85
Pinned against the reference's own MFModel in oracle/gen_golden.py (tests/golden/bprmf_sgd_trace.npz).
"""
import numpy as np


def initialize(n_users, n_items, factors, seed, loc=0.0, scale=0.1):
    """MFModel.__init__/initialize (:24,:40-56)."""
    rs = np.random.RandomState(seed)           # np.random.seed(random_seed) on the global stream
    b = np.zeros(n_items)
    P = rs.normal(loc=loc, scale=scale, size=(n_users, factors))
    Q = rs.normal(loc=loc, scale=scale, size=(n_items, factors))
    return P, Q, b


def update_factors(P, Q, b, u, i, j, lr, reg_bias, reg_user, reg_pos, reg_neg):
    """One triplet, in place (:91-117)."""
    pu = P[u].copy()
    qi = Q[i].copy()
    qj = Q[j].copy()
    bi, bj = b[i], b[j]
    x_ui = 0 + bi + pu @ qi                    # indexed_predict :66-68 (global_bias == 0)
    x_uj = 0 + bj + pu @ qj
    z = 1 / (1 + np.exp(x_ui - x_uj))          # :98
    b[i] = bi + lr * (z - reg_bias * bi)       # :100-101
    b[j] = bj + lr * (-z - reg_bias * bj)      # :104-105
    pu_new = pu + lr * ((qi - qj) * z - reg_user * pu)      # :108-109
    P[u] = pu_new
    Q[i] = qi + lr * (pu_new * z - reg_pos * qi)            # :112-113 (view -> updated user row)
    Q[j] = qj + lr * (-pu_new * z - reg_neg * qj)           # :116-117


def train_sequential(P, Q, b, us, is_, js, **hp):
    """MFModel.train_step over a triplet list (:87-89)."""
    for u, i, j in zip(us, is_, js):
        update_factors(P, Q, b, int(u), int(i), int(j), **hp)


def get_user_predictions(P, Q, b, user, mask_row, k):
    """:70-85 -- scores fp64, masked -> -inf, top-k (order among exact ties unspecified there)."""
    s = b + P[user] @ Q.T
    s = np.where(mask_row, s, -np.inf)
    kk = min(k, s.shape[0])
    part = np.argpartition(s, -kk)[-kk:]
    order = part[np.argsort(s[part])[::-1]]
    return order, s[order]
