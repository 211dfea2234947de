"""NumPy restatement of the reference's LightGCN.  TEST INFRASTRUCTURE -- the TensorFlow library semantics ([TF]) are recalled, not
executed ("parity unpinned" for them, as for oracle/bprmf_batch.py); the FILE's algebra is pinned by executing the reference's own
LightGCN_model.py on the tensorflow stand-in (oracle/gen_golden_tfshim.py -> tests/golden/tfshim_lightgcn.npz) and its
`_create_adj_mat` unmodified under scipy (oracle/gen_golden.py -> tests/golden/lightgcn_laplacian.npz).

Follows elliot/recommender/graph_based/lightgcn/:
  LightGCN._create_adj_mat          LightGCN.py:96-118       symmetric adjacency over U + I nodes (users first); rowsum + 1e-7; d^-1/2;
                                                             adj.dot(D).transpose().dot(D) -- all float32
  _create_weights                   LightGCN_model.py:63-65  Gu, Gi = tf.zeros (the GlorotUniform initializer of :52 is never used: a model
                                                             left to itself keeps all-zero tables -- every gradient of the BPR head is zero at
                                                             zero -- and ranks by item index; parity tests inject weights)
  _propagate_embeddings             :68-94                   E_0 = [Gu; Gi]; E_k = L E_{k-1}; mean_k(alpha_k E_k), alpha_0 = 1, alpha_k = 1/(1+k);
                                                             ASSIGNED to Gu / Gi (:93-94) -- inside the tape, so the gradients of :163 reach the
                                                             variables through the lookups only, not through the propagation
  train_step                        :136-167                 BPRMF_batch's head without a bias; reg_loss = l_w * sum(l2_loss) * 2; Adam
  predict / get_top_k               :131-133, :171-172
[TF] as oracle/bprmf_batch.py: l2_loss = sum(x^2)/2, clip gradient inside [-80, 1e8], IndexedSlices summed, Keras Adam sparse apply.
"""
import numpy as np

from . import bprmf_batch as ob


def create_adj_mat(sp_i_train, n_users, n_items):
    """LightGCN._create_adj_mat (LightGCN.py:96-118) -> (adjacency csr, laplacian csr), float32."""
    import scipy.sparse as sp
    R = sp.csr_matrix(sp_i_train).astype(np.float32)
    N = n_users + n_items
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)            # :97-104
    A.resize((N, N))
    rowsum = np.array(A.sum(1)).astype(np.float32)                                   # :108
    rowsum += np.float32(1e-7)                                                       # :109 (float32 + python float stays float32)
    d_inv_sqrt = np.power(rowsum, np.float32(-0.5)).flatten()                        # :111
    d_inv_sqrt[np.isinf(d_inv_sqrt)] = 0.                                            # :112
    D = sp.diags(d_inv_sqrt)
    lap = A.dot(D).transpose().dot(D)                                                # :114
    return A.tocsr(), lap.tocsr().astype(np.float32)


def propagate(Gu, Gi, lap, n_layers):
    """:68-94 -> the new (Gu, Gi).  fp32 throughout; the sparse product sums a row's terms in column order."""
    f = np.float32
    U = Gu.shape[0]
    ego = np.concatenate([Gu, Gi], axis=0).astype(f)
    embs, alphas = [ego], [1]
    for k in range(1, n_layers + 1):
        ego = (lap @ ego).astype(f)                                                  # :78-83 (folds concatenated: the whole product)
        embs.append(ego)
        alphas.append(1 / (1 + k))                                                   # :87
    embs = [e * f(a) for a, e in zip(alphas, embs)]                                  # :89
    tot = embs[0]
    for e in embs[1:]:
        tot = tot + e
    mean = tot / f(len(embs))                                                        # :90-91 stack + reduce_mean
    return mean[:U].astype(f), mean[U:].astype(f)


class LightGCNOracle:
    def __init__(self, Gu, Gi, lap, lr, l_w, n_layers):
        self.Gu, self.Gi = np.array(Gu, np.float32, copy=True), np.array(Gi, np.float32, copy=True)
        self.lap, self.lr, self.l_w, self.n_layers = lap, lr, l_w, n_layers
        self.t = 0
        self.slots = {"Gu": (np.zeros_like(self.Gu), np.zeros_like(self.Gu)), "Gi": (np.zeros_like(self.Gi), np.zeros_like(self.Gi))}

    def train_step(self, batch):
        u, i, j = (np.asarray(x).reshape(-1).astype(np.int64) for x in batch)
        self.Gu, self.Gi = propagate(self.Gu, self.Gi, self.lap, self.n_layers)      # :148 (assign)
        Bi = np.zeros(self.Gi.shape[0], np.float32)
        # :149-158 == BPRMF_batch's loss with l_w doubled and no bias terms
        loss = ob.forward_loss(self.Gu, self.Gi, Bi, u, i, j, 2.0 * self.l_w, 0.0)
        _, dGu, dGi = ob.gradients(self.Gu, self.Gi, Bi, u, i, j, 2.0 * self.l_w, 0.0)
        self.t += 1
        for name, theta, g in (("Gu", self.Gu, dGu), ("Gi", self.Gi, dGi)):
            m, v = self.slots[name]
            ob.adam_tf_sparse_apply(theta, m, v, g, self.lr, self.t)                 # :163-164
        return float(loss)

    def predict(self, start, stop):
        return self.Gu[start:stop] @ self.Gi.T                                       # :131-133
