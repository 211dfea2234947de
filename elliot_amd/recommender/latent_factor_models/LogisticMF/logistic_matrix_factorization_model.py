"""LogisticMatrixFactorizationModel (elliot/.../LogisticMF/logistic_matrix_factorization_model.py:18-98):
x = <Gu[u], Gi[i]> + Bu[u] + Bi[i]; loss = sum -(alpha y x - (1 + alpha y) log(1 + e^x)) + reg (|Gu[u]|^2 + |Gi[i]|^2) / 2;
GlorotUniform factors, zero biases; Adagrad on (Gi, Bi) or on (Gu, Bu), chosen with set_update_user().

The reference reads `_user_update` as a Python attribute while tracing a @tf.function (:45-47,74-79), so which side a step
really updates there depends on trace timing; this model implements what the plugin's loop states (:96-110)."""
import numpy as np

from .... import ops
from ..pointwise_model import PointwiseFactorModel, glorot_uniform


class LogisticMatrixFactorizationModel(PointwiseFactorModel):
    kind, optimizer, with_biases = "logistic", "adagrad", True

    def __init__(self, num_users, num_items, factors, lambda_weights, alpha, learning_rate=0.01, random_seed=42, name="LMF",
                 ctx=None, init_weights=None, **kwargs):
        super().__init__(num_users, num_items, factors, learning_rate, random_seed, ctx, init_weights, alpha=alpha,
                         l_w=lambda_weights)
        self._side = "items"

    def initial_weights(self, rs, U, I, F):
        return {"Gu": glorot_uniform(rs, U, F), "Gi": glorot_uniform(rs, I, F), "Bu": np.zeros(U, np.float32),
                "Bi": np.zeros(I, np.float32)}

    def set_update_user(self, update_user):
        self._side = "users" if update_user else "items"

    def predict_batch(self, start, stop, **kwargs):
        """[stop-start, I] raw scores (:88-89) -- compatibility path; recommend() never materialises this block."""
        st = self.state
        preds = ops.gemm(self.ctx, st.Gu[start:stop], st.Gi, transB=True, bias=st.Bi)
        st._link(preds, self.num_items, start)
        return preds
