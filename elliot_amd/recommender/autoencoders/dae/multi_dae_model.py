"""DenoisingAutoEncoder on the MI355X -- the Mult-DAE of elliot/recommender/autoencoders/dae/multi_dae_model.py:74-139
(Encoder :19-51 with a tanh latent layer, Decoder :54-72): the Mult-VAE kernels in their `dae` mode (no log-variance head,
no sampling, no KL term), see el_vae_state.dae in include/elliot_hip.h.
"""
import numpy as np

from .... import ops
from ..vae.multi_vae_model import VariationalAutoEncoder, _glorot_normal


class DenoisingAutoEncoder(VariationalAutoEncoder):
    def __init__(self, original_dim, intermediate_dim=600, latent_dim=200, learning_rate=0.001, dropout_rate=0,
                 regularization_lambda=0.01, random_seed=42, name="DenoisingAutoEncoder", ctx=None, train_csr=None,
                 max_batch=512, init_weights=None, **kwargs):
        if init_weights is None:
            rs = np.random.RandomState(random_seed)
            zeros = lambda n: np.zeros(n, np.float32)
            I, H, L = original_dim, intermediate_dim, latent_dim
            init_weights = {"W1": _glorot_normal(rs, I, H), "b1": zeros(H), "Wm": _glorot_normal(rs, H, L), "bm": zeros(L),
                            "W3": _glorot_normal(rs, L, H), "b3": zeros(H), "W4": _glorot_normal(rs, H, I), "b4": zeros(I)}
        if "Wv" in init_weights:
            raise ValueError("MultiDAE has no log-variance head (weights must not contain Wv / bv)")
        super().__init__(original_dim, intermediate_dim, latent_dim, learning_rate, dropout_rate, regularization_lambda,
                         random_seed, name=name, ctx=ctx or ops.get_context(0), train_csr=train_csr, max_batch=max_batch,
                         init_weights=init_weights, eps_mode="zero")

    def train_step(self, batch, **kwargs):
        """multi_dae_model.py:114-127: loss = -mean_b sum_i log_softmax(logits) x (no annealed KL)."""
        return super().train_step(batch, 0.0)
