"""FunkSVDModel (elliot/recommender/latent_factor_models/FunkSVD/funk_svd_model.py:18-121):
<U_MF[u], I_MF[i]> + U_BIAS[u] + I_BIAS[i], batch-mean squared error against the 0/1 label, GlorotUniform for all four
embeddings (the [rows, 1] bias tables included, :40-47), Adam.  `lambda_weights` / `lambda_bias` are accepted and unused,
as in the reference."""
from ..pointwise_model import PointwiseFactorModel


class FunkSVDModel(PointwiseFactorModel):
    kind, optimizer, with_biases = "mse", "adam", True

    def __init__(self, num_users, num_items, embed_mf_size, lambda_weights, lambda_bias, learning_rate=0.01, random_seed=42,
                 name="FunkSVD", ctx=None, init_weights=None, **kwargs):
        self.lambda_weights, self.lambda_bias = lambda_weights, lambda_bias
        super().__init__(num_users, num_items, embed_mf_size, learning_rate, random_seed, ctx, init_weights)
