"""MatrixFactorizationModel (elliot/recommender/latent_factor_models/MF/matrix_factorization_model.py:18-101):
<U_MF[u], I_MF[i]> fitted to the sampler's 0/1 label with a batch-mean squared error, GlorotUniform tables, Adam.
`lambda_weights` is accepted and unused, as in the reference (the Keras embeddings_regularizer never reaches the tape)."""
from ..pointwise_model import PointwiseFactorModel


class MatrixFactorizationModel(PointwiseFactorModel):
    kind, optimizer, with_biases = "mse", "adam", False

    def __init__(self, num_users, num_items, embed_mf_size, lambda_weights, learning_rate=0.01, random_seed=42, name="MF",
                 ctx=None, init_weights=None, **kwargs):
        self.lambda_weights = lambda_weights
        super().__init__(num_users, num_items, embed_mf_size, learning_rate, random_seed, ctx, init_weights)
