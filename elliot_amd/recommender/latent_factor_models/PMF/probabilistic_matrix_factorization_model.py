"""ProbabilisticMatrixFactorizationModel (elliot/.../PMF/probabilistic_matrix_factorization_model.py:20-120):
sigmoid(<U_MF[u], I_MF[i]>) fitted with a batch-mean squared error; RandomNormal(stddev 0.01) tables; Adam.
`gaussian_variance` configures a keras GaussianNoise layer that the reference calls without training=True from its own
train_step (:78), i.e. in inference mode: it adds nothing, and nothing is added here."""
import numpy as np

from ..pointwise_model import PointwiseFactorModel


class ProbabilisticMatrixFactorizationModel(PointwiseFactorModel):
    kind, optimizer, with_biases = "mse_sigmoid", "adam", False

    def __init__(self, num_users, num_items, embed_mf_size, lambda_weights, gaussian_variance, learning_rate=0.01,
                 random_seed=42, name="MF", ctx=None, init_weights=None, **kwargs):
        self.lambda_weights, self.gaussian_variance = lambda_weights, gaussian_variance
        super().__init__(num_users, num_items, embed_mf_size, learning_rate, random_seed, ctx, init_weights)

    def initial_weights(self, rs, U, I, F):
        return {"Gu": rs.normal(scale=0.01, size=(U, F)).astype(np.float32),
                "Gi": rs.normal(scale=0.01, size=(I, F)).astype(np.float32)}
