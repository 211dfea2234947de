"""PMF plugin (YAML key `external.PMF`) -- Probabilistic Matrix Factorization (Salakhutdinov & Mnih, NIPS 2007).

Contract of elliot/recommender/latent_factor_models/PMF/probabilistic_matrix_factorization.py:25-120: hyper-parameters
`lr`, `factors` (default 50), `reg`, `gaussian_variance` (+ base keys), result-file name "PMF_...", the same sampler /
epoch loop / loss normalisation as MF."""
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..pointwise_plugin import PointwisePluginMixin
from .probabilistic_matrix_factorization_model import ProbabilisticMatrixFactorizationModel


class PMF(PointwisePluginMixin, RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("factors", "factors", 50),
            param("reg", "reg", 0.0025, attr="_l_w"),
            param("gaussian_variance", "gvar", 0.1, attr="_gvar"),
        ]
        self.autoset_params()
        self._pointwise_setup()
        self._model = ProbabilisticMatrixFactorizationModel(self._num_users, self._num_items, int(self._factors), self._l_w,
                                                            self._gvar, self._learning_rate, self._seed, ctx=self._ctx,
                                                            init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["PMF", self.get_base_params_shortcut(), self.get_params_shortcut()])
