"""MultiDAE plugin (YAML key `external.MultiDAE`) -- the denoising variant of Variational Autoencoders for Collaborative
Filtering, https://arxiv.org/abs/1802.05814.

Contract of elliot/recommender/autoencoders/dae/multi_dae.py:19-105: hyper-parameters `intermediate_dim`, `latent_dim`,
`reg_lambda`, `lr`, `dropout_pkeep` (+ base keys); `batch_size` < 1 means all users; dropout rate = 1 - dropout_pkeep;
the epoch loss is handed to evaluate() as sum / (epoch + 1) (:105).  SURVEY 8f row N3: a sibling of MultiVAE on the same
kernels; the epoch loop is RecMixin.train() over the users instead of the interactions.
"""
from .... import ops
from ....dataset.samplers import sparse_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .multi_dae_model import DenoisingAutoEncoder


class MultiDAE(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._ctx = ops.get_context(max(int(getattr(config, "gpu", 0) or 0), 0))
        self._sampler = sparse_sampler.Sampler(self._data.sp_i_train, ctx=self._ctx)
        if self._batch_size < 1:
            self._batch_size = self._num_users
        self._params_list = [
            param("intermediate_dim", "intermediate_dim", 600, int),
            param("latent_dim", "latent_dim", 200, int),
            param("reg_lambda", "reg_lambda", 0.01, attr="_lambda"),
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("dropout_pkeep", "dropout_pkeep", 1, attr="_dropout_rate"),
        ]
        self.autoset_params()
        self._dropout_rate = 1. - self._dropout_rate
        self._score_block = min(max(self._batch_size, 1), 2048)
        self._model = DenoisingAutoEncoder(self._num_items, self._intermediate_dim, self._latent_dim, self._learning_rate,
                                           self._dropout_rate, self._lambda, self._seed, ctx=self._ctx,
                                           train_csr=self._sampler.train, max_batch=max(self._batch_size, self._score_block),
                                           init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["MultiDAE", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        return self._score_block

    def _epoch_events(self):
        return self._num_users                                        # one pass over the users (multi_dae.py:95-103)
