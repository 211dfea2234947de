"""GMF plugin (YAML key `external.GMF`) -- the generalised matrix factorisation branch of Neural Collaborative Filtering.

Contract of elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization.py:22-109: hyper-parameters `lr`,
`mf_factors`, `is_edge_weight_train` (+ base keys), result-file name "GeneralizedMF_...", point-wise positive / negative
samples in batches of `batch_size`, epoch loss handed to evaluate() as sum / (epoch + 1) (:94).  The training loop is
RecMixin.train().
"""
from .... import ops
from ....dataset.samplers import pointwise_pos_neg_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..NeuMF.neural_matrix_factorization_model import GeneralizedMatrixFactorizationModel


class GMF(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("mf_factors", "mffactors", 10),
            param("is_edge_weight_train", "isedgeweighttrain", True),
        ]
        self.autoset_params()
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        replay = getattr(self._params, "sampler", "philox") == "replay"      # the reference's exact sample stream
        self._sampler = pointwise_pos_neg_sampler.Sampler(self._data.i_train_dict if replay else self._data.sp_i_train,
                                                          ctx=self._ctx, replay=replay)
        self._model = GeneralizedMatrixFactorizationModel(self._num_users, self._num_items, int(self._mf_factors),
                                                          self._is_edge_weight_train, self._learning_rate, self._seed,
                                                          ctx=self._ctx,
                                                          max_batch=max(min(self._batch_size, 1 << 23), self._num_items),
                                                          init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["GeneralizedMF", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        return max(1, min(4096, (8 * self._model.state.Bmax) // max(self._num_items, 1)))
