"""GMF plugin -- drop-in for elliot/recommender/neural/GeneralizedMF/generalized_matrix_factorization.py:22-109.
Same YAML keys: lr, mf_factors, is_edge_weight_train (+ base epochs / batch_size / seed / meta)."""
from tqdm import tqdm

from ....dataset.samplers import pointwise_pos_neg_sampler as pws
from ...base_recommender_model import BaseRecommenderModel, init_charger
from ...recommender_utils_mixin import RecMixin
from .... import ops
from ..NeuMF.neural_matrix_factorization_model import GeneralizedMatrixFactorizationModel


class GMF(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            ("_learning_rate", "lr", "lr", 0.001, None, None),
            ("_mf_factors", "mf_factors", "mffactors", 10, None, None),
            ("_is_edge_weight_train", "is_edge_weight_train", "isedgeweighttrain", True, None, None)
        ]
        self.autoset_params()
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._sampler = pws.Sampler(self._data.sp_i_train, ctx=self._ctx)
        cap = max(min(self._batch_size, 1 << 20), self._num_items)
        self._model = GeneralizedMatrixFactorizationModel(self._num_users, self._num_items, int(self._mf_factors),
                                                          self._is_edge_weight_train, self._learning_rate, self._seed,
                                                          ctx=self._ctx, max_batch=cap,
                                                          init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "GeneralizedMF" + f"_{self.get_base_params_shortcut()}" + f"_{self.get_params_shortcut()}"

    def _recommendation_block(self):
        return max(1, min(4096, (8 * self._model.state.Bmax) // max(self._num_items, 1)))

    def train(self):
        if self._restore:
            return self.restore_weights()
        for it in self.iterate(self._epochs):
            loss, steps = 0, 0
            with tqdm(total=int(self._data.transactions // self._batch_size), disable=not self._verbose) as t:
                for batch in self._sampler.step(self._data.transactions, self._batch_size):
                    steps += 1
                    loss += self._model.train_step(batch)
                    t.update()
            self.evaluate(it, float(loss) / (it + 1))           # :94
