"""NeuMF plugin -- drop-in for elliot/recommender/neural/NeuMF/neural_matrix_factorization.py:22-124
(Neural Collaborative Filtering, https://arxiv.org/abs/1708.05031).  Same YAML keys: lr, mf_factors, dropout,
is_mf_train, is_mlp_train, m (+ base epochs / batch_size / seed / meta)."""
from tqdm import tqdm

from ...base_recommender_model import BaseRecommenderModel, init_charger
from ...recommender_utils_mixin import RecMixin
from .... import ops
from . import custom_sampler as cs
from .neural_matrix_factorization_model import NeuralMatrixFactorizationModel


class NeuMF(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            ("_learning_rate", "lr", "lr", 0.001, None, None),
            ("_mf_factors", "mf_factors", "mffactors", 10, int, None),
            ("_dropout", "dropout", "drop", 0, None, None),
            ("_is_mf_train", "is_mf_train", "mftrain", True, None, None),
            ("_is_mlp_train", "is_mlp_train", "mlptrain", True, None, None),
            ("_m", "m", "m", 0, int, None)
        ]
        self.autoset_params()
        self._mlp_hidden_size = (self._mf_factors * 4, self._mf_factors * 2, self._mf_factors)   # :71
        self._mlp_factors = self._mf_factors                                                     # :72
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        small = self._data.transactions * (1 + self._m) <= cs.Sampler.PY_LIMIT
        self._sampler = cs.Sampler(self._data.i_train_dict if small else None, self._m, ctx=self._ctx,
                                   sp_i_train=self._data.sp_i_train)
        cap = max(min(self._batch_size, 1 << 20), self._num_items)
        self._model = NeuralMatrixFactorizationModel(self._num_users, self._num_items, self._mf_factors,
                                                     self._mlp_factors, self._mlp_hidden_size, self._dropout,
                                                     self._is_mf_train, self._is_mlp_train, self._learning_rate,
                                                     self._seed, ctx=self._ctx, max_batch=cap,
                                                     init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "NeuMF" + f"_{self.get_base_params_shortcut()}" + f"_{self.get_params_shortcut()}"

    def _recommendation_block(self):
        return max(1, min(4096, (8 * self._model.state.Bmax) // max(self._num_items, 1)))

    def train(self):
        if self._restore:
            return self.restore_weights()
        for it in self.iterate(self._epochs):
            loss, steps = 0, 0
            total = int(self._data.transactions * (self._m + 1) // self._batch_size)
            with tqdm(total=total, disable=not self._verbose) as t:
                for batch in self._sampler.step(self._batch_size):
                    steps += 1
                    loss += self._model.train_step(batch)
                    t.update()
            self.evaluate(it, float(loss) / (it + 1))           # :109
