"""MF2020 plugin (YAML key `external.MF2020`).

Contract of elliot/recommender/latent_factor_models/MF2020/MF.py:23-160: hyper-parameters `factors` (10), `lr` (0.05), `reg` (0), `m` (0
negatives per positive); the "batch size" of 100 000 is only the granularity of the progress display (:69-70) -- every sample is its own
SGD step; per epoch the mean of the batches' mean losses is handed to evaluate() as loss / (epoch + 1) (:123-127).  Extra optional key: `gpu`.
"""
from tqdm import tqdm

from .... import ops
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from . import custom_sampler_rendle as ps
from .MF_model import MFModel


class MF2020(RecMixin, BaseRecommenderModel):
    """Matrix factorisation as in "Neural Collaborative Filtering vs. Matrix Factorization Revisited" (https://dl.acm.org/doi/pdf/10.1145/3383313.3412488)."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "f", 10, int),
            param("lr", "lr", 0.05, attr="_learning_rate"),
            param("reg", "reg", 0, attr="_regularization"),
            param("m", "m", 0, int),
        ]
        self.autoset_params()
        self._ratings = self._data.train_dict
        self._sampler = ps.Sampler(self._data.i_train_dict, self._m, self._data.sp_i_train, self._seed)
        self._batch_size = 100000
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._model = MFModel(self._factors, self._data, self._learning_rate, self._regularization, self._seed, ctx=self._ctx,
                              init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["MF2020", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        return 65536

    def train(self):
        if self._restore:
            return self.restore_weights()
        print(f"Transactions: {self._data.transactions}")
        for it in self.iterate(self._epochs):
            print(f"\n********** Iteration: {it + 1}")
            loss, steps = 0, 0
            with tqdm(total=int(self._data.transactions * (self._m + 1) // self._batch_size), disable=not self._verbose) as t:
                for batch in self._sampler.step(self._batch_size):
                    steps += 1
                    loss += self._model.train_step(batch) / len(batch)
                    t.update()
            self.evaluate(it, loss / (it + 1))
