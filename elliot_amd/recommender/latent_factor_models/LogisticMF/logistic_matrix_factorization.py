"""LogisticMatrixFactorization plugin (YAML key `external.LogisticMatrixFactorization`, alias `external.LMF`) --
Logistic Matrix Factorization for Implicit Feedback Data (Johnson, 2014).

Contract of elliot/recommender/latent_factor_models/LogisticMF/logistic_matrix_factorization.py:23-125: hyper-parameters
`lr`, `factors`, `reg`, `alpha` (+ base keys), result-file name "LMF_...", and an epoch made of TWO sampler passes: the
first updates the item side with the users fixed, the second the user side (:96-110); epoch loss = sum over both passes,
handed to evaluate() as sum / (epoch + 1) (:112)."""
from tqdm import tqdm

from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from ..pointwise_plugin import PointwisePluginMixin
from .logistic_matrix_factorization_model import LogisticMatrixFactorizationModel


class LogisticMatrixFactorization(PointwisePluginMixin, RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("factors", "factors", 10),
            param("reg", "reg", 0.1, attr="_l_w"),
            param("alpha", "alpha", 0.5),
        ]
        self.autoset_params()
        self._pointwise_setup()
        self._model = LogisticMatrixFactorizationModel(self._num_users, self._num_items, int(self._factors), self._l_w,
                                                       self._alpha, self._learning_rate, self._seed, ctx=self._ctx,
                                                       init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["LMF", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def train(self):
        if self._restore:
            return self.restore_weights()
        events, bs = self._data.transactions, self._batch_size
        fused = getattr(self._sampler, "philox", False) and getattr(self._config, "fused_epoch", True) and not self._verbose
        for it in self.iterate(self._epochs):
            epoch_loss = 0
            with tqdm(total=int(events * 2 // bs), disable=not self._verbose) as bar:
                for update_users in (False, True):
                    self._model.set_update_user(update_users)
                    if fused:                                         # the same pass inside the library
                        epoch_loss = self._model.train_epoch(self._sampler, events, bs)
                        continue
                    for batch in self._sampler.step(events, bs):
                        epoch_loss += self._model.train_step(batch)
                        bar.update()
            self.evaluate(it, float(epoch_loss) / (it + 1))


LMF = LogisticMatrixFactorization
