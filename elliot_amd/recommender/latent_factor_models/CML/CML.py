"""CML plugin (YAML key `external.CML`) -- Collaborative Metric Learning (Hsieh et al., WWW 2017).

Contract of elliot/recommender/latent_factor_models/CML/CML.py:22-127: hyper-parameters `factors` (default 100), `lr`, `l_w`,
`l_b`, `margin` (+ base keys), result-file name "CML_...", BPR triplets from custom_sampler in batches of `batch_size`, the
epoch loss handed to evaluate() as sum / (epoch + 1) (:113).  The training loop is RecMixin.train()."""
from .... import ops
from ....dataset.samplers import custom_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .CML_model import CML_model


class CML(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "factors", 100, attr="_user_factors"),
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("l_w", "l_w", 0.001),
            param("l_b", "l_b", 0.001),
            param("margin", "margin", 0.5),
        ]
        self.autoset_params()
        self._item_factors = self._user_factors
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._sampler = custom_sampler.Sampler(self._data.sp_i_train, ctx=self._ctx)
        self._model = CML_model(int(self._user_factors), int(self._item_factors), self._learning_rate, self._l_w, self._l_b,
                                self._margin, self._num_users, self._num_items, self._seed, ctx=self._ctx,
                                init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["CML", self.get_base_params_shortcut(), self.get_params_shortcut()])
