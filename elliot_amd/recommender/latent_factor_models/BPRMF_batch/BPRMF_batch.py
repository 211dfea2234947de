"""BPRMF_batch plugin (YAML key `external.BPRMF_batch`).

Contract of elliot/recommender/latent_factor_models/BPRMF_batch/BPRMF_batch.py:25-120: hyper-parameters `factors`, `lr`,
`l_w`, `l_b` (+ the base `epochs` / `batch_size` / `seed` / `meta`), result-file name "BPRNN_...", batches of
`batch_size` BPR triplets per step, the epoch loss handed to evaluate() as sum / (epoch + 1).  Extra optional keys of this
backend: `optimizer` (adam | adam_lazy | sgd), `sampler` (philox | replay = the reference's exact MT19937 stream), `gpu`.
The training loop itself is RecMixin.train().
"""
from .... import ops
from ....dataset.samplers import custom_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .BPRMF_batch_model import BPRMF_batch_model


class BPRMF_batch(RecMixin, BaseRecommenderModel):
    """Batch BPR matrix factorisation (Rendle et al., https://arxiv.org/abs/1205.2618) on the MI355X kernels."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "factors", 10, int),
            param("lr", "lr", 0.001, float, attr="_learning_rate"),
            param("l_w", "l_w", 0.1, float),
            param("l_b", "l_b", 0.001, float),
        ]
        self.autoset_params()
        if self._batch_size < 1:                                        # "no batch size" = one batch per epoch
            self._batch_size = self._data.transactions
        self._optimizer = getattr(self._params, "optimizer", "adam")
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        replay = getattr(self._params, "sampler", "philox") == "replay"
        positives = self._data.i_train_dict if replay else self._data.sp_i_train
        self._sampler = custom_sampler.Sampler(positives, ctx=self._ctx, replay=replay)
        self._model = BPRMF_batch_model(self._factors, self._learning_rate, self._l_w, self._l_b, self._num_users,
                                        self._num_items, self._seed, ctx=self._ctx, optimizer=self._optimizer,
                                        init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["BPRNN", self.get_base_params_shortcut(), self.get_params_shortcut()])
