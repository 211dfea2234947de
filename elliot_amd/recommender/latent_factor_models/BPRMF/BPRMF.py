"""BPRMF plugin (YAML key `external.BPRMF`): per-sample SGD in fp64, the NumPy-semantics model.

Contract of elliot/recommender/latent_factor_models/BPRMF/BPRMF.py:19-129: the ten hyper-parameters below with the
reference's defaults (:63-76; the four `update_*` switches are read and never used there either), `batch_size` forced to 1
(:80), `transactions` triplets per epoch, no loss value passed to evaluate() (:129).  Here the epoch's triplets come from
one device call and are applied in dependency levels, which gives the parameters of the sequential loop on the same
triplet sequence.  Extra optional keys: `sampler` (philox | replay), `hogwild`, `gpu`.
"""
from .... import ops
from ....dataset.samplers import custom_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .BPRMF_model import MFModel


class BPRMF(RecMixin, BaseRecommenderModel):
    """BPR matrix factorisation, one triplet at a time (Rendle et al., https://arxiv.org/abs/1205.2618)."""

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("factors", "f", 10, int),
            param("lr", "lr", 0.05, attr="_learning_rate"),
            param("bias_regularization", "bias_reg", 0),
            param("user_regularization", "u_reg", 0.0025),
            param("positive_item_regularization", "pos_i_reg", 0.0025),
            param("negative_item_regularization", "neg_i_reg", 0.00025),
            param("update_negative_item_factors", "up_neg_i_f", True),
            param("update_users", "up_u", True),
            param("update_items", "up_i", True),
            param("update_bias", "up_b", True),
        ]
        self.autoset_params()
        self._batch_size = 1
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._model = MFModel(self._factors, self._data, self._learning_rate, self._user_regularization,
                              self._bias_regularization, self._positive_item_regularization,
                              self._negative_item_regularization, self._seed, ctx=self._ctx,
                              hogwild=bool(getattr(self._params, "hogwild", False)),
                              init_weights=kwargs.get("init_weights"))
        if getattr(self._params, "sampler", "philox") == "replay":
            self._sampler = custom_sampler.Sampler(self._data.i_train_dict, ctx=self._ctx, replay=True)
        else:
            self._sampler = custom_sampler.Sampler(self._data.sp_i_train, ctx=self._ctx)

    @property
    def name(self):
        return "_".join(["BPRMF", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        return 65536

    def train(self):
        if self._restore:
            return self.restore_weights()
        n = self._data.transactions
        print(f"Transactions: {n}")
        for it in self.iterate(self._epochs):
            print(f"\n********** Iteration: {it + 1}")
            for epoch_triplets in self._sampler.step(n, n):            # the whole epoch in one sampler call
                self._model.train_step(epoch_triplets)
            self.evaluate(it)
