"""BPRMF plugin -- drop-in for elliot/recommender/latent_factor_models/BPRMF/BPRMF.py:19-129 (per-sample SGD,
fp64).  Same YAML keys and defaults (:63-76); the reference forces batch_size = 1 (:80) and calls train_step
`transactions` times per epoch -- here the epoch's triplets are drawn in one device call and applied in
dependency levels, which yields the same parameters as the sequential loop on the same triplet sequence.
"""
from ....dataset.samplers import custom_sampler as cs
from ...base_recommender_model import BaseRecommenderModel, init_charger
from ...recommender_utils_mixin import RecMixin
from .... import ops
from .BPRMF_model import MFModel


class BPRMF(RecMixin, BaseRecommenderModel):
    r"""Bayesian Personalized Ranking with Matrix Factorization (https://arxiv.org/abs/1205.2618).

    .. code:: yaml

      models:
        external.BPRMF:
          meta:
            save_recs: True
          epochs: 10
          factors: 10
          lr: 0.001
          bias_regularization: 0
          user_regularization: 0.0025
          positive_item_regularization: 0.0025
          negative_item_regularization: 0.0025
    """

    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            ("_factors", "factors", "f", 10, int, None),
            ("_learning_rate", "lr", "lr", 0.05, None, None),
            ("_bias_regularization", "bias_regularization", "bias_reg", 0, None, None),
            ("_user_regularization", "user_regularization", "u_reg", 0.0025, None, None),
            ("_positive_item_regularization", "positive_item_regularization", "pos_i_reg", 0.0025, None, None),
            ("_negative_item_regularization", "negative_item_regularization", "neg_i_reg", 0.00025, None, None),
            ("_update_negative_item_factors", "update_negative_item_factors", "up_neg_i_f", True, None, None),
            ("_update_users", "update_users", "up_u", True, None, None),
            ("_update_items", "update_items", "up_i", True, None, None),
            ("_update_bias", "update_bias", "up_b", True, None, None),   # read but unused, as in the reference
        ]
        self.autoset_params()
        self._batch_size = 1                                           # BPRMF.py:80
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._model = MFModel(self._factors, self._data, self._learning_rate, self._user_regularization,
                              self._bias_regularization, self._positive_item_regularization,
                              self._negative_item_regularization, self._seed, ctx=self._ctx,
                              hogwild=bool(getattr(self._params, "hogwild", False)),
                              init_weights=kwargs.get("init_weights"))
        # `sampler: replay` -> the reference's exact MT19937 triplet stream (seed-exact runs); default: Philox
        if getattr(self._params, "sampler", "philox") == "replay":
            self._sampler = cs.Sampler(self._data.i_train_dict, ctx=self._ctx, replay=True)
        else:
            self._sampler = cs.Sampler(self._data.sp_i_train, ctx=self._ctx)

    @property
    def name(self):
        return "BPRMF" + f"_{self.get_base_params_shortcut()}" + f"_{self.get_params_shortcut()}"

    def _recommendation_block(self):
        return 65536

    def train(self):
        if self._restore:
            return self.restore_weights()
        print(f"Transactions: {self._data.transactions}")
        for it in self.iterate(self._epochs):
            print(f"\n********** Iteration: {it + 1}")
            # one epoch = `transactions` triplets (BPRMF.py:121-127), drawn in one sampler call
            for batch in self._sampler.step(self._data.transactions, self._data.transactions):
                self._model.train_step(batch)
            self.evaluate(it)                                          # loss is not computed by this variant (:129)
