"""MultiVAE plugin -- drop-in for elliot/recommender/autoencoders/vae/multi_vae.py:19-115
(Variational Autoencoders for Collaborative Filtering, https://arxiv.org/abs/1802.05814).
Same YAML keys: intermediate_dim, latent_dim, reg_lambda, lr, dropout_pkeep (+ epochs / batch_size / seed / meta).
"""
from tqdm import tqdm

from ....dataset.samplers import sparse_sampler as sp
from ...base_recommender_model import BaseRecommenderModel, init_charger
from ...recommender_utils_mixin import RecMixin
from .... import ops
from .multi_vae_model import VariationalAutoEncoder


class MultiVAE(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            ("_intermediate_dim", "intermediate_dim", "intermediate_dim", 600, int, None),
            ("_latent_dim", "latent_dim", "latent_dim", 200, int, None),
            ("_lambda", "reg_lambda", "reg_lambda", 0.01, None, None),
            ("_learning_rate", "lr", "lr", 0.001, None, None),
            ("_dropout_rate", "dropout_pkeep", "dropout_pkeep", 1, None, None),
        ]
        self.autoset_params()
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._sampler = sp.Sampler(self._data.sp_i_train, ctx=self._ctx)
        if self._batch_size < 1:
            self._batch_size = self._num_users                      # multi_vae.py:68-69
        self._dropout_rate = 1. - self._dropout_rate                # :71 (pkeep -> rate)
        self._score_block = min(max(self._batch_size, 1), 2048)
        self._model = VariationalAutoEncoder(self._num_items, self._intermediate_dim, self._latent_dim,
                                             self._learning_rate, self._dropout_rate, self._lambda, self._seed,
                                             ctx=self._ctx, train_csr=self._sampler.train,
                                             max_batch=max(self._batch_size, self._score_block),
                                             init_weights=kwargs.get("init_weights"),
                                             eps_mode=getattr(self._params, "eps_mode", "normal"))
        self._total_anneal_steps = 200000                           # :82
        self._anneal_cap = 0.2                                      # :84

    @property
    def name(self):
        return "MultiVAE" + f"_{self.get_base_params_shortcut()}" + f"_{self.get_params_shortcut()}"

    def _recommendation_block(self):
        return self._score_block                                    # dense [block, I] log-softmax buffer

    def train(self):
        if self._restore:
            return self.restore_weights()
        self._update_count = 0
        for it in self.iterate(self._epochs):
            loss, steps = 0, 0
            with tqdm(total=int(self._num_users // self._batch_size), disable=not self._verbose) as t:
                for batch in self._sampler.step(self._num_users, self._batch_size):
                    steps += 1
                    if self._total_anneal_steps > 0:
                        anneal = min(self._anneal_cap, 1. * self._update_count / self._total_anneal_steps)
                    else:
                        anneal = self._anneal_cap
                    loss += self._model.train_step(batch, anneal)
                    t.update()
                    self._update_count += 1
            self.evaluate(it, float(loss) / (it + 1))               # :115
