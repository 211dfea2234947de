"""MultiVAE plugin (YAML key `external.MultiVAE`) -- Variational Autoencoders for Collaborative Filtering,
https://arxiv.org/abs/1802.05814.

Contract of elliot/recommender/autoencoders/vae/multi_vae.py:19-115: hyper-parameters `intermediate_dim`, `latent_dim`,
`reg_lambda`, `lr`, `dropout_pkeep` (+ base keys); `batch_size` < 1 means all users (:68-69); dropout rate =
1 - dropout_pkeep (:71); KL weight annealed linearly over 200 000 updates up to 0.2 (:82-84,106-110); the epoch loss is
handed to evaluate() as sum / (epoch + 1) (:115).
"""
from tqdm import tqdm

from .... import ops
from ....dataset.samplers import sparse_sampler
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from .multi_vae_model import VariationalAutoEncoder

ANNEAL_STEPS, ANNEAL_CAP = 200000, 0.2


class MultiVAE(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("intermediate_dim", "intermediate_dim", 600, int),
            param("latent_dim", "latent_dim", 200, int),
            param("reg_lambda", "reg_lambda", 0.01, attr="_lambda"),
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("dropout_pkeep", "dropout_pkeep", 1, attr="_dropout_rate"),
        ]
        self.autoset_params()
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        self._sampler = sparse_sampler.Sampler(self._data.sp_i_train, ctx=self._ctx)
        if self._batch_size < 1:
            self._batch_size = self._num_users
        self._dropout_rate = 1. - self._dropout_rate                 # the YAML key is a keep probability
        self._score_block = min(max(self._batch_size, 1), 2048)
        self._total_anneal_steps, self._anneal_cap = ANNEAL_STEPS, ANNEAL_CAP
        self._model = VariationalAutoEncoder(self._num_items, self._intermediate_dim, self._latent_dim,
                                             self._learning_rate, self._dropout_rate, self._lambda, self._seed,
                                             ctx=self._ctx, train_csr=self._sampler.train,
                                             max_batch=max(self._batch_size, self._score_block),
                                             init_weights=kwargs.get("init_weights"),
                                             eps_mode=getattr(self._params, "eps_mode", "normal"))

    @property
    def name(self):
        return "_".join(["MultiVAE", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        return self._score_block                                     # dense [block, I] log-softmax buffer

    def _anneal(self):
        if self._total_anneal_steps <= 0:
            return self._anneal_cap
        return min(self._anneal_cap, 1. * self._update_count / self._total_anneal_steps)

    def train(self):
        if self._restore:
            return self.restore_weights()
        self._update_count = 0
        batches_per_epoch = int(self._num_users // self._batch_size)
        for it in self.iterate(self._epochs):
            epoch_loss = 0
            with tqdm(total=batches_per_epoch, disable=not self._verbose) as bar:
                for user_rows in self._sampler.step(self._num_users, self._batch_size):
                    epoch_loss += self._model.train_step(user_rows, self._anneal())
                    self._update_count += 1
                    bar.update()
            self.evaluate(it, float(epoch_loss) / (it + 1))
