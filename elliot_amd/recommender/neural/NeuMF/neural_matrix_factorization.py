"""NeuMF plugin (YAML key `external.NeuMF`) -- Neural Collaborative Filtering, https://arxiv.org/abs/1708.05031.

Contract of elliot/recommender/neural/NeuMF/neural_matrix_factorization.py:22-124: hyper-parameters `lr`, `mf_factors`,
`dropout`, `is_mf_train`, `is_mlp_train`, `m` (negatives per positive; 0 by default, :67) + base keys; the MLP tower is
(4F, 2F, F) on F-dimensional embeddings (:71-72); `batch_size` < 1 means one batch per epoch; the epoch loss is handed to
evaluate() as sum / (epoch + 1) (:109).
"""
from tqdm import tqdm

from .... import ops
from ...base_recommender_model import BaseRecommenderModel, init_charger, param
from ...recommender_utils_mixin import RecMixin
from . import custom_sampler
from .neural_matrix_factorization_model import NeuralMatrixFactorizationModel


class NeuMF(RecMixin, BaseRecommenderModel):
    @init_charger
    def __init__(self, data, config, params, *args, **kwargs):
        self._params_list = [
            param("lr", "lr", 0.001, attr="_learning_rate"),
            param("mf_factors", "mffactors", 10, int),
            param("dropout", "drop", 0),
            param("is_mf_train", "mftrain", True),
            param("is_mlp_train", "mlptrain", True),
            param("m", "m", 0, int),
        ]
        self.autoset_params()
        F = self._mf_factors
        self._mlp_factors, self._mlp_hidden_size = F, (4 * F, 2 * F, F)
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        epoch_samples = self._data.transactions * (1 + self._m)
        host_side = epoch_samples <= custom_sampler.Sampler.PY_LIMIT   # small epochs: the reference's exact bookkeeping
        self._sampler = custom_sampler.Sampler(self._data.i_train_dict if host_side else None, self._m, ctx=self._ctx,
                                               sp_i_train=self._data.sp_i_train)
        self._model = NeuralMatrixFactorizationModel(self._num_users, self._num_items, F, self._mlp_factors,
                                                     self._mlp_hidden_size, self._dropout, self._is_mf_train,
                                                     self._is_mlp_train, self._learning_rate, self._seed, ctx=self._ctx,
                                                     max_batch=max(min(self._batch_size, 1 << 23), self._num_items),
                                                     init_weights=kwargs.get("init_weights"))

    @property
    def name(self):
        return "_".join(["NeuMF", self.get_base_params_shortcut(), self.get_params_shortcut()])

    def _recommendation_block(self):
        pairs = (8 * self._model.state.Bmax) // max(self._num_items, 1)       # the pair route walks a block in chunks of Bmax pairs
        st = self._model.state
        if st.use_mlp and st.fused_supported(min(self._num_items, 10 + 64)):
            # the fused scoring kernels hold no [users, items] activations: blocks of up to 2^27 pairs (128 users x 1 M items; the
            # screened route's per-pair bounds and candidate regions are 8 bytes a pair) amortise the per-call item-side work
            pairs = max(pairs, (1 << 27) // max(self._num_items, 1))
        return max(1, min(4096, pairs))

    def train(self):
        if self._restore:
            return self.restore_weights()
        batches_per_epoch = int(self._data.transactions * (self._m + 1) // self._batch_size)
        for it in self.iterate(self._epochs):
            epoch_loss = 0
            with tqdm(total=batches_per_epoch, disable=not self._verbose) as bar:
                for pairs in self._sampler.step(self._batch_size):
                    epoch_loss += self._model.train_step(pairs)
                    bar.update()
            self.evaluate(it, float(epoch_loss) / (it + 1))
