"""What the four point-wise plugins share beyond RecMixin: context, the device sampler, the default batch size."""
from ... import ops
from ...dataset.samplers import pointwise_pos_neg_sampler


class PointwisePluginMixin:
    def _pointwise_setup(self):
        """matrix_factorization.py:60-67 (identical in PMF / FunkSVD / LogisticMF): batch_size < 1 = one batch per epoch;
        pointwise_pos_neg_sampler over the training interactions."""
        if self._batch_size < 1:
            self._batch_size = self._data.transactions
        self._ctx = ops.get_context(max(int(getattr(self._config, "gpu", 0) or 0), 0))
        # `sampler: replay` = the reference's exact (u, i, label) stream (pointwise_pos_neg_sampler.py:33-46), host-replayed
        replay = getattr(self._params, "sampler", "philox") == "replay"
        self._sampler = pointwise_pos_neg_sampler.Sampler(self._data.i_train_dict if replay else self._data.sp_i_train,
                                                          ctx=self._ctx, replay=replay)
