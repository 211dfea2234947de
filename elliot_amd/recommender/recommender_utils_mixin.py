"""Plugin surface, part 2: RecMixin (train / evaluate / get_recommendations / bookkeeping).

Same method names, signatures and return shapes as elliot/recommender/recommender_utils_mixin.py:9-136.
The scoring half differs in HOW, not WHAT: instead of `predict` -> dense [Ub, I] block -> `get_top_k` with a
dense bool mask slice (reference :63-88), `get_recommendations` asks the model for fused top-k lists
(`self._model.recommend(mask_csr, k, start, stop)`) and converts the resulting [Ub, k] arrays to the
reference's `{public_user: [(public_item, score), ...]}` dicts in one vectorised pass.
"""
import os

import numpy as np
from tqdm import tqdm

from . import _compat


class RecMixin(object):

    # -- training loop and per-epoch evaluation ------------------------------------------------------------------
    def _epoch_events(self):
        """Samples one epoch draws from the sampler: the train interactions (BPR / point-wise models, BPRMF_batch.py:103);
        the auto-encoders override it with the number of users."""
        return self._data.transactions

    def train(self):
        if self._restore:
            return self.restore_weights()
        events = self._epoch_events()
        fused = (hasattr(self._model, "train_epoch") and getattr(self._sampler, "philox", False)
                 and getattr(self._config, "fused_epoch", True) and not self._verbose)
        for it in self.iterate(self._epochs):
            epoch_loss = 0
            if fused:                                                 # the same loop, inside the library (no per-batch host work)
                epoch_loss = self._model.train_epoch(self._sampler, events, self._batch_size)
            else:
                with tqdm(total=int(events // self._batch_size), disable=not self._verbose) as bar:
                    for batch in self._sampler.step(events, self._batch_size):
                        epoch_loss += self._model.train_step(batch)
                        bar.update()
            self.evaluate(it, float(epoch_loss) / (it + 1))          # the reference's normalisation (BPRMF_batch.py:109)

    def evaluate(self, it=None, loss=0):
        """Called after every epoch (it = epoch index) and once after a restore (it = None)."""
        due = it is None or (it + 1) % self._validation_rate == 0
        if not due:
            return
        needed = self.evaluator.get_needed_recommendations()
        recs = None
        if self._device_metrics():
            outcome = self._evaluate_on_device(needed)
        else:
            recs = self.get_recommendations(needed)
            outcome = self.evaluator.eval(recs)
        self._losses.append(loss)
        self._results.append(outcome)
        self.logger.info("Finished" if it is None else f"Epoch {(it + 1)}/{self._epochs} loss {loss/(it + 1):.5f}")
        if self._save_recs and recs is not None:
            self._write_recs(recs[1], it)
        if self.get_best_arg() == len(self._results) - 1:           # this evaluation is the best so far
            self._on_new_best(it)

    def _write_recs(self, recs_test, it):
        folder = self._config.path_output_rec_result
        self.logger.info(f"Writing recommendations at: {folder}")
        stem = self.name if it is None else f"{self.name}_it={it + 1}"
        _compat.store_recommendation(recs_test, os.path.abspath(os.path.join(folder, stem + ".tsv")))

    def _on_new_best(self, it):
        if it is not None:
            self._params.best_iteration = it + 1
        self.logger.info("******************************************")
        self.best_metric_value = self._validation_value(self._results[-1])
        if not self._save_weights:
            return
        if hasattr(self, "_model"):
            self._model.save_weights(self._saving_filepath)
        else:
            self.logger.warning("Saving weights FAILED. No model to save.")

    # -- metrics on the device --------------------------------------------------------------
    def _device_metrics(self):
        """True when evaluate() can skip the {user: [(item, score)...]} dicts: stand-alone evaluator (inside an Elliot
        process the genuine Evaluator needs the dicts), nothing to write to disk, not switched off in the config."""
        return (getattr(self.evaluator, "supports_device", False) and not self._save_recs
                and getattr(self._config, "device_metrics", True) and hasattr(self, "_model")
                and hasattr(self._model, "recommend"))

    def _evaluate_on_device(self, k):
        def blocks():
            block = self._recommendation_block()
            for offset in range(0, self._num_users, block):
                stop = min(offset + block, self._num_users)
                if not self._negative_sampling:
                    idx, _ = self._model.recommend(self.get_candidate_mask(), k, offset, stop)
                    yield offset, idx, idx
                else:
                    idx_t, _ = self._model.recommend(self.get_candidate_mask(), k, offset, stop)
                    idx_v = idx_t
                    if hasattr(self._data, "val_dict"):
                        idx_v, _ = self._model.recommend(self.get_candidate_mask(validation=True), k, offset, stop)
                    yield offset, idx_v, idx_t
        return self.evaluator.eval_device(self._model.ctx, self._data, blocks())

    # -- scoring -------------------------------------------------------------------------------------
    def get_recommendations(self, k: int = 100):
        predictions_top_k_test, predictions_top_k_val = {}, {}
        block = self._recommendation_block()
        for offset in range(0, self._num_users, block):
            offset_stop = min(offset + block, self._num_users)
            # (the reference computes `predictions = self._model.predict(...)` here and hands the dense [Ub, I] block on;
            #  the fused kernels never materialise it -- the slot stays in the signatures, filled with None)
            recs_val, recs_test = self.process_protocol(k, None, offset, offset_stop)
            predictions_top_k_val.update(recs_val)
            predictions_top_k_test.update(recs_test)
        return predictions_top_k_val, predictions_top_k_test

    def _recommendation_block(self):
        """Users per scoring launch: >= the reference's `batch_size` blocks, large enough to fill 256 CUs."""
        return max(int(self._batch_size) if self._batch_size and self._batch_size > 0 else 0, 65536)

    def process_protocol(self, k, *args):
        if not self._negative_sampling:
            recs = self.get_single_recommendation(self.get_candidate_mask(), k, *args)
            return recs, recs
        val = self.get_single_recommendation(self.get_candidate_mask(validation=True), k, *args) \
            if hasattr(self._data, "val_dict") else {}
        return val, self.get_single_recommendation(self.get_candidate_mask(), k, *args)

    def get_single_recommendation(self, mask, k, predictions, offset, offset_stop):
        """Same signature as the reference's (recommender_utils_mixin.py:84).  `mask` is what get_candidate_mask() returned
        (a tagged device CSR, see below); `predictions` -- the reference's dense score block -- is accepted and ignored: the
        model scores and selects in one fused pass."""
        idx, val = self._model.recommend(mask, k, offset, offset_stop)       # [Ub, k] device tensors
        return self._arrays_to_recs(idx.cpu().numpy(), val.cpu().numpy(), offset, offset_stop)

    def _arrays_to_recs(self, idx, val, offset, offset_stop):
        """recommender_utils_mixin.py:86-88: private -> public ids, `{user: [(item, score), ...]}`.  Rows with fewer than k
        candidates are padded by the kernels with (-1, -inf): those entries are dropped here (index -1 would otherwise wrap to
        the last public item and count as a recommendation of it)."""
        pub_items = self._public_item_array()
        pu = self._data.private_users
        short = idx < 0
        if not short.any():
            il, vl = pub_items[idx].tolist(), val.tolist()
            return {pu[u]: list(zip(il[r], vl[r])) for r, u in enumerate(range(offset, offset_stop))}
        out = {}
        for r, u in enumerate(range(offset, offset_stop)):
            keep = ~short[r]
            out[pu[u]] = list(zip(pub_items[idx[r][keep]].tolist(), val[r][keep].tolist()))
        return out

    def _public_item_array(self):
        if getattr(self, "_pub_items_cache", None) is None:
            pi = self._data.private_items
            arr = np.array([pi[p] for p in range(self._num_items)], dtype=object)
            try:
                arr = arr.astype(np.int64)
            except (TypeError, ValueError):
                pass
            self._pub_items_cache = arr
        return self._pub_items_cache

    def get_candidate_mask(self, validation=False):
        """Reference: dense bool [U, I] (`allunrated_mask` / `val_mask` / `test_mask`, :102-109).  Here: a tagged
        device CSR -- ("excl", train CSR) meaning `True where train == 0`, or ("cand", candidate CSR)."""
        from .masks import device_masks
        m = device_masks(self._data, self._model.ctx)
        if self._negative_sampling:
            cand = m.val if validation else m.test
            if cand is None:
                # an unmasked top-k would silently recommend train items and score against the wrong candidate set
                raise Exception(f"negative_sampling is configured but the data set carries no {'validation' if validation else 'test'} "
                                f"candidate mask ({'val' if validation else 'test'}_mask / _cand_csr)")
            return ("cand", cand)
        return ("excl", m.train)

    def restore_weights(self):
        try:
            self._model.load_weights(self._saving_filepath)
            print("Model correctly Restored")
            self.evaluate()
            return True
        except Exception as ex:
            raise Exception(f"Error in model restoring operation! {ex}")

    # -- what the experiment driver reads back (contracts of the reference's :111-136) ---------------------------
    def _validation_value(self, result):
        return result[self._validation_k]["val_results"][self._validation_metric]

    def get_best_arg(self):
        if self._optimize_internal_loss:
            return np.argmin(self._losses)
        return np.argmax([self._validation_value(r) for r in self._results])

    def get_loss(self):
        """What hyperopt minimises: the internal loss, or minus the best validation metric."""
        if self._optimize_internal_loss:
            return min(self._losses)
        return -max(self._validation_value(r) for r in self._results)

    def get_results(self):
        return self._results[self.get_best_arg()]

    def get_params(self):
        return self._params.__dict__

    def iterate(self, epochs):
        """Epoch indices until the early-stopping rule fires (it is consulted BEFORE every epoch)."""
        done = 0
        while done < epochs and not self._early_stopping.stop(self._losses[:], self._results):
            yield done
            done += 1
        if done < epochs:
            self.logger.info(f"Met Early Stopping conditions: {self._early_stopping}")
