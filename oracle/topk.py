"""NumPy restatement of the masked top-k protocol.  TEST INFRASTRUCTURE.

Follows (reference checkout paths):
  get_top_k                      BPRMF_batch_model.py:87-88  -> tf.nn.top_k(tf.where(mask, preds, -inf), k, sorted=True)
  get_single_recommendation      recommender_utils_mixin.py:84-88
  allunrated_mask                dataset/dataset.py:245       -> True where the train matrix is 0
"""
import numpy as np

from . import tf_clauses


def dense_mask_from_excl(indptr, indices, u_start, u_stop, n_items, item_offset=0):
    """allunrated_mask[u_start:u_stop] (dataset.py:245) from the train CSR: True = candidate."""
    m = np.ones((u_stop - u_start, n_items), dtype=bool)
    for r, u in enumerate(range(u_start, u_stop)):
        cols = np.asarray(indices[indptr[u]:indptr[u + 1]]) - item_offset
        cols = cols[(cols >= 0) & (cols < n_items)]
        m[r, cols] = False
    return m


def dense_mask_from_cand(indptr, indices, u_start, u_stop, n_items, item_offset=0):
    """val_mask / test_mask rows (dataset.py:230-243): True only on candidate items."""
    m = np.zeros((u_stop - u_start, n_items), dtype=bool)
    for r, u in enumerate(range(u_start, u_stop)):
        cols = np.asarray(indices[indptr[u]:indptr[u + 1]]) - item_offset
        cols = cols[(cols >= 0) & (cols < n_items)]
        m[r, cols] = True
    return m


def get_top_k(preds, mask, k, item_offset=0):
    """tf.nn.top_k(tf.where(mask, preds, -inf), k, sorted=True): values desc, ties -> lower index."""
    masked = np.where(mask, preds, -np.inf)
    n, I = masked.shape
    idx = np.empty((n, k), np.int64)
    val = np.empty((n, k), masked.dtype)
    cols = np.arange(I)
    for r in range(n):
        tie = cols if tf_clauses.get("top_k_ties_lower_index_first") else -cols      # [TF] clause, oracle/tf_clauses.py
        order = np.lexsort((tie, -masked[r]))  # primary: -score asc (= score desc), secondary: index asc
        idx[r] = order[:k] + item_offset
        val[r] = masked[r, order[:k]]
    return val, idx
