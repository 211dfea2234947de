"""Named switches for every clause of the oracle that restates TensorFlow 2.3.2 LIBRARY behaviour from memory of its
documentation / source (the clauses marked "not verifiable here").  TEST INFRASTRUCTURE.

TensorFlow is not installable in the build container (requirements.txt:3 pins tensorflow==2.3.2; no network), so each clause
below is "parity unpinned" until oracle/gen_golden_tf.py has been run on a box that has it: that script executes the reference's
UNMODIFIED model classes on injected weights and fixed batches and writes tests/golden/tf_*.npz; tests/test_tf_pins.py then checks
the oracle -- with these switches at their defaults -- against what TensorFlow really did, clause by clause.  A clause that
turns out wrong is flipped HERE (one line), the product kernel that mirrors it is fixed, and the test pins it from then on.

The oracles read the switches at call time: `tf_clauses.CLAUSES["..."]`.
"""

CLAUSES = {
    # A.3  tf.clip_by_value(x, -80, 1e8) passes the gradient where -80 <= x <= 1e8 (Minimum / Maximum gradients use >= / <=);
    #      False: strictly inside the interval.                                   BPRMF_batch_model.py:65; CML_model.py
    "clip_gradient_inclusive_at_bound": True,
    # A.3  gradients w.r.t. a variable reached through tf.nn.embedding_lookup are IndexedSlices; OptimizerV2 sums duplicate
    #      indices (unsorted_segment_sum) BEFORE the update, so a row seen c times gets ONE Adam step on the summed gradient.
    #      False: one sparse apply per occurrence.                                BPRMF_batch_model.py:77-78
    "indexed_slices_duplicates_summed_before_apply": True,
    # A.4  Keras Adam._resource_apply_sparse (TF 2.3): m <- m*b1 and v <- v*b2 for ALL rows, scatter-add of the batch's rows, then
    #      var <- var - lr_t * m / (sqrt(v) + eps) for ALL rows.  False: "lazy" Adam, only the rows of the batch move.
    "adam_sparse_apply_moves_all_rows": True,
    # A.4  epsilon = 1e-7 (keras.backend.epsilon()) added OUTSIDE the square root, bias correction folded into lr_t =
    #      lr * sqrt(1 - b2^t) / (1 - b1^t).  False: the "epsilon hat" form  lr * m_hat / (sqrt(v_hat) + eps).
    "adam_epsilon_outside_sqrt_with_folded_bias_correction": True,
    # A.4  dense variables use the fused ApplyAdam op: m += (g - m)(1 - b1); v += (g*g - v)(1 - b2).  False: m*b1 + g*(1 - b1).
    "adam_dense_uses_delta_form": True,
    # A.6  tf.nn.top_k(sorted=True): equal values -> the LOWER index first.      BPRMF_batch_model.py:88 and siblings
    "top_k_ties_lower_index_first": True,
    # A.6  a row with fewer than k finite candidates returns -inf entries for the rest, carrying the lowest MASKED indices in
    #      ascending order (the -inf values tie, the tie rule above orders them).
    "top_k_pads_with_lowest_masked_indices": True,
    # A.7  keras.backend.l2_normalize(x, axis=1) = x * rsqrt(max(sum(x^2), 1e-12)).    multi_vae_model.py:42
    "l2_normalize_epsilon_1e12_inside_max": True,
    # A.7  layers.Dropout(rate) in training: kept units scaled by 1 / (1 - rate).      multi_vae_model.py:43
    "dropout_scales_kept_units": True,
    # A.8  keras.losses.BinaryCrossentropy(): predictions clipped to [1e-7, 1 - 1e-7], mean over the batch; the gradient is
    #      zero where the clip is active.                                        neural_matrix_factorization_model.py:72
    "bce_clips_probabilities_at_1e7": True,
    # A.2/A.5  tf.nn.l2_loss(x) = sum(x^2) / 2.                                   BPRMF_batch_model.py:68-72
    "l2_loss_is_half_sum_of_squares": True,
    # A.4  (round 4, found by executing the reference's files on oracle/tf_shim) Adam's (1 - beta) factors are float32 TENSOR
    #      arithmetic: `1 - beta_1_t` with beta_1_t = identity(hyper('beta_1', float32)) in OptimizerV2._prepare_local, `T(1) - beta1`
    #      inside the fused ApplyAdam kernel -- i.e. fl32(1 - fl32(0.999)) = 0.00099998713, not fl32(1 - 0.999) = 0.001 (1.3e-5 apart,
    #      which v carries).  The device kernels always computed `1.0f - 0.999f`.  False: the double-precision difference, rounded.
    "adam_one_minus_beta_in_fp32": True,
    # A.8  (round 4, same source) K.binary_crossentropy on probabilities adds epsilon() INSIDE the logarithms, after the clip:
    #      -(y log(p + 1e-7) + (1 - y) log(1 - p + 1e-7)) -- 2 % of the loss of a saturated prediction (log 2.2e-7 vs log 1.2e-7),
    #      a relative 1e-7 / p elsewhere; the gradient follows the same expression.  False: plain logarithms of the clipped value.
    #      (The other branch of that function -- a y_pred produced DIRECTLY by a Sigmoid op is traced back to its logits -- does not
    #      apply: the models' outputs reach the loss through a nested tf.function call / tf.squeeze.)
    #                                                                           neural_matrix_factorization_model.py:72,102
    "bce_adds_epsilon_inside_log": True,
}


def get(name):
    return CLAUSES[name]


def one_minus(beta):
    """(1 - beta) as Keras' Adam forms it (clause adam_one_minus_beta_in_fp32), a numpy float32."""
    import numpy as np
    if CLAUSES["adam_one_minus_beta_in_fp32"]:
        return np.float32(1.0) - np.float32(beta)
    return np.float32(1.0 - beta)
