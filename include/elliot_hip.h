/*
 * elliot_hip.h -- C ABI of libelliot_hip.so, the MI355X (gfx950) backend for the
 * latent-factor hot path of sisinflab/elliot (BPRMF, BPRMF_batch, MultiVAE, NeuMF/GMF).
 *
 * This header IS the drop-in boundary.  The reference is pure Python; its "FFI" for
 * this path is the set of TensorFlow / NumPy calls made by the model classes.  Each
 * entry point below names the reference call sites (file:line, relative to the
 * reference checkout) that it replaces.  A Python plugin binds these with ctypes
 * (elliot_amd/_lib.py; INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; el_last_error()
 *     returns a thread-local message for the last failure.
 *   - all data pointers are DEVICE pointers (e.g. torch.Tensor.data_ptr() of a
 *     tensor on the ctx's device) unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous on that stream; the library never synchronises unless documented.
 *   - the library never allocates or frees caller buffers.  Scratch space is passed
 *     in explicitly (ws / ws_bytes) and sized with the matching *_ws_bytes() query.
 *   - embedding tables are row-major [rows, F]; CSR index arrays are int64 indptr /
 *     int32 indices with column indices sorted ascending inside each row.
 *   - one el_ctx per device; a ctx is not thread-safe, distinct ctxs may be used
 *     from distinct threads.
 */
#ifndef ELLIOT_HIP_H
#define ELLIOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct el_ctx el_ctx;

#define EL_ABI_VERSION 8   /* 2: el_score_topk_ws_bytes takes excl_nnz; screened top-k, list / metrics / grads entry points
                            * 3: el_pwmf_* (point-wise factor models), el_bprmf_train_loop, el_cml_*
                            * 4: el_bprmf_state ends in uslot / gGu_rows / gGu_cap (a host built against version 3 passes a
                            *    shorter struct: compare el_abi_version() with EL_ABI_VERSION before the first call);
                            *    el_nmf_score_topk, el_gmf_item_image; el_bprmf_state.Gu_next + el_bprmf_train_step_presorted;
                            *    el_host_split_flags_state; el_nmf_state ends in the deferred-decay fields (row_last ..
                            *    batch_n) and the el_nmf_* calls take it non-const; el_nmf_sync_tables; el_bprmf_state ends
                            *    in Gu_last .. lr_hist_cap, el_bprmf_sync_users
                            * 5: el_bprmf_state ends in Gi_last / Gi_defer (item side of the step fused with its
                            *    Adam pass), el_bprmf_sync_items
                            * 6: el_topk_screen_stats (diagnostics of the screened top-k; no struct changes: a host built
                            *    against 5 runs unchanged)
                            * 7: el_bprmf_state ends in replay_series; el_ctx_set_option / el_ctx_get_option (the library no
                            *    longer reads the environment after el_ctx_create); el_graph_csr, el_spmm_csr_f32,
                            *    el_lightgcn_propagate; el_ngcf_*; el_mf2020_train; el_bprmf_ws_bytes / el_cml_ws_bytes take F, el_bprmf_deterministic
                            * 8: el_nmf_state ends in step_ws / step_ws_bytes (el_nmf_step_ws_bytes) + the el_nmf_presort bookkeeping: the NeuMF / GMF step walks its embedding rows
                            *    as sorted segments and sums every batch reduction in a fixed order (no float atomics); training
                            *    calls REQUIRE the workspace; row_stamp / row_own / claim_seq are no longer read; el_pwmf_ws_bytes takes F (the
                            *    point-wise models' segment sums without float atomics); el_nmf_presort, el_nmf_state.replay_series              */

/* ---- context ---------------------------------------------------------------- */

/* Replaces: device selection in elliot/namespace/namespace_model.py:74
 * (CUDA_VISIBLE_DEVICES) -- one ctx per visible MI355X.                          */
int el_ctx_create(int device, el_ctx** out);
int el_ctx_destroy(el_ctx* ctx);
const char* el_last_error(void);
int el_abi_version(void);
/* Fills name[0..len) with the gcnArchName, returns CU count in *cus.             */
int el_device_info(el_ctx* ctx, char* name, int len, int* cus, int64_t* hbm_bytes);

/* Per-kernel timing for bench.py's roofline leg (new; the reference has no profiler).
 * When enabled every kernel launch is bracketed by hipEvents on its launch stream.
 * el_timing_report synchronises them and writes "kernel_name launches total_ms\n" lines. */
int el_timing_enable(el_ctx* ctx, int on);
int el_timing_report(el_ctx* ctx, char* buf, int len);
/* Restricts the bracketing to the launches reported under `kernel_name` (NULL or "": every launch again).  An event between
 * two kernels costs their back-to-back overlap (measured: 16 events per 1.5 ms training step = +4 % wall), so bench.py times
 * its steps with events on the dominant kernel only and takes the per-kernel breakdown from a separate pass. */
int el_timing_filter(el_ctx* ctx, const char* kernel_name);
/* Launches of the optimiser passes made while this is on (the HBM placement tuner of the host layer times the dense Adam
 * pass on scratch tables) carry their own kernel symbols (k_adam_*<..., true>), so that a rocprofv3 kernel trace of a run
 * lists the product launches and the tuner's probes separately. */
int el_tuning_mode(el_ctx* ctx, int on);
/* Switches of the library (ABI 7).  Each has a name ("ichunk", "uchunk", "loop_graph", "gemm_split", "gemm_xcd", "nmf_side",
 * "nmf_head4", "vae_side", "nmf_screen_maxfrac", "screen_stride", "screen_ka", "screen_prof"); el_ctx_create takes its initial
 * value from the environment variable EL_<NAME IN CAPITALS> when that is set, and that is the only time the library reads the
 * environment: afterwards a switch moves through el_ctx_set_option alone (per context: two contexts of one process may differ).
 * Unknown names fail.  The reference has no counterpart (its switches are YAML fields read by the Python layer).              */
int el_ctx_set_option(el_ctx* ctx, const char* name, double value);
int el_ctx_get_option(el_ctx* ctx, const char* name, double* value);

/* ---- BPR triplet sampler (K1) ------------------------------------------------ */

/* Replaces: elliot/dataset/samplers/custom_sampler.py:31-46 (Sampler.step/sample).
 * Counter-based (Philox4x32-10) device sampler.  Sample n (global sample index
 * `first_sample + n`) draws u ~ U[0,U), i ~ U(pos(u)), j ~ U[0,I) rejected while
 * j in pos(u) -- the reference's distribution; the bit stream is Philox, not
 * MT19937 (el_bpr_sample_mt19937 replays that one exactly).  Users with an empty
 * row are re-drawn; users whose row covers [0,I) are re-drawn.
 * When item_lo/item_hi restrict the negative range (item-sharded training,
 * SURVEY 8e) j ~ U[item_lo,item_hi).  Pass 0, I for the reference behaviour.      */
int el_bpr_sample(el_ctx* ctx, void* stream,
                  const int64_t* pos_indptr, const int32_t* pos_indices,
                  int64_t U, int64_t I, int64_t item_lo, int64_t item_hi,
                  uint64_t seed, uint64_t first_sample, int64_t n,
                  int32_t* out_u, int32_t* out_i, int32_t* out_j);

/* The same sampler with a per-user record (64 bytes: row start, row length, a 384-bit membership signature of the row) built
 * once per dataset: a draw then reads one line for its user instead of indptr + a binary search over the row (the search only
 * runs when the candidate's signature bit is set), 800 -> ~320 bytes of cache lines per triplet.  Same Philox stream, same
 * accept / reject decisions, hence bit-identical output.  meta: device, 64-byte aligned, el_bpr_sampler_meta_bytes(U) bytes;
 * el_bpr_sample_meta with meta == NULL is el_bpr_sample.                                                                   */
size_t el_bpr_sampler_meta_bytes(int64_t U);
int el_bpr_sampler_meta_build(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices, int64_t U, void* meta);
int el_bpr_sample_meta(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices, const void* meta,
                       int64_t U, int64_t I, int64_t item_lo, int64_t item_hi, uint64_t seed, uint64_t first_sample, int64_t n,
                       int32_t* out_u, int32_t* out_i, int32_t* out_j);

/* Exact replay of the reference stream: np.random.seed(s) then the draw order of custom_sampler.py:32-41
 * (u, position of i, j repeated while j in pos(u)) with np.random.randint's masked rejection over successive
 * 32-bit MT19937 outputs (no draw when the range is a single value).
 *   mt_state     : device uint32[625] = 624 state words + position (as RandomState.get_state()); updated in
 *                  place, so consecutive calls continue the stream like consecutive Sampler.step batches
 *   lists_indptr / lists_items : per-user positive list IN THE REFERENCE'S ORDER
 *                  (list(set(...)), custom_sampler.py:21), CSR int64/int32
 *   pos_indptr / pos_indices   : the same rows sorted ascending (membership test `j in ui`)
 * ws: el_bpr_sample_mt19937_ws_bytes(n) bytes.  Synchronises the stream once (the number of consumed words
 * determines the new generator state).                                                              */
size_t el_bpr_sample_mt19937_ws_bytes(int64_t n);
int el_bpr_sample_mt19937(el_ctx* ctx, void* stream, uint32_t* mt_state,
                          const int64_t* lists_indptr, const int32_t* lists_items,
                          const int64_t* pos_indptr, const int32_t* pos_indices,
                          int64_t U, int64_t I, int64_t n,
                          int32_t* out_u, int32_t* out_i, int32_t* out_j,
                          void* ws, size_t ws_bytes);

/* ---- BPR-MF, TF semantics (BPRMF_batch; K2-K4) ------------------------------- */

/* Optimiser applied by el_bprmf_train_step. */
enum {
    EL_OPT_ADAM_TF_DENSE = 0, /* Keras-2.3 Adam sparse apply: every row of m, v and
                                 theta moves every step -- THE REFERENCE */
    EL_OPT_ADAM_LAZY = 2,     /* only touched rows decay/move (NOT the reference)     */
    EL_OPT_SGD = 3            /* theta -= lr * grad            (NOT the reference)     */
};

/* Every el_*_state struct must be ZERO-INITIALISED by the host (memset / = {0}) before its fields are set: optional fields
 * (NULL = feature off) switch code paths, and fields added by a later ABI version sit at the END of a struct.            */
typedef struct el_bprmf_state {
    float* Gu;  /* [U,F] user factors      (BPRMF_batch_model.py:41) */
    float* Gi;  /* [I,F] item factors      (BPRMF_batch_model.py:42) */
    float* Bi;  /* [I]   item bias         (BPRMF_batch_model.py:40) */
    float* gGu; /* [U,F] gradient accumulators, zero on entry, zero on exit */
    float* gGi; /* [I,F] */
    float* gBi; /* [I]   */
    float* mGu; float* vGu; /* Adam slots (NULL for EL_OPT_SGD) */
    float* mGi; float* vGi;
    float* mBi; float* vBi;
    int32_t* tGu; /* [U] per-row claim stamps, zero-initialised (LAZY, sparse SGD), else NULL */
    int32_t* tGi; /* [I] */
    int32_t* tBi; /* [I] */
    int64_t U, I;
    int32_t F;
    /* Optional compact user-gradient rows (NULL = the dense accumulator gGu is used; the default).  TensorFlow hands the
     * optimiser IndexedSlices -- one summed row per DISTINCT user of the batch (BPRMF_batch_model.py:77-78) -- and this is
     * that form: the sorted gradient path writes the row of the user whose segment starts at sorted position h to
     * gGu_rows[h, :] and stamps uslot[user] = (step << 32) | h; the TF-dense Adam pass reads a gradient row only where the
     * stamp carries the current step (every other row has g = 0) and nothing is re-zeroed.  HBM traffic of the pass drops
     * from 32 to ~26.5 bytes per parameter at the BASELINE configs[1] shape.  Requires F % 4 == 0, 16-byte aligned tables,
     * EL_OPT_ADAM_TF_DENSE and the SORTED gradient path (gGu may then be NULL); uslot is zero-initialised by the caller and
     * must be zeroed again whenever the step counter is moved backwards.                                             */
    int64_t* uslot;   /* [U] (step << 32) | slot */
    float* gGu_rows;  /* [gGu_cap, F] */
    int64_t gGu_cap;  /* rows of gGu_rows, >= the batch size of every step */
    /* Optional second user table (NULL = off): with it, the SORTED gradient path and EL_OPT_ADAM_TF_DENSE, el_bprmf_train_step /
     * el_bprmf_train_step_presorted run the user side of the step as ONE kernel -- every user row is read once (theta, m, v),
     * the gradient of a row with triplets in the batch is formed in registers from its sorted segment (gathers of gamma_i,
     * gamma_j), Keras' Adam moves the row, and the NEW value goes to Gu_next while Gu keeps the pre-update rows the item-side
     * gradients still have to read.  No gradient row is written or re-read, the batch's user rows are read once instead of
     * twice: ~1 GB of the 6.6 GB a step moves at the BASELINE configs[1] shape.  Same arithmetic in the same order as the
     * two-kernel form (bit-identical tables).  AFTER SUCH A CALL THE CURRENT TABLE IS Gu_next: the caller swaps the two pointers
     * before its next call (el_bprmf_train_loop does it per batch and leaves the current table in Gu when the number of batches
     * is even, in Gu_next when it is odd).  F % 4 == 0, F <= 512, 16-byte aligned tables.                                  */
    float* Gu_next;   /* [U,F] or NULL */
    /* Optional deferred decay of the user table (Gu_last == NULL = off).  Keras' Adam moves EVERY user row at every step --
     * m <- b1 m, v <- b2 v, theta <- theta - lr_t m / (sqrt(v) + eps) -- with or without a gradient.  For a row
     * without triplets in the batch that update reads nothing but the row itself, so it is postponed and replayed in registers --
     * the same fp32 operations on the same operands in the same order, hence the same bits -- when a batch next contains the user
     * (inside the fused user-side kernel, before the row is used) or when el_bprmf_sync_users is called.  A step then moves only
     * the rows of the batch's users: at 10 M users and 1 M triplets per batch a tenth of the 15 GB the every-row pass streams.
     * Every (element, step) update is still performed exactly once.  The update is IN PLACE (Gu_next is not used); the item-side
     * gradients read the pre-update user rows from Gu_old, where the user-side kernel leaves one row per distinct user of the batch.
     * Needs EL_OPT_ADAM_TF_DENSE, the SORTED gradient path, F % 4 == 0, F <= 512, consecutive `step` values from call to call,
     * and el_bprmf_sync_users(step) before anything else reads Gu / mGu / vGu (el_score_topk, el_bprmf_grads, a checkpoint).
     *   Gu_last  int32[U], zero-initialised: the optimiser step each user row is current at
     *   Gu_old   float[Gu_old_cap, F], Gu_old_cap >= the batch size of every step
     *   lr_hist  float[lr_hist_cap], lr_hist_cap a power of two >= 4: the library keeps lr_t of step s at lr_hist[s % cap] and
     *            brings every row up to date by itself every cap / 2 steps                                              */
    int32_t* Gu_last; float* Gu_old; int64_t Gu_old_cap; float* lr_hist; int32_t lr_hist_cap;
    /* Optional fused item side (Gi_last == NULL = off: the item segments write the dense accumulators gGi / gBi and a dense Adam
     * pass reads and clears them).  With Gi_last, the SORTED gradient path and EL_OPT_ADAM_TF_DENSE, the item-segment kernel takes
     * Keras' Adam step on an item row (factors, bias and their slots, IN PLACE) the moment the row's segment is complete -- the
     * gradient row of an item never goes to HBM; a segment cut by a chunk boundary (the popular items of a Zipf catalogue) adds its
     * partial rows into gGi / gBi with atomics as before, and a short second launch takes the step on exactly those rows from the
     * accumulated gradient and clears it.  Rows without an occurrence in the batch still owe Keras' gradient-free update
     * (m <- b1 m, v <- b2 v, theta moves):
     *   Gi_defer == 0  they are brought to step t by a replay pass at the end of every step (every row current after each call);
     *   Gi_defer != 0  they wait (the deferred decay of the user table, applied to the item table): a row is replayed -- bit for bit
     *                  -- at the start of the next step whose batch contains the item (before the user side gathers it) or by
     *                  el_bprmf_sync_items, which must run before anything else reads Gi / Bi / their slots.  Needs lr_hist.
     * Same operations in the same order as the two-pass form: identical bits wherever that form's own summation order is fixed
     * (segments inside one chunk).  gGi / gBi stay required and are zero on entry and exit; F % 4 == 0, 16-byte aligned tables.
     *   Gi_last  int32[I], zero-initialised: the optimiser step each item row (factors + bias) is current at                 */
    int32_t* Gi_last; int32_t Gi_defer;
    /* How a waiting row (Gu_last / Gi_last) is brought forward over its n gradient-free steps:
     *   0  step by step: the same fp32 operations in the same order as Keras' every-row pass, hence the same bits (one IEEE square
     *      root and one IEEE division per element AND step -- at 10 M users the replay arithmetic, not HBM, bounds the user side);
     *   1  in closed form: m_n = b1^n m, v_n = b2^n v, theta_n = theta - m / (sqrt(v) + eps) * S(row), with S built from four
     *      row-level sums over the lr_t history (elliot_amd/csrc/el_common.h: el_adam_series_sums) -- O(1) per element whatever the
     *      gap.  NOT bit-identical to the step-by-step rounding sequence; as close to the exact-arithmetic recurrence as that sequence
     *      is (relative 1.5e-6 of a row's move), inside north_star's 1e-4 on the loss and the oracle tolerances of the parity tests
     *      (tests/test_gpu_fullsize_c4.py runs both modes).  A state keeps one mode from its first step to its last.            */
    int32_t replay_series;
} el_bprmf_state;

/* How the duplicate-row gradient sum (OptimizerV2's segment-sum of IndexedSlices) is formed. */
enum {
    EL_BPR_AUTO = 0,    /* SORTED when B >= 2048 and a large-enough workspace is given        */
    EL_BPR_ATOMIC = 1,  /* one kernel, float atomics into the dense accumulators              */
    EL_BPR_SORTED = 2   /* radix-sort the batch by row, reduce each segment in registers
                           (deterministic summation order; no atomics on hot rows)           */
};

/* Bytes of scratch el_bprmf_train_step needs for the SORTED path with batch size B. */
size_t el_bprmf_ws_bytes(int64_t B, int64_t U, int64_t I, int32_t F);   /* (ABI 7: + F -- the workspace also holds the partial rows of cut item segments) */
/* 1 when the sorted step sums every gradient row in a fixed order (ABI 7: always): no floating-point atomics on table rows -- cut item
 * segments leave per-chunk partial rows that a second launch adds in chunk order -- so two runs on the same batches give the same bits, and
 * the fused / deferred forms equal the every-row two-pass form bit for bit.  (The batch LOSS is still a sum of per-workgroup partials in
 * arrival order: it may differ in its last bits from run to run; no state depends on it.) */
int el_bprmf_deterministic(void);

/* Replaces: BPRMF_batch_model.train_step (BPRMF_batch_model.py:58-80): two gathers
 * (:49-51), x_ui/x_uj (:53), clip + softplus batch SUM (:65-66), L2 terms (:68-72),
 * tape.gradient + Adam.apply_gradients (:77-78; beta1 .9, beta2 .999, eps 1e-7).
 *   u,i,j     : int32[B] triplets
 *   step      : 1-based optimiser iteration t
 *   lr_t      : bias-corrected step size lr*sqrt(1-beta2^t)/(1-beta1^t) (Adam modes)
 *   loss_out  : device double[1]; the batch loss is ADDED to it.                 */
int el_bprmf_train_step(el_ctx* ctx, void* stream, const el_bprmf_state* st,
                        const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                        float lr, float l_w, float l_b, int opt, int32_t step,
                        float lr_t, double* loss_out, int algo, void* ws, size_t ws_bytes);

/* Deferred decay (el_bprmf_state.Gu_last): replays the postponed gradient-free Adam steps of every user row so that Gu, mGu, vGu
 * hold exactly what the every-row form holds after `step` optimiser steps.  No-op when Gu_last is NULL.                     */
int el_bprmf_sync_users(el_ctx* ctx, void* stream, const el_bprmf_state* st, int32_t step);
/* The same for the item table (el_bprmf_state.Gi_last with Gi_defer): Gi, Bi and their Adam slots as the every-row form holds them
 * after `step` optimiser steps.  No-op when Gi_last is NULL.                                                                */
int el_bprmf_sync_items(el_ctx* ctx, void* stream, const el_bprmf_state* st, int32_t step);

/* Self-test of the arithmetic the replay kernels of the deferred decay run (test infrastructure inside the library: the functions
 * under test are device code).  A postponed gradient-free Adam step costs one IEEE square root and one IEEE division per element;
 * the replay kernels take it on packed fp32 instructions with hand-expanded sequences that must return the compiler's bits:
 *   out3[0]  floats in [2^-96, 2^96] -- ALL 1 610 612 736 of them -- whose packed square root differs from sqrtf()
 *   out3[1]  of ~n_pairs pseudo-random (numerator, denominator) pairs over the guard range, half with neighbouring mantissas,
 *            those whose packed quotient differs from `/`
 *   out3[2]  of ~n_pairs random (theta, m, v, lr_t) inside the guard, those where 8 packed steps differ from the reference step
 * out3: device uint64[3].  All three must be 0 (tests/test_gpu_bpr.py::test_packed_replay_arithmetic_is_exact).             */
int el_selftest_replay_math(el_ctx* ctx, void* stream, int64_t n_pairs, uint64_t* out3);

/* ---- BPR-MF across GPUs: item-sharded tables (new design; the reference is single-device) -- */

/* Step 1 on rank r: st holds the replicated user table (Gu [U,F]) and the LOCAL item shard (Gi [I_r,F],
 * Bi [I_r], their accumulators); i/j are LOCAL item ids (both inside the shard).  Computes the batch loss,
 * the item-row gradients (gGi/gBi, sorted segments) and one user-gradient row per triplet:
 *   dU[b,:] = s_b (gamma_i - gamma_j) + l_w gamma_u      (float [B,F], reduced by user after the all-gather)
 * ws: el_bprmf_ws_bytes(B, U, I_r, F).                                                               */
int el_bprmf_shard_grads(el_ctx* ctx, void* stream, const el_bprmf_state* st,
                         const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                         float l_w, float l_b, int32_t step, float* dU, double* loss_out,
                         void* ws, size_t ws_bytes);

/* Replaces: one epoch of BPRMF_batch.train (BPRMF_batch.py:100-109): `for batch in sampler.step(events, B):
 * loss += model.train_step(batch)` for the batches [0,B), [B,2B), ... of `events` samples (the last one may be short),
 * launched back to back from this one call -- el_bpr_sample(first_sample + start) + el_bprmf_train_step per batch.
 *   first_step : optimiser iteration of the first batch (1-based); lr_t_host[k] = bias-corrected step size of batch k
 *                (HOST array of ceil(events / B) floats; read or copied before the call returns -- it may be freed then)
 *   ws         : as for el_bprmf_train_step(B);  loop_ws: el_bprmf_train_loop_ws_bytes(events, B) bytes of device
 *                scratch (triplets of up to 4M samples per sampler launch -- the sampler does not read the model --, the
 *                step-size table, a control block)
 * Same arithmetic and the same Philox stream as the per-batch calls.  Small batches with EL_OPT_ADAM_TF_DENSE on a
 * small model (the reference's ML-1M defaults) replay a hipGraph of 32 captured steps whose per-step scalars live in
 * device memory (EL_LOOP_GRAPH=0 disables it); everything else launches the per-batch kernels eagerly.            */
size_t el_bprmf_train_loop_ws_bytes(int64_t events, int64_t B);
int el_bprmf_train_loop(el_ctx* ctx, void* stream, const el_bprmf_state* st,
                        const int64_t* pos_indptr, const int32_t* pos_indices, uint64_t seed, uint64_t first_sample,
                        int64_t events, int64_t B, float lr, float l_w, float l_b, int opt, int32_t first_step,
                        const float* lr_t_host, double* loss_out, int algo, void* ws, size_t ws_bytes,
                        void* loop_ws, size_t loop_ws_bytes, const void* sampler_meta /* el_bpr_sampler_meta_build, or NULL */);

/* Step 3: out[ids[p],:] += rows[p,:], p in [0,n) -- the gathered (user id, gradient row) pairs of all ranks
 * reduced into the dense accumulator (stable sort by id: every rank sums in the same order).        */
size_t el_rows_segment_sum_ws_bytes(int64_t n, int64_t n_ids);
int el_rows_segment_sum(el_ctx* ctx, void* stream, const int32_t* ids, const float* rows, int64_t n, int32_t F,
                        int64_t n_ids, float* out, void* ws, size_t ws_bytes);

/* "Dense" multi-GPU mode, step 1: gradients of the rank's batch into the dense accumulators gGu [U,F], gGi, gBi
 * (sorted segments, exactly the first half of el_bprmf_train_step) + loss; no optimiser.  The caller then
 * reduce-scatters gGu over the ranks (RCCL), zeroes it, runs el_bprmf_apply on a state that describes its OWN user rows
 * (Gu/mGu/vGu offset to the shard, gGu = the reduce-scatter output, U = shard rows) and all-gathers the updated rows.
 * ws: el_bprmf_ws_bytes(B, U, I, F).                                                                    */
int el_bprmf_grads(el_ctx* ctx, void* stream, const el_bprmf_state* st,
                   const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                   float l_w, float l_b, int32_t step, double* loss_out, void* ws, size_t ws_bytes);

/* el_bprmf_grads in two halves: ordering a batch reads only its triplets, so a multi-GPU step can do it for the NEXT batch
 * while the item-gradient all-reduce is in flight.  el_bprmf_presort fills ws (el_bprmf_ws_bytes(B, U, I, F)) with the sorted
 * (row, triplet) pairs of (u, i, j); el_bprmf_grads_presorted is el_bprmf_grads on the same triplets and the same ws.     */
int el_bprmf_presort(el_ctx* ctx, void* stream, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                     int64_t U, int64_t I, void* ws, size_t ws_bytes);
int el_bprmf_grads_presorted(el_ctx* ctx, void* stream, const el_bprmf_state* st,
                             const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                             float l_w, float l_b, int32_t step, double* loss_out, void* ws, size_t ws_bytes);

/* el_bprmf_train_step on a batch that el_bprmf_presort already ordered into `ws` (the sampler, the key preparation and the radix
 * sort read only the positives and the triplets, so a single GPU prepares the batch of step t + 1 on a side stream while step t
 * runs): segment kernels + loss + optimiser, with the fused user-side kernel when st->Gu_next is given (see el_bprmf_state).  */
int el_bprmf_train_step_presorted(el_ctx* ctx, void* stream, const el_bprmf_state* st, const int32_t* u, const int32_t* i,
                                  const int32_t* j, int64_t B, float lr, float l_w, float l_b, int opt, int32_t step, float lr_t,
                                  double* loss_out, void* ws, size_t ws_bytes);

/* Step 4: optimiser alone on (Gu, local Gi, local Bi); opt = EL_OPT_ADAM_TF_DENSE or EL_OPT_SGD.  */
int el_bprmf_apply(el_ctx* ctx, void* stream, const el_bprmf_state* st, float lr, int opt, int32_t step, float lr_t);

/* ---- BPR-MF, NumPy semantics (BPRMF; K5) ------------------------------------- */

typedef struct el_bprsgd_state {
    double* P;  /* [U,F] user factors  (BPRMF_model.py:53-54) */
    double* Q;  /* [I,F] item factors  (BPRMF_model.py:55-56) */
    double* b;  /* [I]   item bias     (BPRMF_model.py:52)    */
    int64_t U, I;
    int32_t F;
    double lr, reg_bias, reg_user, reg_pos, reg_neg; /* BPRMF.py:63-71 */
} el_bprsgd_state;

/* Replaces: MFModel.update_factors (BPRMF_model.py:91-117), fp64, including the
 * in-place aliasing order (item rows see the UPDATED user row).  Triplets
 * [first, first+n) are applied concurrently: the caller guarantees they are
 * mutually conflict-free (level schedule, el_bprsgd_levels_host) for the result to
 * equal the sequential reference; otherwise the update is Hogwild.               */
int el_bprsgd_apply(el_ctx* ctx, void* stream, const el_bprsgd_state* st,
                    const int32_t* u, const int32_t* i, const int32_t* j,
                    int64_t first, int64_t n);

/* One launch per level: segments [level_start_host[L], level_start_host[L+1]) of the
 * (already level-ordered) device triplet arrays, L = 0..n_levels-1, in stream order. */
int el_bprsgd_apply_levels(el_ctx* ctx, void* stream, const el_bprsgd_state* st,
                           const int32_t* u, const int32_t* i, const int32_t* j,
                           const int64_t* level_start_host, int64_t n_levels);

/* Host helper (no GPU): dependency levels that make concurrent application equal to
 * the sequential order of MFModel.train_step (BPRMF_model.py:87-89).
 *   level[t]   = 1 + max(level of the last earlier triplet sharing u, i or j)
 *   order_host = stable argsort of level (int32[n]); level L (0-based) is
 *                order_host[level_start_host[L] .. level_start_host[L+1])
 * Returns the number of levels in *n_levels (level_start_host holds n_levels+1
 * entries; level_start_cap is its capacity).                                     */
int el_bprsgd_levels_host(const int32_t* u_host, const int32_t* i_host, const int32_t* j_host,
                          int64_t n, int64_t U, int64_t I,
                          int32_t* order_host, int64_t* level_start_host,
                          int64_t level_start_cap, int64_t* n_levels);

/* ---- full-catalog scoring + masked top-k (K6/K7) ------------------------------ */

enum {
    EL_TOPK_AUTO = 0,   /* SCREEN when eligible and a workspace is given, else MFMA, else the wave kernel */
    EL_TOPK_MFMA = 1,   /* force the fp32 kernel, v_mfma_f32_32x32x2_f32 (F<=256, k<=40)                */
    EL_TOPK_SIMPLE = 2, /* force the wave-per-user VALU kernel (any F, k<=4032, candidate protocol)     */
    EL_TOPK_SCREEN = 3, /* force the bf16-screened / fp32-exact kernels (F<=256, k<=128); same results  */
    /* flag, OR-ed into algo: the caller asserts that Gi / Bi have not been written since its previous el_score_topk call
     * with this workspace (scoring block after block of users against one table): the screened kernels then keep the
     * item-side bf16 image of that call instead of deriving it again.  Ignored unless workspace, tables and shape match --
     * and VERIFIED: a 64-bit hash of every element of Gi / Bi (read-only pass on the device, no host synchronisation) is
     * compared with the hash the image was built from; tables updated in place at the same address rebuild the image. */
    EL_TOPK_ITEMS_UNCHANGED = 0x100,
    /* el_nmf_score_topk only: layers 2-3 first on the half-precision matrix instruction with a per-pair error bound (spectral norms
     * of the rounded weights and of their rounding errors, the pair's own activation norms and measured rounding residuals), the
     * fp32 kernel then only on the pairs whose upper bound reaches the user's k-th best lower bound: the same index lists and logit
     * bits.  Needs the workspace of el_nmf_score_ws_bytes(..., with_cand = 2); synchronises the stream once (a 16-byte flag read),
     * so a call with this flag is refused on a stream that is being captured into a graph; takes the unscreened route when the
     * bound leaves more than half of the pairs (el_nmf_screen_stats tells).  With EL_TOPK_ITEMS_UNCHANGED the half-precision item
     * image is kept across calls on one workspace whatever their user ranges and k are (it sits in front of every region sized by
     * them) -- as long as the previous call on that workspace was a screened one; otherwise it is rebuilt.                 */
    EL_NMF_SCREEN = 0x200
};

/* Replaces: BPRMF_batch_model.predict + get_top_k (BPRMF_batch_model.py:83-88) and
 * MFModel.get_user_predictions (BPRMF_model.py:70-85) for users [u_start,u_stop):
 *   score(u,i) = Bi[i] + sum_f Gu[u,f]*Gi[i,f]   (fp32, k-ordered fma chain)
 *   mask       = NOT (item_offset+i in excl row u)      [allunrated_mask, dataset.py:245]
 *                or, when cand_indptr != NULL, (item in cand row u) [val/test mask]
 *   top-k by (score desc, item index asc)  [tf.nn.top_k sorted=True tie rule];
 *   rows with fewer than k unmasked items are padded with -inf and the lowest
 *   masked item indices, as tf.where(mask, preds, -inf) + top_k would return.
 * Gi / Bi describe the LOCAL item shard [item_offset, item_offset+I_local); out_idx
 * holds GLOBAL item indices.  CSR rows are indexed by absolute user id.
 *   out_idx int32[(u_stop-u_start), k], out_val float[(u_stop-u_start), k]
 * ws: el_score_topk_ws_bytes(...) bytes (0 unless the screened kernel can run);
 *     excl_nnz = entries of the exclusion rows [u_start,u_stop) (0 without a mask): the
 *     screened kernel keeps 64 + nnz_u candidate slots per user.  With ws == NULL, AUTO
 *     uses the fp32 kernels.                                                           */
size_t el_score_topk_ws_bytes(int64_t n_users, int64_t I_local, int32_t F, int32_t k, int64_t excl_nnz, int algo);
int el_score_topk(el_ctx* ctx, void* stream,
                  const float* Gu, const float* Gi, const float* Bi,
                  int64_t u_start, int64_t u_stop, int64_t item_offset, int64_t I_local, int32_t F,
                  const int64_t* excl_indptr, const int32_t* excl_indices,
                  const int64_t* cand_indptr, const int32_t* cand_indices,
                  int32_t k, int32_t* out_idx, float* out_val,
                  int algo, void* ws, size_t ws_bytes);

/* Same contract for fp64 tables (BPRMF NumPy model, BPRMF_model.py:70-85).
 * Scores are fp64 k-ordered fma chains; out_val is double.                        */
int el_score_topk_f64(el_ctx* ctx, void* stream,
                      const double* P, const double* Q, const double* b,
                      int64_t u_start, int64_t u_stop, int64_t item_offset, int64_t I_local, int32_t F,
                      const int64_t* excl_indptr, const int32_t* excl_indices,
                      const int64_t* cand_indptr, const int32_t* cand_indices,
                      int32_t k, int32_t* out_idx, double* out_val);

/* Diagnostics of the last SCREENED el_score_topk call on this ctx (synchronises the stream; the call's workspace must still be
 * alive): users of the block, 64-bit records the bf16 pass appended for them (one record = the up-to-16 items of one lane that
 * reached the user's threshold: what the exact fp32 re-scoring starts from), users that took the exact fallback.  Zeros when no
 * screened call has run.  No reference counterpart (the reference materialises every score).                                    */
int el_topk_screen_stats(el_ctx* ctx, void* stream, int64_t* users, int64_t* records, int64_t* flagged_users);

/* Merge G partial top-k lists per user (item shards / item splits) into one, with the
 * same (score desc, index asc) rule.  parts_idx int32[G, n_users, k] (index -1 =
 * empty), parts_val float[G, n_users, k].  New design: the reference has
 * no multi-device path; the merged result equals the single-shard result.         */
int el_topk_merge(el_ctx* ctx, void* stream,
                  const int32_t* parts_idx, const float* parts_val,
                  int32_t G, int64_t n_users, int32_t k,
                  int32_t* out_idx, float* out_val);

/* Top-k over a materialised score block (MultiVAE / NeuMF predict outputs):
 * Replaces get_top_k (multi_vae_model.py:158-159,
 * neural_matrix_factorization_model.py:147-148) on preds float[n_users, I].        */
int el_dense_topk(el_ctx* ctx, void* stream, const float* preds, int64_t ld,
                  int64_t u_start, int64_t u_stop, int64_t I,
                  const int64_t* excl_indptr, const int32_t* excl_indices,
                  const int64_t* cand_indptr, const int32_t* cand_indices,
                  int32_t k, int32_t* out_idx, float* out_val);

/* ---- dense layers: fp32 MFMA GEMM with fused bias + activation (K9, K12) ----------------- */

/* Replaces: keras.layers.Dense forward/backward products of the neural latent-factor models
 * (multi_vae_model.py:44-53,72-78; neural_matrix_factorization_model.py:59-64; tf.matmul).
 *   C[M,N] = act(op(A) op(B) + bias[N]);  act: 0 none, 1 tanh, 2 relu, 3 sigmoid
 *   transA = 0: A is [M,K] row-major (lda >= K); 1: A is stored [K,M] (lda >= M)
 *   transB = 0: B is [K,N] row-major (ldb >= N); 1: B is stored [N,K] (ldb >= K)
 * ws: el_gemm_ws_bytes(...) bytes enable the deterministic split-K path for small M*N.
 * Arithmetic: fp32 operands and results.  Aligned products of 2 GFLOP and more run on v_mfma_f32_32x32x16_bf16 with every
 * operand split into three bf16 planes -- six bf16 products per fp32 product, fp32 accumulation, error at the level of fp32
 * rounding (csrc/el_gemm.hip: k_gemm_b3; tests/test_gpu_dense.py); EL_GEMM_SPLIT=0 in the environment keeps everything on
 * v_mfma_f32_32x32x2_f32.  Neither form promises a summation order (the scoring kernels do: el_score_topk*, el_nmf_score_topk).
 * Non-finite operands: the split form computes the lower planes as a - upper(a), so an INFINITE operand yields NaN planes and the
 * entries it reaches come out NaN where the fp32 instruction returns +-Inf (or NaN: Inf * 0); a NaN operand gives NaN in both
 * forms.  Finite operands never produce a non-finite plane.  (The models' activations and weights are finite; a caller that relies
 * on Inf propagation sets EL_GEMM_SPLIT=0.)                                                                                    */
size_t el_gemm_ws_bytes(el_ctx* ctx, int64_t M, int64_t N, int64_t K);
int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                const float* bias, int act, void* ws, size_t ws_bytes);

/* ---- Mult-VAE (K9-K11) -------------------------------------------------------------------- */

typedef struct el_vae_state {
    int64_t I;       /* items = original_dim                (multi_vae_model.py:101)           */
    int32_t H, L;    /* intermediate_dim, latent_dim         (multi_vae.py:57-58)               */
    int64_t Bmax;    /* rows the activation buffers can hold                                   */
    /* variables, gradients and Adam slots in the order
     * W1[I,H] b1[H] Wmv[H,2L] bmv[2L] W3[L,H] b3[H] W4[H,I] b4[I]
     * (Wmv = [dense_mean.kernel | dense_log_var.kernel], multi_vae_model.py:48-53)            */
    float* w[8];
    float* g[8];     /* gradients (overwritten every step)                                      */
    float* m[8];
    float* v[8];
    /* activations / backward buffers, row-major, Bmax rows */
    float* h;        /* [Bmax,H]   tanh(x~ W1 + b1)             */
    float* mv;       /* [Bmax,2L]  [mu | logvar]                */
    float* z;        /* [Bmax,L]                                */
    float* dz;       /* [Bmax,L]                                */
    float* h2;       /* [Bmax,H]   tanh(z W3 + b3)              */
    float* logits;   /* [Bmax,I]   logits -> dlogits / log_softmax (in place); reused as the dense x~ image */
    float* dh2;      /* [Bmax,H]                                */
    float* dmv;      /* [Bmax,2L]                               */
    float* dh;       /* [Bmax,H]                                */
    float* rnorm;    /* [Bmax]     1/||x_b||                    */
    void* ws;        /* GEMM workspace (may be NULL): max of el_gemm_ws_bytes over the step's products.  With 2 x that (+ 2 x the
                      * 16 (I + 1) + 16 bytes of the sparse first-layer gradient's index, each half rounded up to 256) el_vae_grads /
                      * el_vae_train_step run the weight-gradient products and the bias column sums on the library's second
                      * stream beside the chain that produces the input gradients (same kernels, same results)            */
    size_t ws_bytes;
    int32_t dae;     /* 1: MultiDAE (autoencoders/dae/multi_dae_model.py:19-139): the encoder head is
                      * z = tanh(h Wm + bm) with w[2] = Wm [H,L], w[3] = bm [L] (mv / dmv unused), no sampling,
                      * no KL; eps / anneal are ignored                                                  */
} el_vae_state;

/* Replaces: VariationalAutoEncoder.train_step (multi_vae_model.py:125-142) on the batch whose rows
 * are users rows[0..B) of the train CSR (what sparse_sampler.py:19-25 yields as dense rows).
 *   eps          : device float[B,L] standard-normal draws (Sampling, :28) or NULL for eps = 0
 *   anneal       : KL weight of this step (multi_vae.py:105-108)
 *   dropout_rate : 1 - dropout_pkeep (multi_vae.py:71); the mask is Philox(dropout_seed, step)
 *   lr_t         : bias-corrected Adam step size; loss_out: device double[1], loss is ADDED. */
int el_vae_train_step(el_ctx* ctx, void* stream, const el_vae_state* st,
                      const int64_t* indptr, const int32_t* indices, const int32_t* rows, int64_t B,
                      const float* eps, float anneal, float dropout_rate, uint64_t dropout_seed,
                      int32_t step, float lr_t, double* loss_out);

/* Multi-GPU form (data parallel over the user rows of a batch, weights replicated): el_vae_grads = forward + loss +
 * backward with both batch means (multinomial NLL over rows, KL over rows x latent units) taken over B_global rows (the sum of
 * all ranks' B), the eight gradient buffers st->g[] complete on exit; the caller all-reduces them over RCCL; el_vae_apply =
 * Adam on the ten variables.  grads + apply with B_global = B is el_vae_train_step.                                      */
int el_vae_grads(el_ctx* ctx, void* stream, const el_vae_state* st,
                 const int64_t* indptr, const int32_t* indices, const int32_t* rows, int64_t B, int64_t B_global,
                 const float* eps, float anneal, float dropout_rate, uint64_t dropout_seed, int32_t step, double* loss_out);
int el_vae_apply(el_ctx* ctx, void* stream, const el_vae_state* st, float lr_t);

/* Replaces: VariationalAutoEncoder.predict (multi_vae_model.py:144-155): st->logits[0..B) receives
 * log_softmax(logits) of users rows[0..B) (dropout off; eps as above).                        */
int el_vae_predict(el_ctx* ctx, void* stream, const el_vae_state* st,
                   const int64_t* indptr, const int32_t* indices, const int32_t* rows, int64_t B,
                   const float* eps);

/* ---- NeuMF / GMF (K12-K13) ----------------------------------------------------------------- */

typedef struct el_nmf_state {
    int64_t U, I, Bmax;      /* users, items, samples the activation buffers can hold                   */
    int32_t F, E;            /* mf embedding size, mlp embedding size (NeuMF: E = F)                    */
    int32_t n_layers;        /* hidden Dense(relu) layers of the MLP tower (NeuMF: 3 = 4F, 2F, F)       */
    int32_t use_mf, use_mlp; /* is_mf_train / is_mlp_train (neural_matrix_factorization.py:64-65)       */
    int32_t head_bias;       /* NeuMF predict_layer = Dense(1) has a bias; GMF's edge weight h has none */
    int32_t units[4];
    /* embedding tables U_MF[U,F] I_MF[I,F] U_MLP[U,E] I_MLP[I,E] (+ gradient accumulators, zero on entry
     * and exit, and Adam slots): neural_matrix_factorization_model.py:40-51                            */
    float* tab[4];  float* gtab[4]; float* mtab[4]; float* vtab[4];
    /* Dense layers, Keras layout kernel [in, out], bias [out]                                          */
    float* W[4];  float* b[4];  float* gW[4]; float* gb[4];
    float* mW[4]; float* vW[4]; float* mb[4]; float* vb[4];
    /* head: w [F + units[last]] (GMF: h [F]), optional scalar bias                                     */
    float* hw; float* hb; float* ghw; float* ghb; float* mhw; float* vhw; float* mhb; float* vhb;
    /* activations, Bmax rows */
    float* X0;      /* [Bmax, 2E]  concat(U_MLP[u], I_MLP[i])     */
    float* dX0;     /* [Bmax, 2E]                                 */
    float* MF;      /* [Bmax, F]   U_MF[u] * I_MF[i]              */
    float* dlogit;  /* [Bmax]                                     */
    float* act[4];  /* [Bmax, units[l]]                           */
    float* dact[4];
    void* ws; size_t ws_bytes;   /* GEMM workspace: max of el_gemm_ws_bytes over the tower's products; with 2 x that, el_nmf_grads /
                                  * el_nmf_train_step run the tower's weight-gradient products on the library's second stream
                                  * beside the embedding kernels (same kernels, same results)                              */
    /* keras Dropout(dropout) in front of every Dense of the MLP tower (neural_matrix_factorization_model.py:58-61),
     * active in train_step only: x <- x * keep / (1 - dropout), keep ~ Bernoulli(1 - dropout) from Philox4x32-10 with
     * counter (sample row, column / 4, drop_step, layer) and key drop_seed; the caller advances drop_step per step.
     * TensorFlow's own random stream cannot be reproduced; the distribution is.  0 = off (the reference's default). */
    float dropout; int32_t drop_step; uint64_t drop_seed;
    /* Deferred decay of the embedding tables (optional; row_last[0] == NULL = the eager form: every step streams theta, g, m, v
     * of every row).  Keras' Adam moves EVERY row of an embedding table at every step -- m <- b1 m, v <- b2 v,
     * theta <- theta - lr_t m / (sqrt(v) + eps) -- gradient or not.  For a row without a gradient that update
     * reads nothing but the row itself, so the library postpones it and replays the missed steps in registers -- the same fp32
     * operations on the same operands in the same order, hence the same bits -- when a batch contains the row again or when the
     * tables are read as a whole (el_nmf_forward, el_nmf_score_topk and el_nmf_sync_tables bring every row up to date first).
     * A caller that reads tab[] / mtab[] / vtab[] itself calls el_nmf_sync_tables before; one that all-reduces gtab[] of a
     * replicated table over several ranks (el_nmf_grads / el_nmf_apply) leaves the feature off: the rows of other ranks'
     * samples are not known here.  With it on, every el_nmf_grads is followed by its el_nmf_apply.
     *   row_last[0] int32[U], row_last[1] int32[I]   zero-initialised: the optimiser step the row is current at
     *   row_stamp, row_own                           unused since ABI 8 (may be NULL)
     *   lr_hist      float[lr_hist_cap]              the library records lr_t per step here (cap >= 2; when it is full every
     *                                                row is brought up to date and the history restarts)
     *   hist_base = 1, opt_step = flushed_step = claim_seq = 0, batch_* = NULL/0 at creation; maintained by the library.      */
    int32_t* row_last[2]; int32_t* row_stamp[2]; uint8_t* row_own; float* lr_hist;
    int32_t lr_hist_cap, hist_base, opt_step, flushed_step, claim_seq;
    const int32_t* batch_u; const int32_t* batch_i; int64_t batch_n;
    /* Step workspace (ABI 8; REQUIRED by el_nmf_train_step / el_nmf_grads / el_nmf_apply): el_nmf_step_ws_bytes(ctx, st) bytes of
     * device memory, 256-byte aligned, owned by this state (one per state: el_nmf_apply reads what el_nmf_grads left in it).
     * Holds the sorted (row, sample) keys of the batch, the two factors of the MF product per sample and the partial rows of the
     * batch reductions: the step sorts the batch by embedding row and walks the segments (catch-up + gather one way, gradient sums
     * + Adam the other), and every reduction over the batch (embedding rows, Dense biases, head weights, loss) is added in a fixed
     * order -- two runs from the same state give the same bits.                                                              */
    void* step_ws; size_t step_ws_bytes;
    /* el_nmf_presort bookkeeping (zero at creation; maintained by the library): the batch whose keys are ordered ahead, and which of
     * the workspace's two sort sets the current step uses */
    const int32_t* pre_u; const int32_t* pre_i; int64_t pre_n; int32_t sort_set;
    /* Deferred decay only: 0 = waiting rows are replayed step by step (the bits of Keras' every-row update); 1 = in closed form from
     * four row-level sums over the lr_t history, O(1) per element (as el_bprmf_state.replay_series: inside the parity tolerances). */
    int32_t replay_series;
} el_nmf_state;

/* Bytes of el_nmf_state.step_ws for the state's shape (U, I, Bmax, F, E, n_layers, units, use_mf, use_mlp must be filled in). */
size_t el_nmf_step_ws_bytes(el_ctx* ctx, const el_nmf_state* st);

/* Orders the (embedding row, sample) keys of the batch (u, i) AHEAD of its step (the sort reads u and i only): call it on another
 * stream while the previous step trains; the el_nmf_train_step / el_nmf_grads that follows with the same u, i, n -- arrays unchanged
 * in between, this call complete before the step in stream order (the caller's event) -- skips its own sort.  One batch pending. */
int el_nmf_presort(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n);

/* Replaces: pointwise_pos_neg_sampler.Sampler.step (dataset/samplers/pointwise_pos_neg_sampler.py:26-50):
 * u uniform, fair coin, positive item of u (label 1) or rejected-uniform negative (label 0); Philox stream. */
int el_pointwise_sample(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                        int64_t U, int64_t I, uint64_t seed, uint64_t first_sample, int64_t n,
                        int32_t* out_u, int32_t* out_i, float* out_label);

/* The same draws through the per-user sampler records (el_bpr_sampler_meta_build); meta == NULL is el_pointwise_sample. */
int el_pointwise_sample_meta(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices, const void* meta,
                             int64_t U, int64_t I, uint64_t seed, uint64_t first_sample, int64_t n,
                             int32_t* out_u, int32_t* out_i, float* out_label);

/* Replaces: NeuralMatrixFactorizationModel.get_recs / GeneralizedMatrixFactorizationModel.get_recs on an
 * explicit pair list (neural_matrix_factorization_model.py:120-144): out_prob[b] = sigmoid(...) of (u[b], i[b]). */
int el_nmf_forward(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                   int64_t n, float* out_prob);

/* Replaces: train_step of both models (neural_matrix_factorization_model.py:96-106;
 * generalized_matrix_factorization_model.py:68-79): forward, keras BinaryCrossentropy (batch mean), backward,
 * Adam.  label: float[n] in {0,1}.  loss_out: device double[1], loss is ADDED.                          */
int el_nmf_train_step(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                      const float* label, int64_t n, int32_t step, float lr_t, double* loss_out);

/* Multi-GPU form of the step (data parallel over samples, item tables sharded): el_nmf_grads = forward +
 * loss + backward with the BinaryCrossentropy mean taken over n_global samples (the sum of all ranks' n), every gradient
 * buffer of the state complete on exit; the caller all-reduces the gradients of the replicated variables (user tables,
 * Dense layers, head) over RCCL; el_nmf_apply = Keras Adam on every variable.  grads + apply with n_global = n is
 * el_nmf_train_step.                                                                                            */
int el_nmf_grads(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                 const float* label, int64_t n, int64_t n_global, double* loss_out);
int el_nmf_apply(el_ctx* ctx, void* stream, el_nmf_state* st, int32_t step, float lr_t);

/* Deferred decay (el_nmf_state.row_last): replays the postponed gradient-free Adam steps of every embedding row so that tab[],
 * mtab[], vtab[] hold exactly what the eager form holds after st->opt_step steps.  No-op when the feature is off or nothing is
 * pending.  The state is not const in the el_nmf_* calls: the library keeps its step counters in it.                         */
int el_nmf_sync_tables(el_ctx* ctx, void* stream, el_nmf_state* st);

/* Full-catalogue scoring fused with the masked top-k for NeuMF.
 * Replaces: NeuMF.get_recommendations' [Ub, I] index grids (neural/NeuMF/neural_matrix_factorization.py:111-119) +
 * NeuralMatrixFactorizationModel.get_recs (neural_matrix_factorization_model.py:119-144) + get_top_k (:146-148) for users
 * [u_start, u_stop) against the item shard [item_offset, item_offset + I_local) of the state's tables (rows of tab[1] / tab[3]):
 *   logit(u,i) = w . [Umf[u]*Imf[i] ; MLP([Umlp[u] ; Imlp[i]])] + b       (the network of el_nmf_forward, no Dropout)
 *   out_idx / out_val [n_users, k]: the k best unmasked items by (LOGIT desc, item asc) and their logits; mask / padding
 *   semantics of el_score_topk.  The model's output is sigmoid(logit): apply el_pwmf_link_values(EL_PW_MSE_SIGMOID) to the
 *   list and re-rank with el_topk_rerank where distinct logits collapse to one float (take the list a few entries longer).
 * Layer 1 is evaluated in its separable form (W1[:E]^T u + W1[E:]^T i, two projections), layers 2-3 and the head per pair on
 * fp32 MFMA tiles with the selection fused; nothing of size [users, I] is written.  Numerics: every dot product is the k-ordered
 * fp32 fma chain from +0, `+ bias`, relu; layer 1 = (chain_u + chain_i) + b1; head = two interleaved chains (even / odd
 * positions of [mf ; mlp]) summed, + b, + 0.0f (pinned by oracle/c/el_oracle.c: orc_nmf_logits).
 * Supported: use_mlp with n_layers == 3, units <= (1024, 256, 128), units[2] <= half of units[1] rounded up to {32,64,128,256},
 * E <= 256, F <= 256, k <= 448 (el_nmf_score_supported; anything else: el_nmf_forward on pair lists + el_dense_topk).
 *   flags: EL_TOPK_ITEMS_UNCHANGED = the caller asserts that Imlp and W1 are what the previous call with this workspace
 *          projected (block after block of one evaluation): the [I_local, units[0]] item projection is kept -- verified by a
 *          device-side hash of both arrays, no host synchronisation
 *   ws   : el_nmf_score_ws_bytes(...) bytes, 16-byte aligned (item projection I_local x units[0] x 4 bytes + per-block scratch)  */
int el_nmf_score_supported(const el_nmf_state* st, int32_t k);
/* with_cand: 0 = full catalogue, 1 = candidate lists, 2 = full catalogue with room for the screened route (EL_NMF_SCREEN)        */
size_t el_nmf_score_ws_bytes(el_ctx* ctx, const el_nmf_state* st, int64_t n_users, int64_t I_local, int32_t k, int with_cand);
/* Diagnostics of the last el_nmf_score_topk call on this ctx: pairs the exact fp32 kernel scored (-1 for a candidate-list call),
 * and whether a call that asked for EL_NMF_SCREEN went without it (1: too many survivors, a user short of k, a small shard). */
int el_nmf_screen_stats(el_ctx* ctx, int64_t* exact_pairs, int* fell_back);
int el_nmf_score_topk(el_ctx* ctx, void* stream, el_nmf_state* st, int64_t u_start, int64_t u_stop,
                      int64_t item_offset, int64_t I_local, const int64_t* excl_indptr, const int32_t* excl_indices,
                      const int64_t* cand_indptr, const int32_t* cand_indices, int32_t k, int32_t* out_idx, float* out_val,
                      int flags, void* ws, size_t ws_bytes);

/* GMF (generalized_matrix_factorization_model.py:59-66,81-93): score = sigmoid(sum_f h_f u_f i_f) = sigmoid(<u, i * h>).
 * out[i, f] = Imf[i, f] * hw[f]: the item image that turns GMF's full-catalogue scoring into el_score_topk(Umf, out, NULL)
 * (screened bf16 / exact fp32 kernels), followed by el_pwmf_link_values(EL_PW_MSE_SIGMOID) + el_topk_rerank.            */
int el_gmf_item_image(el_ctx* ctx, void* stream, const float* Imf, const float* hw, int64_t I, int32_t F, float* out);

/* ---- point-wise factor models: MF, PMF, FunkSVD, LogisticMF -----------------------------
 * One kernel family for the reference's TF models that score a (user, item) sample with a dot product of two
 * embedding rows (+ optional user / item bias), push it through a link, and fit it to the sampler's 0/1 label:
 *   EL_PW_MSE          out = <Gu[u],Gi[i]> (+ Bu[u] + Bi[i]);  loss = mean_b (y - out)^2
 *                      MF       latent_factor_models/MF/matrix_factorization_model.py:52-71   (no biases)
 *                      FunkSVD  latent_factor_models/FunkSVD/funk_svd_model.py:62-85          (both biases)
 *   EL_PW_MSE_SIGMOID  out = sigmoid(<Gu[u],Gi[i]>);           loss = mean_b (y - out)^2
 *                      PMF      latent_factor_models/PMF/probabilistic_matrix_factorization_model.py:64-89
 *                      (its GaussianNoise layer is called without training=True, hence inactive)
 *   EL_PW_LOGISTIC     x = <Gu[u],Gi[i]> + Bu[u] + Bi[i];      loss = sum_b -(alpha y x - (1 + alpha y) softplus(x))
 *                                                                     + l_w (|Gu[u_b]|^2 + |Gi[i_b]|^2) / 2
 *                      LogisticMF latent_factor_models/LogisticMF/logistic_matrix_factorization_model.py:52-85
 * (the Keras embeddings_regularizer of MF / PMF / FunkSVD never reaches their tape loss, so those have no L2 term).
 * Duplicate rows of a batch are reduced in sorted segments (stable radix sort -> deterministic sums), as TF's
 * IndexedSlices de-duplication does, then the optimiser runs over the whole table:
 *   EL_PW_ADAM     Keras Adam, TF 2.3 sparse-apply semantics: every row's m, v decay and move each step
 *                  (beta1 .9, beta2 .999, eps 1e-7); slots m*, v*;  lr_t = lr sqrt(1-beta2^t)/(1-beta1^t)
 *   EL_PW_ADAGRAD  Keras Adagrad: acc += g^2, theta -= lr g / (sqrt(acc) + 1e-7); slot m* = acc (start 0.1), lr_t = lr
 * side: EL_PW_BOTH updates user and item variables, EL_PW_ITEMS only (Gi, Bi), EL_PW_USERS only (Gu, Bu) -- LogisticMF
 * alternates the two (logistic_matrix_factorization.py:96-110).                                                  */
enum { EL_PW_MSE = 0, EL_PW_MSE_SIGMOID = 1, EL_PW_LOGISTIC = 2 };
enum { EL_PW_ADAM = 0, EL_PW_ADAGRAD = 1 };
enum { EL_PW_BOTH = 0, EL_PW_ITEMS = 1, EL_PW_USERS = 2 };

typedef struct el_pwmf_state {
    int64_t U, I;
    int32_t F;
    int32_t kind;            /* EL_PW_MSE | EL_PW_MSE_SIGMOID | EL_PW_LOGISTIC                              */
    float alpha, l_w;        /* EL_PW_LOGISTIC only                                                          */
    float* Gu; float* Gi;    /* [U,F], [I,F]                                                                 */
    float* Bu; float* Bi;    /* [U], [I]; both NULL = no bias terms (MF, PMF)                                */
    float* gGu; float* gGi; float* gBu; float* gBi;   /* dense gradient accumulators, zero on entry and exit */
    float* mGu; float* mGi; float* mBu; float* mBi;   /* Adam m / Adagrad accumulator                        */
    float* vGu; float* vGi; float* vBu; float* vBi;   /* Adam v (NULL for Adagrad)                           */
} el_pwmf_state;

size_t el_pwmf_ws_bytes(int64_t n, int64_t U, int64_t I, int32_t F);   /* (ABI 8: takes F -- partial rows of cut segments, summed in a fixed order) */

/* out[b] = link(score of (u[b], i[b])): model.predict / get_recs on an explicit pair list
 * (matrix_factorization_model.py:74-99).  EL_PW_LOGISTIC returns x (predict_batch has no link, :88-89).         */
int el_pwmf_forward(el_ctx* ctx, void* stream, const el_pwmf_state* st, const int32_t* u, const int32_t* i,
                    int64_t n, float* out);

/* One train_step on n samples; label float[n]; step = 1-based optimiser iteration; loss_out: device double[1],
 * the batch loss is ADDED.  ws: el_pwmf_ws_bytes(n, U, I, F).                                                     */
int el_pwmf_train_step(el_ctx* ctx, void* stream, const el_pwmf_state* st, const int32_t* u, const int32_t* i,
                       const float* label, int64_t n, int opt, int side, int32_t step, float lr_t,
                       double* loss_out, void* ws, size_t ws_bytes);

/* Multi-GPU form of the step (data parallel over samples, user rows sharded, item table replicated):
 * el_pwmf_grads = forward + loss + gradient sums with the batch MEAN taken over n_global samples (the sum of all ranks' n;
 * EL_PW_LOGISTIC sums, n_global is ignored), the accumulators of `side` complete on exit; the caller all-reduces gGi / gBi
 * over RCCL; el_pwmf_apply = the optimiser on the variables of `side`.  grads + apply with n_global = n is
 * el_pwmf_train_step.                                                                                                */
int el_pwmf_grads(el_ctx* ctx, void* stream, const el_pwmf_state* st, const int32_t* u, const int32_t* i,
                  const float* label, int64_t n, int64_t n_global, int side, double* loss_out, void* ws, size_t ws_bytes);
int el_pwmf_apply(el_ctx* ctx, void* stream, const el_pwmf_state* st, int opt, int side, int32_t step, float lr_t);

/* Replaces: one sampler pass of a point-wise plugin's epoch (matrix_factorization.py:85-97: `for batch in sampler.step(events,
 * B): loss += model.train_step(batch)`) from ONE call, as el_bprmf_train_loop does for BPRMF_batch: el_pointwise_sample_meta
 * (first_sample + start; up to 4 M draws per launch, sampler_meta may be NULL) + el_pwmf_train_step per batch.
 *   lr_t_host[k]: step size of batch k (Adam: bias-corrected; Adagrad: lr); first_step: optimiser iteration of batch 0
 *   ws: el_pwmf_ws_bytes(B, U, I, F); loop_ws: el_pwmf_train_loop_ws_bytes(events, B).                                        */
size_t el_pwmf_train_loop_ws_bytes(int64_t events, int64_t B);
int el_pwmf_train_loop(el_ctx* ctx, void* stream, const el_pwmf_state* st, const int64_t* pos_indptr, const int32_t* pos_indices,
                       const void* sampler_meta, uint64_t seed, uint64_t first_sample, int64_t events, int64_t B, int opt, int side,
                       int32_t first_step, const float* lr_t_host, double* loss_out, void* ws, size_t ws_bytes,
                       void* loop_ws, size_t loop_ws_bytes);

/* Turn the top-k values of el_score_topk (Bi[i] + <Gu[u],Gi[i]>) into the model's scores, in place:
 * vals[r, c] <- link(vals[r, c] + Bu[u_start + r]) (Bu may be NULL; -inf padding stays -inf).  The link is monotone,
 * so the ranking can only change where distinct inputs collapse to one float -- the host re-ranks those (ops.py). */
int el_pwmf_link_values(el_ctx* ctx, void* stream, float* vals, int64_t n_rows, int64_t ld, int32_t k, int kind,
                        const float* Bu, int64_t u_start);

/* Re-order the first kk <= 4096 entries of every row of (idx int32[n_rows, ld], vals float[n_rows, ld]) in place by
 * (value desc, index asc): the order tf.nn.top_k(sorted=True) returns (matrix_factorization_model.py:100-101) -- applied to
 * lists whose values were transformed after the selection (el_pwmf_link_values, el_cml_rescore).  NaN-free input assumed. */
int el_topk_rerank(el_ctx* ctx, void* stream, int32_t* idx, float* vals, int64_t n_rows, int64_t ld, int32_t kk);

/* ---- Collaborative Metric Learning ---------------------------------------------------------
 * Replaces: CML_model.train_step (latent_factor_models/CML/CML_model.py:69-95) on BPR triplets u,i,j int32[B], with the
 * reference's shapes taken literally: the squared distances keep their [B,1] shape while the squeezed biases are [B], so
 * score = -dist + beta is a [B,B] matrix (:60-66) and the hinge sums over all pairs (triplet a's distances, triplet b's
 * biases):  loss = sum_{a,b} max(margin - clip(D_a + E_b, -80, 1e8), 0) + l_w (|u|^2+|i|^2+|j|^2)/2 + l_b b_i^2/2 + l_b b_j^2/20,
 * D_a = |u_a - j_a|^2 - |u_a - i_a|^2, E_b = b(i_b) - b(j_b).  Evaluated in O(B log B) (two float sorts + binary searches;
 * el_cml.hip).  Variables, gradient accumulators and Adam slots are an el_bprmf_state (Gu, Gi, Bi); optimiser = Keras
 * Adam with TF 2.3 sparse-apply semantics (EL_OPT_ADAM_TF_DENSE).  loss_out: device double[1], ADDED to.            */
size_t el_cml_ws_bytes(int64_t B, int64_t B_all, int64_t U, int64_t I, int32_t F);   /* B_all = B on one GPU */
int el_cml_train_step(el_ctx* ctx, void* stream, const el_bprmf_state* st, const int32_t* u, const int32_t* i,
                      const int32_t* j, int64_t B, float l_w, float l_b, float margin, int32_t step, float lr_t,
                      double* loss_out, void* ws, size_t ws_bytes);

/* Multi-GPU form (data parallel over triplets).  The [B,B] hinge couples EVERY distance of the global batch with
 * EVERY bias difference, so the ranks have a real exchange step: el_cml_forward writes the rank's D_a / E_a (device float[B]) and
 * adds its share of the regulariser; the caller ALL-GATHERS D and E (2 x B floats per rank); el_cml_grads evaluates the rank's
 * triplets against the gathered vectors (D_all / E_all, B_all values: sorted copies + binary searches), adds its share of the
 * hinge sum and leaves the row gradients in the accumulators; item-side gradients are all-reduced, el_bprmf_apply is the
 * optimiser.  forward + grads(D_all = D) + apply on one rank is el_cml_train_step.                                          */
int el_cml_forward(el_ctx* ctx, void* stream, const el_bprmf_state* st, const int32_t* u, const int32_t* i, const int32_t* j,
                   int64_t B, float l_w, float l_b, float* D, float* E, double* loss_out);
int el_cml_grads(el_ctx* ctx, void* stream, const el_bprmf_state* st, const int32_t* u, const int32_t* i, const int32_t* j,
                 int64_t B, float l_w, float l_b, float margin, const float* D, const float* E, const float* D_all,
                 const float* E_all, int64_t B_all, double* loss_out, void* ws, size_t ws_bytes);

/* Replaces: CML_model.predict (:97-102), score(u,i) = -|Gu[u] - Gi[i]|^2 + Bi[i], through the fused scoring kernel:
 * el_cml_prepare_items writes Gi2 = 2 Gi and Bi2 = Bi - |Gi|^2, so that el_score_topk(Gu, Gi2, Bi2) ranks by
 * score + |Gu[u]|^2 (a per-user constant); el_cml_rescore then evaluates the reference's formula directly for the listed
 * candidates (idx int32[n_rows, ld], first kk columns; -1 -> -inf) and the host re-ranks them (ops.py).            */
int el_cml_prepare_items(el_ctx* ctx, void* stream, const float* Gi, const float* Bi, int64_t I, int32_t F,
                         float* Gi2, float* Bi2);
int el_cml_rescore(el_ctx* ctx, void* stream, const float* Gu, const float* Gi, const float* Bi, int32_t F,
                   const int32_t* idx, int64_t n_rows, int64_t ld, int32_t kk, int64_t u_start, float* val);

/* ---- accuracy metrics from the top-k index tensor --------------------------------------
 * Replaces: get_single_recommendation's dict building (recommender_utils_mixin.py:84-88) + Evaluator.eval
 * and its metric classes (evaluation/evaluator.py:117-147; metrics/accuracy/ndcg/ndcg.py:68-125;
 * relevance/relevance.py:49-96; precision.py:66; recall.py:66; hit_rate.py:66; map.py:69-80; mrr.py:63-70;
 * f1.py:56-68) for users [u_start, u_stop):
 *   rec_idx   int32 [u_stop-u_start, ld]  item ids as written by el_score_topk (-1 = none); first `cutoff` used
 *   test CSR  rows by absolute user id, indices ascending, ids in the same id space as rec_idx;
 *             test_ratings float (NULL = 1.0 each); an item is relevant when rating >= threshold
 *   discount  device double[cutoff] = ln 2 / ln(rank + 2) (computed by the host so that both sides share it)
 *   sums      device double[8], ADDED to: sum over the users with >= 1 relevant item of
 *             nDCG, Precision, Recall, HR, MAP, MRR, F1, and the number of such users (means = sums[m] / sums[7])
 *   per_user  optional device double[(u_stop-u_start), 8] with the individual rows (last column 0/1 = counted)
 * fp64 arithmetic; the reduction has a fixed shape (run-to-run identical).  cutoff <= 512.                 */
size_t el_rec_metrics_ws_bytes(int64_t n_users);
int el_rec_metrics(el_ctx* ctx, void* stream, const int32_t* rec_idx, int64_t ld, int64_t u_start, int64_t u_stop,
                   const int64_t* test_indptr, const int32_t* test_indices, const float* test_ratings,
                   double threshold, int32_t cutoff, const double* discount, double* sums, double* per_user,
                   void* ws, size_t ws_bytes);

/* Fragile-user report (BASELINE.md "fragile near-tie users reported separately").  The reference scores
 * with tf.matmul (BPRMF_batch_model.py:83-84), whose fp32 summation order is unknowable here; the kernels pin the k-ordered
 * fma chain.  Any two fp32 evaluations of <u,i> differ by at most F 2^-23 |u||i|, so a user's top-k SET does not depend on
 * the order when  val[k-1] - val[k] >= F 2^-23 |u| max(|i_k|, |i_k+1|).  From [u_stop-u_start, ld] lists of el_score_topk
 * with ld >= k + 1 (ask for k + 1 entries):
 *   counts[0] += users for whom the bound does NOT hold ("fragile"), counts[1] += users with fewer than k + 1 candidates
 *   flags     optional uint8 [u_stop-u_start]: 1 = fragile                                                              */
int el_topk_fragile(el_ctx* ctx, void* stream, const float* Gu, const float* Gi, int32_t F, int64_t u_start, int64_t u_stop,
                    const int32_t* idx, const float* val, int64_t ld, int32_t k, int64_t item_offset,
                    unsigned char* flags, uint64_t* counts);

/* ---- Collectives of the sharded paths (new design; the reference is single-device) --------------------
 * RCCL over xGMI, one communicator per rank = per process = per GPU.  RCCL is bound at run time (the nccl* symbols already in
 * the process, else librccl.so.1): no link-time dependency.  Rank 0 obtains the 128-byte id (el_comm_unique_id) and hands it
 * to the other ranks out of band (file, TCP store, MPI -- the host's business), every rank calls el_comm_init with it.
 * All calls are asynchronous on `stream`; buffers are device pointers.
 *   el_allreduce_rows       buf[0..count) <- sum over ranks, in place (fp32): the item gradients gGi / gBi of the user-sharded
 *                           step, the dense gradients of NeuMF / Mult-VAE -- north_star's "all-reduce of user-row gradients"
 *                           when buf is the dense user-gradient table
 *   el_reduce_scatter_rows  own[0..count_per_rank) <- block `rank` of the element-wise sum of full[0..world*count_per_rank)
 *   el_allgather_rows       full <- concatenation of every rank's `part` (bytes_per_rank bytes each; part may alias its block)
 *   el_allgather_topk       all_idx / all_val [world, n_users, k] <- every rank's partial lists of el_score_topk on its item
 *                           shard; el_topk_merge(all_idx, all_val, G = world) follows                                    */
typedef struct el_comm el_comm;
int el_comm_unique_id(void* id128);
int el_comm_init(el_ctx* ctx, const void* id128, int rank, int world, el_comm** out);
int el_comm_destroy(el_comm* comm);
int el_comm_rank(const el_comm* comm, int* rank, int* world);
int el_allreduce_rows(el_ctx* ctx, el_comm* comm, void* stream, float* buf, int64_t count);
int el_reduce_scatter_rows(el_ctx* ctx, el_comm* comm, void* stream, const float* full, float* own, int64_t count_per_rank);
int el_allgather_rows(el_ctx* ctx, el_comm* comm, void* stream, const void* part, void* full, int64_t bytes_per_rank);
int el_allgather_topk(el_ctx* ctx, el_comm* comm, void* stream, const int32_t* part_idx, const float* part_val,
                      int64_t n_users, int32_t k, int32_t* all_idx, float* all_val);

/* ---- Host-side data plane: ratings file -> split -> id maps, without per-user Python -------------------------
 * Plain CPU functions (no context, no stream, no GPU needed): the two sequential integer loops of the reference's loader that
 * NumPy cannot vectorise, bit-identical to the reference's own output.
 *   el_host_split_flags   replaces Splitter.splitting_randomsubsampling_kfolds / subsampling_list_generator
 *                         (elliot/splitter/base_splitter.py:256-274) and the leave-n-out variant (:276-294):
 *                         seg_len[n_seg] = rows per user in groupby (= ascending user id) order; flags[n_folds][sum seg_len] <- 0
 *                         train / 1 test, laid over each user's rows in file order (fold after fold on one stream).  mode 0: param = test_ratio, train = floor(n (1 - r));
 *                         mode 1: param = n held out.  seed = Splitter.random_seed (np.random.seed(seed), legacy MT19937 stream:
 *                         one shuffle per user, Fisher-Yates from the top with masked-rejection draws).
 *   el_host_pyset_order   replaces `list({k for a in train_dict.values() for k in a.keys()})` (elliot/dataset/dataset.py:202):
 *                         keys[n] = the train items in user-major file order (non-negative ints < 2^61 - 1); out[*n_out] <- the
 *                         distinct keys in the iteration order of the CPython set they were inserted into = the reference's
 *                         private item ids (dataset.py:211-214).  out needs room for n entries.
 *   el_host_negative_sample  replaces NegativeSampler.sample_by_random_uniform (elliot/negative_sampling/negative_sampling.py:
 *                         95-105; strategy "random", num_items): per user random.sample(range(n_candidates), num) on Python's
 *                         `random` MT19937 stream, candidates = the items in neither train nor test, ascending.  excl = CSR of the
 *                         sorted private ids in train or test; py_state625 = random.getstate()[1] (624 key words + position),
 *                         updated in place (hand it back with random.setstate); setsize = the pool / selection-set switch of
 *                         random.sample (21, + 4 ** ceil(log(3 num, 4)) for num > 5); out[n_users * num] <- private ids in sample
 *                         order.  Fails like random.sample (ValueError) when a user has fewer than num candidates.             */
int el_host_negative_sample(const int64_t* excl_indptr, const int32_t* excl_indices, int64_t n_users, int64_t n_items, int32_t num,
                            int64_t setsize, uint32_t* py_state625, int32_t* out);
int el_host_split_flags(const int64_t* seg_len, int64_t n_seg, int mode, double param, uint32_t seed, int32_t n_folds, int8_t* flags);
/* el_host_split_flags on a generator state the caller carries: np_state625 = the legacy MT19937 state (624 key words + position,
 * np.random.get_state()[1:3]; after np.random.seed(s): init_genrand(s), position 624), updated in place.  The reference seeds
 * np.random ONCE per Splitter.process_splitting (base_splitter.py:73); the test split and then the validation split of every test
 * fold's train part (:86-98) continue that one stream -- a host that splits a hierarchy passes the same state to every call.   */
int el_host_split_flags_state(const int64_t* seg_len, int64_t n_seg, int mode, double param, uint32_t* np_state625, int32_t n_folds,
                              int8_t* flags);
int el_host_pyset_order(const int64_t* keys, int64_t n, int64_t* out, int64_t* n_out);

/* ---- graph propagation (LightGCN / NGCF BPR heads; ABI 7) ------------------------------------------------ */

/* The normalised adjacency D^-1/2 A D^-1/2 of the user-item graph over N = U + I nodes (users first), as the reference builds it
 * (graph_based/lightgcn/LightGCN.py:96-118: rowsum + 1e-7, power -1/2, two sparse products, all in fp32), in CSR with ascending column
 * indices -- plus the work decomposition of the product, built once per graph by the host (elliot_amd/ops.py: GraphCSR): every row is
 * cut into chunks of at most 512 consecutive non-zeros (an empty row keeps one empty chunk), one lane group per chunk; a row of several
 * chunks sums their partial rows in chunk order in a second launch (no floating-point atomics: a fixed summation order).            */
typedef struct el_graph_csr {
    const int64_t* indptr;       /* [N + 1] */
    const int32_t* indices;      /* [nnz]   */
    const float* vals;           /* [nnz]   */
    int64_t N, n0;               /* nodes; rows / columns [0, n0) are users (the first table), the rest items (the second) */
    const int32_t* chunk_row;    /* [n_chunks] row of the chunk                                                        */
    const int64_t* chunk_lo;     /* [n_chunks] its first non-zero (it ends 512 further on, or at the end of the row)    */
    const int32_t* chunk_slot;   /* [n_chunks] -1: the only chunk of its row; else its partial row in `part`            */
    int64_t n_chunks;
    const int32_t* multi_row;    /* [n_multi] rows cut into several chunks ...                                          */
    const int32_t* multi_slot;   /* [n_multi] ... their first partial slot (consecutive slots, chunk order) ...         */
    const int32_t* multi_cnt;    /* [n_multi] ... and how many                                                          */
    int64_t n_multi;
    float* part;                 /* [n_partials, F] scratch of the product, F = the widest table it is used with        */
} el_graph_csr;

/* Y = L X for the stacked table X = [X0 (n0 rows); X1 (N - n0 rows)] of width F (a multiple of 4, 16-byte aligned tables).
 * Replaces: tf.sparse.sparse_dense_matmul(A_fold_hat[f], ego_embeddings) over all folds (LightGCN_model.py:78-82, NGCF_model.py:118-121). */
int el_spmm_csr_f32(el_ctx* ctx, void* stream, const el_graph_csr* g, const float* X0, const float* X1, int32_t F, float* Y0, float* Y1);

/* Replaces: LightGCN_model._propagate_embeddings (LightGCN_model.py:68-94), in place:
 *   [Gu; Gi] <- mean over k = 0 .. n_layers of alpha_k L^k [Gu; Gi],  alpha_0 = 1, alpha_k = 1 / (1 + k)
 * (the reference ASSIGNS the result to the variables inside train_step: the BPR step that follows moves the propagated tables, and
 * the next step propagates them again).  ws: el_lightgcn_ws_bytes(U, I, F, n_layers) bytes, 16-byte aligned.                       */
size_t el_lightgcn_ws_bytes(int64_t U, int64_t I, int32_t F, int32_t n_layers);
int el_lightgcn_propagate(el_ctx* ctx, void* stream, const el_graph_csr* g, float* Gu, float* Gi, int32_t F, int32_t n_layers,
                          void* ws, size_t ws_bytes);

/* NGCF (graph_based/ngcf/NGCF_model.py:106-142), the dense half of one embedding-propagation layer around el_spmm_csr_f32 and
 * el_gemm_f32:  el_ngcf_pre   X2 [N, 2k] = [lap + ego | ego * lap]   (the two operands of W_1 and W_2, :123-133, as ONE product's input)
 *               el_ngcf_post  ego' = dropout(leaky_relu(S), rate) -> ego_next [N, kout] and l2_normalize(ego') into columns
 *                             [col_off, col_off + kout) of Gu / Gi (row stride W): the concat + assign of :139-142.  The dropout mask is a
 *                             counter-based draw (seed, step, row, column): TensorFlow's stream cannot be reproduced outside TensorFlow.
 *               el_adam_l2_dense  Keras Adam on a variable whose only gradient is its L2 term (g = two_lw * theta): the GraphLayers,
 *                             which the tape of train_step (:199-215) reaches through reg_loss alone.                              */
int el_ngcf_pre(el_ctx* ctx, void* stream, const float* ego, const float* lap, int64_t N, int32_t k, float* X2);
int el_ngcf_post(el_ctx* ctx, void* stream, const float* S, int64_t N, int64_t n0, int32_t kout, float rate, uint64_t seed, uint32_t step,
                 float* ego_next, float* Gu, float* Gi, int32_t W, int32_t col_off);
int el_adam_l2_dense(el_ctx* ctx, void* stream, float* theta, float* m, float* v, int64_t n, float lr_t, float two_lw);

/* ---- MF2020: point-wise logistic SGD, fp64, strictly sequential (ABI 7) ------------------------------------ */

/* Parameters of latent_factor_models/MF2020/MF_model.py:37-56 in HBM (fp64, as NumPy holds them). */
typedef struct el_mf2020_state {
    double* P;      /* [U,F] _user_factors */
    double* Q;      /* [I,F] _item_factors */
    double* bu;     /* [U]   _user_bias    */
    double* bi;     /* [I]   _item_bias    */
    double* gb;     /* [1]   _global_bias  */
    int64_t U, I;
    int32_t F;
    double lr, reg;
} el_mf2020_state;

/* Replaces: MFModel.train_step (MF2020/MF_model.py:80-113): the (user, item, rating) rows of `samples` (int32 [n,3], the rows
 * custom_sampler_rendle.Sampler.step yields) taken ONE AFTER THE OTHER -- prediction with the global bias, the two stable branches
 * of the logistic loss, the five in-place updates (the item row sees the UPDATED user row: `uf_` is a view, :103-104) -- exactly
 * the reference's order: every sample reads and writes _global_bias, so the chain is strict and one workgroup walks it on LDS
 * (elliot_amd/csrc/el_mf2020.hip).  *loss_out (device double, may be NULL) += the sum of the samples' losses (:108).           */
int el_mf2020_train(el_ctx* ctx, void* stream, const el_mf2020_state* st, const int32_t* samples, int64_t n, double* loss_out);

#ifdef __cplusplus
}
#endif
#endif /* ELLIOT_HIP_H */
