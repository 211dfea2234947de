/*
 * el_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's scoring + masked top-k for the latent-factor path:
 *   BPRMF_batch_model.predict / get_top_k   (BPRMF_batch_model.py:83-88)
 *   MFModel.get_user_predictions            (BPRMF_model.py:70-85)
 *   RecMixin mask semantics                 (recommender_utils_mixin.py:75-88,102-109;
 *                                            dataset.py:245  allunrated_mask = (train == 0))
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Pinned conventions (DESIGN.md "numerics contract"):
 *   dot   = fmaf chain over f = 0..F-1 starting from +0  (tf.matmul's own summation order is
 *           Eigen-internal and unknowable without TF; see tests for the fp64 cross-check)
 *   score = (Bi ? dot + Bi[i] : dot) + 0.0f
 *   order = score desc, item index asc (tf.nn.top_k sorted=True); NaN never selected;
 *           rows with < k unmasked items are padded with -inf entries carrying the lowest
 *           masked item indices (what top_k(where(mask, preds, -inf)) returns).
 * Parity status: the tie rule / -inf padding follow TF's documented top_k contract, which no
 * reference test pins ("parity unpinned" for that clause).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>


static int64_t lower_bound(const int32_t* idx, int64_t lo, int64_t hi, int64_t x) {
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)idx[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* fmaf-chain scores for users [u0,u1) x items [0,I): out[(u-u0)*I + i] */
void orc_scores_f32(const float* Gu, const float* Gi, const float* Bi, int64_t u0, int64_t u1, int64_t I,
                    int32_t F, float* out) {
    for (int64_t u = u0; u < u1; ++u) {
        const float* gu = Gu + u * F;
        float* o = out + (u - u0) * I;
        for (int64_t i = 0; i < I; ++i) {
            const float* gi = Gi + i * F;
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc = fmaf(gi[f], gu[f], acc);
            o[i] = (Bi ? acc + Bi[i] : acc) + 0.0f;
        }
    }
}

void orc_scores_f64(const double* P, const double* Q, const double* b, int64_t u0, int64_t u1, int64_t I,
                    int32_t F, double* out) {
    for (int64_t u = u0; u < u1; ++u) {
        const double* pu = P + u * F;
        double* o = out + (u - u0) * I;
        for (int64_t i = 0; i < I; ++i) {
            const double* qi = Q + i * F;
            double acc = 0.0;
            for (int f = 0; f < F; ++f) acc = fma(qi[f], pu[f], acc);
            o[i] = (b ? acc + b[i] : acc) + 0.0;
        }
    }
}

/* better(a,b): a ranks before b */
static int better_f32(float sa, int32_t ia, float sb, int32_t ib) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ia < ib;
}
static int better_f64(double sa, int32_t ia, double sb, int32_t ib) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ia < ib;
}

/* r-th masked item (ascending) of one row inside [off, off+I): -1 if none */
static int32_t fill_masked(const int32_t* excl, int64_t e0, int64_t e1, const int32_t* cand, int64_t c0, int64_t c1,
                           int use_cand, int64_t off, int64_t I, int64_t r) {
    int64_t end = off + I;
    if (use_cand) {
        /* masked = every item that is not a candidate */
        int64_t seen = 0;
        int64_t q = lower_bound(cand, c0, c1, off);
        for (int64_t g = off; g < end; ++g) {
            while (q < c1 && cand[q] < g) ++q;
            if (q < c1 && cand[q] == g) continue;
            if (seen == r) return (int32_t)g;
            ++seen;
        }
        return -1;
    }
    if (excl) {
        int64_t lo = lower_bound(excl, e0, e1, off), hi = lower_bound(excl, e0, e1, end);
        return (lo + r < hi) ? excl[lo + r] : -1;
    }
    return -1;
}

/*
 * Masked top-k of one score row per user.
 *   scores : [n_users, ld] local item columns 0..I-1 (global item = item_offset + column)
 *   excl_* : CSR by ABSOLUTE user id, global item ids, mask = NOT in row   (may be NULL)
 *   cand_* : CSR by ABSOLUTE user id, global item ids, mask = in row       (may be NULL; wins)
 */
#define DEFINE_TOPK(NAME, T, BETTER, NEGINF)                                                                   \
    void NAME(const T* scores, int64_t ld, int64_t u_start, int64_t n_users, int64_t item_offset, int64_t I,    \
              const int64_t* excl_indptr, const int32_t* excl_idx, const int64_t* cand_indptr,                  \
              const int32_t* cand_idx, int32_t k, int32_t* out_idx, T* out_val) {                               \
        T* bs = (T*)malloc(sizeof(T) * (size_t)k);                                                              \
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);                                            \
        for (int64_t r = 0; r < n_users; ++r) {                                                                 \
            int64_t user = u_start + r;                                                                         \
            const T* s = scores + r * ld;                                                                       \
            int64_t e0 = 0, e1 = 0, c0 = 0, c1 = 0;                                                             \
            if (excl_indptr) { e0 = excl_indptr[user]; e1 = excl_indptr[user + 1]; }                            \
            if (cand_indptr) { c0 = cand_indptr[user]; c1 = cand_indptr[user + 1]; }                            \
            int n = 0;                                                                                          \
            int64_t ep = excl_indptr ? lower_bound(excl_idx, e0, e1, item_offset) : 0;                          \
            int64_t cp = cand_indptr ? lower_bound(cand_idx, c0, c1, item_offset) : 0;                          \
            for (int64_t i = 0; i < I; ++i) {                                                                   \
                int64_t g = item_offset + i;                                                                    \
                int ok;                                                                                         \
                if (cand_indptr) {                                                                              \
                    while (cp < c1 && cand_idx[cp] < g) ++cp;                                                   \
                    ok = (cp < c1 && cand_idx[cp] == g);                                                        \
                } else if (excl_indptr) {                                                                       \
                    while (ep < e1 && excl_idx[ep] < g) ++ep;                                                   \
                    ok = !(ep < e1 && excl_idx[ep] == g);                                                       \
                } else ok = 1;                                                                                  \
                T v = s[i];                                                                                     \
                if (!ok || v != v) continue;                                                                    \
                if (n == k && !BETTER(v, (int32_t)g, bs[k - 1], bi[k - 1])) continue;                           \
                int p = (n < k) ? n : k - 1;                                                                    \
                while (p > 0 && BETTER(v, (int32_t)g, bs[p - 1], bi[p - 1])) {                                  \
                    bs[p] = bs[p - 1]; bi[p] = bi[p - 1]; --p;                                                  \
                }                                                                                               \
                bs[p] = v; bi[p] = (int32_t)g;                                                                  \
                if (n < k) ++n;                                                                                 \
            }                                                                                                   \
            for (int t = 0; t < k; ++t) {                                                                       \
                if (t < n) { out_idx[r * k + t] = bi[t]; out_val[r * k + t] = bs[t]; }                          \
                else {                                                                                          \
                    out_idx[r * k + t] = fill_masked(excl_idx, e0, e1, cand_idx, c0, c1, cand_indptr != 0,      \
                                                     item_offset, I, t - n);                                    \
                    out_val[r * k + t] = NEGINF;                                                                \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
        free(bs); free(bi);                                                                                     \
    }

DEFINE_TOPK(orc_topk_rows_f32, float, better_f32, -INFINITY)
DEFINE_TOPK(orc_topk_rows_f64, double, better_f64, -INFINITY)

/* fused convenience: score block + top-k, processing `ub` users at a time (the reference's
 * get_recommendations loop, BPRMF_batch.py:111-120) */
void orc_score_topk_f32(const float* Gu, const float* Gi, const float* Bi, int64_t u_start, int64_t u_stop,
                        int64_t item_offset, int64_t I, int32_t F, const int64_t* excl_indptr,
                        const int32_t* excl_idx, const int64_t* cand_indptr, const int32_t* cand_idx, int32_t k,
                        int32_t* out_idx, float* out_val) {
    const int64_t ub = 64;
    float* blk = (float*)malloc(sizeof(float) * (size_t)(ub * (I > 0 ? I : 1)));
    for (int64_t s = u_start; s < u_stop; s += ub) {
        int64_t e = s + ub < u_stop ? s + ub : u_stop;
        orc_scores_f32(Gu, Gi, Bi, s, e, I, F, blk);
        orc_topk_rows_f32(blk, I, s, e - s, item_offset, I, excl_indptr, excl_idx, cand_indptr, cand_idx, k,
                          out_idx + (s - u_start) * k, out_val + (s - u_start) * k);
    }
    free(blk);
}

void orc_score_topk_f64(const double* P, const double* Q, const double* b, int64_t u_start, int64_t u_stop,
                        int64_t item_offset, int64_t I, int32_t F, const int64_t* excl_indptr,
                        const int32_t* excl_idx, const int64_t* cand_indptr, const int32_t* cand_idx, int32_t k,
                        int32_t* out_idx, double* out_val) {
    const int64_t ub = 64;
    double* blk = (double*)malloc(sizeof(double) * (size_t)(ub * (I > 0 ? I : 1)));
    for (int64_t s = u_start; s < u_stop; s += ub) {
        int64_t e = s + ub < u_stop ? s + ub : u_stop;
        orc_scores_f64(P, Q, b, s, e, I, F, blk);
        orc_topk_rows_f64(blk, I, s, e - s, item_offset, I, excl_indptr, excl_idx, cand_indptr, cand_idx, k,
                          out_idx + (s - u_start) * k, out_val + (s - u_start) * k);
    }
    free(blk);
}

/*
 * NeuMF full-catalogue logits (neural/NeuMF/neural_matrix_factorization_model.py:75-93 `call`, :119-144 `get_recs`):
 *   mf   = Umf[u] * Imf[i];  mlp = relu-Dense chain on concat(Umlp[u], Imlp[i]) (3 layers);  logit = w . [mf ; mlp] + b
 * (the model's output is sigmoid(logit); the ranking is taken on the logit, the link is applied to the survivors).
 * Pinned summation order (= what elliot_amd/csrc/el_nmf_score.hip computes; TF's own order inside tf.matmul is unknowable
 * without TF -- "parity unpinned" for the order, checked against fp64 maths in tests/test_oracle_neumf.py):
 *   layer 1   pu[k] = fmaf chain over e of Umlp[u][e] W1[e][k],  pi[k] = chain over e of Imlp[i][e] W1[E+e][k]  (both from +0),
 *             a1[k] = max((pu[k] + pi[k]) + b1[k], 0)                              (Dense on a concat, evaluated separably)
 *   layer l   a[m]  = max(chain over k of W[k][m] a_prev[k] (from +0) + b[m], 0)
 *   head      two interleaved chains over the positions p of x = [mf ; a3]: acc_(p & 1) = fmaf(w[p], x[p], acc_(p & 1)),
 *             p ascending, mf[f] = Umf[u][f] * Imf[i][f] rounded to fp32 first; note the parity is that of f inside mf and of
 *             the feature index inside a3 (F even in every configuration of the reference; the kernel pairs the same way)
 *             logit = ((acc_0 + acc_1) + b) + 0.0f
 * out[(r * n_items) + (i - i0)] for users[r], items [i0, i1).
 */
static void dense_chain(const float* W, const float* a, int K, int N, float* acc) {
    for (int n = 0; n < N; ++n) acc[n] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float ak = a[k];
        const float* w = W + (int64_t)k * N;
        for (int n = 0; n < N; ++n) acc[n] = fmaf(w[n], ak, acc[n]);
    }
}

void orc_nmf_logits(const float* Umf, const float* Imf, const float* Umlp, const float* Imlp, int32_t F, int32_t E,
                    const float* W1, const float* b1, int32_t H1, const float* W2, const float* b2, int32_t H2,
                    const float* W3, const float* b3, int32_t H3, const float* hw, const float* hb,
                    const int64_t* users, int64_t n_users, int64_t i0, int64_t i1, float* out) {
    const int64_t nI = i1 - i0;
    for (int64_t r = 0; r < n_users; ++r) {
        const int64_t u = users[r];
        float* pu = (float*)malloc(sizeof(float) * (size_t)H1);
        dense_chain(W1, Umlp + u * E, E, H1, pu);
#pragma omp parallel
        {
            float* pi = (float*)malloc(sizeof(float) * (size_t)H1);
            float* a1 = (float*)malloc(sizeof(float) * (size_t)H1);
            float* a2 = (float*)malloc(sizeof(float) * (size_t)H2);
            float* a3 = (float*)malloc(sizeof(float) * (size_t)H3);
#pragma omp for schedule(static)
            for (int64_t i = i0; i < i1; ++i) {
                dense_chain(W1 + (int64_t)E * H1, Imlp + i * E, E, H1, pi);
                for (int k = 0; k < H1; ++k) {
                    const float t = (pu[k] + pi[k]) + b1[k];
                    a1[k] = t > 0.0f ? t : 0.0f;
                }
                dense_chain(W2, a1, H1, H2, a2);
                for (int m = 0; m < H2; ++m) {
                    const float t = a2[m] + b2[m];
                    a2[m] = t > 0.0f ? t : 0.0f;
                }
                dense_chain(W3, a2, H2, H3, a3);
                for (int m = 0; m < H3; ++m) {
                    const float t = a3[m] + b3[m];
                    a3[m] = t > 0.0f ? t : 0.0f;
                }
                float acc[2] = {0.0f, 0.0f};
                for (int f = 0; f < F; ++f) {
                    const float term = Umf[u * F + f] * Imf[i * F + f];
                    acc[f & 1] = fmaf(hw[f], term, acc[f & 1]);
                }
                for (int m = 0; m < H3; ++m) acc[m & 1] = fmaf(hw[F + m], a3[m], acc[m & 1]);
                out[r * nI + (i - i0)] = ((acc[0] + acc[1]) + (hb ? hb[0] : 0.0f)) + 0.0f;
            }
            free(pi); free(a1); free(a2); free(a3);
        }
        free(pu);
    }
}
