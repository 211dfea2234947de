// Accuracy metrics on the device, straight from the [users, k] top-k index tensor.
//
// Replaces, for the simple accuracy metrics, the reference's per-user Python: get_single_recommendation's dict building
// (recommender_utils_mixin.py:84-88) + Evaluator.eval -> metric classes (evaluation/evaluator.py:117-147):
//   relevance   test items with rating >= threshold                         relevance/relevance.py:87-96
//   gain        2^(rating - threshold + 1) - 1, discount ln2 / ln(rank + 2)  relevance/relevance.py:49-55,71-82
//   nDCG        DCG / IDCG, IDCG over the user's gains sorted descending      metrics/accuracy/ndcg/ndcg.py:68-125
//   Precision   hits / cutoff (precision.py:66)   Recall  hits / #relevant (recall.py:66)   HR  hits > 0 (hit_rate.py:66)
//   MAP         mean over ranks r <= cutoff of (hits in the first r) / r  (map.py:69-80, the reference's definition)
//   MRR         1 / rank of the first hit (mrr.py:63-70)        F1  harmonic mean of Precision and Recall (f1.py:56-68)
//   averaged over the users with >= 1 relevant test item (ndcg.py:124-125)
// Arithmetic in fp64 like the reference; the discount table comes from the host (Python's math.log values).
// One wave per user; the per-user rows are reduced by a fixed-shape tree so that the sums are run-to-run identical.
#include "el_common.h"
#include "el_topk_common.h"

#define MET_N 8          // nDCG, Precision, Recall, HR, MAP, MRR, F1, valid
#define MET_BUF 1024     // LDS gain buffer per wave (kept <= 512 + incoming)
#define MET_MAXCUT 512

__device__ __forceinline__ double el_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_rec_metrics(const int32_t* __restrict__ rec, int64_t ld, int64_t u_start, int64_t n_users,
                                                     const int64_t* __restrict__ tp, const int32_t* __restrict__ ti,
                                                     const float* __restrict__ tr, double thr, int cutoff,
                                                     const double* __restrict__ disc, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ur = (int64_t)blockIdx.x * 4 + wv;
    if (ur >= n_users) return;
    __shared__ u64 gb_[4][MET_BUF];
    u64* gb = gb_[wv];
    const int64_t user = u_start + ur;
    const int64_t t0 = tp[user], t1 = tp[user + 1];

    // ---- relevant test items: count, and the `cutoff` largest gains (IDCG) -------------------------------------
    int cnt = 0;
    double nrel = 0.0;
    auto sort_keep = [&]() {
        int n2 = 64;
        while (n2 < cnt) n2 <<= 1;
        for (int t = cnt + lane; t < n2; t += 64) gb[t] = 0ull;
        el_wave_lds_sync();
        el_wave_bitonic_desc(gb, n2, lane);               // positive doubles order as their bit patterns
        cnt = cnt < cutoff ? cnt : cutoff;
    };
    for (int64_t b = t0; b < t1; b += 64) {
        const int64_t e = b + lane;
        bool relv = false;
        double g = 0.0;
        if (e < t1) {
            const double r = tr ? (double)tr[e] : 1.0;
            relv = r >= thr;
            if (relv) g = exp2(r - thr + 1.0) - 1.0;
        }
        const u64 bal = __ballot(relv);
        if (relv) gb[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (u64)__double_as_longlong(g);
        cnt += __popcll(bal);
        nrel += (double)__popcll(bal);
        el_wave_lds_sync();
        if (cnt > MET_BUF - 64) sort_keep();
    }
    sort_keep();
    double idcg = 0.0;
    for (int t = lane; t < cnt; t += 64) idcg += __longlong_as_double((long long)gb[t]) * disc[t];
    idcg = el_wave_sum(idcg);

    // ---- the recommendation list ------------------------------------------------------------------------------------
    double dcg = 0.0, map = 0.0;
    int nh = 0, first = -1;
    for (int c0 = 0; c0 < cutoff; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        double g = 0.0;
        if (c < cutoff) {
            const int32_t item = rec[ur * ld + c];
            if (item >= 0) {
                const int64_t pos = el_lower_bound(ti, t0, t1, item);
                if (pos < t1 && ti[pos] == item) {
                    const double r = tr ? (double)tr[pos] : 1.0;
                    if (r >= thr) {
                        hit = true;
                        g = exp2(r - thr + 1.0) - 1.0;
                    }
                }
            }
        }
        const u64 bal = __ballot(hit);
        if (c < cutoff) {
            const int cum = nh + __popcll(bal & ((2ull << lane) - 1ull));     // hits in ranks 1..c+1
            map += (double)cum / (double)(c + 1);
            dcg += g * disc[c];
        }
        if (first < 0 && bal) first = c0 + __ffsll((long long)bal) - 1;
        nh += __popcll(bal);
    }
    dcg = el_wave_sum(dcg);
    map = el_wave_sum(map);
    if (lane == 0) {
        double* o = out + ur * MET_N;
        const double valid = nrel > 0.0 ? 1.0 : 0.0;
        const double prec = (double)nh / (double)cutoff;
        const double rcl = nrel > 0.0 ? (double)nh / nrel : 0.0;
        o[0] = (dcg > 0.0 && idcg > 0.0) ? dcg / idcg : 0.0;
        o[1] = prec;
        o[2] = rcl;
        o[3] = nh > 0 ? 1.0 : 0.0;
        o[4] = map / (double)cutoff;
        o[5] = first >= 0 ? 1.0 / (double)(first + 1) : 0.0;
        o[6] = (prec + rcl) > 0.0 ? 2.0 * prec * rcl / (prec + rcl) : 0.0;
        o[7] = valid;
    }
}

// sums over the valid users, fixed shape: G workgroups x contiguous row ranges -> partial[G][8]; then one wave adds the G rows
__global__ __launch_bounds__(256) void k_metrics_partial(const double* __restrict__ rows, int64_t n, int64_t per, double* __restrict__ part) {
    __shared__ double sh[256];
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    double acc[MET_N];
#pragma unroll
    for (int m = 0; m < MET_N; ++m) acc[m] = 0.0;
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
        const double v = rows[r * MET_N + 7];
        if (v != 0.0) {
#pragma unroll
            for (int m = 0; m < MET_N; ++m) acc[m] += rows[r * MET_N + m];
        }
    }
    for (int m = 0; m < MET_N; ++m) {
        sh[threadIdx.x] = acc[m];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * MET_N + m] = sh[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_metrics_final(const double* __restrict__ part, int G, double* __restrict__ out) {
    const int m = threadIdx.x;
    if (m >= MET_N) return;
    double a = 0.0;
    for (int g = 0; g < G; ++g) a += part[(int64_t)g * MET_N + m];
    out[m] += a;
}

static int met_groups(int64_t n) {
    int64_t g = (n + 4095) / 4096;
    return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

extern "C" size_t el_rec_metrics_ws_bytes(int64_t n_users) {
    return n_users <= 0 ? 0 : ((size_t)n_users * MET_N * 8 + (size_t)met_groups(n_users) * MET_N * 8);
}

extern "C" int el_rec_metrics(el_ctx* ctx, void* stream, const int32_t* rec_idx, int64_t ld, int64_t u_start, int64_t u_stop,
                              const int64_t* test_indptr, const int32_t* test_indices, const float* test_ratings,
                              double threshold, int32_t cutoff, const double* discount, double* sums, double* per_user,
                              void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(u_stop >= u_start, "el_rec_metrics: u_stop < u_start");
    const int64_t n = u_stop - u_start;
    if (n == 0) return 0;
    EL_REQUIRE(rec_idx && test_indptr && test_indices && discount && sums, "el_rec_metrics: null pointer");
    EL_REQUIRE(cutoff >= 1 && cutoff <= MET_MAXCUT && (int64_t)cutoff <= ld, "el_rec_metrics: cutoff %d unsupported (1..%d, <= ld)", cutoff,
               MET_MAXCUT);
    EL_REQUIRE(ws != nullptr && ws_bytes >= el_rec_metrics_ws_bytes(n), "el_rec_metrics: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* rows = per_user ? per_user : (double*)ws;
    double* part = (double*)((char*)ws + (size_t)n * MET_N * 8);
    EL_LAUNCH("k_rec_metrics", k_rec_metrics, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, rec_idx, ld, u_start, n, test_indptr,
              test_indices, test_ratings, threshold, (int)cutoff, discount, rows);
    const int G = met_groups(n);
    const int64_t per = (n + G - 1) / G;
    EL_LAUNCH("k_metrics_partial", k_metrics_partial, dim3(G), dim3(256), 0, st, (const double*)rows, n, per, part);
    EL_LAUNCH("k_metrics_final", k_metrics_final, dim3(1), dim3(64), 0, st, (const double*)part, G, sums);
    EL_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================
// Fragile-user report.  The fused top-k pins ONE summation order for the fp32 scores (the k-ordered fma
// chain); TensorFlow / Eigen's order in BPRMF_batch_model.py:83-84 (`tf.matmul`) cannot be known here.  Two correct fp32
// evaluations of <u, i> differ by at most gamma = F 2^-23 |u| |i| (both sums within F 2^-24 |u||i| of the exact value), so a
// user's top-k SET is independent of the summation order whenever score_k - score_{k+1} >= F 2^-23 |u| max(|i_k|, |i_{k+1}|).
// This kernel counts the users for whom that does not hold, from a [n, k+1] list of el_score_topk: one wave per user.
//   counts[0] += fragile users, counts[1] += users with fewer than k+1 finite entries (nothing beyond rank k can displace)
// =====================================================================================================
__global__ __launch_bounds__(256) void k_topk_fragile(const float* __restrict__ Gu, const float* __restrict__ Gi, int F,
                                                      int64_t u_start, int64_t n, const int32_t* __restrict__ idx,
                                                      const float* __restrict__ val, int64_t ld, int k, int64_t item_offset,
                                                      unsigned char* __restrict__ flags, unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t ia = idx[r * ld + (k - 1)], ib = idx[r * ld + k];
    const float va = val[r * ld + (k - 1)], vb = val[r * ld + k];
    const bool full = ia >= 0 && ib >= 0 && va > -INFINITY && vb > -INFINITY;
    int fragile = 0;
    if (full) {
        const float* pu = Gu + (u_start + r) * (int64_t)F;
        const float* pa = Gi + ((int64_t)ia - item_offset) * F;
        const float* pb = Gi + ((int64_t)ib - item_offset) * F;
        double nu = 0.0, na = 0.0, nb = 0.0;
        for (int f = lane; f < F; f += 64) {
            const double x = pu[f], y = pa[f], z = pb[f];
            nu += x * x;
            na += y * y;
            nb += z * z;
        }
        nu = el_group_sum(nu, 64);
        na = el_group_sum(na, 64);
        nb = el_group_sum(nb, 64);
        const double bound = (double)F * 1.1920928955078125e-07 * sqrt(nu) * sqrt(na > nb ? na : nb);
        fragile = ((double)va - (double)vb) < bound ? 1 : 0;
    }
    if (lane == 0) {
        if (flags) flags[r] = (unsigned char)fragile;
        if (fragile) atomicAdd(counts, 1ull);
        if (!full) atomicAdd(counts + 1, 1ull);
    }
}

extern "C" int el_topk_fragile(el_ctx* ctx, void* stream, const float* Gu, const float* Gi, int32_t F, int64_t u_start,
                               int64_t u_stop, const int32_t* idx, const float* val, int64_t ld, int32_t k, int64_t item_offset,
                               unsigned char* flags, uint64_t* counts) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(u_stop >= u_start, "el_topk_fragile: u_stop < u_start");
    const int64_t n = u_stop - u_start;
    if (n == 0) return 0;
    EL_REQUIRE(Gu && Gi && idx && val && counts, "el_topk_fragile: null pointer");
    EL_REQUIRE(F >= 1 && k >= 1 && ld >= (int64_t)k + 1, "el_topk_fragile: needs lists of at least k + 1 entries (ld %lld, k %d)", (long long)ld, k);
    EL_LAUNCH("k_topk_fragile", k_topk_fragile, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Gu, Gi, (int)F, u_start, n,
              idx, val, ld, (int)k, item_offset, flags, reinterpret_cast<unsigned long long*>(counts));
    EL_CHECK_LAUNCH();
    return 0;
}
