// NeuMF / GMF: point-wise neural matrix factorisation.
//
// Replaces NeuralMatrixFactorizationModel.call / train_step / get_recs
// (neural/NeuMF/neural_matrix_factorization_model.py:75-93,96-106,120-144) and
// GeneralizedMatrixFactorizationModel.call / train_step (neural/GeneralizedMF/
// generalized_matrix_factorization_model.py:59-79), plus the GMF sampler
// (dataset/samplers/pointwise_pos_neg_sampler.py:26-50).
//
//   mf   = Umf[u] * Imf[i]                               (element-wise)
//   mlp  = relu-MLP([Umlp[u] ; Imlp[i]])                 (Dense(relu) x n_layers, el_gemm.hip, MFMA fp32)
//   y    = sigmoid(w . [mf ; mlp] + b0)                  (GMF: mf only, no bias, w = edge_weight h)
//   loss = keras BinaryCrossentropy: mean_b -(t log(p + 1e-7) + (1-t) log(1 - p + 1e-7)), p clipped to [1e-7, 1-1e-7] first
// Embedding tables receive IndexedSlices gradients -> Keras Adam sparse apply (every row decays/moves, SURVEY
// A.4, k_adam_dense of el_bpr.hip); Dense kernels/biases and the head use the dense ApplyAdam arithmetic.
// Gathers/scatters are HBM-bound row operations (one lane group per sample, 16 B per lane); the MLP is GEMM-bound.
#include "el_common.h"

extern "C" int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           const float* bias, int act, void* ws, size_t ws_bytes);
// el_bpr.hip / el_vae.hip
__global__ void k_adam_dense(float* th, float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps);
__global__ void k_adam_apply_dense(float* th, float* g, float* m, float* v, int64_t n, float alpha, float b1, float b2,
                                   float eps, int zero_g);
__global__ void k_colsum(const float* X, int64_t B, int64_t N, int64_t ld, float* out);

// ---- point-wise sampler (GMF) --------------------------------------------------------------------------
struct PwPhilox {
    u32 n_lo, n_hi, k0, k1, a, w[4];
    int have;
    __device__ __forceinline__ void init(u64 n, u64 seed) {
        n_lo = (u32)n;
        n_hi = (u32)(n >> 32);
        k0 = (u32)seed;
        k1 = (u32)(seed >> 32);
        a = 0;
        have = 0;
    }
    __device__ __forceinline__ u32 next() {
        if (have == 0) {
            el_philox4 r = el_philox4x32_10(n_lo, n_hi, a, 1u /* stream id: point-wise sampler */, k0, k1);
            w[0] = r.x;
            w[1] = r.y;
            w[2] = r.z;
            w[3] = r.w;
            a++;
            have = 4;
        }
        u32 v = w[4 - have];
        have--;
        return v;
    }
    __device__ __forceinline__ u32 bounded(u32 n) {
        u32 m = n - 1u;
        m |= m >> 1;
        m |= m >> 2;
        m |= m >> 4;
        m |= m >> 8;
        m |= m >> 16;
        u32 v;
        do {
            v = next() & m;
        } while (v >= n);
        return v;
    }
};

// u ~ U[0,U); coin; positive: i ~ U(pos(u)), label 1; negative: i ~ U[0,I) \ pos(u), label 0
// (pointwise_pos_neg_sampler.py:33-46)
template <bool META>
__global__ __launch_bounds__(256) void k_pw_sample(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                   int64_t U, int64_t I, u64 seed, u64 first, int64_t n, int32_t* out_u,
                                                   int32_t* out_i, float* out_y, const SamplerRec* __restrict__ meta) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    PwPhilox ps;
    ps.init(first + (u64)t, seed);
    for (;;) {
        u32 u = ps.bounded((u32)U);
        int64_t r0, r1;
        SamplerUser su;
        if (META) {
            su.load(meta, u);                                 // one 64-byte record: row bounds + membership signature
            r0 = su.r0, r1 = su.r1;
        } else {
            r0 = indptr[u];
            r1 = indptr[u + 1];
        }
        int64_t lui = r1 - r0;
        if (lui <= 0 || lui >= I) continue;
        u32 coin = ps.next() >> 31;                       // random.getrandbits(1)
        int32_t it = -1;
        if (coin) {
            it = indices[r0 + ps.bounded((u32)lui)];
        } else {
            for (int attempt = 0; attempt < 4096 && it < 0; ++attempt) {
                int32_t cand = (int32_t)ps.bounded((u32)I);
                if ((META && !su.maybe(cand)) || !el_row_contains(indices, r0, r1, cand)) it = cand;
            }
            if (it < 0) continue;
        }
        out_u[t] = (int32_t)u;
        out_i[t] = it;
        out_y[t] = coin ? 1.0f : 0.0f;
        return;
    }
}

extern "C" int el_pointwise_sample(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                                   int64_t U, int64_t I, uint64_t seed, uint64_t first_sample, int64_t n, int32_t* out_u,
                                   int32_t* out_i, float* out_label) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(pos_indptr && pos_indices && out_u && out_i && out_label, "el_pointwise_sample: null pointer");
    EL_REQUIRE(U >= 1 && U < 0xffffffffLL && I >= 2 && I < 0x7fffffffLL, "el_pointwise_sample: U/I out of range");
    if (n <= 0) return 0;
    EL_LAUNCH("k_pw_sample", k_pw_sample<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos_indptr,
              pos_indices, U, I, (u64)seed, (u64)first_sample, n, out_u, out_i, out_label, (const SamplerRec*)nullptr);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_pointwise_sample_meta(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                                        const void* meta, int64_t U, int64_t I, uint64_t seed, uint64_t first_sample, int64_t n,
                                        int32_t* out_u, int32_t* out_i, float* out_label) {
    if (meta == nullptr) return el_pointwise_sample(ctx, stream, pos_indptr, pos_indices, U, I, seed, first_sample, n, out_u, out_i, out_label);
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(pos_indptr && pos_indices && out_u && out_i && out_label, "el_pointwise_sample_meta: null pointer");
    EL_REQUIRE(U >= 1 && U < 0xffffffffLL && I >= 2 && I < 0x7fffffffLL, "el_pointwise_sample_meta: U/I out of range");
    EL_REQUIRE(((uintptr_t)meta & 63) == 0, "el_pointwise_sample_meta: meta must be 64-byte aligned");
    if (n <= 0) return 0;
    EL_LAUNCH("k_pw_sample", k_pw_sample<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos_indptr,
              pos_indices, U, I, (u64)seed, (u64)first_sample, n, out_u, out_i, out_label, (const SamplerRec*)meta);
    EL_CHECK_LAUNCH();
    return 0;
}

// ---- gather: X0[b] = [Umlp[u] ; Imlp[i]], MF[b] = Umf[u] * Imf[i] -------------------------------------------
__global__ __launch_bounds__(256) void k_nmf_gather(el_nmf_state st, const int32_t* __restrict__ bu,
                                                    const int32_t* __restrict__ bi, int64_t n) {
    // one wave per sample; generic strided loops (F, E need not be multiples of 4)
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int64_t u = bu[b], i = bi[b];
    if (st.use_mf) {
        const float* pu = st.tab[0] + u * st.F;
        const float* pi = st.tab[1] + i * st.F;
        float* o = st.MF + b * st.F;
        for (int f = lane; f < st.F; f += 64) o[f] = pu[f] * pi[f];
    }
    if (st.use_mlp) {
        const float* pu = st.tab[2] + u * st.E;
        const float* pi = st.tab[3] + i * st.E;
        float* o = st.X0 + b * 2 * st.E;
        for (int f = lane; f < st.E; f += 64) {
            o[f] = pu[f];
            o[st.E + f] = pi[f];
        }
    }
}

// ---- head: logit, probability, BCE, d logit, head gradients --------------------------------------------------
// One wave per sample, persistent waves.  mode 0: forward only (out_prob[b] = p).  mode 1: training.
// The head-weight gradient ghw[f] = sum_b dlogit_b cat[b, f] is a reduction over the WHOLE batch onto F + Hl addresses:
// every wave keeps its share in registers (feature f = lane + 64 q), the four waves of a workgroup are combined in LDS and
// each workgroup issues one atomic per feature -- per-sample atomics on those few addresses cost 6.4 ms at B = 262 144.
#define NMF_HEAD_Q 16                                    // features per lane held in registers: F + Hl <= 1024
// Q = ceil((F + Hl) / 64) rounded up to a power of two (host): the feature loops carry no dead iterations; the head weights
// sit in registers; the NEXT sample's row is fetched while this one's reduction / exp / log chain runs.
template <int Q>
__global__ __launch_bounds__(256) void k_nmf_head(el_nmf_state st, const float* __restrict__ label, int64_t n, int mode,
                                                  float* out_prob, double* loss_out, int64_t n_div) {
    __shared__ float wsum[4];
    __shared__ float facc[4][64 * Q];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int F = st.use_mf ? st.F : 0;
    const int Hl = st.use_mlp ? st.units[st.n_layers - 1] : 0;
    const int NF = F + Hl;
    const float hbias = st.head_bias ? st.hb[0] : 0.f;
    const float* act_last = st.use_mlp ? st.act[st.n_layers - 1] : nullptr;
    float* dact_last = st.use_mlp ? st.dact[st.n_layers - 1] : nullptr;
    float acc[Q], hwr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc[q] = 0.f;
        const int f = lane + 64 * q;
        hwr[q] = f < NF ? st.hw[f] : 0.f;
    }
    auto fetch = [&](int64_t b, float (&v)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int f = lane + 64 * q;
            v[q] = 0.f;
            if (f < NF) v[q] = f < F ? st.MF[b * F + f] : act_last[b * (int64_t)Hl + (f - F)];
        }
    };
    float bacc = 0.f, myloss = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wv;
    float nxt[Q];
    if (b < n) fetch(b, nxt);
    for (; b < n; b += stride) {
        float val[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) val[q] = nxt[q];
        if (b + stride < n) fetch(b + stride, nxt);
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) part += val[q] * hwr[q];    // (features beyond NF: 0 * 0; same order of the live terms as before)
        const float logit = el_group_sum(part, 64) + hbias;
        const float p = 1.0f / (1.0f + expf(-logit));
        if (mode == 0) {
            if (lane == 0) out_prob[b] = p;
            continue;
        }
        const float t = label[b];
        // K.binary_crossentropy (oracle/tf_clauses.py: bce_clips_probabilities_at_1e7, bce_adds_epsilon_inside_log): the probability
        // clipped to [1e-7, 1 - 1e-7], epsilon added again inside both logarithms
        const float pc = fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f);
        if (lane == 0) myloss += -(t * logf(pc + 1e-7f) + (1.0f - t) * logf((1.0f - pc) + 1e-7f)) / (float)n_div;
        float dlogit = 0.f;                                             // d/dlogit: zero where the clip is active
        if (p > 1e-7f && p < 1.0f - 1e-7f) {
            const float dp = -(t / (p + 1e-7f) - (1.0f - t) / ((1.0f - p) + 1e-7f));      // through the logarithms ...
            dlogit = dp * (p * (1.0f - p)) / (float)n_div;                                 // ... and the sigmoid
        }
        if (lane == 0) st.dlogit[b] = dlogit;
        // d loss / d (pre-activation of the last Dense(relu) layer) = dlogit w_f where its output is positive: the ReLU derivative
        // is taken here, where the output row is in registers
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            acc[q] += dlogit * val[q];
            const int f = lane + 64 * q;
            if (f >= F && f < NF) dact_last[b * (int64_t)Hl + (f - F)] = val[q] > 0.f ? dlogit * hwr[q] : 0.f;
        }
        bacc += dlogit;
    }
    if (mode == 0) return;
#pragma unroll
    for (int q = 0; q < Q; ++q) facc[wv][lane + 64 * q] = acc[q];
    float wl = el_group_sum(myloss, 64);
    if (lane == 0) wsum[wv] = wl;
    __syncthreads();
    for (int f = threadIdx.x; f < NF; f += 256) {
        const float g = (facc[0][f] + facc[1][f]) + (facc[2][f] + facc[3][f]);
        if (g != 0.f) atomicAdd(st.ghw + f, g);
    }
    if (st.head_bias && lane == 0 && bacc != 0.f) atomicAdd(st.ghb, bacc);
    if (threadIdx.x == 0 && loss_out) {
        const double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0) atomicAdd(loss_out, tot);
    }
}

static void launch_nmf_head(el_ctx* ctx, hipStream_t s, const el_nmf_state* st, const float* label, int64_t n, int mode, float* out_prob,
                            double* loss_out, int64_t n_div, unsigned grid) {
    const int NF = (st->use_mf ? st->F : 0) + (st->use_mlp ? st->units[st->n_layers - 1] : 0);
    const int nq = (NF + 63) / 64;
#define EL_HEAD(Q_) EL_LAUNCH("k_nmf_head", k_nmf_head<Q_>, dim3(grid), dim3(256), 0, s, *st, label, n, mode, out_prob, loss_out, n_div)
    if (nq <= 1) EL_HEAD(1);
    else if (nq <= 2) EL_HEAD(2);
    else if (nq <= 4) EL_HEAD(4);
    else if (nq <= 8) EL_HEAD(8);
    else EL_HEAD(16);
#undef EL_HEAD
    (void)ctx;
}

// in place: d <- d * (y > 0)   (relu backward)
__global__ __launch_bounds__(256) void k_relu_bwd(float* __restrict__ d, const float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        if (!(y[t] > 0.f)) d[t] = 0.f;
}

// relu backward + bias gradient in one pass over d [n, units]:  d <- d * (y > 0);  gb[c] += sum_b d[b, c]
// (gb zeroed by the caller).  A workgroup owns the column block blockIdx.x (W = min(units, 256) columns, 256 / W rows at a
// time) and walks the rows blockIdx.y, blockIdx.y + gridDim.y, ...; the column sums stay in registers until the end.
__global__ __launch_bounds__(256) void k_relu_bwd_colsum(float* __restrict__ d, const float* __restrict__ y, int64_t n,
                                                         int64_t units, float* __restrict__ gb) {
    __shared__ float part[256];
    const int W = units < 256 ? (int)units : 256, R = 256 / W;
    const int tc = threadIdx.x % W, tr = threadIdx.x / W;
    const int64_t c = (int64_t)blockIdx.x * W + tc;
    float s = 0.f;
    if (tr < R && c < units) {
        // four rows per trip: eight loads in flight per thread (one row per trip left the pass latency-bound at ~4 TB/s)
        const int64_t stride = (int64_t)gridDim.y * R;
        int64_t b = (int64_t)blockIdx.y * R + tr;
        for (; b + 3 * stride < n; b += 4 * stride) {
            float v[4], yy[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t e = (b + k * stride) * units + c;
                v[k] = d[e];
                yy[k] = y[e];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(yy[k] > 0.f)) {
                    v[k] = 0.f;
                    d[(b + k * stride) * units + c] = 0.f;
                }
                s += v[k];
            }
        }
        for (; b < n; b += stride) {
            const int64_t e = b * units + c;
            float v = d[e];
            if (!(y[e] > 0.f)) {
                v = 0.f;
                d[e] = 0.f;
            }
            s += v;
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0 && c < units) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += part[r * W + tc];
        if (t != 0.f) atomicAdd(gb + c, t);
    }
}

// bias gradient alone: gb[c] += sum_b d[b, c] (gb zeroed by the caller); d already carries the ReLU derivative (k_nmf_head
// applies it for the last layer).  Same walk as k_relu_bwd_colsum, one read of d and nothing else.
__global__ __launch_bounds__(256) void k_nmf_colsum(const float* __restrict__ d, int64_t n, int64_t units, float* __restrict__ gb) {
    __shared__ float part[256];
    const int W = units < 256 ? (int)units : 256, R = 256 / W;
    const int tc = threadIdx.x % W, tr = threadIdx.x / W;
    const int64_t c = (int64_t)blockIdx.x * W + tc;
    float s0 = 0.f, s1 = 0.f;
    if (tr < R && c < units) {
        int64_t b = (int64_t)blockIdx.y * R + tr;
        const int64_t stride = (int64_t)gridDim.y * R;
        for (; b + stride < n; b += 2 * stride) {
            s0 += d[b * units + c];
            s1 += d[(b + stride) * units + c];
        }
        if (b < n) s0 += d[b * units + c];
    }
    part[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (tr == 0 && c < units) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += part[r * W + tc];
        if (t != 0.f) atomicAdd(gb + c, t);
    }
}

// keras Dropout in training mode, in place on x [n, width]: Philox counter (row, column / 4, step, layer) -> four uniforms.
// The same call on the gradient of x reproduces the mask in the backward pass (nothing is stored).
__global__ __launch_bounds__(256) void k_nmf_dropout(float* __restrict__ x, int64_t n, int width, float rate, u64 seed, u32 step,
                                                     u32 layer) {
    const int w4 = (width + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * w4) return;
    const int64_t b = t / w4;
    const int c4 = (int)(t - b * w4);
    const el_philox4 r = el_philox4x32_10((u32)b, (u32)c4, step, layer, (u32)seed, (u32)(seed >> 32));
    const u32 rv[4] = {r.x, r.y, r.z, r.w};
    const float keep = 1.0f / (1.0f - rate);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        if (c < width) {
            const float uni = (float)(rv[k] >> 8) * (1.0f / 16777216.0f);
            float* px = x + b * width + c;
            *px = uni < rate ? 0.f : *px * keep;
        }
    }
}

// embedding gradients (IndexedSlices, duplicates summed): scatter-add one row per sample and table
__global__ __launch_bounds__(256) void k_nmf_scatter(el_nmf_state st, const int32_t* __restrict__ bu,
                                                     const int32_t* __restrict__ bi, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int64_t u = bu[b], i = bi[b];
    if (st.use_mf) {
        const float dl = st.dlogit[b];
        const float* pu = st.tab[0] + u * st.F;
        const float* pi = st.tab[1] + i * st.F;
        float* gu = st.gtab[0] + u * st.F;
        float* gi = st.gtab[1] + i * st.F;
        for (int f = lane; f < st.F; f += 64) {
            const float s = dl * st.hw[f];
            atomicAdd(gu + f, s * pi[f]);
            atomicAdd(gi + f, s * pu[f]);
        }
    }
    if (st.use_mlp) {
        const float* dx = st.dX0 + b * 2 * st.E;
        float* gu = st.gtab[2] + u * st.E;
        float* gi = st.gtab[3] + i * st.E;
        for (int f = lane; f < st.E; f += 64) {
            atomicAdd(gu + f, dx[f]);
            atomicAdd(gi + f, dx[st.E + f]);
        }
    }
}

// ---- deferred decay of the embedding tables ---------------------------------------------------------------------------
// Keras' Adam moves EVERY row of an embedding table at every step: m <- b1 m, v <- b2 v,
// theta <- theta - lr_t m / (sqrt(v) + eps), gradient or not.  The eager form streams theta, g, m, v of all (U + I)(F + E)
// parameters per step: 3.3 of the 10 ms of a step at 1.25 M x 1 M x 128, for a batch that touches a fifth of the rows.
// For a row WITHOUT a gradient that update reads nothing but the row itself, so it can be postponed and replayed in
// registers -- the same fp32 operations on the same operands in the same order, hence the same bits -- at the moment the row is
// needed again: by a batch that contains it (k_nmf_catchup, before the forward pass reads it) or by anything that reads the
// tables as a whole (k_nmf_flush_rows: scoring, weights(), a checkpoint).  row_last[side][r] = the optimiser step row r of
// that side's tables is current at; lr_hist[s - hist_base] = lr_t of step s.  Every (element, step) update is still performed
// exactly once; what disappears is the HBM round trip of the rows a step does not touch.
//
// One wave per (sample, side: 0 user / 1 item).  The first wave to stamp a row with this batch's claim number owns it: it
// replays the row's missed steps (last, t-1] now and applies step t with the accumulated gradient row after the backward pass
// (k_nmf_apply_rows); the other occurrences of the row do nothing.
struct NmfRowTabs {
    float* th[2];
    float* g[2];
    float* m[2];
    float* v[2];
    int D[2];
    int n;
};

__device__ __forceinline__ NmfRowTabs nmf_row_tabs(const el_nmf_state& st, int side) {
    NmfRowTabs r;
    r.n = 0;
    if (st.use_mf) {
        r.th[r.n] = st.tab[side], r.g[r.n] = st.gtab[side], r.m[r.n] = st.mtab[side], r.v[r.n] = st.vtab[side], r.D[r.n] = st.F;
        r.n++;
    }
    if (st.use_mlp) {
        r.th[r.n] = st.tab[2 + side], r.g[r.n] = st.gtab[2 + side], r.m[r.n] = st.mtab[2 + side], r.v[r.n] = st.vtab[2 + side], r.D[r.n] = st.E;
        r.n++;
    }
    return r;
}

// A wave owns one row of every table of a side at a time: NT tables x Q float-pairs (or floats) per lane.
//   VW = 2: lane holds elements 2 lane + 128 q, +1 (rows of an even dimension start 8-byte aligned); VW = 1: lane + 64 q.
template <int VW, int Q>
struct NmfRowRegs {
    float a[2][Q][VW], m[2][Q][VW], v[2][Q][VW];
};

template <int VW, int Q, bool WITH_TH>
__device__ __forceinline__ void nmf_rows_load(NmfRowRegs<VW, Q>& r, const NmfRowTabs& rt, int64_t row, int f0, int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int f = f0 + (lane + 64 * q) * VW;
#pragma unroll
            for (int x = 0; x < VW; ++x) r.a[k][q][x] = r.m[k][q][x] = r.v[k][q][x] = 0.f;
            if (k < rt.n && f < rt.D[k]) {
                const int64_t o = row * rt.D[k] + f;
                if (VW == 2) {
                    if (WITH_TH) {
                        const float2 t = *reinterpret_cast<const float2*>(rt.th[k] + o);
                        r.a[k][q][0] = t.x, r.a[k][q][VW - 1] = t.y;
                    }
                    const float2 mm = *reinterpret_cast<const float2*>(rt.m[k] + o), vv = *reinterpret_cast<const float2*>(rt.v[k] + o);
                    r.m[k][q][0] = mm.x, r.m[k][q][VW - 1] = mm.y;
                    r.v[k][q][0] = vv.x, r.v[k][q][VW - 1] = vv.y;
                } else {
                    if (WITH_TH) r.a[k][q][0] = rt.th[k][o];
                    r.m[k][q][0] = rt.m[k][o];
                    r.v[k][q][0] = rt.v[k][o];
                }
            }
        }
}

template <int VW, int Q>
__device__ __forceinline__ void nmf_rows_store(const float (&val)[2][Q][VW], float* const (&dst)[2], const NmfRowTabs& rt, int64_t row,
                                               int f0, int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int f = f0 + (lane + 64 * q) * VW;
            if (k < rt.n && f < rt.D[k]) {
                float* d = dst[k] + row * rt.D[k] + f;
                if (VW == 2) *reinterpret_cast<float2*>(d) = make_float2(val[k][q][0], val[k][q][VW - 1]);
                else d[0] = val[k][q][0];
            }
        }
}

// nsteps gradient-free steps on the registers (theta too when WITH_TH): exactly el_adam_elem with g = 0
template <int VW, int Q, bool WITH_TH>
__device__ __forceinline__ void nmf_rows_replay(NmfRowRegs<VW, Q>& r, const float* __restrict__ lr_from, int nsteps) {
    const float b1 = 0.9f, b2 = 0.999f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    if (WITH_TH) {
        // (el_common.h: the step on packed fp32 instructions where the values allow it, el_adam_elem otherwise -- the same bits;
        // the element groups are independent, so each runs its own loop over the steps)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q) el_adam_replay<VW>(r.a[k][q], r.m[k][q], r.v[k][q], nsteps, [&](int s) { return lr_from[s]; });
        return;
    }
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int x = 0; x < VW; ++x) {
                    r.m[k][q][x] = r.m[k][q][x] * b1 + 0.0f * omb1;           // the first two lines of el_adam_elem
                    r.v[k][q][x] = r.v[k][q][x] * b2 + (0.0f * 0.0f) * omb2;
                }
    }
}

// MODE 0 (catch-up): theta of the row to step t - 1 (m, v are read, replayed in registers and NOT written: k_nmf_apply_rows
//                    replays them again -- two multiplications per step -- when it rewrites them anyway)
// MODE 1 (apply):    m, v to step t - 1, then step t with the gradient row; theta, m, v written, gradient row zeroed
// MODE 2 (flush):    theta, m, v to step t
template <int VW, int Q, int MODE>
__device__ __forceinline__ void nmf_row_pass(const el_nmf_state& st, const NmfRowTabs& rt, int64_t row, int lane, int last, int32_t t, float lr_t) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int nsteps = (MODE == 2 ? t : t - 1) - last;
    const float* lr_from = st.lr_hist + (last + 1 - st.hist_base);
    const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
    for (int f0 = 0; f0 < Dmax; f0 += 64 * Q * VW) {
        NmfRowRegs<VW, Q> r;
        nmf_rows_load<VW, Q, true>(r, rt, row, f0, lane);
        float g[2][Q][VW];
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int f = f0 + (lane + 64 * q) * VW;
#pragma unroll
                    for (int x = 0; x < VW; ++x) g[k][q][x] = 0.f;
                    if (k < rt.n && f < rt.D[k]) {
                        const float* gp = rt.g[k] + row * rt.D[k] + f;
                        if (VW == 2) {
                            const float2 t2 = *reinterpret_cast<const float2*>(gp);
                            g[k][q][0] = t2.x, g[k][q][VW - 1] = t2.y;
                        } else {
                            g[k][q][0] = gp[0];
                        }
                    }
                }
            nmf_rows_replay<VW, Q, false>(r, lr_from, nsteps);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) el_adam_elem(r.a[k][q][x], r.m[k][q][x], r.v[k][q][x], g[k][q][x], lr_t, b1, b2, omb1, omb2, eps);
        } else {
            // m = v = 0 (rows that never had a gradient) is a fixed point of the gradient-free step -- m <- 0, v <- 0,
            // theta <- theta - lr 0 / (0 + eps) = theta: nothing to replay and nothing to write, whatever the gap
            bool nz = false;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) nz = nz || r.m[k][q][x] != 0.f || r.v[k][q][x] != 0.f;
            if (__ballot(nz) == 0ull) continue;
            nmf_rows_replay<VW, Q, true>(r, lr_from, nsteps);
        }
        float* const dth[2] = {rt.th[0], rt.th[1]};
        nmf_rows_store<VW, Q>(r.a, dth, rt, row, f0, lane);
        if (MODE != 0) {
            float* const dm[2] = {rt.m[0], rt.m[1]};
            float* const dv[2] = {rt.v[0], rt.v[1]};
            nmf_rows_store<VW, Q>(r.m, dm, rt, row, f0, lane);
            nmf_rows_store<VW, Q>(r.v, dv, rt, row, f0, lane);
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) g[k][q][x] = 0.f;
            float* const dg[2] = {rt.g[0], rt.g[1]};
            nmf_rows_store<VW, Q>(g, dg, rt, row, f0, lane);
        }
    }
}

// VW = 2 when every table row starts 8-byte aligned (even dimensions); one chunk of 128 (VW 2) / 64 (VW 1) elements per lane pass
template <int MODE>
__device__ __forceinline__ void nmf_row_dispatch(const el_nmf_state& st, const NmfRowTabs& rt, int64_t row, int lane, int last, int32_t t,
                                                 float lr_t) {
    const bool even = (rt.D[0] % 2 == 0) && (rt.n < 2 || rt.D[1] % 2 == 0);
    if (even) nmf_row_pass<2, 1, MODE>(st, rt, row, lane, last, t, lr_t);
    else nmf_row_pass<1, 1, MODE>(st, rt, row, lane, last, t, lr_t);
}

__global__ __launch_bounds__(256) void k_nmf_catchup(el_nmf_state st, const int32_t* __restrict__ bu, const int32_t* __restrict__ bi,
                                                     int64_t n, int32_t t, int32_t claim) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= 2 * n) return;
    const int side = p >= n ? 1 : 0;
    const int64_t b = p - (side ? n : 0);
    const int64_t row = side ? bi[b] : bu[b];
    int own = 0, last = 0;
    if (lane == 0) {
        own = atomicExch(st.row_stamp[side] + row, claim) != claim;
        st.row_own[(int64_t)side * st.Bmax + b] = (uint8_t)own;
        if (own) last = st.row_last[side][row];
    }
    own = __builtin_amdgcn_readfirstlane(own);
    last = __builtin_amdgcn_readfirstlane(last);
    if (!own || (t - 1) - last <= 0) return;
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    nmf_row_dispatch<0>(st, rt, row, lane, last, t, 0.f);
}

// step t on the rows this batch owns: Keras sparse apply with the accumulated gradient row (duplicates already summed); the
// gradient row is zero again afterwards
__global__ __launch_bounds__(256) void k_nmf_apply_rows(el_nmf_state st, const int32_t* __restrict__ bu, const int32_t* __restrict__ bi,
                                                        int64_t n, int32_t t, float lr_t) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p == 0 && lane == 0) st.lr_hist[t - st.hist_base] = lr_t;        // this step's lr_t for later replays (the replays of this
    //                                                                      kernel read steps < t only)
    if (p >= 2 * n) return;
    const int side = p >= n ? 1 : 0;
    const int64_t b = p - (side ? n : 0);
    if (!st.row_own[(int64_t)side * st.Bmax + b]) return;
    const int64_t row = side ? bi[b] : bu[b];
    const int last = __builtin_amdgcn_readfirstlane(st.row_last[side][row]);
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    nmf_row_dispatch<1>(st, rt, row, lane, last, t, lr_t);
    if (lane == 0) st.row_last[side][row] = t;
}

// every row of one side up to step t (one wave per row)
__global__ __launch_bounds__(256) void k_nmf_flush_rows(el_nmf_state st, int side, int64_t rows, int32_t t) {
    const int lane = threadIdx.x & 63;
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int last = __builtin_amdgcn_readfirstlane(st.row_last[side][row]);
        if (t - last <= 0) continue;
        nmf_row_dispatch<2>(st, rt, row, lane, last, t, 0.f);
        if (lane == 0) st.row_last[side][row] = t;
    }
}

// ---- host -------------------------------------------------------------------------------------------------------
static unsigned g1(int64_t n, el_ctx* ctx) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->cus * 16;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

static unsigned head_grid(int64_t n, el_ctx* ctx) {
    const int64_t want = (n + 3) / 4, cap = (int64_t)ctx->cus * 8;
    return (unsigned)(want < cap ? want : cap);
}

static int nmf_check(const el_nmf_state* st, int64_t n, bool train) {
    EL_REQUIRE(st != nullptr, "el_nmf: null state");
    EL_REQUIRE(st->use_mf || st->use_mlp, "el_nmf: mf_train and mlp_train can not be False at the same time");
    EL_REQUIRE(n >= 1 && n <= st->Bmax, "el_nmf: %lld samples exceed Bmax %lld", (long long)n, (long long)st->Bmax);
    EL_REQUIRE(st->hw != nullptr && st->dlogit != nullptr, "el_nmf: head buffers missing");
    EL_REQUIRE(st->dropout >= 0.f && st->dropout < 1.f, "el_nmf: dropout must be in [0, 1)");
    EL_REQUIRE((st->use_mf ? st->F : 0) + (st->use_mlp ? st->units[st->n_layers > 0 ? st->n_layers - 1 : 0] : 0) <= 64 * NMF_HEAD_Q,
               "el_nmf: head input wider than %d features", 64 * NMF_HEAD_Q);
    if (st->use_mf) EL_REQUIRE(st->tab[0] && st->tab[1] && st->MF && st->F >= 1, "el_nmf: MF tables missing");
    if (st->use_mlp) {
        EL_REQUIRE(st->tab[2] && st->tab[3] && st->X0 && st->E >= 1, "el_nmf: MLP tables missing");
        EL_REQUIRE(st->n_layers >= 1 && st->n_layers <= 4, "el_nmf: 1..4 hidden layers supported");
        for (int l = 0; l < st->n_layers; ++l) EL_REQUIRE(st->W[l] && st->b[l] && st->act[l] && st->units[l] >= 1, "el_nmf: layer %d missing", l);
    }
    if (train) {
        EL_REQUIRE(st->ghw && st->mhw && st->vhw, "el_nmf: head optimiser buffers missing");
        if (st->use_mlp) EL_REQUIRE(st->dX0 != nullptr, "el_nmf: dX0 missing");
    }
    if (st->row_last[0] || st->row_last[1]) {
        EL_REQUIRE(st->row_last[0] && st->row_last[1] && st->row_stamp[0] && st->row_stamp[1] && st->row_own && st->lr_hist &&
                   st->lr_hist_cap >= 2, "el_nmf: deferred decay needs row_last[2], row_stamp[2], row_own and lr_hist");
        EL_REQUIRE(st->opt_step >= 0 && st->flushed_step >= 0 && st->flushed_step <= st->opt_step && st->hist_base >= 1,
                   "el_nmf: deferred-decay counters corrupt (zero-initialise opt_step / flushed_step / claim_seq, hist_base = 1)");
    }
    return 0;
}

static inline bool nmf_deferred(const el_nmf_state* st) { return st->row_last[0] != nullptr; }

// every embedding row current at st->opt_step (no-op in the eager form and when nothing is pending)
static int nmf_sync(el_ctx* ctx, hipStream_t s, el_nmf_state* st, bool inside_step = false) {
    if (!nmf_deferred(st)) return 0;
    EL_REQUIRE(inside_step || st->batch_n == 0, "el_nmf: the tables cannot be read between el_nmf_grads and el_nmf_apply (deferred decay)");
    if (st->flushed_step >= st->opt_step) return 0;
    const int64_t rows[2] = {st->U, st->I};
    for (int side = 0; side < 2; ++side) {
        int64_t g = (rows[side] + 3) / 4;
        const int64_t cap = (int64_t)ctx->cus * 32;
        if (g > cap) g = cap;
        EL_LAUNCH("k_nmf_flush_rows", k_nmf_flush_rows, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, s, *st, side, rows[side], st->opt_step);
    }
    EL_CHECK_LAUNCH();
    st->flushed_step = st->opt_step;
    st->hist_base = st->opt_step + 1;                    // steps <= opt_step are never replayed again
    return 0;
}

extern "C" int el_nmf_sync_tables(el_ctx* ctx, void* stream, el_nmf_state* st) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = nmf_check(st, 1, false)) return rc;
    return nmf_sync(ctx, (hipStream_t)stream, st);
}

// deferred decay, start of step t = opt_step + 1: record lr_t, elect the owners of the batch's rows and bring those rows to t - 1
static int nmf_begin_rows(el_ctx* ctx, hipStream_t s, el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n) {
    const int32_t t = st->opt_step + 1;
    if (t - st->hist_base >= st->lr_hist_cap)            // history full: bring every row to t - 1, restart the history at t
        if (int rc = nmf_sync(ctx, s, st)) return rc;
    if (st->claim_seq >= 0x7ffffffe || st->claim_seq < 0) {     // 2^31 gradient evaluations: start the claim numbers again
        EL_CHECK_HIP(hipMemsetAsync(st->row_stamp[0], 0, (size_t)st->U * 4, s));
        EL_CHECK_HIP(hipMemsetAsync(st->row_stamp[1], 0, (size_t)st->I * 4, s));
        st->claim_seq = 0;
    }
    st->claim_seq += 1;
    EL_LAUNCH("k_nmf_catchup", k_nmf_catchup, dim3((unsigned)((2 * n + 3) / 4)), dim3(256), 0, s, *st, u, i, n, t, st->claim_seq);
    EL_CHECK_LAUNCH();
    st->batch_u = u, st->batch_i = i, st->batch_n = n;
    return 0;
}

static void nmf_dropout(hipStream_t s, const el_nmf_state* st, float* x, int64_t n, int64_t width, int layer) {
    const int64_t threads = n * ((width + 3) / 4);
    EL_LAUNCH("k_nmf_dropout", k_nmf_dropout, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, n, (int)width, st->dropout,
              (u64)st->drop_seed, (u32)st->drop_step, (u32)layer);
}

static int nmf_forward(el_ctx* ctx, hipStream_t s, const el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n,
                       bool train = false) {
    EL_LAUNCH("k_nmf_gather", k_nmf_gather, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, *st, u, i, n);
    const bool drop = train && st->dropout > 0.f;
    if (st->use_mlp) {
        float* in = st->X0;
        int64_t kin = 2 * (int64_t)st->E;
        for (int l = 0; l < st->n_layers; ++l) {
            if (drop) nmf_dropout(s, st, in, n, kin, l);            // Dropout in front of Dense l: the input buffer itself is dropped
            if (int rc = el_gemm_f32(ctx, s, 0, 0, n, st->units[l], kin, in, kin, st->W[l], st->units[l], st->act[l],
                                     st->units[l], st->b[l], 2 /*relu*/, st->ws, st->ws_bytes)) return rc;
            in = st->act[l];
            kin = st->units[l];
        }
    }
    EL_CHECK_LAUNCH();
    return 0;
}

// probabilities of the pairs (u[b], i[b]) -- get_recs (neural_matrix_factorization_model.py:120-144)
extern "C" int el_nmf_forward(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                              int64_t n, float* out_prob) {
    if (int rc = el_bind(ctx)) return rc;
    if (n == 0) return 0;
    if (int rc = nmf_check(st, n, false)) return rc;
    EL_REQUIRE(u && i && out_prob, "el_nmf_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = nmf_sync(ctx, s, st)) return rc;
    if (int rc = nmf_forward(ctx, s, st, u, i, n)) return rc;
    launch_nmf_head(ctx, s, st, nullptr, n, 0, out_prob, nullptr, n, head_grid(n, ctx));
    EL_CHECK_LAUNCH();
    return 0;
}

// forward, BinaryCrossentropy (mean over n_div samples: n_div = n, or the global batch when several ranks share a step),
// backward: every gradient buffer of the state is complete on exit
// defer_join (el_nmf_train_step): the weight-gradient products may still be running on the library's second stream when this returns;
// nmf_apply waits for them after it has launched the embedding rows' step (ctx->side_join_pending)
static int nmf_grads(el_ctx* ctx, hipStream_t s, el_nmf_state* st, const int32_t* u, const int32_t* i, const float* label,
                     int64_t n, int64_t n_div, double* loss_out, bool defer_join = false) {
    const int F = st->use_mf ? st->F : 0;
    const int Hl = st->use_mlp ? st->units[st->n_layers - 1] : 0;
    if (nmf_deferred(st)) {
        EL_REQUIRE(st->batch_n == 0, "el_nmf_grads: the previous el_nmf_grads has not been applied (deferred decay: the batch's rows are "
                   "half-way between two steps until el_nmf_apply)");
        if (int rc = nmf_begin_rows(ctx, s, st, u, i, n)) return rc;
    }
    if (int rc = nmf_forward(ctx, s, st, u, i, n, true)) return rc;
    EL_CHECK_HIP(hipMemsetAsync(st->ghw, 0, (size_t)(F + Hl) * 4, s));
    if (st->head_bias) EL_CHECK_HIP(hipMemsetAsync(st->ghb, 0, 4, s));
    if (st->use_mlp)
        for (int l = 0; l < st->n_layers; ++l) EL_CHECK_HIP(hipMemsetAsync(st->gb[l], 0, (size_t)st->units[l] * 4, s));
    // the head leaves dact[last] = d loss / d pre-activation of the last layer (its ReLU derivative applied where the output row is
    // in registers anyway); the layers below take theirs in k_relu_bwd_colsum together with the bias gradient
    launch_nmf_head(ctx, s, st, label, n, 1, nullptr, loss_out, n_div, head_grid(n, ctx));
    if (st->use_mlp) {
        // (round 5) the ReLU derivative and the bias gradient of layer l - 1 ride in the epilogue of the product that writes its input
        // gradient (el_gemm_f32_x: one write of d instead of write + read + write, no read of d for the column sums) -- without Dropout
        // (its mask comes between the product and the ReLU test) and where the product runs on the split kernel unsplit
        // Two streams (round 5): the weight gradients gW[l] = in^T dact[l] (K = the batch: split-K products, 1.0 of the 3.2 ms the
        // tower's products take) are needed by the optimiser only; the chain dact[last] -> ... -> dX0 -> embedding scatter -> embedding
        // rows' Adam step does not wait for them.  They run on the library's second stream, forked as each dact[l] becomes final, and
        // overlap with the chain's bandwidth-bound tail (k_nmf_scatter, k_nmf_apply_rows).  Their split-K partials take the upper half of
        // the workspace (a host that sizes it 2 x el_gemm_ws_bytes gets the overlap; the option nmf_side = 0 turns it off: bench.py's
        // per-kernel breakdown runs one stream).
        size_t need_w = 0;
        for (int l = 0; l < st->n_layers; ++l) {
            const size_t a = el_gemm_ws_bytes(ctx, l == 0 ? 2 * (int64_t)st->E : st->units[l - 1], st->units[l], n);
            need_w = a > need_w ? a : need_w;
        }
        const size_t half = (st->ws_bytes / 2) & ~(size_t)255;
        bool side_on = ctx->opt.nmf_side != 0 && st->ws != nullptr && half >= need_w && n >= 4096 && !ctx->side_join_pending;
        if (side_on) side_on = el_side_stream_ready(ctx);
        hipStream_t ss = side_on ? ctx->side : s;
        void* ws_w = side_on ? (void*)((char*)st->ws + half) : st->ws;
        const size_t wsb_w = side_on ? half : st->ws_bytes, wsb_d = side_on ? half : st->ws_bytes;
        // a return out of the layer loop (a failing product or launch) joins whatever was forked to the side stream so far: the
        // caller's next call -- or its free of the activations / the workspace -- must not race with work still in flight there
        struct SideGuard {
            el_ctx* c; hipStream_t s, ss; bool armed;
            ~SideGuard() {
                if (!armed) return;
                (void)hipEventRecord(c->side_ev[7], ss);
                (void)hipStreamWaitEvent(s, c->side_ev[7], 0);
            }
        } guard{ctx, s, ss, side_on};
        bool done_below = false;                  // dact[l] already carries layer l's ReLU derivative and gb[l] its column sums
        for (int l = st->n_layers - 1; l >= 0; --l) {
            const int64_t units = st->units[l];
            const float* in = (l == 0) ? st->X0 : st->act[l - 1];
            const int64_t kin = (l == 0) ? 2 * (int64_t)st->E : st->units[l - 1];
            if (!done_below) {
                const int W = units < 256 ? (int)units : 256, R = 256 / W;
                const unsigned gx = (unsigned)((units + W - 1) / W);
                int64_t gy = ((int64_t)ctx->cus * 8 + gx - 1) / gx, rows = (n + R - 1) / R;
                if (gy > rows) gy = rows;
                if (l == st->n_layers - 1) {     // the head applied this layer's ReLU derivative already: column sums only
                    if (side_on) {               // (nothing on the chain needs them: second stream, behind the head)
                        EL_CHECK_HIP(hipEventRecord(ctx->side_ev[4], s));
                        EL_CHECK_HIP(hipStreamWaitEvent(ss, ctx->side_ev[4], 0));
                    }
                    EL_LAUNCH("k_nmf_colsum", k_nmf_colsum, dim3(gx, (unsigned)gy), dim3(256), 0, ss, st->dact[l], n, units, st->gb[l]);
                } else
                    EL_LAUNCH("k_relu_bwd_colsum", k_relu_bwd_colsum, dim3(gx, (unsigned)gy), dim3(256), 0, s, st->dact[l], st->act[l], n,
                              units, st->gb[l]);
            }
            done_below = false;
            if (side_on) {                        // dact[l] is final: its weight gradient goes to the side stream
                EL_CHECK_HIP(hipEventRecord(ctx->side_ev[l & 3], s));
                EL_CHECK_HIP(hipStreamWaitEvent(ss, ctx->side_ev[l & 3], 0));
            }
            if (int rc = el_gemm_f32(ctx, ss, 1, 0, kin, units, n, in, kin, st->dact[l], units, st->gW[l], units, nullptr, 0, ws_w, wsb_w)) return rc;
            float* din = (l == 0) ? st->dX0 : st->dact[l - 1];
            if (l >= 1 && !(st->dropout > 0.f)) {
                int fused = 0;
                if (int rc = el_gemm_f32_x(ctx, s, 0, 1, n, kin, units, st->dact[l], units, st->W[l], units, din, kin, nullptr, 0, st->act[l - 1], kin,
                                           st->gb[l - 1], st->ws, wsb_d, &fused)) return rc;
                done_below = fused != 0;
            } else if (int rc = el_gemm_f32(ctx, s, 0, 1, n, kin, units, st->dact[l], units, st->W[l], units, din, kin, nullptr, 0, st->ws, wsb_d)) return rc;
            // gradient w.r.t. the DROPPED input -> w.r.t. the layer below: the same mask again.  (The relu test of the layer
            // below then reads its dropped output: zero exactly where this mask is zero, positive where it was positive.)
            if (st->dropout > 0.f) nmf_dropout(s, st, din, n, kin, l);
        }
        if (side_on) {
            EL_CHECK_HIP(hipEventRecord(ctx->side_ev[7], ss));           // every weight gradient is complete
            ctx->side_join_pending = true;
            guard.armed = false;                                         // (the regular join: below, or inside the apply half)
        }
    }
    EL_LAUNCH("k_nmf_scatter", k_nmf_scatter, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, *st, u, i, n);
    EL_CHECK_LAUNCH();
    if (ctx->side_join_pending && !defer_join) {                         // el_nmf_grads: the caller reads the gradients next
        EL_CHECK_HIP(hipStreamWaitEvent(s, ctx->side_ev[7], 0));
        ctx->side_join_pending = false;
    }
    return 0;
}

// Keras Adam on every variable (dense apply; the embedding gradients are dense accumulators, zero again on exit)
static int nmf_apply(el_ctx* ctx, hipStream_t s, el_nmf_state* st, float lr_t) {
    const int F = st->use_mf ? st->F : 0;
    const int Hl = st->use_mlp ? st->units[st->n_layers - 1] : 0;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    const int64_t rows[4] = {st->U, st->I, st->U, st->I};
    const int64_t dims[4] = {st->F, st->F, st->E, st->E};
    if (nmf_deferred(st)) {
        EL_REQUIRE(st->batch_u && st->batch_i && st->batch_n >= 1, "el_nmf_apply: deferred decay applies the rows of the preceding el_nmf_grads");
        const int32_t t = st->opt_step + 1;
        EL_REQUIRE(t - st->hist_base < st->lr_hist_cap, "el_nmf_apply: lr history overrun");        // (nmf_begin_rows made room)
        EL_LAUNCH("k_nmf_apply_rows", k_nmf_apply_rows, dim3((unsigned)((2 * st->batch_n + 3) / 4)), dim3(256), 0, s, *st, st->batch_u,
                  st->batch_i, st->batch_n, t, lr_t);
        st->batch_u = st->batch_i = nullptr, st->batch_n = 0;
    }
    if (ctx->side_join_pending) {                  // the Dense layers' weight gradients (second stream) before their Adam step
        EL_CHECK_HIP(hipStreamWaitEvent(s, ctx->side_ev[7], 0));
        ctx->side_join_pending = false;
    }
    for (int t = 0; t < 4 && !nmf_deferred(st); ++t) {
        const bool on = (t < 2) ? st->use_mf : st->use_mlp;
        if (!on) continue;
        const int64_t cnt = rows[t] * dims[t];
        EL_LAUNCH("k_adam_dense_tab", k_adam_dense, dim3(g1(cnt / 4 + 1, ctx)), dim3(256), 0, s, st->tab[t], st->gtab[t],
                  st->mtab[t], st->vtab[t], cnt, lr_t, b1, b2, eps);
    }
    if (st->use_mlp) {
        for (int l = 0; l < st->n_layers; ++l) {
            const int64_t kin = (l == 0) ? 2 * (int64_t)st->E : st->units[l - 1];
            EL_LAUNCH("k_adam_apply_dense", k_adam_apply_dense, dim3(g1(kin * st->units[l], ctx)), dim3(256), 0, s, st->W[l],
                      st->gW[l], st->mW[l], st->vW[l], kin * st->units[l], lr_t, b1, b2, eps, 0);
            EL_LAUNCH("k_adam_apply_dense", k_adam_apply_dense, dim3(g1(st->units[l], ctx)), dim3(256), 0, s, st->b[l], st->gb[l],
                      st->mb[l], st->vb[l], (int64_t)st->units[l], lr_t, b1, b2, eps, 0);
        }
    }
    EL_LAUNCH("k_adam_apply_dense", k_adam_apply_dense, dim3(g1(F + Hl, ctx)), dim3(256), 0, s, st->hw, st->ghw, st->mhw, st->vhw,
              (int64_t)(F + Hl), lr_t, b1, b2, eps, 0);
    if (st->head_bias)
        EL_LAUNCH("k_adam_apply_dense", k_adam_apply_dense, dim3(1), dim3(256), 0, s, st->hb, st->ghb, st->mhb, st->vhb, (int64_t)1,
                  lr_t, b1, b2, eps, 0);
    EL_CHECK_LAUNCH();
    st->opt_step += 1;
    if (!nmf_deferred(st)) st->flushed_step = st->opt_step;
    return 0;
}

extern "C" int el_nmf_train_step(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                                 const float* label, int64_t n, int32_t step, float lr_t, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    if (n == 0) return 0;
    if (int rc = nmf_check(st, n, true)) return rc;
    EL_REQUIRE(u && i && label && loss_out && step >= 1, "el_nmf_train_step: bad arguments");
    if (nmf_deferred(st)) EL_REQUIRE(step == st->opt_step + 1, "el_nmf_train_step: step %d does not follow the state's %d applied steps", (int)step, (int)st->opt_step);
    if (int rc = nmf_grads(ctx, (hipStream_t)stream, st, u, i, label, n, n, loss_out, true)) return rc;
    return nmf_apply(ctx, (hipStream_t)stream, st, lr_t);
}

extern "C" int el_nmf_grads(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                            const float* label, int64_t n, int64_t n_global, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = nmf_check(st, n, true)) return rc;
    EL_REQUIRE(n >= 1 && n_global >= n && u && i && label && loss_out, "el_nmf_grads: bad arguments");
    EL_REQUIRE(!nmf_deferred(st) || n_global == n, "el_nmf_grads: a batch shared with other ranks (n_global > n) cannot use the deferred "
               "decay -- the rows other ranks' samples move are not known here; leave el_nmf_state.row_last NULL");
    return nmf_grads(ctx, (hipStream_t)stream, st, u, i, label, n, n_global, loss_out);
}

extern "C" int el_nmf_apply(el_ctx* ctx, void* stream, el_nmf_state* st, int32_t step, float lr_t) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = nmf_check(st, 1, true)) return rc;
    EL_REQUIRE(step >= 1, "el_nmf_apply: step >= 1");
    if (nmf_deferred(st)) EL_REQUIRE(step == st->opt_step + 1, "el_nmf_apply: step %d does not follow the state's %d applied steps", (int)step, (int)st->opt_step);
    return nmf_apply(ctx, (hipStream_t)stream, st, lr_t);
}
