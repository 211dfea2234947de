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
#include <rocprim/device/device_radix_sort.hpp>

extern "C" int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           const float* bias, int act, void* ws, size_t ws_bytes);
// el_bpr.hip / el_vae.hip
__global__ void k_adam_dense(float* th, float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps);
__global__ void k_adam_apply_dense(float* th, float* g, float* m, float* v, int64_t n, float alpha, float b1, float b2,
                                   float eps, int zero_g);

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
// each workgroup stores ONE row of partial sums (features, then the bias gradient; its loss share beside it): k_nmf_head_finish adds
// the workgroups' rows in a fixed order -- no float atomics, the same bits on every run.
// mfu / mfi (training): the two factors of the MF product as k_nmf_seg_fwd left them; NULL: the product itself in st.MF (k_nmf_gather).
#define NMF_HEAD_Q 16                                    // features per lane held in registers: F + Hl <= 1024
// Q = ceil((F + Hl) / 64) rounded up to a power of two (host): the feature loops carry no dead iterations; the head weights
// sit in registers; the NEXT sample's row is fetched while this one's reduction / exp / log chain runs.
template <int Q>
__global__ __launch_bounds__(256) void k_nmf_head(el_nmf_state st, const float* __restrict__ label, int64_t n, int mode,
                                                  float* out_prob, int64_t n_div, const float* __restrict__ mfu,
                                                  const float* __restrict__ mfi, float* __restrict__ part, double* __restrict__ ploss) {
    __shared__ float wsum[4], bsum[4];
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
            if (f < NF) v[q] = f < F ? (mfu ? mfu[b * F + f] * mfi[b * F + f] : st.MF[b * F + f]) : act_last[b * (int64_t)Hl + (f - F)];
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
    if (lane == 0) wsum[wv] = wl, bsum[wv] = bacc;
    __syncthreads();
    float* prow = part + (int64_t)blockIdx.x * (NF + 1);
    for (int f = threadIdx.x; f < NF; f += 256) prow[f] = (facc[0][f] + facc[1][f]) + (facc[2][f] + facc[3][f]);
    if (threadIdx.x == 0) {
        prow[NF] = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
        ploss[blockIdx.x] = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
    }
}

// The same head with FOUR samples per wave (16 lanes each; lane `sub` holds the float4 chunks sub + 16 q of the sample's F + Hl
// features): the reduction over a sample's features is four row operations instead of six, and the exp / log / division chain of the
// loss runs once per wave-instruction for four samples -- the one-sample form issues ~85 instructions per sample and is bound by
// instruction issue (0.21 ms at B = 262 144, eight waves per SIMD), this one ~35.  Needs F % 4 == 0 and Hl % 4 == 0 (16-byte chunks that
// do not straddle the MF / MLP boundary); summation orders differ from the one-sample form (results at fp32 rounding level).
template <int Q4>
__global__ __launch_bounds__(256) void k_nmf_head4(el_nmf_state st, const float* __restrict__ label, int64_t n, int mode,
                                                   float* out_prob, int64_t n_div, const float* __restrict__ mfu,
                                                   const float* __restrict__ mfi, float* __restrict__ part, double* __restrict__ ploss) {
    __shared__ float wsum[4], bsum[4];
    __shared__ float facc[4][64 * Q4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane & 15, grp = lane >> 4;
    const int F = st.use_mf ? st.F : 0;
    const int Hl = st.use_mlp ? st.units[st.n_layers - 1] : 0;
    const int NF = F + Hl;
    const float hbias = st.head_bias ? st.hb[0] : 0.f;
    const float* act_last = st.use_mlp ? st.act[st.n_layers - 1] : nullptr;
    float* dact_last = st.use_mlp ? st.dact[st.n_layers - 1] : nullptr;
    float4 acc[Q4], hwr[Q4];
#pragma unroll
    for (int q = 0; q < Q4; ++q) {
        acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int f = (sub + 16 * q) * 4;
        hwr[q] = f < NF ? *reinterpret_cast<const float4*>(st.hw + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto fetch = [&](int64_t b, float4 (&v)[Q4]) {
#pragma unroll
        for (int q = 0; q < Q4; ++q) {
            const int f = (sub + 16 * q) * 4;
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < F) {
                if (mfu) {
                    const float4 a = *reinterpret_cast<const float4*>(mfu + b * F + f), c = *reinterpret_cast<const float4*>(mfi + b * F + f);
                    v[q] = make_float4(a.x * c.x, a.y * c.y, a.z * c.z, a.w * c.w);
                } else v[q] = *reinterpret_cast<const float4*>(st.MF + b * F + f);
            } else if (f < NF) v[q] = *reinterpret_cast<const float4*>(act_last + b * (int64_t)Hl + (f - F));
        }
    };
    float bacc = 0.f, myloss = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 16;
    int64_t b = ((int64_t)blockIdx.x * 4 + wv) * 4 + grp;
    float4 nxt[Q4];
    if (b < n) fetch(b, nxt);
    for (; b < n; b += stride) {
        float4 val[Q4];
#pragma unroll
        for (int q = 0; q < Q4; ++q) val[q] = nxt[q];
        if (b + stride < n) fetch(b + stride, nxt);
        float pr = 0.f;
#pragma unroll
        for (int q = 0; q < Q4; ++q) pr += (val[q].x * hwr[q].x + val[q].y * hwr[q].y) + (val[q].z * hwr[q].z + val[q].w * hwr[q].w);
        const float logit = el_group_sum(pr, 16) + hbias;
        const float p = 1.0f / (1.0f + expf(-logit));
        if (mode == 0) {
            if (sub == 0) out_prob[b] = p;
            continue;
        }
        const float t = label[b];
        // K.binary_crossentropy (oracle/tf_clauses.py): the probability clipped to [1e-7, 1 - 1e-7], epsilon added again inside both logarithms
        const float pc = fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f);
        if (sub == 0) myloss += -(t * logf(pc + 1e-7f) + (1.0f - t) * logf((1.0f - pc) + 1e-7f)) / (float)n_div;
        float dlogit = 0.f;                                             // d/dlogit: zero where the clip is active
        if (p > 1e-7f && p < 1.0f - 1e-7f) {
            const float dp = -(t / (p + 1e-7f) - (1.0f - t) / ((1.0f - p) + 1e-7f));
            dlogit = dp * (p * (1.0f - p)) / (float)n_div;
        }
        if (sub == 0) st.dlogit[b] = dlogit, bacc += dlogit;
#pragma unroll
        for (int q = 0; q < Q4; ++q) {
            acc[q].x += dlogit * val[q].x, acc[q].y += dlogit * val[q].y, acc[q].z += dlogit * val[q].z, acc[q].w += dlogit * val[q].w;
            const int f = (sub + 16 * q) * 4;
            if (f >= F && f < NF)                                       // the last Dense(relu) layer's pre-activation gradient (its ReLU derivative here)
                *reinterpret_cast<float4*>(dact_last + b * (int64_t)Hl + (f - F)) =
                    make_float4(val[q].x > 0.f ? dlogit * hwr[q].x : 0.f, val[q].y > 0.f ? dlogit * hwr[q].y : 0.f,
                                val[q].z > 0.f ? dlogit * hwr[q].z : 0.f, val[q].w > 0.f ? dlogit * hwr[q].w : 0.f);
        }
    }
    if (mode == 0) return;
    // the four lane groups of a wave (different samples, same features) in a fixed order, then the four waves through LDS
#pragma unroll
    for (int q = 0; q < Q4; ++q) {
        float* a4 = reinterpret_cast<float*>(&acc[q]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float v = a4[x];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            a4[x] = v;
        }
        if (grp == 0) *reinterpret_cast<float4*>(&facc[wv][(sub + 16 * q) * 4]) = acc[q];
    }
    float wl = el_group_sum(myloss, 64), wb = el_group_sum(bacc, 64);
    if (lane == 0) wsum[wv] = wl, bsum[wv] = wb;
    __syncthreads();
    float* prow = part + (int64_t)blockIdx.x * (NF + 1);
    for (int f = threadIdx.x; f < NF; f += 256) prow[f] = (facc[0][f] + facc[1][f]) + (facc[2][f] + facc[3][f]);
    if (threadIdx.x == 0) {
        prow[NF] = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
        ploss[blockIdx.x] = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
    }
}

// ghw[f] = sum over the workgroups' partial rows (column f), ghb = column NF: el_gemm.hip's ordered column sums with the last column
// going to its own address
__global__ __launch_bounds__(1024) void k_nmf_head_finish(const float* __restrict__ part, int P, int NF, float* __restrict__ ghw,
                                                          float* __restrict__ ghb) {
    __shared__ float red[64][17];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc, C = NF + 1;
    const int per = (P + 63) / 64;
    const int r0 = tr * per, r1 = r0 + per < P ? r0 + per : P;
    float acc = 0.f;
    if (c < C) {
        for (int r = r0; r < r1; r += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (r + k < r1) ? part[(int64_t)(r + k) * C + c] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (r + k < r1) acc += v[k];
        }
    }
    red[tr][tc] = acc;
    __syncthreads();
    if (tr == 0 && c < C) {
        const int used = (P + per - 1) / per;
        float t = 0.f;
        for (int h = 0; h < used; ++h) t += red[h][tc];
        if (c < NF) ghw[c] = t;
        else if (ghb) ghb[0] = t;
    }
}

// *loss_out += the workgroups' loss shares, in workgroup order (one wave: lane l adds a contiguous share, lane 0 the 64 shares)
__global__ __launch_bounds__(64) void k_nmf_loss_finish(const double* __restrict__ ploss, int P, double* loss_out) {
    __shared__ double sh[64];
    const int per = (P + 63) / 64, r0 = threadIdx.x * per, r1 = r0 + per < P ? r0 + per : P;
    double a = 0.0;
    for (int r = r0; r < r1; ++r) a += ploss[r];
    sh[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int h = 0; h < 64; ++h) t += sh[h];
        *loss_out += t;
    }
}

// returns the number of workgroups launched (= partial rows written in training mode)
static unsigned launch_nmf_head(el_ctx* ctx, hipStream_t s, const el_nmf_state* st, const float* label, int64_t n, int mode, float* out_prob,
                            int64_t n_div, unsigned grid, const float* mfu, const float* mfi, float* part, double* ploss) {
    const int NF = (st->use_mf ? st->F : 0) + (st->use_mlp ? st->units[st->n_layers - 1] : 0);
    const int nq = (NF + 63) / 64;
    const int F = st->use_mf ? st->F : 0, Hl = NF - F;
    const bool al = [&] {                                             // 16-byte chunks: every row of every source starts aligned
        auto ok = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        if (F % 4 || Hl % 4 || !ok(st->hw)) return false;
        if (F && !(mfu ? ok(mfu) && ok(mfi) : ok(st->MF))) return false;
        if (Hl && !(ok(st->act[st->n_layers - 1]) && (mode == 0 || ok(st->dact[st->n_layers - 1])))) return false;
        return true;
    }();
    if (al && ctx->opt.nmf_head4 != 0) {
        unsigned g4 = (unsigned)((n + 15) / 16);
        if (g4 > grid) g4 = grid;
        if (g4 < 1) g4 = 1;
#define EL_HEAD4(Q_) EL_LAUNCH("k_nmf_head", k_nmf_head4<Q_>, dim3(g4), dim3(256), 0, s, *st, label, n, mode, out_prob, n_div, mfu, mfi, part, ploss)
        if (nq <= 1) EL_HEAD4(1);
        else if (nq <= 2) EL_HEAD4(2);
        else if (nq <= 4) EL_HEAD4(4);
        else if (nq <= 8) EL_HEAD4(8);
        else EL_HEAD4(16);
#undef EL_HEAD4
        return g4;
    }
#define EL_HEAD(Q_) EL_LAUNCH("k_nmf_head", k_nmf_head<Q_>, dim3(grid), dim3(256), 0, s, *st, label, n, mode, out_prob, n_div, mfu, mfi, part, ploss)
    if (nq <= 1) EL_HEAD(1);
    else if (nq <= 2) EL_HEAD(2);
    else if (nq <= 4) EL_HEAD(4);
    else if (nq <= 8) EL_HEAD(8);
    else EL_HEAD(16);
#undef EL_HEAD
    return grid;
}

// in place: d <- d * (y > 0)   (relu backward)
__global__ __launch_bounds__(256) void k_relu_bwd(float* __restrict__ d, const float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        if (!(y[t] > 0.f)) d[t] = 0.f;
}

// relu backward + bias gradient in one pass over d [n, units]:  d <- d * (y > 0);  part[blockIdx.y, c] = this workgroup's share of
// sum_b d[b, c] (el_colsum_finish adds the gridDim.y shares in order).  A workgroup owns the column block blockIdx.x (W = min(units,
// 256) columns, 256 / W rows at a time) and walks the rows blockIdx.y, blockIdx.y + gridDim.y, ...; the column sums stay in
// registers until the end.
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
        gb[(int64_t)blockIdx.y * units + c] = t;
    }
}

// bias gradient alone (partial rows as above); d already carries the ReLU derivative (k_nmf_head applies it for the last layer).
// Same walk as k_relu_bwd_colsum, one read of d and nothing else.
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
        gb[(int64_t)blockIdx.y * units + c] = t;
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

// ---- deferred decay of the embedding tables ---------------------------------------------------------------------------
// Keras' Adam moves EVERY row of an embedding table at every step: m <- b1 m, v <- b2 v,
// theta <- theta - lr_t m / (sqrt(v) + eps), gradient or not.  The eager form streams theta, g, m, v of all (U + I)(F + E)
// parameters per step: 3.3 of the 10 ms of a step at 1.25 M x 1 M x 128, for a batch that touches a fifth of the rows.
// For a row WITHOUT a gradient that update reads nothing but the row itself, so it can be postponed and replayed in
// registers -- the same fp32 operations on the same operands in the same order, hence the same bits -- at the moment the row is
// needed again: by a batch that contains it (k_nmf_seg_fwd, before the forward pass reads it) or by anything that reads the
// tables as a whole (k_nmf_flush_rows: scoring, weights(), a checkpoint).  row_last[side][r] = the optimiser step row r of
// that side's tables is current at; lr_hist[s - hist_base] = lr_t of step s.  Every (element, step) update is still performed
// exactly once; what disappears is the HBM round trip of the rows a step does not touch.
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

// one array of the side's tables (theta, or the gradient rows) alone
template <int VW>
__device__ __forceinline__ void nmf_rows_load_arr(float (&dst)[2][1][VW], const float* const (&src)[2], const NmfRowTabs& rt, int64_t row, int f0,
                                                  int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f = f0 + lane * VW;
#pragma unroll
        for (int x = 0; x < VW; ++x) dst[k][0][x] = 0.f;
        if (k < rt.n && f < rt.D[k]) {
            const float* sp = src[k] + row * rt.D[k] + f;
            if (VW == 2) {
                const float2 t = *reinterpret_cast<const float2*>(sp);
                dst[k][0][0] = t.x, dst[k][0][VW - 1] = t.y;
            } else dst[k][0][0] = sp[0];
        }
    }
}
template <int VW>
__device__ __forceinline__ void nmf_rows_load_theta(NmfRowRegs<VW, 1>& r, const NmfRowTabs& rt, int64_t row, int f0, int lane) {
    const float* const src[2] = {rt.th[0], rt.th[1]};
    nmf_rows_load_arr<VW>(r.a, src, rt, row, f0, lane);
}
template <int VW>
__device__ __forceinline__ void nmf_rows_load_g(float (&g)[2][1][VW], const NmfRowTabs& rt, int64_t row, int f0, int lane) {
    const float* const src[2] = {rt.g[0], rt.g[1]};
    nmf_rows_load_arr<VW>(g, src, rt, row, f0, lane);
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

// nsteps gradient-free steps on the registers (theta too when WITH_TH): exactly el_adam_elem with g = 0 -- or, ser (el_nmf_state.
// replay_series), in closed form from four row-level sums over the lr_t history (el_common.h: el_adam_series_*; O(1) per element
// whatever the gap, as close to the exact recurrence as the step-by-step fp32 form, not its bits)
template <int VW, int Q, bool WITH_TH>
__device__ __forceinline__ void nmf_rows_replay(NmfRowRegs<VW, Q>& r, const float* __restrict__ lr_from, int nsteps, bool ser = false) {
    const float b1 = 0.9f, b2 = 0.999f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    if (ser) {
        if (nsteps <= 0) return;
        const el_series sr = el_adam_series_sums(nsteps, [&](int s) { return lr_from[s]; });
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (WITH_TH) el_adam_series_apply<VW>(r.a[k][q], r.m[k][q], r.v[k][q], sr);
                else {
#pragma unroll
                    for (int x = 0; x < VW; ++x) r.m[k][q][x] = r.m[k][q][x] * sr.p1, r.v[k][q][x] = r.v[k][q][x] * sr.p2;
                }
            }
        return;
    }
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

// flush: theta, m, v of one row (both tables of the side) from step `last` to step t
template <int VW>
__device__ __forceinline__ void nmf_row_flush(const el_nmf_state& st, const NmfRowTabs& rt, int64_t row, int lane, int last, int32_t t) {
    const int nsteps = t - last;
    const float* lr_from = st.lr_hist + (last + 1 - st.hist_base);
    const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
    for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
        NmfRowRegs<VW, 1> r;
        nmf_rows_load<VW, 1, true>(r, rt, row, f0, lane);
        // m = v = 0 (rows that never had a gradient) is a fixed point of the gradient-free step -- m <- 0, v <- 0,
        // theta <- theta - lr 0 / (0 + eps) = theta: nothing to replay and nothing to write, whatever the gap
        bool nz = false;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int x = 0; x < VW; ++x) nz = nz || r.m[k][0][x] != 0.f || r.v[k][0][x] != 0.f;
        if (__ballot(nz) == 0ull) continue;
        nmf_rows_replay<VW, 1, true>(r, lr_from, nsteps, st.replay_series != 0);
        float* const dth[2] = {rt.th[0], rt.th[1]};
        float* const dm[2] = {rt.m[0], rt.m[1]};
        float* const dv[2] = {rt.v[0], rt.v[1]};
        nmf_rows_store<VW, 1>(r.a, dth, rt, row, f0, lane);
        nmf_rows_store<VW, 1>(r.m, dm, rt, row, f0, lane);
        nmf_rows_store<VW, 1>(r.v, dv, rt, row, f0, lane);
    }
}

// ---- the embedding side on sorted segments --------------------------------------------------------------------------------------------
// A batch names its rows through (u[b], i[b]); several samples may name the same row.  The step sorts the 2 n keys (u[b], then U + i[b];
// value = b; stable radix sort: ascending b inside a row's segment) once and walks the SEGMENTS both ways:
//   forward   k_nmf_seg_fwd: one wave per segment (the wave of the segment's first sorted position; the others leave).  Deferred decay:
//             the row's postponed gradient-free steps (last, t - 1] are replayed on theta now (m, v are read for it, not written: the
//             backward pass replays them -- two multiplications per step -- when it rewrites them anyway).  The row then goes to the
//             activation rows of the segment's samples: X0[b] = [Umlp[u] ; Imlp[i]], and the two factors of MF[b] = Umf[u] * Imf[i]
//             to mfp[0][b] / mfp[1][b] (the head multiplies them; the backward pass needs each factor again AFTER the partner's row
//             has moved on).  Samples 64, 65, ... of a long segment: k_nmf_seg_fwd_long (the rows go on a list here).
//   backward  k_nmf_seg_bwd: the gradient row of a segment = the sum of its samples' rows in ascending b -- no atomics, the same bits
//             on every run -- and then   MODE 0: stored to gtab (el_nmf_grads; the eager every-row Adam or el_nmf_apply consume it)
//                                        MODE 1: Keras' Adam step t on the row at once (el_nmf_train_step with the deferred decay)
//                                        MODE 2: no sums: step t with the row el_nmf_grads left in gtab (el_nmf_apply, deferred)
//             Segments of more than 64 samples (popular items; listed with their length by the forward pass): k_nmf_seg_bwd_part, a 16-wave
//             workgroup per block of 1 024 samples (wave w: samples [64 w, 64 w + 64) in ascending order, the waves' rows added in
//             wave order), then k_nmf_seg_bwd_long, a wave per row adding the blocks' rows in block order.  The order is a function
//             of the segment alone: MODE 0 + 2 and MODE 1 give the same bits.
// Replaces the rounds 3-5 pair k_nmf_catchup / k_nmf_gather and k_nmf_scatter (float atomics) / k_nmf_apply_rows.
struct NmfSeg {
    const u32* keys;       // [2 n] sorted
    const int32_t* perm;   // [2 n] sample of the sorted position
    int64_t n2;
    float* mfp[2];         // [n, F] copies of Umf[u[b]] / Imf[i[b]] as the forward pass saw them
    int32_t* llist;        // long segments: [0] count, [1] partial rows handed out, then (head position, samples, first partial row) triples
    float* lpart;          // partial gradient rows of the long segments' blocks of NMF_LBLK samples: [rows, Dsum]
    int32_t t;
    float lr_t;
};

__global__ __launch_bounds__(256) void k_nmf_keys(const int32_t* __restrict__ bu, const int32_t* __restrict__ bi, int64_t n, int64_t U,
                                                  u32* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    keys[b] = (u32)bu[b], keys[n + b] = (u32)(U + bi[b]);
    vals[b] = vals[n + b] = (int32_t)b;
}

// What the wave of sorted position p needs to know about its segment, with every load issued before the first use (one round trip to
// memory instead of three: the kernels below are bound by the length of their dependent-load chain, not by bytes): is p the first
// position of its segment (head), the segment's samples len = 1 .. 64, or 65 for "more than 64", and the sample of position p + lane.
#define NMF_LBLK 1024        // samples of a long segment per workgroup (16 waves x 64)
#define NMF_LY 8             // blocks of one segment in flight (gridDim.y of the long kernels)
struct NmfHead {
    bool head;
    u32 key;
    int len;
    int32_t myb;
};
__device__ __forceinline__ NmfHead nmf_seg_head(const NmfSeg& sg, int64_t p, int lane) {
    const int64_t last = sg.n2 - 1;
    const int64_t q = p - 1 + lane;                                          // lane 0: the position in front, lane l: p + l - 1
    const u32 kk = sg.keys[q < 0 ? 0 : (q < last ? q : last)];
    const int64_t q2 = p + 63 + (lane & 1);                                  // (lane 0: p + 63, lane 1: p + 64)
    const u32 kt = sg.keys[q2 < last ? q2 : last];
    const int64_t q3 = p + lane;
    NmfHead h;
    h.myb = sg.perm[q3 < last ? q3 : last];
    h.key = (u32)__shfl((int)kk, 1, 64);
    const u32 prev = (u32)__shfl((int)kk, 0, 64);
    h.head = p == 0 || prev != h.key;
    const bool same = lane >= 1 && q <= last && kk == h.key;                 // positions p .. p + 62
    const unsigned long long bal = __ballot(same) >> 1;
    const u32 k63 = (u32)__shfl((int)kt, 0, 64), k64 = (u32)__shfl((int)kt, 1, 64);
    if (bal != 0x7fffffffffffffffull) h.len = __builtin_ctzll(~bal);
    else if (!(p + 63 <= last && k63 == h.key)) h.len = 63;
    else h.len = (p + 64 <= last && k64 == h.key) ? 65 : 64;
    return h;
}

// theta of one row chunk (both tables of the side) -> the activation rows of sample b
template <int VW>
__device__ __forceinline__ void nmf_act_store(const el_nmf_state& st, const NmfSeg& sg, const NmfRowTabs& rt, int side, int64_t b, int f0, int lane,
                                              const float (&a)[2][1][VW]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f = f0 + lane * VW;
        if (k < rt.n && f < rt.D[k]) {
            const bool mf = st.use_mf && k == 0;
            float* d = mf ? sg.mfp[side] + b * st.F + f : st.X0 + b * 2 * st.E + (int64_t)side * st.E + f;
            if (VW == 2) *reinterpret_cast<float2*>(d) = make_float2(a[k][0][0], a[k][0][VW - 1]);
            else d[0] = a[k][0][0];
        }
    }
}

template <int VW, bool DEFER>
__global__ __launch_bounds__(256) void k_nmf_seg_fwd(el_nmf_state st, NmfSeg sg) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= sg.n2) return;
    const NmfHead hd = nmf_seg_head(sg, p, lane);
    if (!hd.head) return;
    const u32 key = hd.key;
    const int side = (int64_t)key >= st.U ? 1 : 0;
    const int64_t row = (int64_t)key - (side ? st.U : 0);
    int len = hd.len;
    if (len > 64) {                                                          // the rest: k_nmf_seg_fwd_long; the list serves the backward pass too
        int64_t lo = p + 64, hi = sg.n2;                                     // first position past the segment (keys ascend)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sg.keys[mid] <= key) lo = mid + 1;
            else hi = mid;
        }
        if (lane == 0) {
            const int total = (int)(lo - p), nblk = (total + NMF_LBLK - 1) / NMF_LBLK;
            const int ent = atomicAdd(sg.llist, 1), slot0 = atomicAdd(sg.llist + 1, nblk);
            sg.llist[2 + 3 * ent] = (int32_t)p, sg.llist[3 + 3 * ent] = total, sg.llist[4 + 3 * ent] = slot0;
        }
        len = 64;
    }
    const int32_t myb = hd.myb;
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    // the row stamp and the row itself are fetched TOGETHER (the stamp is not needed to address the row; m, v of the rare row that
    // needs no replay are wasted): one round trip instead of two in front of the replay
    const int last_v = DEFER ? st.row_last[side][row] : 0;
    const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
    for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
        NmfRowRegs<VW, 1> r;
        bool fresh = false;
        if (DEFER) nmf_rows_load<VW, 1, true>(r, rt, row, f0, lane);
        const int last = __builtin_amdgcn_readfirstlane(last_v);
        const int nsteps = DEFER ? (sg.t - 1) - last : 0;
        const float* lr_from = st.lr_hist + (last + 1 - st.hist_base);
        if (DEFER && nsteps > 0) {
            // m = v = 0 (rows that never had a gradient) is a fixed point of the gradient-free step: nothing to replay, nothing to write
            bool nz = false;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int x = 0; x < VW; ++x) nz = nz || r.m[k][0][x] != 0.f || r.v[k][0][x] != 0.f;
            if (__ballot(nz) != 0ull) {
                nmf_rows_replay<VW, 1, true>(r, lr_from, nsteps, st.replay_series != 0);
                fresh = true;
            }
        } else if (!DEFER) {
            nmf_rows_load_theta<VW>(r, rt, row, f0, lane);
        }
        if (fresh) {
            float* const dth[2] = {rt.th[0], rt.th[1]};
            nmf_rows_store<VW, 1>(r.a, dth, rt, row, f0, lane);
        }
        for (int j = 0; j < len; ++j) nmf_act_store<VW>(st, sg, rt, side, (int64_t)__shfl(myb, j, 64), f0, lane, r.a);
    }
}

// samples 64, 65, ... of the long segments: a workgroup per (listed row, block of NMF_LBLK samples), 64 samples per wave
template <int VW>
__global__ __launch_bounds__(1024) void k_nmf_seg_fwd_long(el_nmf_state st, NmfSeg sg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nlist = sg.llist[0];
    for (int ent = blockIdx.x; ent < nlist; ent += gridDim.x) {
        const int64_t p = sg.llist[2 + 3 * ent];
        const int total = sg.llist[3 + 3 * ent];
        const u32 key = sg.keys[p];
        const int side = (int64_t)key >= st.U ? 1 : 0;
        const int64_t row = (int64_t)key - (side ? st.U : 0);
        const NmfRowTabs rt = nmf_row_tabs(st, side);
        const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
        for (int j0 = blockIdx.y * NMF_LBLK + wv * 64; j0 < total; j0 += NMF_LY * NMF_LBLK) {
            if (j0 < 64) continue;                                           // (k_nmf_seg_fwd wrote the first 64)
            const int cnt = total - j0 < 64 ? total - j0 : 64;
            const int32_t b = lane < cnt ? sg.perm[p + j0 + lane] : 0;
            for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
                NmfRowRegs<VW, 1> r;
                nmf_rows_load_theta<VW>(r, rt, row, f0, lane);               // (k_nmf_seg_fwd has brought the row to t - 1)
                for (int j = 0; j < cnt; ++j) nmf_act_store<VW>(st, sg, rt, side, (int64_t)__shfl(b, j, 64), f0, lane, r.a);
            }
        }
    }
}

// g += the gradient rows of the cnt <= 64 samples the lanes hold in myb (lane j: the j-th), ascending, NB samples' loads in flight
//   MF tables: d loss / d Umf[u] = (dlogit_b h_f) Imf[i]_f (the partner's factor as the forward pass saw it); MLP tables: the halves of dX0[b]
template <int VW, int NB>
__device__ __forceinline__ void nmf_seg_sum(const el_nmf_state& st, const NmfSeg& sg, int side, int32_t myb, int cnt, int f0, int lane,
                                            float (&g)[2][VW]) {
    const int f = f0 + lane * VW;
    float hwv[VW];
#pragma unroll
    for (int x = 0; x < VW; ++x) hwv[x] = 0.f;
    const bool has_mf = st.use_mf && f < st.F;
    if (has_mf) {
#pragma unroll
        for (int x = 0; x < VW; ++x) hwv[x] = st.hw[f + x];
    }
    const int kml = st.use_mf ? 1 : 0;                                       // table slot of the MLP embedding
    const bool has_ml = st.use_mlp && f < st.E;
    const float* part = sg.mfp[side ^ 1];
    for (int j = 0; j < cnt; j += NB) {
        float pv[NB][VW], dv[NB][VW], dl[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int64_t b = __shfl(myb, j + q < cnt ? j + q : j, 64);
            dl[q] = st.dlogit[b];
#pragma unroll
            for (int x = 0; x < VW; ++x) pv[q][x] = dv[q][x] = 0.f;
            if (has_mf) {
                if (VW == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(part + b * st.F + f);
                    pv[q][0] = t2.x, pv[q][VW - 1] = t2.y;
                } else pv[q][0] = part[b * st.F + f];
            }
            if (has_ml) {
                const float* dx = st.dX0 + b * 2 * st.E + (int64_t)side * st.E + f;
                if (VW == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(dx);
                    dv[q][0] = t2.x, dv[q][VW - 1] = t2.y;
                } else dv[q][0] = dx[0];
            }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (j + q >= cnt) continue;
#pragma unroll
            for (int x = 0; x < VW; ++x) {
                if (has_mf) {
                    const float s = dl[q] * hwv[x];
                    g[0][x] += s * pv[q][x];
                }
                if (has_ml) g[kml][x] += dv[q][x];
            }
        }
    }
}

// the same for a segment of ONE sample b (nine rows in ten): 0 + term, as the loop above adds it
template <int VW>
__device__ __forceinline__ void nmf_seg_one(const el_nmf_state& st, const NmfSeg& sg, int side, int64_t b, int f0, int lane, float (&g)[2][VW]) {
    const int f = f0 + lane * VW;
    const int kml = st.use_mf ? 1 : 0;
    if (st.use_mf && f < st.F) {
        const float dl = st.dlogit[b];
        const float* part = sg.mfp[side ^ 1] + b * st.F + f;
#pragma unroll
        for (int x = 0; x < VW; ++x) {
            const float s = dl * st.hw[f + x];
            g[0][x] += s * part[x];
        }
    }
    if (st.use_mlp && f < st.E) {
        const float* dx = st.dX0 + b * 2 * st.E + (int64_t)side * st.E + f;
#pragma unroll
        for (int x = 0; x < VW; ++x) g[kml][x] += dx[x];
    }
}

// the row chunk's gradient g -> gtab (MODE 0) or Keras' Adam step t on theta, m, v (MODE 1, 2; m, v first replayed to t - 1)
template <int VW, int MODE>
__device__ __forceinline__ void nmf_seg_finish(const el_nmf_state& st, const NmfSeg& sg, const NmfRowTabs& rt, int64_t row, int f0, int lane,
                                               NmfRowRegs<VW, 1>& r, float (&g)[2][1][VW], int nsteps, const float* lr_from) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    if (MODE == 0) {
        float* const dg[2] = {rt.g[0], rt.g[1]};
        nmf_rows_store<VW, 1>(g, dg, rt, row, f0, lane);
        return;
    }
    nmf_rows_replay<VW, 1, false>(r, lr_from, nsteps, st.replay_series != 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int x = 0; x < VW; ++x) el_adam_elem(r.a[k][0][x], r.m[k][0][x], r.v[k][0][x], g[k][0][x], sg.lr_t, b1, b2, omb1, omb2, eps);
    float* const dth[2] = {rt.th[0], rt.th[1]};
    float* const dm[2] = {rt.m[0], rt.m[1]};
    float* const dv[2] = {rt.v[0], rt.v[1]};
    nmf_rows_store<VW, 1>(r.a, dth, rt, row, f0, lane);
    nmf_rows_store<VW, 1>(r.m, dm, rt, row, f0, lane);
    nmf_rows_store<VW, 1>(r.v, dv, rt, row, f0, lane);
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int x = 0; x < VW; ++x) g[k][0][x] = 0.f;
        float* const dg[2] = {rt.g[0], rt.g[1]};
        nmf_rows_store<VW, 1>(g, dg, rt, row, f0, lane);
    }
}

template <int VW, int MODE>
__global__ __launch_bounds__(256) void k_nmf_seg_bwd(el_nmf_state st, NmfSeg sg) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (MODE != 0 && p == 0 && lane == 0) st.lr_hist[sg.t - st.hist_base] = sg.lr_t;     // this step's lr_t for later replays (this
    //                                                                                        kernel's own replays read steps < t only)
    if (p >= sg.n2) return;
    const NmfHead hd = nmf_seg_head(sg, p, lane);
    if (!hd.head) return;
    const u32 key = hd.key;
    const int side = (int64_t)key >= st.U ? 1 : 0;
    const int64_t row = (int64_t)key - (side ? st.U : 0);
    const int len = hd.len;
    if (MODE != 2 && len > 64) return;                                        // k_nmf_seg_bwd_long's (on the list since the forward pass)
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    const int last_v = MODE != 0 ? st.row_last[side][row] : 0;              // (consumed after the row loads below are in flight)
    const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
    for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
        NmfRowRegs<VW, 1> r;
        if (MODE != 0) nmf_rows_load<VW, 1, true>(r, rt, row, f0, lane);
        float g[2][1][VW];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int x = 0; x < VW; ++x) g[k][0][x] = 0.f;
        if (MODE == 2) {
            nmf_rows_load_g<VW>(g, rt, row, f0, lane);
        } else {
            float gs[2][VW];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int x = 0; x < VW; ++x) gs[k][x] = 0.f;
            if (len == 1) nmf_seg_one<VW>(st, sg, side, (int64_t)__shfl(hd.myb, 0, 64), f0, lane, gs);
            else nmf_seg_sum<VW, 4>(st, sg, side, hd.myb, len, f0, lane, gs);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int x = 0; x < VW; ++x) g[k][0][x] = gs[k][x];
        }
        const int last = __builtin_amdgcn_readfirstlane(last_v);
        nmf_seg_finish<VW, MODE>(st, sg, rt, row, f0, lane, r, g, (sg.t - 1) - last, st.lr_hist + (last + 1 - st.hist_base));
    }
    if (MODE != 0 && lane == 0) st.row_last[side][row] = sg.t;
}

// long segments, step 1: the partial gradient row of every block of NMF_LBLK samples -- wave w of the block's workgroup adds samples
// [64 w, 64 w + 64) of the block in ascending order, the 16 waves' rows are added in wave order -> lpart[first row of the segment + block]
template <int VW>
__global__ __launch_bounds__(1024) void k_nmf_seg_bwd_part(el_nmf_state st, NmfSeg sg) {
    __shared__ float s_red[16][2][64 * VW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nlist = sg.llist[0];
    const int Fa = st.use_mf ? st.F : 0, Dsum = Fa + (st.use_mlp ? st.E : 0);
    for (int ent = blockIdx.x; ent < nlist; ent += gridDim.x) {
        const int64_t p = sg.llist[2 + 3 * ent];
        const int total = sg.llist[3 + 3 * ent], slot0 = sg.llist[4 + 3 * ent];
        const u32 key = sg.keys[p];
        const int side = (int64_t)key >= st.U ? 1 : 0;
        const NmfRowTabs rt = nmf_row_tabs(st, side);
        const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
        const int nblk = (total + NMF_LBLK - 1) / NMF_LBLK;
        for (int blk = blockIdx.y; blk < nblk; blk += NMF_LY) {
            const int j0 = blk * NMF_LBLK + wv * 64;
            const int cnt = j0 >= total ? 0 : (total - j0 < 64 ? total - j0 : 64);
            const int used = ((total - blk * NMF_LBLK < NMF_LBLK ? total - blk * NMF_LBLK : NMF_LBLK) + 63) / 64;
            const int32_t myb = lane < cnt ? sg.perm[p + j0 + lane] : 0;
            for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
                float gs[2][VW];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int x = 0; x < VW; ++x) gs[k][x] = 0.f;
                if (cnt > 0) nmf_seg_sum<VW, 8>(st, sg, side, myb, cnt, f0, lane, gs);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int x = 0; x < VW; ++x) s_red[wv][k][lane * VW + x] = gs[k][x];
                __syncthreads();
                if (wv < 2 && wv < rt.n) {                                   // wave k adds table k's 16 rows in wave order
                    const int f = f0 + lane * VW;
                    if (f < rt.D[wv]) {
                        float* d = sg.lpart + (int64_t)(slot0 + blk) * Dsum + (wv == 1 ? Fa : 0) + f;
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            float t = 0.f;
                            for (int h = 0; h < used; ++h) t += s_red[h][wv][lane * VW + x];
                            d[x] = t;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

// long segments, step 2: one wave per listed row adds its blocks' partial rows in block order, then as k_nmf_seg_bwd (MODE 0 / 1)
template <int VW, int MODE>
__global__ __launch_bounds__(256) void k_nmf_seg_bwd_long(el_nmf_state st, NmfSeg sg) {
    const int lane = threadIdx.x & 63;
    const int nlist = sg.llist[0];
    const int Fa = st.use_mf ? st.F : 0, Dsum = Fa + (st.use_mlp ? st.E : 0);
    for (int ent = blockIdx.x * 4 + (threadIdx.x >> 6); ent < nlist; ent += gridDim.x * 4) {
        const int64_t p = sg.llist[2 + 3 * ent];
        const int total = sg.llist[3 + 3 * ent], slot0 = sg.llist[4 + 3 * ent];
        const u32 key = sg.keys[p];
        const int side = (int64_t)key >= st.U ? 1 : 0;
        const int64_t row = (int64_t)key - (side ? st.U : 0);
        const NmfRowTabs rt = nmf_row_tabs(st, side);
        const int nblk = (total + NMF_LBLK - 1) / NMF_LBLK;
        int nsteps = 0;
        const float* lr_from = nullptr;
        if (MODE != 0) {
            const int last = __builtin_amdgcn_readfirstlane(st.row_last[side][row]);
            nsteps = (sg.t - 1) - last;
            lr_from = st.lr_hist + (last + 1 - st.hist_base);
        }
        const int Dmax = rt.n == 2 ? (rt.D[0] > rt.D[1] ? rt.D[0] : rt.D[1]) : rt.D[0];
        for (int f0 = 0; f0 < Dmax; f0 += 64 * VW) {
            NmfRowRegs<VW, 1> r;
            if (MODE != 0) nmf_rows_load<VW, 1, true>(r, rt, row, f0, lane);
            float g[2][1][VW];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int x = 0; x < VW; ++x) g[k][0][x] = 0.f;
            const int f = f0 + lane * VW;
            for (int blk = 0; blk < nblk; blk += 8) {
                float v[8][2][VW];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float* src = sg.lpart + (int64_t)(slot0 + (blk + q < nblk ? blk + q : blk)) * Dsum + (k == 1 ? Fa : 0) + f;
#pragma unroll
                        for (int x = 0; x < VW; ++x) v[q][k][x] = (k < rt.n && f < rt.D[k]) ? src[x] : 0.f;
                    }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (blk + q >= nblk) continue;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int x = 0; x < VW; ++x) g[k][0][x] += v[q][k][x];
                }
            }
            nmf_seg_finish<VW, MODE>(st, sg, rt, row, f0, lane, r, g, nsteps, lr_from);
        }
        if (MODE != 0 && lane == 0) st.row_last[side][row] = sg.t;
    }
}

// every row of one side up to step t (one wave per row)
__global__ __launch_bounds__(256) void k_nmf_flush_rows(el_nmf_state st, int side, int64_t rows, int32_t t) {
    const int lane = threadIdx.x & 63;
    const NmfRowTabs rt = nmf_row_tabs(st, side);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int last = __builtin_amdgcn_readfirstlane(st.row_last[side][row]);
        if (t - last <= 0) continue;
        const bool even = (rt.D[0] % 2 == 0) && (rt.n < 2 || rt.D[1] % 2 == 0);      // VW = 2 when every table row starts 8-byte aligned
        if (even) nmf_row_flush<2>(st, rt, row, lane, last, t);
        else nmf_row_flush<1>(st, rt, row, lane, last, t);
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

// ---- the step workspace (el_nmf_state.step_ws): sort buffers, the long-segment list, the MF factor copies, partial sums ------------------
static int nmf_bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && (1LL << b) < n) ++b;
    return b;
}
struct NmfStepWs {
    // two complete sort sets: el_nmf_presort orders the NEXT batch into the set the current step does not use
    u32 *kin[2], *kout[2];
    int32_t *vin[2], *vout[2];
    void* tmp[2];
    size_t tmp_bytes;
    int32_t* llist;
    float* lpart;
    float* mfp[2];
    float* hpart;          // head: [P, NF + 1]
    double* hloss;         // head: [P]
    float* cpart[4];       // per layer: column-sum partial rows (k_nmf_colsum / k_relu_bwd_colsum: <= P rows; the GEMM epilogue: ceil(B / 128))
};
static int nmf_carve(el_ctx* ctx, const el_nmf_state* st, void* base, NmfStepWs* w, size_t* total) {
    const int64_t B = st->Bmax;
    const int64_t P = (int64_t)ctx->cus * 8;                      // workgroups of the head / rows of the column-sum grids at most
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* r = base ? (void*)((char*)base + off) : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return r;
    };
    for (int k = 0; k < 2; ++k) {
        w->kin[k] = (u32*)take((size_t)2 * B * 4), w->kout[k] = (u32*)take((size_t)2 * B * 4);
        w->vin[k] = (int32_t*)take((size_t)2 * B * 4), w->vout[k] = (int32_t*)take((size_t)2 * B * 4);
    }
    size_t tb = 0;
    u32* np = nullptr;
    int32_t* nv = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, tb, np, np, nv, nv, (unsigned)(2 * B), 0, nmf_bits_for(st->U + st->I), (hipStream_t)0) != hipSuccess) {
        el_set_error("el_nmf: rocprim::radix_sort_pairs size query failed");
        return 1;
    }
    w->tmp_bytes = tb;
    for (int k = 0; k < 2; ++k) w->tmp[k] = take(tb + 256);
    w->llist = (int32_t*)take((size_t)(3 * (2 * B / 65 + 2) + 8) * 4);
    {   // partial rows of the long segments' blocks: at most one per NMF_LBLK positions + one per long segment
        const int64_t Dsum = (st->use_mf ? st->F : 0) + (st->use_mlp ? st->E : 0);
        w->lpart = (float*)take((size_t)(2 * B / NMF_LBLK + 2 * B / 65 + 4) * Dsum * 4);
    }
    for (int side = 0; side < 2; ++side) w->mfp[side] = st->use_mf ? (float*)take((size_t)B * st->F * 4) : nullptr;
    const int NF = (st->use_mf ? st->F : 0) + (st->use_mlp ? st->units[st->n_layers > 0 ? st->n_layers - 1 : 0] : 0);
    w->hpart = (float*)take((size_t)P * (NF + 1) * 4);
    w->hloss = (double*)take((size_t)P * 8);
    for (int l = 0; l < 4; ++l) {
        w->cpart[l] = nullptr;
        if (!st->use_mlp || l >= st->n_layers) continue;
        const int64_t rows = P > (B + 127) / 128 ? P : (B + 127) / 128;
        w->cpart[l] = (float*)take((size_t)rows * st->units[l] * 4);
    }
    *total = off;
    return 0;
}

// Bytes of el_nmf_state.step_ws for this state's shape (U, I, Bmax, F, E, units; 256-byte aligned base)
extern "C" size_t el_nmf_step_ws_bytes(el_ctx* ctx, const el_nmf_state* st) {
    if (el_bind(ctx) || !st || st->Bmax < 1 || st->n_layers < 0 || st->n_layers > 4) return 0;
    NmfStepWs w;
    size_t total = 0;
    if (nmf_carve(ctx, st, nullptr, &w, &total)) return 0;
    return total;
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
    if (train) {
        EL_REQUIRE(st->step_ws != nullptr && (uintptr_t)st->step_ws % 256 == 0, "el_nmf: training needs el_nmf_state.step_ws (el_nmf_step_ws_bytes, 256-byte aligned)");
        EL_REQUIRE(st->U + st->I < (1LL << 31), "el_nmf: U + I must stay below 2^31 (sort keys)");
    }
    if (st->row_last[0] || st->row_last[1]) {
        EL_REQUIRE(st->row_last[0] && st->row_last[1] && st->lr_hist && st->lr_hist_cap >= 2, "el_nmf: deferred decay needs row_last[2] and lr_hist");
        EL_REQUIRE(st->opt_step >= 0 && st->flushed_step >= 0 && st->flushed_step <= st->opt_step && st->hist_base >= 1,
                   "el_nmf: deferred-decay counters corrupt (zero-initialise opt_step / flushed_step, hist_base = 1)");
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

// The step's segments: sort keys (u[b], U + i[b]) -> (keys, perm); with the deferred decay, room in the lr history first.
static inline bool nmf_even(const el_nmf_state* st) {      // every table row starts 8-byte aligned: float2 accesses
    return (!st->use_mf || st->F % 2 == 0) && (!st->use_mlp || st->E % 2 == 0);
}

static int nmf_sort_into(el_ctx* ctx, hipStream_t s, const el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n, const NmfStepWs* w,
                         int set) {
    EL_LAUNCH("k_nmf_keys", k_nmf_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, i, n, st->U, w->kin[set], w->vin[set]);
    ElKernelTimer tm("rocprim_radix_sort_pairs", s);
    size_t tb = w->tmp_bytes;
    EL_CHECK_HIP(rocprim::radix_sort_pairs(w->tmp[set], tb, w->kin[set], w->kout[set], w->vin[set], w->vout[set], (unsigned)(2 * n), 0,
                                           nmf_bits_for(st->U + st->I), s));
    (void)ctx;
    return 0;
}

static int nmf_sort_batch(el_ctx* ctx, hipStream_t s, el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n, NmfStepWs* w,
                          NmfSeg* sg) {
    size_t total = 0;
    if (int rc = nmf_carve(ctx, st, st->step_ws, w, &total)) return rc;
    EL_REQUIRE(st->step_ws_bytes >= total, "el_nmf: step_ws holds %zu bytes, el_nmf_step_ws_bytes asks for %zu", st->step_ws_bytes, total);
    EL_REQUIRE(st->sort_set == 0 || st->sort_set == 1, "el_nmf: sort_set corrupt (zero-initialise the state)");
    if (nmf_deferred(st) && (st->opt_step + 1) - st->hist_base >= st->lr_hist_cap)     // history full: every row to t - 1, the history restarts at t
        if (int rc = nmf_sync(ctx, s, st)) return rc;
    if (st->pre_u == u && st->pre_i == i && st->pre_n == n && u != nullptr) {           // el_nmf_presort ordered this batch already
        st->sort_set ^= 1;
        st->pre_u = st->pre_i = nullptr, st->pre_n = 0;
    } else if (int rc = nmf_sort_into(ctx, s, st, u, i, n, w, st->sort_set)) return rc;
    sg->keys = w->kout[st->sort_set], sg->perm = w->vout[st->sort_set], sg->n2 = 2 * n;
    sg->mfp[0] = w->mfp[0], sg->mfp[1] = w->mfp[1];
    sg->llist = w->llist, sg->lpart = w->lpart;
    sg->t = st->opt_step + 1, sg->lr_t = 0.f;
    return 0;
}

static void nmf_dropout(hipStream_t s, const el_nmf_state* st, float* x, int64_t n, int64_t width, int layer) {
    const int64_t threads = n * ((width + 3) / 4);
    EL_LAUNCH("k_nmf_dropout", k_nmf_dropout, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, n, (int)width, st->dropout,
              (u64)st->drop_seed, (u32)st->drop_step, (u32)layer);
}

static int nmf_forward(el_ctx* ctx, hipStream_t s, const el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n,
                       bool train = false, const NmfSeg* sg = nullptr) {
    if (sg) {                                        // training: catch-up (deferred decay) + gather, one wave per segment
        EL_CHECK_HIP(hipMemsetAsync(sg->llist, 0, 8, s));
        const bool even = nmf_even(st);
        const dim3 grid((unsigned)((sg->n2 + 3) / 4));
        const bool defer = nmf_deferred(st);
        if (even && defer) EL_LAUNCH("k_nmf_seg_fwd", (k_nmf_seg_fwd<2, true>), grid, dim3(256), 0, s, *st, *sg);
        else if (even) EL_LAUNCH("k_nmf_seg_fwd", (k_nmf_seg_fwd<2, false>), grid, dim3(256), 0, s, *st, *sg);
        else if (defer) EL_LAUNCH("k_nmf_seg_fwd", (k_nmf_seg_fwd<1, true>), grid, dim3(256), 0, s, *st, *sg);
        else EL_LAUNCH("k_nmf_seg_fwd", (k_nmf_seg_fwd<1, false>), grid, dim3(256), 0, s, *st, *sg);
        const dim3 gl((unsigned)ctx->cus, NMF_LY);
        if (even) EL_LAUNCH("k_nmf_seg_fwd_long", k_nmf_seg_fwd_long<2>, gl, dim3(1024), 0, s, *st, *sg);
        else EL_LAUNCH("k_nmf_seg_fwd_long", k_nmf_seg_fwd_long<1>, gl, dim3(1024), 0, s, *st, *sg);
    } else
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
    launch_nmf_head(ctx, s, st, nullptr, n, 0, out_prob, n, head_grid(n, ctx), nullptr, nullptr, nullptr, nullptr);
    EL_CHECK_LAUNCH();
    return 0;
}

// forward, BinaryCrossentropy (mean over n_div samples: n_div = n, or the global batch when several ranks share a step),
// backward: every gradient buffer of the state is complete on exit
// defer_join (el_nmf_train_step): the weight-gradient products may still be running on the library's second stream when this returns;
// nmf_apply waits for them after it has launched the embedding rows' step (ctx->side_join_pending)
// the embedding rows of the step's segments: MODE 0 gradient rows -> gtab, 1 sums + Adam, 2 Adam on the gtab rows (header of the kernels)
static void nmf_launch_bwd(hipStream_t s, const el_nmf_state* st, const NmfSeg* sg, int mode, el_ctx* ctx) {
    const bool even = nmf_even(st);
    const dim3 grid((unsigned)((sg->n2 + 3) / 4)), gl((unsigned)ctx->cus, NMF_LY), gc((unsigned)ctx->cus);
#define NMF_BWD(VW_, M_)                                                                                              \
    do {                                                                                                              \
        EL_LAUNCH("k_nmf_seg_bwd", (k_nmf_seg_bwd<VW_, M_>), grid, dim3(256), 0, s, *st, *sg);                        \
        if (M_ != 2) {                                                                                                \
            EL_LAUNCH("k_nmf_seg_bwd_part", k_nmf_seg_bwd_part<VW_>, gl, dim3(1024), 0, s, *st, *sg);                 \
            EL_LAUNCH("k_nmf_seg_bwd_long", (k_nmf_seg_bwd_long<VW_, (M_ == 2 ? 0 : M_)>), gc, dim3(256), 0, s, *st, *sg);   \
        }                                                                                                             \
    } while (0)
    if (even) {
        if (mode == 0) NMF_BWD(2, 0);
        else if (mode == 1) NMF_BWD(2, 1);
        else NMF_BWD(2, 2);
    } else {
        if (mode == 0) NMF_BWD(1, 0);
        else if (mode == 1) NMF_BWD(1, 1);
        else NMF_BWD(1, 2);
    }
#undef NMF_BWD
}

// fused_rows (el_nmf_train_step with the deferred decay): the embedding rows' sums and their Adam step in one pass, lr_t known here
static int nmf_grads(el_ctx* ctx, hipStream_t s, el_nmf_state* st, const int32_t* u, const int32_t* i, const float* label,
                     int64_t n, int64_t n_div, double* loss_out, bool defer_join = false, bool fused_rows = false, float lr_t = 0.f) {
    const int F = st->use_mf ? st->F : 0;
    const int Hl = st->use_mlp ? st->units[st->n_layers - 1] : 0;
    if (nmf_deferred(st))
        EL_REQUIRE(st->batch_n == 0, "el_nmf_grads: the previous el_nmf_grads has not been applied (deferred decay: the batch's rows are "
                   "half-way between two steps until el_nmf_apply)");
    NmfStepWs w;
    NmfSeg sg;
    if (int rc = nmf_sort_batch(ctx, s, st, u, i, n, &w, &sg)) return rc;
    if (int rc = nmf_forward(ctx, s, st, u, i, n, true, &sg)) return rc;
    // the head leaves dact[last] = d loss / d pre-activation of the last layer (its ReLU derivative applied where the output row is
    // in registers anyway); the layers below take theirs in k_relu_bwd_colsum together with the bias gradient
    const unsigned hg = launch_nmf_head(ctx, s, st, label, n, 1, nullptr, n_div, head_grid(n, ctx), w.mfp[0], w.mfp[1], w.hpart, w.hloss);
    {   // the workgroups' partial rows, added in order: head weights; the bias gradient is column F + Hl of the same rows
        EL_LAUNCH("k_nmf_head_finish", k_nmf_head_finish, dim3((unsigned)((F + Hl + 1 + 15) / 16)), dim3(1024), 0, s, w.hpart, (int)hg, F + Hl, st->ghw,
                  st->head_bias ? st->ghb : nullptr);
        EL_LAUNCH("k_nmf_loss_finish", k_nmf_loss_finish, dim3(1), dim3(64), 0, s, w.hloss, (int)hg, loss_out);
    }
    if (st->use_mlp) {
        // (round 5) the ReLU derivative and the bias gradient of layer l - 1 ride in the epilogue of the product that writes its input
        // gradient (el_gemm_f32_x: one write of d instead of write + read + write, no read of d for the column sums) -- without Dropout
        // (its mask comes between the product and the ReLU test) and where the product runs on the split kernel unsplit
        // Two streams (round 5): the weight gradients gW[l] = in^T dact[l] (K = the batch: split-K products, 1.0 of the 3.2 ms the
        // tower's products take) are needed by the optimiser only; the chain dact[last] -> ... -> dX0 -> embedding scatter -> embedding
        // rows' Adam step does not wait for them.  They run on the library's second stream, forked as each dact[l] becomes final, and
        // overlap with the chain's bandwidth-bound tail (the embedding rows' segment pass).  Their split-K partials take the upper half of
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
                    EL_LAUNCH("k_nmf_colsum", k_nmf_colsum, dim3(gx, (unsigned)gy), dim3(256), 0, ss, st->dact[l], n, units, w.cpart[l]);
                    if (int rc = el_colsum_finish(ss, w.cpart[l], (int)gy, units, st->gb[l])) return rc;
                } else {
                    EL_LAUNCH("k_relu_bwd_colsum", k_relu_bwd_colsum, dim3(gx, (unsigned)gy), dim3(256), 0, s, st->dact[l], st->act[l], n,
                              units, w.cpart[l]);
                    if (int rc = el_colsum_finish(s, w.cpart[l], (int)gy, units, st->gb[l])) return rc;
                }
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
                                           st->gb[l - 1], w.cpart[l - 1], st->ws, wsb_d, &fused)) return rc;
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
    // the embedding rows: gradient rows of the segments -> gtab (the eager Adam passes or el_nmf_apply take them from there), or sums
    // + step t at once
    sg.lr_t = lr_t;
    nmf_launch_bwd(s, st, &sg, fused_rows ? 1 : 0, ctx);
    EL_CHECK_LAUNCH();
    if (nmf_deferred(st) && !fused_rows) st->batch_u = u, st->batch_i = i, st->batch_n = n;
    if (ctx->side_join_pending && !defer_join) {                         // el_nmf_grads: the caller reads the gradients next
        EL_CHECK_HIP(hipStreamWaitEvent(s, ctx->side_ev[7], 0));
        ctx->side_join_pending = false;
    }
    return 0;
}

// Keras Adam on every variable (dense apply; the embedding gradients are dense accumulators, zero again on exit)
static int nmf_apply(el_ctx* ctx, hipStream_t s, el_nmf_state* st, float lr_t, bool rows_done = false) {
    const int F = st->use_mf ? st->F : 0;
    const int Hl = st->use_mlp ? st->units[st->n_layers - 1] : 0;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    const int64_t rows[4] = {st->U, st->I, st->U, st->I};
    const int64_t dims[4] = {st->F, st->F, st->E, st->E};
    if (nmf_deferred(st) && !rows_done) {
        EL_REQUIRE(st->batch_u && st->batch_i && st->batch_n >= 1, "el_nmf_apply: deferred decay applies the rows of the preceding el_nmf_grads");
        const int32_t t = st->opt_step + 1;
        EL_REQUIRE(t - st->hist_base < st->lr_hist_cap, "el_nmf_apply: lr history overrun");        // (nmf_sort_batch made room)
        NmfStepWs w;                                  // the sorted keys of the el_nmf_grads call are still in the step workspace
        size_t total = 0;
        if (int rc = nmf_carve(ctx, st, st->step_ws, &w, &total)) return rc;
        NmfSeg sg;
        sg.keys = w.kout[st->sort_set], sg.perm = w.vout[st->sort_set], sg.n2 = 2 * st->batch_n, sg.mfp[0] = w.mfp[0], sg.mfp[1] = w.mfp[1], sg.llist = w.llist, sg.lpart = w.lpart;
        sg.t = t, sg.lr_t = lr_t;
        nmf_launch_bwd(s, st, &sg, 2, ctx);
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

// The (row, sample) keys of a batch ordered AHEAD of its step: the sort reads u and i only, so a training loop whose sampler runs a batch
// ahead (it never reads the model) calls this on another stream while the previous step trains; the el_nmf_train_step / el_nmf_grads
// call that follows with the SAME u, i, n (the arrays unchanged in between, this call complete in the step's stream order: the caller's
// event) then skips its own sort.  One batch can be pending; a step on other arrays sorts for itself and leaves the pending one alone.
extern "C" int el_nmf_presort(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i, int64_t n) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = nmf_check(st, n, true)) return rc;
    EL_REQUIRE(u && i && n >= 1, "el_nmf_presort: bad arguments");
    NmfStepWs w;
    size_t total = 0;
    if (int rc = nmf_carve(ctx, st, st->step_ws, &w, &total)) return rc;
    EL_REQUIRE(st->step_ws_bytes >= total, "el_nmf: step_ws holds %zu bytes, el_nmf_step_ws_bytes asks for %zu", st->step_ws_bytes, total);
    EL_REQUIRE(st->sort_set == 0 || st->sort_set == 1, "el_nmf: sort_set corrupt (zero-initialise the state)");
    if (int rc = nmf_sort_into(ctx, (hipStream_t)stream, st, u, i, n, &w, st->sort_set ^ 1)) return rc;
    EL_CHECK_LAUNCH();
    st->pre_u = u, st->pre_i = i, st->pre_n = n;
    return 0;
}

extern "C" int el_nmf_train_step(el_ctx* ctx, void* stream, el_nmf_state* st, const int32_t* u, const int32_t* i,
                                 const float* label, int64_t n, int32_t step, float lr_t, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    if (n == 0) return 0;
    if (int rc = nmf_check(st, n, true)) return rc;
    EL_REQUIRE(u && i && label && loss_out && step >= 1, "el_nmf_train_step: bad arguments");
    if (nmf_deferred(st)) EL_REQUIRE(step == st->opt_step + 1, "el_nmf_train_step: step %d does not follow the state's %d applied steps", (int)step, (int)st->opt_step);
    const bool fused_rows = nmf_deferred(st);
    if (int rc = nmf_grads(ctx, (hipStream_t)stream, st, u, i, label, n, n, loss_out, true, fused_rows, lr_t)) return rc;
    return nmf_apply(ctx, (hipStream_t)stream, st, lr_t, fused_rows);
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
