// Graph propagation of the BPR heads that sit on a user-item graph: LightGCN and NGCF.
//
// Replaces LightGCN_model._propagate_embeddings (graph_based/lightgcn/LightGCN_model.py:68-94) and the sparse half of
// NGCF_model._propagate_embeddings (graph_based/ngcf/NGCF_model.py:106-142): tf.sparse.sparse_dense_matmul of the normalised
// adjacency (LightGCN.py:96-118: D^-1/2 A D^-1/2 over the U + I nodes, 2 T non-zeros) with the stacked embedding table [Gu; Gi].
// The reference cuts the matrix into n_fold row blocks only to fit TensorFlow's memory; the product is the same.
//
// k_spmm_csr: Y[r, :] = sum_p vals[p] X[indices[p], :] over the CSR row r -- an HBM-bound gather: per non-zero one 4 F-byte row of X
// (random), 4 + 4 bytes of index and value (streamed), per row one 4 F-byte store: nnz (8 + 4 F) + N 4 F bytes per layer.
//   * one LANE GROUP (lpt lanes x 16 bytes = a row of F floats) per CHUNK of at most SPMM_CH consecutive non-zeros of one row: item
//     rows of a Zipf catalogue hold 10^4 .. 10^5 neighbours, user rows 10 .. 10^3 -- a group per ROW would leave the launch waiting
//     for a handful of groups.  The chunk list (row, first non-zero, slot) is built once per graph by the host (the Laplacian does
//     not change during training).
//   * a group reads its chunk's (index, value) pairs lpt at a time, one per lane (coalesced), and broadcasts them by shuffle: four row
//     gathers in flight per lane;
//   * a row that fits one chunk gets its epilogue (below) in the same kernel; a row cut into several chunks leaves one partial row per
//     chunk in `part` and k_spmm_finish adds them IN CHUNK ORDER (no floating-point atomics: the sum order is fixed, runs repeat bit for
//     bit) and applies the epilogue.
// Epilogue (LightGCN's layer combination fused into the product): with y = (L X)[r],
//     acc_new = (first ? x_self * 1 : acc_old) + alpha * y          (all_embeddings[k] * all_alphas[k], summed: :88-90)
//     store_y    -> Y[r]   = y          (the next layer's input)
//     store_acc  -> ACC[r] = acc_new
//     last       -> OUT[r] = acc_new / n_terms                      (reduce_mean over the n_layers + 1 stacked tables)
// The stacked table is two tables (users, then items): row r < n0 lives in X0, the rest in X1 -- no concatenated copy is made.
#include "el_common.h"

#define SPMM_CH 512

int el_pick_lpt(int F, int vw, int* cpl);      // el_bpr.hip

struct SpmmParams {
    const int64_t* indptr;     // [N + 1]
    const int32_t* indices;    // [nnz]
    const float* vals;         // [nnz]
    const int32_t* chunk_row;  // [n_chunks]
    const int64_t* chunk_lo;   // [n_chunks] first non-zero of the chunk
    const int32_t* chunk_slot; // [n_chunks] -1: the row's only chunk; else index of the partial row in `part`
    int64_t n_chunks;
    const float *X0, *X1;      // input rows [0, n0) and [n0, N)
    int64_t n0, N;
    int F, lpt;
    float* part;               // [n_partials, F]
    // epilogue
    const float *S0, *S1;      // x_self (first layer: the layer-0 table itself), split like X
    float *Y0, *Y1;            // the product (next layer's input) or NULL
    float* ACC;                // [N, F] running weighted sum or NULL
    float *O0, *O1;            // final table (split) or NULL
    float alpha, n_terms;
    int first, last;
    // finish pass
    const int32_t* mrow;       // [n_multi] rows cut into several chunks
    const int32_t* mslot;      // [n_multi] first partial slot of the row
    const int32_t* mcnt;       // [n_multi] number of partials
    int64_t n_multi;
};

__device__ __forceinline__ const float* spmm_row(const float* A0, const float* A1, int64_t n0, int64_t r, int F) {
    return r < n0 ? A0 + r * F : A1 + (r - n0) * F;
}

template <int CPL>
__device__ __forceinline__ void spmm_epilogue(const SpmmParams& p, int64_t r, int sub, const float (&y)[CPL][4]) {
    const int F = p.F;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * p.lpt) * 4;
        if (e >= F) continue;
        float4 yv = make_float4(y[q][0], y[q][1], y[q][2], y[q][3]);
        if (p.Y0) *reinterpret_cast<float4*>(const_cast<float*>(spmm_row(p.Y0, p.Y1, p.n0, r, F)) + e) = yv;
        if (p.ACC || p.O0) {
            float4 a;
            if (p.first) {
                const float4 s = *reinterpret_cast<const float4*>(spmm_row(p.S0, p.S1, p.n0, r, F) + e);
                a = make_float4(s.x * 1.0f, s.y * 1.0f, s.z * 1.0f, s.w * 1.0f);
            } else {
                a = *reinterpret_cast<const float4*>(p.ACC + r * F + e);
            }
            a.x = a.x + p.alpha * yv.x, a.y = a.y + p.alpha * yv.y, a.z = a.z + p.alpha * yv.z, a.w = a.w + p.alpha * yv.w;
            if (p.last) {
                // reduce_mean over the stacked tables: the sum divided by their number
                const float d = p.n_terms;
                a.x = a.x / d, a.y = a.y / d, a.z = a.z / d, a.w = a.w / d;
                *reinterpret_cast<float4*>(const_cast<float*>(spmm_row(p.O0, p.O1, p.n0, r, F)) + e) = a;
            } else {
                *reinterpret_cast<float4*>(p.ACC + r * F + e) = a;
            }
        }
    }
}

template <int CPL>
__global__ __launch_bounds__(256) void k_spmm_csr(SpmmParams p) {
    const int lpt = p.lpt, F = p.F;
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    if (c >= p.n_chunks) return;
    const int64_t r = p.chunk_row[c];
    const int64_t lo = p.chunk_lo[c];
    int64_t hi = p.indptr[r + 1];
    if (hi > lo + SPMM_CH) hi = lo + SPMM_CH;
    float y[CPL][4];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int x = 0; x < 4; ++x) y[q][x] = 0.f;
    for (int64_t base = lo; base < hi; base += lpt) {
        const int cnt = (int)((hi - base < lpt) ? hi - base : lpt);
        int32_t my_j = 0;
        float my_w = 0.f;
        if (sub < cnt) {
            my_j = p.indices[base + sub];
            my_w = p.vals[base + sub];
        }
        const int lane0 = (int)(threadIdx.x & 63) & ~(lpt - 1);          // first lane of this group inside the wave
        for (int t0 = 0; t0 < cnt; t0 += 4) {
            float4 xr[4][CPL];
            float w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tt = t0 + t < cnt ? t0 + t : t0;               // (group-uniform; a repeated entry is weighted 0 below)
                const int64_t j = (int64_t)__shfl(my_j, lane0 + tt, 64);
                w[t] = t0 + t < cnt ? __shfl(my_w, lane0 + tt, 64) : 0.f;
                const float* src = spmm_row(p.X0, p.X1, p.n0, j, F);
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * 4;
                    xr[t][q] = e < F ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t0 + t >= cnt) continue;                             // (group-uniform: the sum holds exactly the row's terms, in order)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    y[q][0] += w[t] * xr[t][q].x;
                    y[q][1] += w[t] * xr[t][q].y;
                    y[q][2] += w[t] * xr[t][q].z;
                    y[q][3] += w[t] * xr[t][q].w;
                }
            }
        }
    }
    const int slot = p.chunk_slot[c];
    if (slot < 0) {
        spmm_epilogue<CPL>(p, r, sub, y);
    } else {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * 4;
            if (e < F) *reinterpret_cast<float4*>(p.part + (int64_t)slot * F + e) = make_float4(y[q][0], y[q][1], y[q][2], y[q][3]);
        }
    }
}

// rows cut into several chunks: their partial rows added in chunk order, then the epilogue.  One lane group per row, eight partials in
// flight.
template <int CPL>
__global__ __launch_bounds__(256) void k_spmm_finish(SpmmParams p) {
    const int lpt = p.lpt, F = p.F;
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    if (g >= p.n_multi) return;
    const int64_t r = p.mrow[g];
    const int64_t s0 = p.mslot[g];
    const int n = p.mcnt[g];
    float y[CPL][4];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int x = 0; x < 4; ++x) y[q][x] = 0.f;
    for (int k0 = 0; k0 < n; k0 += 8) {
        float4 v[8][CPL];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = k0 + k < n ? k0 + k : k0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * 4;
                v[k][q] = e < F ? *reinterpret_cast<const float4*>(p.part + (s0 + kk) * F + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k0 + k >= n) continue;
#pragma unroll
            for (int q = 0; q < CPL; ++q) y[q][0] += v[k][q].x, y[q][1] += v[k][q].y, y[q][2] += v[k][q].z, y[q][3] += v[k][q].w;
        }
    }
    spmm_epilogue<CPL>(p, r, sub, y);
}

static int spmm_launch(hipStream_t s, const SpmmParams& p) {
    const int groups = (p.F / 4 + p.lpt - 1) / p.lpt;
    const unsigned g1 = (unsigned)((p.n_chunks * p.lpt + 255) / 256), g2 = (unsigned)((p.n_multi * p.lpt + 255) / 256);
    if (groups <= 1) {
        if (p.n_chunks) EL_LAUNCH("k_spmm_csr", k_spmm_csr<1>, dim3(g1), dim3(256), 0, s, p);
        if (p.n_multi) EL_LAUNCH("k_spmm_finish", k_spmm_finish<1>, dim3(g2), dim3(256), 0, s, p);
    } else if (groups == 2) {
        if (p.n_chunks) EL_LAUNCH("k_spmm_csr", k_spmm_csr<2>, dim3(g1), dim3(256), 0, s, p);
        if (p.n_multi) EL_LAUNCH("k_spmm_finish", k_spmm_finish<2>, dim3(g2), dim3(256), 0, s, p);
    } else {
        if (p.n_chunks) EL_LAUNCH("k_spmm_csr", k_spmm_csr<4>, dim3(g1), dim3(256), 0, s, p);
        if (p.n_multi) EL_LAUNCH("k_spmm_finish", k_spmm_finish<4>, dim3(g2), dim3(256), 0, s, p);
    }
    EL_CHECK_LAUNCH();
    return 0;
}

static int spmm_check(const char* fn, const el_graph_csr* g, int F) {
    EL_REQUIRE(g && g->indptr && g->indices && g->vals && g->N >= 1 && g->n0 >= 0 && g->n0 <= g->N, "%s: incomplete graph", fn);
    EL_REQUIRE(g->n_chunks >= 0 && (g->n_chunks == 0 || (g->chunk_row && g->chunk_lo && g->chunk_slot)), "%s: chunk list missing", fn);
    EL_REQUIRE(g->n_multi >= 0 && (g->n_multi == 0 || (g->multi_row && g->multi_slot && g->multi_cnt && g->part)), "%s: multi-chunk rows need their list and the partial buffer", fn);
    EL_REQUIRE(F >= 4 && F % 4 == 0 && F <= 1024, "%s: F=%d (multiples of 4 up to 1024)", fn, F);
    return 0;
}

static void spmm_fill(SpmmParams& p, const el_graph_csr* g, int F) {
    memset(&p, 0, sizeof(p));
    p.indptr = g->indptr, p.indices = g->indices, p.vals = g->vals;
    p.chunk_row = g->chunk_row, p.chunk_lo = g->chunk_lo, p.chunk_slot = g->chunk_slot, p.n_chunks = g->n_chunks;
    p.mrow = g->multi_row, p.mslot = g->multi_slot, p.mcnt = g->multi_cnt, p.n_multi = g->n_multi;
    p.part = g->part;
    p.n0 = g->n0, p.N = g->N;
    p.F = F;
    int cpl = 1;
    p.lpt = el_pick_lpt(F, 4, &cpl);
    p.n_terms = 1.f;
}

// Y = L X  (rows [0, n0) of X / Y in X0 / Y0, the rest in X1 / Y1)
extern "C" int el_spmm_csr_f32(el_ctx* ctx, void* stream, const el_graph_csr* g, const float* X0, const float* X1, int32_t F, float* Y0,
                               float* Y1) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = spmm_check("el_spmm_csr_f32", g, F)) return rc;
    EL_REQUIRE(X0 && Y0 && (g->n0 == g->N || (X1 && Y1)), "el_spmm_csr_f32: null table");
    EL_REQUIRE((((uintptr_t)X0 | (uintptr_t)X1 | (uintptr_t)Y0 | (uintptr_t)Y1 | (uintptr_t)g->part) & 15) == 0, "el_spmm_csr_f32: tables must be 16-byte aligned");
    SpmmParams p;
    spmm_fill(p, g, F);
    p.X0 = X0, p.X1 = X1 ? X1 : X0, p.Y0 = Y0, p.Y1 = Y1 ? Y1 : Y0;
    return spmm_launch((hipStream_t)stream, p);
}

extern "C" size_t el_lightgcn_ws_bytes(int64_t U, int64_t I, int32_t F, int32_t n_layers) {
    if (U < 0 || I < 0 || F < 1 || n_layers < 0) return 0;
    const size_t tab = (((size_t)(U + I) * F * 4) + 255) & ~(size_t)255;
    return n_layers == 0 ? 0 : (n_layers == 1 ? tab : (n_layers == 2 ? 2 * tab : 3 * tab));
}

// [Gu; Gi] <- mean_k alpha_k L^k [Gu; Gi], alpha_0 = 1, alpha_k = 1 / (1 + k)   (LightGCN_model.py:68-94), in place
extern "C" int el_lightgcn_propagate(el_ctx* ctx, void* stream, const el_graph_csr* g, float* Gu, float* Gi, int32_t F, int32_t n_layers,
                                     void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = spmm_check("el_lightgcn_propagate", g, F)) return rc;
    EL_REQUIRE(Gu && Gi && n_layers >= 0 && n_layers <= 16, "el_lightgcn_propagate: bad arguments");
    if (n_layers == 0) return 0;                               // mean of the one-element stack [E * 1]
    const int64_t U = g->n0, I = g->N - g->n0;
    const size_t need = el_lightgcn_ws_bytes(U, I, F, n_layers);
    EL_REQUIRE(ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0, "el_lightgcn_propagate: workspace too small (%zu < %zu)", ws_bytes, need);
    EL_REQUIRE((((uintptr_t)Gu | (uintptr_t)Gi | (uintptr_t)g->part) & 15) == 0, "el_lightgcn_propagate: tables must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t tab = (((size_t)(U + I) * F * 4) + 255) & ~(size_t)255;
    float* T = (float*)ws;                                     // the result (it replaces Gu / Gi at the end) -- and the running sum
    float* P[2] = {n_layers >= 2 ? (float*)((char*)ws + tab) : nullptr, n_layers >= 3 ? (float*)((char*)ws + 2 * tab) : nullptr};
    for (int k = 1; k <= n_layers; ++k) {
        SpmmParams p;
        spmm_fill(p, g, F);
        const float* in0 = k == 1 ? Gu : P[(k - 2) % (n_layers >= 3 ? 2 : 1)];
        p.X0 = in0, p.X1 = k == 1 ? Gi : in0 + (size_t)U * F;
        p.S0 = Gu, p.S1 = Gi;
        p.first = k == 1, p.last = k == n_layers;
        p.alpha = (float)(1.0 / (1.0 + (double)k));            // (Python float 1 / (1 + k), cast to fp32 at the multiplication)
        p.n_terms = (float)(n_layers + 1);
        p.ACC = T;
        if (!p.last) {
            float* out = P[(k - 1) % (n_layers >= 3 ? 2 : 1)];
            p.Y0 = out, p.Y1 = out + (size_t)U * F;
        } else {
            p.O0 = T, p.O1 = T + (size_t)U * F;
        }
        if (int rc = spmm_launch(s, p)) return rc;
    }
    EL_CHECK_HIP(hipMemcpyAsync(Gu, T, (size_t)U * F * 4, hipMemcpyDeviceToDevice, s));
    EL_CHECK_HIP(hipMemcpyAsync(Gi, T + (size_t)U * F, (size_t)I * F * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---- NGCF (graph_based/ngcf/NGCF_model.py:106-142): the dense half of one embedding-propagation layer ------------------------------
// With ego = E_{k-1} [N, kin] and lap = L ego (el_spmm_csr_f32), the reference forms
//     first  = (lap + ego) W1 + b1,   second = (ego * lap) W2 + b2,   ego' = leaky_relu(first + second),   ego' = dropout(ego', rate),
//     all_embeddings += l2_normalize(ego', axis = 1)
// Here: k_ngcf_pre writes X2 = [lap + ego | ego * lap] ([N, 2 kin]), ONE product X2 [W1; W2] + (b1 + b2) on the library's GEMM gives
// first + second, and k_ngcf_post applies leaky_relu (slope 0.2, tf.nn.leaky_relu's default) and the dropout mask, stores ego' for the
// next layer and its row-normalised copy into the column block [col_off, col_off + kout) of Gu / Gi (row stride W): the `tf.concat(...,
// 1)` + `assign` of :139-142 without a concatenated temporary.
__global__ __launch_bounds__(256) void k_ngcf_pre(const float* __restrict__ ego, const float* __restrict__ lap, int64_t n4, int k4, float* __restrict__ X2) {
    // one thread per float4 of ego: row r = t / k4, chunk c = t % k4
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const int64_t r = t / k4;
    const int c = (int)(t - r * k4);
    const float4 e = reinterpret_cast<const float4*>(ego)[t], l = reinterpret_cast<const float4*>(lap)[t];
    float4* o = reinterpret_cast<float4*>(X2) + r * 2 * k4;
    o[c] = make_float4(l.x + e.x, l.y + e.y, l.z + e.z, l.w + e.w);
    o[k4 + c] = make_float4(e.x * l.x, e.y * l.y, e.z * l.z, e.w * l.w);
}

// one wave per row
__global__ __launch_bounds__(256) void k_ngcf_post(const float* __restrict__ S, int64_t N, int64_t n0, int kout, float slope, float rate,
                                                   u64 seed, u32 step, float* __restrict__ ego_next, float* __restrict__ Gu,
                                                   float* __restrict__ Gi, int W, int col_off) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const float keep_scale = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    float ss = 0.f;
    for (int c = lane; c < kout; c += 64) {
        float v = S[r * kout + c];
        v = v > 0.f ? v : slope * v;
        if (rate > 0.f) {
            // counter-based mask (the reference's is TensorFlow's stateful uniform stream: not reproducible outside TensorFlow)
            const el_philox4 q = el_philox4x32_10((u32)(r & 0xffffffffu), (u32)((u64)r >> 32), (u32)c, step, (u32)seed, (u32)(seed >> 32));
            const float uni = (float)(q.x >> 8) * (1.0f / 16777216.0f);
            v = uni >= rate ? v * keep_scale : 0.f;
        }
        ego_next[r * kout + c] = v;
        ss += v * v;
    }
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));                  // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, epsilon = 1e-12))
    float* dst = r < n0 ? Gu + r * W + col_off : Gi + (r - n0) * W + col_off;
    for (int c = lane; c < kout; c += 64) dst[c] = ego_next[r * kout + c] * inv;
}

extern "C" int el_ngcf_pre(el_ctx* ctx, void* stream, const float* ego, const float* lap, int64_t N, int32_t k, float* X2) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(ego && lap && X2 && N >= 0 && k >= 4 && k % 4 == 0, "el_ngcf_pre: bad arguments (k=%d must be a multiple of 4)", k);
    EL_REQUIRE((((uintptr_t)ego | (uintptr_t)lap | (uintptr_t)X2) & 15) == 0, "el_ngcf_pre: tables must be 16-byte aligned");
    const int64_t n4 = N * (k / 4);
    if (n4 == 0) return 0;
    EL_LAUNCH("k_ngcf_pre", k_ngcf_pre, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ego, lap, n4, k / 4, X2);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_ngcf_post(el_ctx* ctx, void* stream, const float* S, int64_t N, int64_t n0, int32_t kout, float rate, uint64_t seed,
                            uint32_t step, float* ego_next, float* Gu, float* Gi, int32_t W, int32_t col_off) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(S && ego_next && Gu && Gi && N >= 0 && n0 >= 0 && n0 <= N && kout >= 1 && col_off >= 0 && col_off + kout <= W,
               "el_ngcf_post: bad arguments");
    EL_REQUIRE(rate >= 0.f && rate < 1.f, "el_ngcf_post: dropout rate %g outside [0, 1)", (double)rate);
    if (N == 0) return 0;
    EL_LAUNCH("k_ngcf_post", k_ngcf_post, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, N, n0, (int)kout, 0.2f, rate,
              (u64)seed, (u32)step, ego_next, Gu, Gi, (int)W, (int)col_off);
    EL_CHECK_LAUNCH();
    return 0;
}

// Keras Adam (dense apply) on a variable whose only gradient is its L2 term: g = two_lw * theta.  (NGCF's GraphLayers: the tape of
// NGCF_model.train_step reaches them through reg_loss alone -- the propagation's result is ASSIGNED to Gu / Gi, :141-142.)
__global__ __launch_bounds__(256) void k_adam_l2_dense(float* __restrict__ th, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_t,
                                                       float two_lw) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    const float g = two_lw * th[t];
    const float mm = m[t] + (g - m[t]) * (1.0f - b1);           // Keras dense apply: m += (g - m) (1 - b1)
    const float vv = v[t] + (g * g - v[t]) * (1.0f - b2);
    th[t] = th[t] - (mm * lr_t) / (sqrtf(vv) + eps);
    m[t] = mm, v[t] = vv;
}

extern "C" int el_adam_l2_dense(el_ctx* ctx, void* stream, float* theta, float* m, float* v, int64_t n, float lr_t, float two_lw) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(theta && m && v && n >= 0, "el_adam_l2_dense: bad arguments");
    if (n == 0) return 0;
    EL_LAUNCH("k_adam_l2_dense", k_adam_l2_dense, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, theta, m, v, n, lr_t, two_lw);
    EL_CHECK_LAUNCH();
    return 0;
}
