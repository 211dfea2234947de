// Collaborative Metric Learning -- replaces CML_model.train_step / predict
// (elliot/recommender/latent_factor_models/CML/CML_model.py:58-106); contract in include/elliot_hip.h.
//
// What the reference computes.  user / pos / neg arrive as [B,1]; gamma_* are squeezed to [B,F] but the squared distance is
// reduced with keepdims=True ([B,1]) while the squeezed item bias is [B], so `score = -dist + beta` BROADCASTS to [B,B]
// (:60-66): score[a,b] = -dist_a + beta_b.  The hinge therefore runs over all B^2 pairs of (triplet a's distances, triplet
// b's biases):
//     diff[a,b] = D_a + E_b,   D_a = |u_a - j_a|^2 - |u_a - i_a|^2,   E_b = beta(i_b) - beta(j_b)
//     loss = sum_{a,b} max(margin - clip(diff[a,b], -80, 1e8), 0) + l_w (|u|^2 + |i|^2 + |j|^2)/2 + l_b beta_i^2/2 + l_b beta_j^2/20
// That sum separates: with Es / Ds the sorted E / D values,
//     n_a = #{b : -80 - D_a <= E_b <= margin - D_a}   (pairs whose hinge is active and inside the clip)  -> dloss/dD_a = -n_a
//     m_b = #{a : -80 - E_b <= D_a <= margin - E_b}                                                     -> dloss/dE_b = -m_b
//     low_a = #{b : E_b < -80 - D_a}                   (clipped pairs: constant margin + 80 each)
//     sum_{a,b} = sum_a [ n_a (margin - D_a) - m_a E_a + low_a (margin + 80) ]
// so one step is O(B log B): k_cml_fwd (gathers, D, E, regulariser) -> two float radix sorts -> k_cml_coef (four binary
// searches per triplet) -> row gradients (sorted segments of el_bpr_sorted.hip with the coefficients given; k_cml_bwd with
// float atomics below 2048 triplets) -> the TF-semantics dense
// Adam of the BPR path (same state struct).  dD/du = 2 (i - j), dD/di = 2 (u - i), dD/dj = -2 (u - j).
#include "el_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

int el_pick_lpt(int F, int vw, int* cpl);                                                                       // el_bpr.hip
int el_bpr_sorted_cml_grads(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, const int32_t* u, const int32_t* i,
                            const int32_t* j, int64_t B, float l_w, float l_b, float* cD, const float* cE, void* ws,
                            size_t ws_bytes);                                                                    // el_bpr_sorted.hip
int el_bprmf_apply_optimizer(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, const int32_t* u, const int32_t* i,
                             const int32_t* j, int64_t B, float lr, int opt, int32_t step, float lr_t);          // el_bpr.hip

namespace {

template <int VW>
__device__ __forceinline__ void cml_ld(const float* p, float* dst) {
    if (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        dst[0] = t.x, dst[1] = t.y, dst[2] = t.z, dst[3] = t.w;
    } else {
        dst[0] = p[0];
    }
}

struct CmlArgs {
    el_bprmf_state st;
    const int32_t* bu;
    const int32_t* bi;
    const int32_t* bj;
    int64_t B;
    float l_w, l_b, margin;
    int lpt;
    float* D;       // [B]
    float* E;       // [B]
    float* Ds;      // sorted copies of the GLOBAL batch's D / E (n_all values; = the local batch on one GPU)
    float* Es;
    int64_t n_all;
    float* cD;      // [B] dloss/dD
    float* cE;      // [B] dloss/dE
    double* loss_out;
};

template <typename T>
__device__ __forceinline__ void block_add(T v, double* out) {
    __shared__ double part[4];
    double l = (double)v;
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (part[0] + part[1]) + (part[2] + part[3]);
        if (t != 0.0) atomicAdd(out, t);
    }
}

// phase 0: D, E, regulariser.  phase 1: row gradients from cD / cE.
template <int VW, int CPL, int PHASE>
__global__ __launch_bounds__(256) void k_cml_rows(CmlArgs p) {
    const int F = p.st.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t groups = (int64_t)gridDim.x * blockDim.x / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    double reg = 0.0;
    for (int64_t a = gid / lpt; a < p.B; a += groups) {
        const int32_t u = p.bu[a], i = p.bi[a], j = p.bj[a];
        const float* pu = p.st.Gu + (int64_t)u * F;
        const float* pi = p.st.Gi + (int64_t)i * F;
        const float* pj = p.st.Gi + (int64_t)j * F;
        float gu[CPL][VW], gi[CPL][VW], gj[CPL][VW];
        float dp = 0.f, dn = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
#pragma unroll
            for (int x = 0; x < VW; ++x) gu[q][x] = gi[q][x] = gj[q][x] = 0.f;
            if (e < F) {
                cml_ld<VW>(pu + e, gu[q]);
                cml_ld<VW>(pi + e, gi[q]);
                cml_ld<VW>(pj + e, gj[q]);
            }
            if (PHASE == 0) {
#pragma unroll
                for (int x = 0; x < VW; ++x) {
                    const float a1 = gu[q][x] - gi[q][x], a2 = gu[q][x] - gj[q][x];
                    dp = fmaf(a1, a1, dp);
                    dn = fmaf(a2, a2, dn);
                    sq += gu[q][x] * gu[q][x] + gi[q][x] * gi[q][x] + gj[q][x] * gj[q][x];
                }
            }
        }
        if (PHASE == 0) {
            dp = el_group_sum(dp, lpt);
            dn = el_group_sum(dn, lpt);
            sq = el_group_sum(sq, lpt);
            if (sub == 0) {
                const float bi_ = p.st.Bi[i], bj_ = p.st.Bi[j];
                p.D[a] = dn - dp;
                p.E[a] = bi_ - bj_;
                reg += (double)(0.5f * p.l_w * sq + 0.5f * p.l_b * bi_ * bi_ + (0.5f * p.l_b * bj_ * bj_) / 10.0f);
            }
        } else {
            const float c = 2.0f * p.cD[a];
            float* gpu = p.st.gGu + (int64_t)u * F;
            float* gpi = p.st.gGi + (int64_t)i * F;
            float* gpj = p.st.gGi + (int64_t)j * F;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
                if (e < F) {
#pragma unroll
                    for (int x = 0; x < VW; ++x) {
                        atomicAdd(gpu + e + x, c * (gi[q][x] - gj[q][x]) + p.l_w * gu[q][x]);
                        atomicAdd(gpi + e + x, c * (gu[q][x] - gi[q][x]) + p.l_w * gi[q][x]);
                        atomicAdd(gpj + e + x, -c * (gu[q][x] - gj[q][x]) + p.l_w * gj[q][x]);
                    }
                }
            }
            if (sub == 0) {
                const float ce = p.cE[a];
                atomicAdd(p.st.gBi + i, ce + p.l_b * p.st.Bi[i]);
                atomicAdd(p.st.gBi + j, -ce + (p.l_b / 10.0f) * p.st.Bi[j]);
            }
        }
    }
    if (PHASE == 0) block_add(reg, p.loss_out);
}

// first position with xs[pos] >= v (lower) / > v (upper), xs ascending
__device__ __forceinline__ int64_t cml_bound(const float* __restrict__ xs, int64_t n, float v, bool upper) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const float x = xs[mid];
        if (upper ? (x <= v) : (x < v))
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_cml_coef(CmlArgs p) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double l = 0.0;
    for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < p.B; a += stride) {
        const float Da = p.D[a], Ea = p.E[a];
        const int64_t lowE = cml_bound(p.Es, p.n_all, -80.0f - Da, false);
        const int64_t nA = cml_bound(p.Es, p.n_all, p.margin - Da, true) - lowE;
        const int64_t lowD = cml_bound(p.Ds, p.n_all, -80.0f - Ea, false);
        const int64_t mA = cml_bound(p.Ds, p.n_all, p.margin - Ea, true) - lowD;
        p.cD[a] = -(float)(nA > 0 ? nA : 0);
        p.cE[a] = -(float)(mA > 0 ? mA : 0);
        l += (double)(nA > 0 ? nA : 0) * (double)(p.margin - Da) - (double)(mA > 0 ? mA : 0) * (double)Ea +
             (double)lowE * (double)(p.margin + 80.0f);
    }
    block_add(l, p.loss_out);
}

// item side of the scoring identity  -|u - i|^2 + b_i = (b_i - |i|^2) + <u, 2 i> - |u|^2
__global__ __launch_bounds__(256) void k_cml_items(const float* __restrict__ Gi, const float* __restrict__ Bi, int64_t I, int F,
                                                   float* __restrict__ Gi2, float* __restrict__ Bi2) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= I) return;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float v = Gi[i * F + f];
        Gi2[i * F + f] = 2.0f * v;
        s = fmaf(v, v, s);
    }
    s = el_group_sum(s, 64);
    if (lane == 0) Bi2[i] = Bi[i] - s;
}

// val[r, c] = -sum_f (Gu[u_start + r, f] - Gi[idx[r, c], f])^2 + Bi[idx[r, c]]   (idx < 0: -inf) -- the reference's formula
// (CML_model.py:98-102) evaluated directly for the listed candidates.  One wave per (row, candidate).
__global__ __launch_bounds__(256) void k_cml_rescore(const float* __restrict__ Gu, const float* __restrict__ Gi,
                                                     const float* __restrict__ Bi, int F, const int32_t* __restrict__ idx,
                                                     int64_t n_rows, int64_t ld, int kk, int64_t u_start, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_rows * kk) return;
    const int64_t r = t / kk;
    const int c = (int)(t - r * kk);
    const int32_t it = idx[r * ld + c];
    float s = 0.f;
    if (it >= 0) {
        const float* pu = Gu + (u_start + r) * F;
        const float* pi = Gi + (int64_t)it * F;
        for (int f = lane; f < F; f += 64) {
            const float d = pu[f] - pi[f];
            s = fmaf(d, d, s);
        }
    }
    s = el_group_sum(s, 64);
    if (lane == 0) val[r * ld + c] = it >= 0 ? Bi[it] - s : -INFINITY;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct CmlWs {
    float *D, *E, *Ds, *Es, *cD, *cE;
    void* tmp;
    size_t tmp_bytes, total;
};

int carve(int64_t B, int64_t B_all, char* base, CmlWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += al256(bytes);
        return p;
    };
    float** slots[4] = {&w->D, &w->E, &w->cD, &w->cE};
    for (auto s : slots) *s = (float*)take((size_t)B * 4);
    w->Ds = (float*)take((size_t)B_all * 4);
    w->Es = (float*)take((size_t)B_all * 4);
    size_t t = 0;
    float* np = nullptr;
    if (rocprim::radix_sort_keys(nullptr, t, np, np, (unsigned)B_all, 0, 32, (hipStream_t)0) != hipSuccess) return 1;
    w->tmp_bytes = t;
    w->tmp = take(t);
    w->total = off;
    return 0;
}

bool vec_ok(const el_bprmf_state& st) {
    auto ok = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return st.F % 4 == 0 && ok(st.Gu) && ok(st.Gi) && ok(st.gGu) && ok(st.gGi);
}

template <int VW, int PHASE>
int launch_rows(CmlArgs p, el_ctx* ctx, hipStream_t s) {
    int cpl = 1;
    p.lpt = el_pick_lpt(p.st.F, VW, &cpl);
    EL_REQUIRE(cpl <= 4, "el_cml_train_step: F=%d too large for this build (max %d)", p.st.F, 64 * 4 * VW);
    const int64_t want = (p.B * p.lpt + 255) / 256, cap = (int64_t)ctx->cus * 16;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    const char* nm = PHASE == 0 ? "k_cml_fwd" : "k_cml_bwd";
    if (cpl == 1) EL_LAUNCH(nm, (k_cml_rows<VW, 1, PHASE>), dim3(grid), dim3(256), 0, s, p);
    else if (cpl == 2) EL_LAUNCH(nm, (k_cml_rows<VW, 2, PHASE>), dim3(grid), dim3(256), 0, s, p);
    else EL_LAUNCH(nm, (k_cml_rows<VW, 4, PHASE>), dim3(grid), dim3(256), 0, s, p);
    EL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

static int cml_check(const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, const char* who) {
    EL_REQUIRE(stp && stp->Gu && stp->Gi && stp->Bi && stp->gGu && stp->gGi && stp->gBi, "%s: null state", who);
    EL_REQUIRE(stp->U > 0 && stp->I > 0 && stp->F > 0, "%s: bad shape", who);
    EL_REQUIRE(u && i && j, "%s: null triplet arrays", who);
    EL_REQUIRE(B < (1LL << 31), "%s: batch too large", who);
    return 0;
}

static CmlArgs cml_args(const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B, float l_w, float l_b,
                        float margin, double* loss_out) {
    CmlArgs p;
    memset(&p, 0, sizeof(p));
    p.st = *stp;
    p.st.tGu = p.st.tGi = p.st.tBi = nullptr;
    p.bu = u, p.bi = i, p.bj = j, p.B = B, p.l_w = l_w, p.l_b = l_b, p.margin = margin, p.loss_out = loss_out;
    return p;
}

extern "C" size_t el_cml_ws_bytes(int64_t B, int64_t B_all, int64_t U, int64_t I, int32_t F) {
    if (B <= 0) return 0;
    CmlWs w;
    if (carve(B, B_all < B ? B : B_all, nullptr, &w)) return 0;
    return w.total + el_bprmf_ws_bytes(B, U, I, F);             // + the sort / segment scratch of the gradient pass
}

// phase 1: D_a, E_a of the rank's triplets (+ the regulariser into loss_out)
extern "C" int el_cml_forward(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                              const int32_t* j, int64_t B, float l_w, float l_b, float* D, float* E, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    if (B <= 0) return 0;
    if (int rc = cml_check(stp, u, i, j, B, "el_cml_forward")) return rc;
    EL_REQUIRE(D && E && loss_out, "el_cml_forward: null output");
    CmlArgs p = cml_args(stp, u, i, j, B, l_w, l_b, 0.f, loss_out);
    p.D = D, p.E = E;
    return vec_ok(p.st) ? launch_rows<4, 0>(p, ctx, (hipStream_t)stream) : launch_rows<1, 0>(p, ctx, (hipStream_t)stream);
}

// phase 2: coefficients of the rank's triplets against the GLOBAL batch's D / E, hinge part of the loss, row gradients
static int cml_grads(el_ctx* ctx, hipStream_t s, const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j,
                     int64_t B, float l_w, float l_b, float margin, const float* D, const float* E, const float* D_all,
                     const float* E_all, int64_t B_all, double* loss_out, char* ws, size_t ws_bytes, const CmlWs& w) {
    CmlArgs p = cml_args(stp, u, i, j, B, l_w, l_b, margin, loss_out);
    p.D = const_cast<float*>(D), p.E = const_cast<float*>(E), p.Ds = w.Ds, p.Es = w.Es, p.cD = w.cD, p.cE = w.cE, p.n_all = B_all;
    {
        ElKernelTimer t("rocprim_radix_sort_keys", s);
        size_t tb = w.tmp_bytes;
        EL_CHECK_HIP(rocprim::radix_sort_keys(w.tmp, tb, D_all, w.Ds, (unsigned)B_all, 0, 32, s));
        tb = w.tmp_bytes;
        EL_CHECK_HIP(rocprim::radix_sort_keys(w.tmp, tb, E_all, w.Es, (unsigned)B_all, 0, 32, s));
    }
    {
        const int64_t want = (B + 255) / 256, cap = (int64_t)ctx->cus * 16;
        EL_LAUNCH("k_cml_coef", k_cml_coef, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, s, p);
        EL_CHECK_LAUNCH();
    }
    // row gradients: batches worth sorting take the segment kernels of the BPR path (no float atomics on hot item rows:
    // 7.3 -> ~0.7 ms at B = 1M under Zipf popularity), small ones the atomic kernel
    if (B >= 2048) return el_bpr_sorted_cml_grads(ctx, s, p.st, u, i, j, B, l_w, l_b, w.cD, w.cE, ws + w.total, ws_bytes - w.total);
    return vec_ok(p.st) ? launch_rows<4, 1>(p, ctx, s) : launch_rows<1, 1>(p, ctx, s);
}

extern "C" int el_cml_grads(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                            const int32_t* j, int64_t B, float l_w, float l_b, float margin, const float* D, const float* E,
                            const float* D_all, const float* E_all, int64_t B_all, double* loss_out, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    if (B <= 0) return 0;
    if (int rc = cml_check(stp, u, i, j, B, "el_cml_grads")) return rc;
    EL_REQUIRE(D && E && D_all && E_all && B_all >= B && B_all < (1LL << 31) && loss_out, "el_cml_grads: bad arguments");
    CmlWs w;
    EL_REQUIRE(carve(B, B_all, (char*)ws, &w) == 0, "el_cml_grads: rocprim size query failed");
    const size_t need = w.total + (B >= 2048 ? el_bprmf_ws_bytes(B, stp->U, stp->I, stp->F) : 0);
    EL_REQUIRE(ws != nullptr && ws_bytes >= need, "el_cml_grads: workspace too small (%zu < %zu)", ws_bytes, need);
    return cml_grads(ctx, (hipStream_t)stream, stp, u, i, j, B, l_w, l_b, margin, D, E, D_all, E_all, B_all, loss_out, (char*)ws, ws_bytes, w);
}

extern "C" int el_cml_train_step(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                                 const int32_t* j, int64_t B, float l_w, float l_b, float margin, int32_t step, float lr_t,
                                 double* loss_out, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    if (B <= 0) return 0;
    if (int rc = cml_check(stp, u, i, j, B, "el_cml_train_step")) return rc;
    EL_REQUIRE(stp->mGu && stp->vGu && stp->mGi && stp->vGi && stp->mBi && stp->vBi, "el_cml_train_step: Adam slots missing");
    EL_REQUIRE(loss_out && step >= 1, "el_cml_train_step: bad loss pointer / step");
    CmlWs w;
    EL_REQUIRE(carve(B, B, (char*)ws, &w) == 0, "el_cml_train_step: rocprim size query failed");
    const size_t need = w.total + (B >= 2048 ? el_bprmf_ws_bytes(B, stp->U, stp->I, stp->F) : 0);
    EL_REQUIRE(ws != nullptr && ws_bytes >= need, "el_cml_train_step: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    CmlArgs p = cml_args(stp, u, i, j, B, l_w, l_b, margin, loss_out);
    p.D = w.D, p.E = w.E;
    if (int rc = vec_ok(p.st) ? launch_rows<4, 0>(p, ctx, s) : launch_rows<1, 0>(p, ctx, s)) return rc;
    if (int rc = cml_grads(ctx, s, stp, u, i, j, B, l_w, l_b, margin, w.D, w.E, w.D, w.E, B, loss_out, (char*)ws, ws_bytes, w)) return rc;
    return el_bprmf_apply_optimizer(ctx, s, p.st, nullptr, nullptr, nullptr, 0, 0.f, EL_OPT_ADAM_TF_DENSE, step, lr_t);
}

extern "C" int el_cml_prepare_items(el_ctx* ctx, void* stream, const float* Gi, const float* Bi, int64_t I, int32_t F,
                                    float* Gi2, float* Bi2) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(Gi && Bi && Gi2 && Bi2 && I > 0 && F > 0, "el_cml_prepare_items: bad arguments");
    EL_LAUNCH("k_cml_items", k_cml_items, dim3((unsigned)((I + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Gi, Bi, I, (int)F, Gi2, Bi2);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_cml_rescore(el_ctx* ctx, void* stream, const float* Gu, const float* Gi, const float* Bi, int32_t F,
                              const int32_t* idx, int64_t n_rows, int64_t ld, int32_t kk, int64_t u_start, float* val) {
    if (int rc = el_bind(ctx)) return rc;
    if (n_rows <= 0 || kk <= 0) return 0;
    EL_REQUIRE(Gu && Gi && Bi && idx && val && F > 0 && ld >= kk, "el_cml_rescore: bad arguments");
    const int64_t waves = n_rows * kk;
    EL_LAUNCH("k_cml_rescore", k_cml_rescore, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Gu, Gi, Bi, (int)F,
              idx, n_rows, ld, (int)kk, u_start, val);
    EL_CHECK_LAUNCH();
    return 0;
}
