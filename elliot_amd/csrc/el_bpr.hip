// BPR triplet sampler + BPR-MF training steps.
//
//   el_bpr_sample        <- dataset/samplers/custom_sampler.py:31-46
//   el_bprmf_train_step  <- BPRMF_batch_model.train_step (BPRMF_batch_model.py:58-80)
//   el_bprsgd_apply      <- MFModel.update_factors       (BPRMF_model.py:91-117)
//
// All three are HBM-bound gather/scatter kernels: a group of LPT lanes (LPT = 8..64, a
// power of two) owns one triplet, each lane moves 16 bytes of every row it touches, dot
// products are reduced with wavefront shuffles, nothing is staged through LDS because a
// row is consumed exactly once per triplet.
#include <stdlib.h>
#include <vector>
#include "el_common.h"

// ---------------------------------------------------------------------------------------
// K1: Philox sampler
// ---------------------------------------------------------------------------------------
struct PhiloxStream {
    u32 n_lo, n_hi, k0, k1, a;
    u32 w[4];
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
            el_philox4 r = el_philox4x32_10(n_lo, n_hi, a, 0u, k0, k1);
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
    // uniform integer in [0, n), n >= 1: masked rejection (np.random.randint's scheme)
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

// Optional per-user record of the sampler (el_bpr_sampler_meta_build): everything a draw needs about its user in ONE 64-byte
// line -- row start, row length and a 384-bit membership signature of the row.  A negative candidate whose signature bit is
// clear is certainly not a positive (the usual case: rows fill ~18 % of the bits), so the binary search over the row -- three
// more cache lines per triplet -- only runs on a signature hit.  Accept / reject decisions are those of the plain path.
__global__ __launch_bounds__(256) void k_bpr_sampler_meta(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          int64_t U, SamplerRec* __restrict__ meta) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    SamplerRec r;
    r.r0 = indptr[u];
    const int64_t r1 = indptr[u + 1];
    r.len = (int32_t)(r1 - r.r0);
    r.pad = 0u;
#pragma unroll
    for (int w = 0; w < 12; ++w) r.sig[w] = 0u;
    for (int64_t e = r.r0; e < r1; ++e) {
        const u32 b = el_sig_bit(indices[e]);
        r.sig[b >> 5] |= 1u << (b & 31u);
    }
    meta[u] = r;
}

template <bool META>
__global__ __launch_bounds__(256) void k_bpr_sample(const int64_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices, int64_t U, int64_t I,
                                                    int64_t item_lo, int64_t item_hi, u64 seed, u64 first, int64_t n,
                                                    int32_t* out_u, int32_t* out_i, int32_t* out_j,
                                                    const SamplerRec* __restrict__ meta) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    PhiloxStream ps;
    ps.init(first + (u64)t, seed);
    const u32 range = (u32)(item_hi - item_lo);
    for (;;) {
        u32 u = ps.bounded((u32)U);
        int64_t r0, r1;
        SamplerUser su;
        if (META) {
            su.load(meta, u);
            r0 = su.r0, r1 = su.r1;
        } else {
            r0 = indptr[u];
            r1 = indptr[u + 1];
        }
        int64_t lui = r1 - r0;
        if (lui <= 0 || lui >= I) continue;  // no positive / no negative available: re-draw the user
        u32 ipos = ps.bounded((u32)lui);
        int32_t it = indices[r0 + ipos];
        int32_t jt = -1;
        for (int attempt = 0; attempt < 4096; ++attempt) {
            int32_t cand = (int32_t)(item_lo + (int64_t)ps.bounded(range));
            const bool maybe = META ? su.maybe(cand) : true;
            if (!maybe || !el_row_contains(indices, r0, r1, cand)) {
                jt = cand;
                break;
            }
        }
        if (jt < 0) continue;  // (sharded range fully positive for this user)
        out_u[t] = (int32_t)u;
        out_i[t] = it;
        out_j[t] = jt;
        return;
    }
}

extern "C" int el_bpr_sample(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                             int64_t U, int64_t I, int64_t item_lo, int64_t item_hi, uint64_t seed,
                             uint64_t first_sample, int64_t n, int32_t* out_u, int32_t* out_i, int32_t* out_j) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(pos_indptr && pos_indices && out_u && out_i && out_j, "el_bpr_sample: null pointer");
    EL_REQUIRE(U >= 1 && U < 0xffffffffLL && I >= 2 && I < 0x7fffffffLL, "el_bpr_sample: U/I out of range");
    EL_REQUIRE(item_lo >= 0 && item_hi <= I && item_hi > item_lo, "el_bpr_sample: bad negative range");
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + 255) / 256);
    EL_LAUNCH("k_bpr_sample", k_bpr_sample<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pos_indptr, pos_indices, U, I,
                       item_lo, item_hi, (u64)seed, (u64)first_sample, n, out_u, out_i, out_j, (const SamplerRec*)nullptr);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t el_bpr_sampler_meta_bytes(int64_t U) { return U > 0 ? (size_t)U * sizeof(SamplerRec) : 0; }

extern "C" int el_bpr_sampler_meta_build(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                                         int64_t U, void* meta) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(pos_indptr && pos_indices && meta && U >= 1, "el_bpr_sampler_meta_build: bad arguments");
    EL_REQUIRE(((uintptr_t)meta & 63) == 0, "el_bpr_sampler_meta_build: meta must be 64-byte aligned");
    EL_LAUNCH("k_bpr_sampler_meta", k_bpr_sampler_meta, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
              pos_indptr, pos_indices, U, (SamplerRec*)meta);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_bpr_sample_meta(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices, const void* meta,
                                  int64_t U, int64_t I, int64_t item_lo, int64_t item_hi, uint64_t seed, uint64_t first_sample,
                                  int64_t n, int32_t* out_u, int32_t* out_i, int32_t* out_j) {
    if (meta == nullptr) return el_bpr_sample(ctx, stream, pos_indptr, pos_indices, U, I, item_lo, item_hi, seed, first_sample, n, out_u, out_i, out_j);
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(pos_indptr && pos_indices && out_u && out_i && out_j, "el_bpr_sample_meta: null pointer");
    EL_REQUIRE(U >= 1 && U < 0xffffffffLL && I >= 2 && I < 0x7fffffffLL, "el_bpr_sample_meta: U/I out of range");
    EL_REQUIRE(item_lo >= 0 && item_hi <= I && item_hi > item_lo, "el_bpr_sample_meta: bad negative range");
    EL_REQUIRE(((uintptr_t)meta & 63) == 0, "el_bpr_sample_meta: meta must be 64-byte aligned");
    if (n <= 0) return 0;
    EL_LAUNCH("k_bpr_sample", k_bpr_sample<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos_indptr,
              pos_indices, U, I, item_lo, item_hi, (u64)seed, (u64)first_sample, n, out_u, out_i, out_j, (const SamplerRec*)meta);
    EL_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// K2-K4: BPRMF_batch train step
// ---------------------------------------------------------------------------------------
template <int VW>
__device__ __forceinline__ void ld_vec(const float* p, float* dst) {
    if (VW == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        dst[0] = t.x;
        dst[1] = t.y;
        dst[2] = t.z;
        dst[3] = t.w;
    } else {
        dst[0] = p[0];
    }
}

__device__ __forceinline__ float el_softplus(float x) {
    // tf.nn.softplus: x for large x, exp(x) for very negative x, log1p(exp(x)) otherwise
    if (x > 15.0f) return x;
    if (x < -15.0f) return expf(x);
    return log1pf(expf(x));
}

// forward + backward of one batch; gradients are scatter-ADDED into the dense accumulators
// gGu/gGi/gBi (duplicates sum, as OptimizerV2 de-duplicates IndexedSlices by segment-sum).
template <int VW, int CPL>
__device__ __forceinline__ void bprmf_fwd_bwd_body(const el_bprmf_state& st, const int32_t* __restrict__ bu,
                                                   const int32_t* __restrict__ bi_, const int32_t* __restrict__ bj,
                                                   int64_t B, float l_w, float l_b, int lpt, int32_t step,
                                                   double* loss_out) {
    const int F = st.F;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    const bool active = b < B;
    int32_t uu = 0, ii = 0, jj = 0;
    if (active) {
        uu = bu[b];
        ii = bi_[b];
        jj = bj[b];
    }
    const float* pu = st.Gu + (int64_t)uu * F;
    const float* pi = st.Gi + (int64_t)ii * F;
    const float* pj = st.Gi + (int64_t)jj * F;
    float gu[CPL][VW], gi[CPL][VW], gj[CPL][VW];
    float dpi = 0.f, dpj = 0.f, nu = 0.f, ni = 0.f, nj = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
        const bool ok = active && e < F;
#pragma unroll
        for (int x = 0; x < VW; ++x) gu[q][x] = gi[q][x] = gj[q][x] = 0.f;
        if (ok) {
            ld_vec<VW>(pu + e, gu[q]);
            ld_vec<VW>(pi + e, gi[q]);
            ld_vec<VW>(pj + e, gj[q]);
        }
#pragma unroll
        for (int x = 0; x < VW; ++x) {
            dpi += gu[q][x] * gi[q][x];
            dpj += gu[q][x] * gj[q][x];
            nu += gu[q][x] * gu[q][x];
            ni += gi[q][x] * gi[q][x];
            nj += gj[q][x] * gj[q][x];
        }
    }
    dpi = el_group_sum(dpi, lpt);
    dpj = el_group_sum(dpj, lpt);
    nu = el_group_sum(nu, lpt);
    ni = el_group_sum(ni, lpt);
    nj = el_group_sum(nj, lpt);
    float beta_i = 0.f, beta_j = 0.f;
    if (active) {
        beta_i = st.Bi[ii];
        beta_j = st.Bi[jj];
    }
    const float xui = beta_i + dpi, xuj = beta_j + dpj;
    const float d = xui - xuj;
    const float dc = fminf(fmaxf(d, -80.0f), 1e8f);
    float s = 0.f;  // d loss / d difference
    if (d >= -80.0f) s = -1.0f / (1.0f + expf(d));
    float myloss = 0.f;
    if (active && sub == 0) {
        myloss = el_softplus(-dc) + l_w * 0.5f * (nu + ni + nj) + l_b * 0.5f * beta_i * beta_i +
                 (l_b * 0.5f * beta_j * beta_j) / 10.0f;
    }
    if (active) {
        float* gpu = st.gGu + (int64_t)uu * F;
        float* gpi = st.gGi + (int64_t)ii * F;
        float* gpj = st.gGi + (int64_t)jj * F;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
#pragma unroll
                for (int x = 0; x < VW; ++x) {
                    atomicAdd(gpu + e + x, s * (gi[q][x] - gj[q][x]) + l_w * gu[q][x]);
                    atomicAdd(gpi + e + x, s * gu[q][x] + l_w * gi[q][x]);
                    atomicAdd(gpj + e + x, -s * gu[q][x] + l_w * gj[q][x]);
                }
            }
        }
        if (sub == 0) {
            atomicAdd(st.gBi + ii, s + l_b * beta_i);
            atomicAdd(st.gBi + jj, -s + (l_b / 10.0f) * beta_j);
            if (st.tGu) {
                st.tGu[uu] = step;
                st.tGi[ii] = step;
                st.tGi[jj] = step;
                st.tBi[ii] = step;
                st.tBi[jj] = step;
            }
        }
    }
    // block loss -> one double atomic
    __shared__ float wsum[4];
    float wl = el_group_sum(myloss, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = wl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0) atomicAdd(loss_out, tot);
    }
}

template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_bprmf_fwd_bwd(el_bprmf_state st, const int32_t* __restrict__ bu,
                                                       const int32_t* __restrict__ bi_, const int32_t* __restrict__ bj,
                                                       int64_t B, float l_w, float l_b, int lpt, int32_t step,
                                                       double* loss_out) {
    bprmf_fwd_bwd_body<VW, CPL>(st, bu, bi_, bj, B, l_w, l_b, lpt, step, loss_out);
}

// ---- graph-replayable forms (el_bprmf_train_loop) -----------------------------------------------------------------
// A captured launch cannot take this step's batch offset or Adam step size by value, so both live in device memory:
// ctl[0] is read by the gradient kernel (and copied to ctl[1]), ctl[1] by the optimiser kernel, whose first thread then
// writes the NEXT step into ctl[0] -- no kernel ever reads a word that a concurrently running block writes.
struct LoopCtl {
    int64_t off;   // first triplet of the step inside the sampled chunk
    int64_t cn;    // triplets in the chunk
    int32_t k;     // index into lr_tab
    int32_t pad;
};

template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_bprmf_fwd_bwd_g(el_bprmf_state st, const int32_t* __restrict__ bu,
                                                         const int32_t* __restrict__ bi_, const int32_t* __restrict__ bj,
                                                         int64_t B, float l_w, float l_b, int lpt, double* loss_out,
                                                         LoopCtl* ctl) {
    const LoopCtl c = ctl[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl[1] = c;
    const int64_t left = c.cn - c.off;
    if (left <= 0) return;                                     // padding step of the last replay
    const int64_t n = left < B ? left : B;
    bprmf_fwd_bwd_body<VW, CPL>(st, bu + c.off, bi_ + c.off, bj + c.off, n, l_w, l_b, lpt, 0, loss_out);
}

__global__ void k_loop_ctl_set(LoopCtl* ctl, int64_t cn, int32_t k) {
    ctl[0].off = 0;
    ctl[0].cn = cn;
    ctl[0].k = k;
    ctl[1] = ctl[0];
}

// (el_adam_elem -- Keras-2.3 Adam, sparse-apply arithmetic, one element -- lives in el_common.h)

// dense pass: EVERY element of the variable decays and moves (TF sparse-apply semantics);
// the gradient accumulator is reset on the way.  Pure streaming: 4 read + 3(4) write streams.
// NT: non-temporal loads/stores (nothing here is re-read before the next step); UNR: float4 groups per iteration.
template <bool NT, int UNR>
__device__ __forceinline__ void adam_dense_body(float* __restrict__ th, float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                float eps) {
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* th4 = reinterpret_cast<f4*>(th);
    f4* g4 = reinterpret_cast<f4*>(g);
    f4* m4 = reinterpret_cast<f4*>(m);
    f4* v4 = reinterpret_cast<f4*>(v);
    for (int64_t e0 = t; e0 < n4; e0 += stride * UNR) {
        f4 a[UNR], gg[UNR], mm[UNR], vv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t e = e0 + q * stride;
            if (e < n4) {
                if (NT) {
                    a[q] = __builtin_nontemporal_load(th4 + e);
                    gg[q] = __builtin_nontemporal_load(g4 + e);
                    mm[q] = __builtin_nontemporal_load(m4 + e);
                    vv[q] = __builtin_nontemporal_load(v4 + e);
                } else {
                    a[q] = th4[e];
                    gg[q] = g4[e];
                    mm[q] = m4[e];
                    vv[q] = v4[e];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t e = e0 + q * stride;
            if (e < n4) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float ax = a[q][x], mx = mm[q][x], vx = vv[q][x];
                    el_adam_elem(ax, mx, vx, gg[q][x], lr_t, b1, b2, omb1, omb2, eps);
                    a[q][x] = ax;
                    mm[q][x] = mx;
                    vv[q][x] = vx;
                }
                const bool nz = gg[q][0] != 0.f || gg[q][1] != 0.f || gg[q][2] != 0.f || gg[q][3] != 0.f;
                const f4 zero = {0.f, 0.f, 0.f, 0.f};
                if (NT) {
                    __builtin_nontemporal_store(a[q], th4 + e);
                    __builtin_nontemporal_store(mm[q], m4 + e);
                    __builtin_nontemporal_store(vv[q], v4 + e);
                    if (nz) __builtin_nontemporal_store(zero, g4 + e);
                } else {
                    th4[e] = a[q];
                    m4[e] = mm[q];
                    v4[e] = vv[q];
                    if (nz) g4[e] = zero;
                }
            }
        }
    }
    for (int64_t e = (n4 << 2) + t; e < n; e += stride) {
        float a = th[e], gg = g[e], mm = m[e], vv = v[e];
        el_adam_elem(a, mm, vv, gg, lr_t, b1, b2, omb1, omb2, eps);
        th[e] = a;
        m[e] = mm;
        v[e] = vv;
        if (gg != 0.f) g[e] = 0.f;
    }
}

// Measured on the 128M-element user table (MI355X): plain 0.82 ms, non-temporal 0.76 ms, non-temporal + 2 groups in
// flight 0.75 ms per pass (28-32 B/element of HBM traffic -> ~5.2 TB/s).
__global__ __launch_bounds__(256) void k_adam_dense(float* __restrict__ th, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    float lr_t, float b1, float b2, float eps) {
    adam_dense_body<true, 2>(th, g, m, v, n, lr_t, b1, b2, eps);
}
// two tensors in one launch (item factors + item bias of the fused BPR step)
__global__ __launch_bounds__(256) void k_adam_dense_pair(float* __restrict__ th, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n, float* __restrict__ th2, float* __restrict__ g2,
                                                         float* __restrict__ m2, float* __restrict__ v2, int64_t n2, float lr_t, float b1,
                                                         float b2, float eps) {
    adam_dense_body<true, 2>(th, g, m, v, n, lr_t, b1, b2, eps);
    adam_dense_body<true, 2>(th2, g2, m2, v2, n2, lr_t, b1, b2, eps);
}
// the same code under another symbol (el_tuning_mode: probes of the placement tuner stay apart in kernel traces)
__global__ __launch_bounds__(256) void k_adam_dense_tune(float* __restrict__ th, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         float lr_t, float b1, float b2, float eps) {
    adam_dense_body<true, 2>(th, g, m, v, n, lr_t, b1, b2, eps);
}

// The same pass reading COMPACT gradient rows (el_bprmf_state.uslot / gGu_rows): every element still decays m, v and moves
// theta (Keras sparse apply), but a gradient is fetched only for rows stamped with this step and nothing is zeroed afterwards:
// 12 + 12 bytes per parameter + 4 per parameter of a touched row, against 16 + 12 (+ 4 re-zeroed) of the dense form.
// n4 = U F / 4 float4 elements; row of element e = e / F4 (a shift when F4 is a power of two).
template <int UNR, bool TUNE>
__global__ __launch_bounds__(256) void k_adam_rows(float* __restrict__ th, const float* __restrict__ grows,
                                                   const int64_t* __restrict__ uslot, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int F4, int f4_shift, int32_t step,
                                                   float lr_t, float b1, float b2, float eps) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4* th4 = reinterpret_cast<f4*>(th);
    const f4* g4 = reinterpret_cast<const f4*>(grows);
    f4* m4 = reinterpret_cast<f4*>(m);
    f4* v4 = reinterpret_cast<f4*>(v);
    const bool small = n4 < (1LL << 31);
    for (int64_t e0 = t; e0 < n4; e0 += stride * UNR) {
        f4 a[UNR], gg[UNR], mm[UNR], vv[UNR];
        int64_t gsrc[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t e = e0 + q * stride;
            gsrc[q] = -1;
            if (e < n4) {
                const int64_t row = f4_shift >= 0 ? (e >> f4_shift) : (small ? (int64_t)((u32)e / (u32)F4) : e / F4);
                const int64_t ent = uslot[row];
                if ((int32_t)(ent >> 32) == step) gsrc[q] = (ent & 0xffffffffLL) * F4 + (e - row * F4);
                a[q] = __builtin_nontemporal_load(th4 + e);
                mm[q] = __builtin_nontemporal_load(m4 + e);
                vv[q] = __builtin_nontemporal_load(v4 + e);
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            gg[q] = gsrc[q] >= 0 ? __builtin_nontemporal_load(g4 + gsrc[q]) : zero;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int64_t e = e0 + q * stride;
            if (e < n4) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float ax = a[q][x], mx = mm[q][x], vx = vv[q][x];
                    el_adam_elem(ax, mx, vx, gg[q][x], lr_t, b1, b2, omb1, omb2, eps);
                    a[q][x] = ax;
                    mm[q][x] = mx;
                    vv[q][x] = vx;
                }
                __builtin_nontemporal_store(a[q], th4 + e);
                __builtin_nontemporal_store(mm[q], m4 + e);
                __builtin_nontemporal_store(vv[q], v4 + e);
            }
        }
    }
}

// Small models (ML-1M: 0.6 M parameters): the three dense passes are launch-bound, so they share one launch.
struct AdamTriple {
    float* th[3];
    float* g[3];
    float* m[3];
    float* v[3];
    int64_t n[3];
};
__global__ __launch_bounds__(256) void k_adam_dense3(AdamTriple t, float lr_t, float b1, float b2, float eps) {
#pragma unroll
    for (int k = 0; k < 3; ++k) adam_dense_body<false, 1>(t.th[k], t.g[k], t.m[k], t.v[k], t.n[k], lr_t, b1, b2, eps);
}

__global__ __launch_bounds__(256) void k_adam_dense3_g(AdamTriple t, const float* __restrict__ lr_tab, float b1, float b2,
                                                       float eps, LoopCtl* ctl, int64_t B) {
    const LoopCtl c = ctl[1];
    if (c.off >= c.cn) return;                                 // padding step: no batch, no optimiser iteration
    const float lr_t = lr_tab[c.k];
#pragma unroll
    for (int k = 0; k < 3; ++k) adam_dense_body<false, 1>(t.th[k], t.g[k], t.m[k], t.v[k], t.n[k], lr_t, b1, b2, eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[0].off = c.off + B;
        ctl[0].k = c.k + 1;
    }
}

// touched-row pass (EL_OPT_ADAM_LAZY / EL_OPT_SGD): one lane group per batch entry and
// role; the first group to claim a row (stamp step -> -step) applies the update once.
// role 0: Gu[u], 1: Gi[i] (+Bi[i]), 2: Gi[j] (+Bi[j])
template <int VW, int CPL, bool ADAM>
__global__ __launch_bounds__(256) void k_rows_apply(el_bprmf_state st, const int32_t* __restrict__ bu,
                                                    const int32_t* __restrict__ bi_, const int32_t* __restrict__ bj,
                                                    int64_t B, int lpt, int32_t step, float lr, float lr_t, float b1,
                                                    float b2, float eps) {
    const int F = st.F;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    const int64_t b = grp / 3;
    const int role = (int)(grp - b * 3);
    const bool active = b < B;
    int32_t row = 0;
    if (active) row = (role == 0) ? bu[b] : (role == 1 ? bi_[b] : bj[b]);
    int32_t* stamp = (role == 0) ? st.tGu : st.tGi;
    int claimed = 0;
    if (active && sub == 0) claimed = (atomicCAS(stamp + row, step, -step) == step) ? 1 : 0;
    const int lane = threadIdx.x & 63;
    claimed = __shfl(claimed, lane & ~(lpt - 1), 64);
    if (!claimed) return;
    float* th = ((role == 0) ? st.Gu : st.Gi) + (int64_t)row * F;
    float* g = ((role == 0) ? st.gGu : st.gGi) + (int64_t)row * F;
    float* m = ADAM ? ((role == 0) ? st.mGu : st.mGi) + (int64_t)row * F : nullptr;
    float* v = ADAM ? ((role == 0) ? st.vGu : st.vGi) + (int64_t)row * F : nullptr;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
        if (e < F) {
            float a[VW], gg[VW], mm[VW], vv[VW];
            ld_vec<VW>(th + e, a);
            ld_vec<VW>(g + e, gg);
            if (ADAM) {
                ld_vec<VW>(m + e, mm);
                ld_vec<VW>(v + e, vv);
            }
#pragma unroll
            for (int x = 0; x < VW; ++x) {
                if (ADAM)
                    el_adam_elem(a[x], mm[x], vv[x], gg[x], lr_t, b1, b2, omb1, omb2, eps);
                else
                    a[x] = a[x] - lr * gg[x];
                th[e + x] = a[x];
                g[e + x] = 0.f;
                if (ADAM) {
                    m[e + x] = mm[x];
                    v[e + x] = vv[x];
                }
            }
        }
    }
    if (role != 0 && sub == 0) {
        // item bias shares the item row's claim only if its own stamp is still pending
        if (atomicCAS(st.tBi + row, step, -step) == step) {
            float a = st.Bi[row], gg = st.gBi[row];
            if (ADAM) {
                float mm = st.mBi[row], vv = st.vBi[row];
                el_adam_elem(a, mm, vv, gg, lr_t, b1, b2, omb1, omb2, eps);
                st.mBi[row] = mm;
                st.vBi[row] = vv;
            } else {
                a = a - lr * gg;
            }
            st.Bi[row] = a;
            st.gBi[row] = 0.f;
        }
    }
}

// plain dense SGD (EL_OPT_SGD without stamp arrays)
__global__ __launch_bounds__(256) void k_sgd_dense(float* __restrict__ th, float* __restrict__ g, int64_t n,
                                                   float lr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        float gg = g[e];
        if (gg != 0.f) {
            th[e] = th[e] - lr * gg;
            g[e] = 0.f;
        }
    }
}

int el_pick_lpt(int F, int vw, int* cpl) {
    int groups = (F + vw - 1) / vw;  // vector chunks per row
    int lpt = 8;
    while (lpt < groups && lpt < 64) lpt <<= 1;
    int c = (groups + lpt - 1) / lpt;
    int cp = 1;
    while (cp < c) cp <<= 1;
    *cpl = cp;
    return lpt;
}

static bool rows_aligned16(const void* p, int F) { return (F % 4 == 0) && (((uintptr_t)p) % 16 == 0); }

template <int VW>
static int launch_fwd_bwd(const el_bprmf_state& st, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                          float l_w, float l_b, int32_t step, double* loss_out, hipStream_t s) {
    int cpl = 1;
    int lpt = el_pick_lpt(st.F, VW, &cpl);
    EL_REQUIRE(cpl <= 4, "el_bprmf_train_step: F=%d too large for this build (max %d)", st.F, 64 * 4 * VW);
    int64_t threads = B * lpt;
    unsigned grid = (unsigned)((threads + 255) / 256);
    if (cpl == 1)
        EL_LAUNCH("k_bprmf_fwd_bwd", (k_bprmf_fwd_bwd<VW, 1>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, l_w, l_b, lpt, step, loss_out);
    else if (cpl == 2)
        EL_LAUNCH("k_bprmf_fwd_bwd", (k_bprmf_fwd_bwd<VW, 2>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, l_w, l_b, lpt, step, loss_out);
    else
        EL_LAUNCH("k_bprmf_fwd_bwd", (k_bprmf_fwd_bwd<VW, 4>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, l_w, l_b, lpt, step, loss_out);
    EL_CHECK_LAUNCH();
    return 0;
}

template <int VW, bool ADAM>
static int launch_rows_apply(const el_bprmf_state& st, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                             int32_t step, float lr, float lr_t, hipStream_t s) {
    int cpl = 1;
    int lpt = el_pick_lpt(st.F, VW, &cpl);
    EL_REQUIRE(cpl <= 4, "el_bprmf_train_step: F=%d too large for this build", st.F);
    int64_t threads = B * 3 * lpt;
    unsigned grid = (unsigned)((threads + 255) / 256);
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    if (cpl == 1)
        EL_LAUNCH("k_rows_apply", (k_rows_apply<VW, 1, ADAM>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, lpt, step, lr, lr_t, b1, b2, eps);
    else if (cpl == 2)
        EL_LAUNCH("k_rows_apply", (k_rows_apply<VW, 2, ADAM>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, lpt, step, lr, lr_t, b1, b2, eps);
    else
        EL_LAUNCH("k_rows_apply", (k_rows_apply<VW, 4, ADAM>), dim3(grid), dim3(256), 0, s, st, u, i, j, B, lpt, step, lr, lr_t, b1, b2, eps);
    EL_CHECK_LAUNCH();
    return 0;
}

static unsigned stream_grid(el_ctx* ctx, int64_t n_threads) {
    int64_t blocks = (n_threads + 255) / 256;
    const int mult = 8;                                     // workgroups per CU of the streaming passes
    int64_t cap = (int64_t)ctx->cus * mult;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// TF-dense Adam on the item side alone (Gi, Bi): what is left of the optimiser phase after the fused user-side kernel
int el_bprmf_apply_items_adam(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, float lr_t) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    const int64_t ni = st.I * (int64_t)st.F;
    // the factors and the bias vector in ONE launch (the bias pass is 8 us of work behind a launch gap of its own)
    EL_LAUNCH("k_adam_dense_Gi", k_adam_dense_pair, dim3(stream_grid(ctx, ni / 4 + 1)), dim3(256), 0, s, st.Gi, st.gGi, st.mGi, st.vGi, ni, st.Bi,
              st.gBi, st.mBi, st.vBi, st.I, lr_t, b1, b2, eps);
    EL_CHECK_LAUNCH();
    return 0;
}

// shared argument validation of both train-step paths
int el_bprmf_check_state(const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j,
                         double* loss_out, int opt, int32_t step, bool* vec, bool* rows_mode) {
    EL_REQUIRE(stp != nullptr, "el_bprmf_train_step: null state");
    const el_bprmf_state& st = *stp;
    EL_REQUIRE(st.Gu && st.Gi && st.Bi && (st.gGu || st.uslot) && st.gGi && st.gBi, "el_bprmf_train_step: null table/accumulator");
    EL_REQUIRE(st.F >= 1 && st.U >= 1 && st.I >= 1, "el_bprmf_train_step: bad shape");
    EL_REQUIRE(u && i && j && loss_out, "el_bprmf_train_step: null batch/loss pointer");
    EL_REQUIRE(step >= 1, "el_bprmf_train_step: step is 1-based");
    const bool adam = (opt == EL_OPT_ADAM_TF_DENSE || opt == EL_OPT_ADAM_LAZY);
    EL_REQUIRE(adam || opt == EL_OPT_SGD, "el_bprmf_train_step: unknown optimiser %d", opt);
    if (adam) EL_REQUIRE(st.mGu && st.vGu && st.mGi && st.vGi && st.mBi && st.vBi, "el_bprmf_train_step: Adam slots missing");
    *rows_mode = (opt == EL_OPT_ADAM_LAZY) || (opt == EL_OPT_SGD && st.tGu != nullptr);
    if (*rows_mode) EL_REQUIRE(st.tGu && st.tGi && st.tBi, "el_bprmf_train_step: stamp arrays missing");
    *vec = rows_aligned16(st.Gu, st.F) && rows_aligned16(st.Gi, st.F) && rows_aligned16(st.uslot ? st.gGu_rows : st.gGu, st.F) &&
           rows_aligned16(st.gGi, st.F) &&
           (!adam || (rows_aligned16(st.mGu, st.F) && rows_aligned16(st.vGu, st.F) &&
                      rows_aligned16(st.mGi, st.F) && rows_aligned16(st.vGi, st.F)));
    return 0;
}

// optimiser phase shared by both gradient paths (gradients are in gGu/gGi/gBi, stamps set in rows mode)
int el_bprmf_apply_optimizer(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, const int32_t* u, const int32_t* i,
                             const int32_t* j, int64_t B, float lr, int opt, int32_t step, float lr_t) {
    const bool adam = (opt == EL_OPT_ADAM_TF_DENSE || opt == EL_OPT_ADAM_LAZY);
    const bool rows_mode = (opt == EL_OPT_ADAM_LAZY) || (opt == EL_OPT_SGD && st.tGu != nullptr);
    const bool vec = rows_aligned16(st.Gu, st.F) && rows_aligned16(st.Gi, st.F) && rows_aligned16(st.gGu, st.F) &&
                     rows_aligned16(st.gGi, st.F) &&
                     (!adam || (rows_aligned16(st.mGu, st.F) && rows_aligned16(st.vGu, st.F) &&
                                rows_aligned16(st.mGi, st.F) && rows_aligned16(st.vGi, st.F)));
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    if (opt == EL_OPT_ADAM_TF_DENSE) {
        const int64_t nu = st.U * (int64_t)st.F, ni = st.I * (int64_t)st.F;
        auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        if (st.uslot) {
            // user table: compact gradient rows of this step (written by the sorted segment kernels)
            EL_REQUIRE(st.F % 4 == 0 && al16(st.Gu) && al16(st.mGu) && al16(st.vGu) && st.gGu_rows && al16(st.gGu_rows),
                       "el_bprmf_apply: compact user-gradient rows need F %% 4 == 0 and 16-byte aligned tables");
            const int F4 = st.F / 4;
            int sh = -1;
            if ((F4 & (F4 - 1)) == 0) { sh = 0; while ((1 << sh) < F4) ++sh; }
            if (ctx->tuning) {
                EL_LAUNCH("k_adam_rows_Gu", (k_adam_rows<2, true>), dim3(stream_grid(ctx, nu / 4 + 1)), dim3(256), 0, s, st.Gu, st.gGu_rows, st.uslot,
                          st.mGu, st.vGu, nu / 4, F4, sh, step, lr_t, b1, b2, eps);
                EL_LAUNCH("k_adam_dense_Gi", k_adam_dense_tune, dim3(stream_grid(ctx, ni / 4 + 1)), dim3(256), 0, s, st.Gi, st.gGi, st.mGi, st.vGi, ni, lr_t, b1, b2, eps);
                EL_LAUNCH("k_adam_dense_Bi", k_adam_dense_tune, dim3(stream_grid(ctx, st.I / 4 + 1)), dim3(256), 0, s, st.Bi, st.gBi, st.mBi, st.vBi, st.I, lr_t, b1, b2, eps);
                EL_CHECK_LAUNCH();
                return 0;
            }
            EL_LAUNCH("k_adam_rows_Gu", (k_adam_rows<2, false>), dim3(stream_grid(ctx, nu / 4 + 1)), dim3(256), 0, s, st.Gu, st.gGu_rows, st.uslot,
                      st.mGu, st.vGu, nu / 4, F4, sh, step, lr_t, b1, b2, eps);
            EL_LAUNCH("k_adam_dense_Gi", k_adam_dense, dim3(stream_grid(ctx, ni / 4 + 1)), dim3(256), 0, s, st.Gi, st.gGi, st.mGi, st.vGi, ni, lr_t, b1, b2, eps);
            EL_LAUNCH("k_adam_dense_Bi", k_adam_dense, dim3(stream_grid(ctx, st.I / 4 + 1)), dim3(256), 0, s, st.Bi, st.gBi, st.mBi, st.vBi, st.I, lr_t, b1, b2, eps);
            EL_CHECK_LAUNCH();
            return 0;
        }
        if (nu + ni <= (4LL << 20) && al16(st.Gu) && al16(st.gGu) && al16(st.mGu) && al16(st.vGu) && al16(st.Gi) && al16(st.gGi) &&
            al16(st.mGi) && al16(st.vGi) && al16(st.Bi) && al16(st.gBi) && al16(st.mBi) && al16(st.vBi)) {
            AdamTriple t = {{st.Gu, st.Gi, st.Bi}, {st.gGu, st.gGi, st.gBi}, {st.mGu, st.mGi, st.mBi}, {st.vGu, st.vGi, st.vBi}, {nu, ni, st.I}};
            const int64_t big = nu > ni ? nu : ni;
            EL_LAUNCH("k_adam_dense3", k_adam_dense3, dim3(stream_grid(ctx, big / 4 + 1)), dim3(256), 0, s, t, lr_t, b1, b2, eps);
            EL_CHECK_LAUNCH();
            return 0;
        }
        if (ctx->tuning) {
            EL_LAUNCH("k_adam_dense_Gu", k_adam_dense_tune, dim3(stream_grid(ctx, nu / 4 + 1)), dim3(256), 0, s, st.Gu, st.gGu, st.mGu, st.vGu, nu, lr_t, b1, b2, eps);
            EL_LAUNCH("k_adam_dense_Gi", k_adam_dense_tune, dim3(stream_grid(ctx, ni / 4 + 1)), dim3(256), 0, s, st.Gi, st.gGi, st.mGi, st.vGi, ni, lr_t, b1, b2, eps);
            EL_LAUNCH("k_adam_dense_Bi", k_adam_dense_tune, dim3(stream_grid(ctx, st.I / 4 + 1)), dim3(256), 0, s, st.Bi, st.gBi, st.mBi, st.vBi, st.I, lr_t, b1, b2, eps);
            EL_CHECK_LAUNCH();
            return 0;
        }
        EL_LAUNCH("k_adam_dense_Gu", k_adam_dense, dim3(stream_grid(ctx, nu / 4 + 1)), dim3(256), 0, s, st.Gu, st.gGu, st.mGu, st.vGu, nu, lr_t, b1, b2, eps);
        EL_LAUNCH("k_adam_dense_Gi", k_adam_dense, dim3(stream_grid(ctx, ni / 4 + 1)), dim3(256), 0, s, st.Gi, st.gGi, st.mGi, st.vGi, ni, lr_t, b1, b2, eps);
        EL_LAUNCH("k_adam_dense_Bi", k_adam_dense, dim3(stream_grid(ctx, st.I / 4 + 1)), dim3(256), 0, s, st.Bi, st.gBi, st.mBi, st.vBi, st.I, lr_t, b1, b2, eps);
        EL_CHECK_LAUNCH();
        return 0;
    }
    if (rows_mode) {
        if (opt == EL_OPT_ADAM_LAZY)
            return vec ? launch_rows_apply<4, true>(st, u, i, j, B, step, lr, lr_t, s)
                       : launch_rows_apply<1, true>(st, u, i, j, B, step, lr, lr_t, s);
        return vec ? launch_rows_apply<4, false>(st, u, i, j, B, step, lr, lr_t, s)
                   : launch_rows_apply<1, false>(st, u, i, j, B, step, lr, lr_t, s);
    }
    const int64_t nu = st.U * (int64_t)st.F, ni = st.I * (int64_t)st.F;
    EL_LAUNCH("k_sgd_dense", k_sgd_dense, dim3(stream_grid(ctx, nu)), dim3(256), 0, s, st.Gu, st.gGu, nu, lr);
    EL_LAUNCH("k_sgd_dense", k_sgd_dense, dim3(stream_grid(ctx, ni)), dim3(256), 0, s, st.Gi, st.gGi, ni, lr);
    EL_LAUNCH("k_sgd_dense", k_sgd_dense, dim3(stream_grid(ctx, st.I)), dim3(256), 0, s, st.Bi, st.gBi, st.I, lr);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" __attribute__((visibility("hidden"))) int el_bprmf_train_step_sorted(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u,
                                          const int32_t* i, const int32_t* j, int64_t B, float lr, float l_w,
                                          float l_b, int opt, int32_t step, float lr_t, double* loss_out, void* ws,
                                          size_t ws_bytes);
extern "C" size_t el_bprmf_ws_bytes(int64_t B, int64_t U, int64_t I, int32_t F);

extern "C" int el_bprmf_train_step(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u,
                                   const int32_t* i, const int32_t* j, int64_t B, float lr, float l_w, float l_b,
                                   int opt, int32_t step, float lr_t, double* loss_out, int algo, void* ws,
                                   size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    bool vec = false, rows_mode = false;
    if (int rc = el_bprmf_check_state(stp, u, i, j, loss_out, opt, step, &vec, &rows_mode)) return rc;
    EL_REQUIRE(algo == EL_BPR_AUTO || algo == EL_BPR_ATOMIC || algo == EL_BPR_SORTED, "el_bprmf_train_step: bad algo %d", algo);
    if (B <= 0) return 0;
    const el_bprmf_state st = *stp;
    bool sorted = (algo == EL_BPR_SORTED);
    if (algo == EL_BPR_AUTO) sorted = (B >= 2048 || st.uslot) && ws != nullptr && ws_bytes >= el_bprmf_ws_bytes(B, st.U, st.I, st.F);
    if (st.Gu_next && opt == EL_OPT_ADAM_TF_DENSE && algo != EL_BPR_ATOMIC && ws != nullptr && ws_bytes >= el_bprmf_ws_bytes(B, st.U, st.I, st.F)) sorted = true;
    if (st.Gu_last) {
        EL_REQUIRE(algo != EL_BPR_ATOMIC && ws != nullptr && ws_bytes >= el_bprmf_ws_bytes(B, st.U, st.I, st.F),
                   "el_bprmf_train_step: the deferred decay (Gu_last) needs the SORTED path and its workspace");
        sorted = true;
    }
    if (st.Gi_last) {
        // (the fused item side stamps Gi_last per row: a step on another path would leave the stamps behind and a later sorted step
        //  would replay gradient-free updates on rows that are already current)
        EL_REQUIRE(algo != EL_BPR_ATOMIC && ws != nullptr && ws_bytes >= el_bprmf_ws_bytes(B, st.U, st.I, st.F),
                   "el_bprmf_train_step: the fused item side (Gi_last) needs the SORTED path and its workspace");
        sorted = true;
    }
    EL_REQUIRE(sorted || !st.uslot, "el_bprmf_train_step: compact user-gradient rows (uslot) need the SORTED path and its workspace");
    EL_REQUIRE(sorted || !(st.Gu_next && opt == EL_OPT_ADAM_TF_DENSE), "el_bprmf_train_step: a second user table (Gu_next) needs the SORTED path and its workspace");
    if (sorted)
        return el_bprmf_train_step_sorted(ctx, stream, stp, u, i, j, B, lr, l_w, l_b, opt, step, lr_t, loss_out, ws, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    el_bprmf_state fst = st;
    if (!rows_mode) fst.tGu = fst.tGi = fst.tBi = nullptr;
    int rc = vec ? launch_fwd_bwd<4>(fst, u, i, j, B, l_w, l_b, step, loss_out, s)
                 : launch_fwd_bwd<1>(fst, u, i, j, B, l_w, l_b, step, loss_out, s);
    if (rc) return rc;
    return el_bprmf_apply_optimizer(ctx, s, st, u, i, j, B, lr, opt, step, lr_t);
}

// ---------------------------------------------------------------------------------------
// K5: NumPy-semantics per-sample SGD, fp64
// ---------------------------------------------------------------------------------------
template <int VW>
__device__ __forceinline__ void ld_vecd(const double* p, double* dst) {
    if (VW == 2) {
        double2 t = *reinterpret_cast<const double2*>(p);
        dst[0] = t.x;
        dst[1] = t.y;
    } else {
        dst[0] = p[0];
    }
}
template <int VW>
__device__ __forceinline__ void st_vecd(double* p, const double* src) {
    if (VW == 2) {
        *reinterpret_cast<double2*>(p) = make_double2(src[0], src[1]);
    } else {
        p[0] = src[0];
    }
}

template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_bprsgd_apply(el_bprsgd_state st, const int32_t* __restrict__ bu,
                                                      const int32_t* __restrict__ bi_,
                                                      const int32_t* __restrict__ bj, int64_t first, int64_t n,
                                                      int lpt) {
    const int F = st.F;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    const bool active = t < n;
    int32_t uu = 0, ii = 0, jj = 0;
    if (active) {
        uu = bu[first + t];
        ii = bi_[first + t];
        jj = bj[first + t];
    }
    double* pu = st.P + (int64_t)uu * F;
    double* qi = st.Q + (int64_t)ii * F;
    double* qj = st.Q + (int64_t)jj * F;
    double vu[CPL][VW], vi[CPL][VW], vj[CPL][VW];
    double di = 0.0, dj = 0.0;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
#pragma unroll
        for (int x = 0; x < VW; ++x) vu[q][x] = vi[q][x] = vj[q][x] = 0.0;
        if (active && e < F) {
            ld_vecd<VW>(pu + e, vu[q]);
            ld_vecd<VW>(qi + e, vi[q]);
            ld_vecd<VW>(qj + e, vj[q]);
        }
#pragma unroll
        for (int x = 0; x < VW; ++x) {
            di += vu[q][x] * vi[q][x];
            dj += vu[q][x] * vj[q][x];
        }
    }
    di = el_group_sum(di, lpt);
    dj = el_group_sum(dj, lpt);
    if (!active) return;
    const double b_i = st.b[ii], b_j = st.b[jj];
    // BPRMF_model.py:66-68,98: z = 1/(1+exp(x_ui - x_uj)), x = global_bias(0) + b + p.q
    const double xui = (0.0 + b_i) + di, xuj = (0.0 + b_j) + dj;
    const double z = 1.0 / (1.0 + exp(xui - xuj));
    const double lr = st.lr;
    if (sub == 0) {
        st.b[ii] = b_i + lr * (z - st.reg_bias * b_i);      // :100-101
        st.b[jj] = b_j + lr * (-z - st.reg_bias * b_j);     // :104-105
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
        if (e < F) {
            double nu[VW], ni[VW], nj[VW];
#pragma unroll
            for (int x = 0; x < VW; ++x) {
                // :108-109 user row first ...
                nu[x] = vu[q][x] + lr * ((vi[q][x] - vj[q][x]) * z - st.reg_user * vu[q][x]);
                // ... :112-117 item rows see the UPDATED user row (view aliasing)
                ni[x] = vi[q][x] + lr * (nu[x] * z - st.reg_pos * vi[q][x]);
                nj[x] = vj[q][x] + lr * (-nu[x] * z - st.reg_neg * vj[q][x]);
            }
            st_vecd<VW>(pu + e, nu);
            st_vecd<VW>(qi + e, ni);
            st_vecd<VW>(qj + e, nj);
        }
    }
}

static int launch_bprsgd(const el_bprsgd_state& st, const int32_t* u, const int32_t* i, const int32_t* j,
                         int64_t first, int64_t n, hipStream_t s) {
    const bool vec = (st.F % 2 == 0) && (((uintptr_t)st.P) % 16 == 0) && (((uintptr_t)st.Q) % 16 == 0);
    const int vw = vec ? 2 : 1;
    int cpl = 1;
    int lpt = el_pick_lpt(st.F, vw, &cpl);
    EL_REQUIRE(cpl <= 4, "el_bprsgd_apply: F=%d too large for this build", st.F);
    int64_t threads = n * lpt;
    unsigned grid = (unsigned)((threads + 255) / 256);
#define EL_SGD_LAUNCH(VW_, CPL_) \
    EL_LAUNCH("k_bprsgd_apply", (k_bprsgd_apply<VW_, CPL_>), dim3(grid), dim3(256), 0, s, st, u, i, j, first, n, lpt)
    if (vec) {
        if (cpl == 1) EL_SGD_LAUNCH(2, 1);
        else if (cpl == 2) EL_SGD_LAUNCH(2, 2);
        else EL_SGD_LAUNCH(2, 4);
    } else {
        if (cpl == 1) EL_SGD_LAUNCH(1, 1);
        else if (cpl == 2) EL_SGD_LAUNCH(1, 2);
        else EL_SGD_LAUNCH(1, 4);
    }
#undef EL_SGD_LAUNCH
    EL_CHECK_LAUNCH();
    return 0;
}

static int check_sgd_state(const el_bprsgd_state* stp) {
    EL_REQUIRE(stp != nullptr, "el_bprsgd: null state");
    EL_REQUIRE(stp->P && stp->Q && stp->b, "el_bprsgd: null table");
    EL_REQUIRE(stp->F >= 1 && stp->U >= 1 && stp->I >= 1, "el_bprsgd: bad shape");
    return 0;
}

extern "C" int el_bprsgd_apply(el_ctx* ctx, void* stream, const el_bprsgd_state* stp, const int32_t* u,
                               const int32_t* i, const int32_t* j, int64_t first, int64_t n) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_sgd_state(stp)) return rc;
    EL_REQUIRE(u && i && j && first >= 0, "el_bprsgd_apply: bad triplet arrays");
    if (n <= 0) return 0;
    return launch_bprsgd(*stp, u, i, j, first, n, (hipStream_t)stream);
}

extern "C" int el_bprsgd_apply_levels(el_ctx* ctx, void* stream, const el_bprsgd_state* stp, const int32_t* u,
                                      const int32_t* i, const int32_t* j, const int64_t* level_start_host,
                                      int64_t n_levels) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_sgd_state(stp)) return rc;
    EL_REQUIRE(u && i && j && level_start_host && n_levels >= 0, "el_bprsgd_apply_levels: bad arguments");
    for (int64_t L = 0; L < n_levels; ++L) {
        int64_t a = level_start_host[L], b = level_start_host[L + 1];
        if (b > a)
            if (int rc = launch_bprsgd(*stp, u, i, j, a, b - a, (hipStream_t)stream)) return rc;
    }
    return 0;
}

// Host-only scheduling helper: dependency levels of a triplet sequence.
extern "C" int el_bprsgd_levels_host(const int32_t* u, const int32_t* i, const int32_t* j, int64_t n, int64_t U,
                                     int64_t I, int32_t* order, int64_t* level_start, int64_t level_start_cap,
                                     int64_t* n_levels) {
    EL_REQUIRE(u && i && j && order && level_start && n_levels, "el_bprsgd_levels_host: null pointer");
    EL_REQUIRE(n >= 0 && n < 0x7fffffffLL, "el_bprsgd_levels_host: n out of range");
    std::vector<int32_t> lu((size_t)U, 0), li((size_t)I, 0), lev((size_t)n);
    int32_t maxl = 0;
    for (int64_t t = 0; t < n; ++t) {
        EL_REQUIRE(u[t] >= 0 && u[t] < U && i[t] >= 0 && i[t] < I && j[t] >= 0 && j[t] < I,
                   "el_bprsgd_levels_host: triplet %lld out of range", (long long)t);
        int32_t a = lu[u[t]], b = li[i[t]], c = li[j[t]];
        int32_t l = a > b ? a : b;
        l = (l > c ? l : c) + 1;
        lev[t] = l;
        lu[u[t]] = l;
        li[i[t]] = l;
        li[j[t]] = l;
        if (l > maxl) maxl = l;
    }
    EL_REQUIRE((int64_t)maxl + 1 <= level_start_cap, "el_bprsgd_levels_host: %d levels exceed capacity %lld", maxl,
               (long long)level_start_cap);
    // level l (1-based) occupies order[level_start[l-1] .. level_start[l])
    std::vector<int64_t> count((size_t)maxl + 1, 0);
    for (int64_t t = 0; t < n; ++t) count[lev[t]]++;
    level_start[0] = 0;
    for (int32_t l = 1; l <= maxl; ++l) level_start[l] = level_start[l - 1] + count[l];
    std::vector<int64_t> cur((size_t)maxl + 1, 0);
    for (int32_t l = 1; l <= maxl; ++l) cur[l] = level_start[l - 1];
    for (int64_t t = 0; t < n; ++t) order[cur[lev[t]]++] = (int32_t)t;
    *n_levels = maxl;
    return 0;
}

// ---------------------------------------------------------------------------------------
// The epoch loop of BPRMF_batch.train (BPRMF_batch.py:100-109) without host round trips: sampler.step + train_step for
// consecutive batches from one call.  At the reference's default batch size (512) a step is two microsecond kernels;
// what an epoch used to consist of was the Python / ctypes cost per step (~40 us) and a sampler launch whose 15 us
// chain of dependent CSR probes nothing overlapped.  Here
//   * the sampler does not read the model, so the triplets of up to LOOP_CHUNK/B batches come from ONE launch
//     (sample t of the epoch is Philox counter first_sample + t whichever launch draws it);
//   * the small-batch step (atomic gradient kernel + the fused three-tensor Adam) is captured ONCE in a hipGraph of
//     LOOP_GRAPH_STEPS steps whose per-step scalars (batch offset, Adam step size) live in device memory (LoopCtl), and
//     replayed -- one hipGraphLaunch per 32 steps instead of 64 kernel launches;
//   * large batches (sorted path) and the other optimisers run the same kernels eagerly.
// ---------------------------------------------------------------------------------------
static const int64_t LOOP_CHUNK = 4 << 20;     // triplets per sampler launch (48 MB of indices)
static const int LOOP_GRAPH_STEPS = 32;

static int64_t loop_cap(int64_t events, int64_t B) {
    int64_t most = (LOOP_CHUNK / B) * B;
    if (most < B) most = B;
    const int64_t want = ((events + B - 1) / B) * B;
    return want < most ? (want < B ? B : want) : most;
}
static size_t loop_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t el_bprmf_train_loop_ws_bytes(int64_t events, int64_t B) {
    if (events <= 0 || B <= 0) return 0;
    const int64_t steps = (events + B - 1) / B;
    return loop_align((size_t)loop_cap(events, B) * 12) + loop_align((size_t)steps * 4) + 256;
}

struct LoopGraphKey {
    el_bprmf_state st;
    const int32_t* bu;
    int64_t cap, B;
    float l_w, l_b;
    double* loss_out;
    const float* lr_tab;
    LoopCtl* ctl;
    int vec, lpt, cpl;
    unsigned grid_f, grid_a;
};

template <int VW>
static void launch_fwd_g(int cpl, unsigned grid, hipStream_t s, const el_bprmf_state& st, const int32_t* bu, const int32_t* bi,
                         const int32_t* bj, int64_t B, float l_w, float l_b, int lpt, double* loss_out, LoopCtl* ctl) {
    if (cpl == 1) hipLaunchKernelGGL((k_bprmf_fwd_bwd_g<VW, 1>), dim3(grid), dim3(256), 0, s, st, bu, bi, bj, B, l_w, l_b, lpt, loss_out, ctl);
    else if (cpl == 2) hipLaunchKernelGGL((k_bprmf_fwd_bwd_g<VW, 2>), dim3(grid), dim3(256), 0, s, st, bu, bi, bj, B, l_w, l_b, lpt, loss_out, ctl);
    else hipLaunchKernelGGL((k_bprmf_fwd_bwd_g<VW, 4>), dim3(grid), dim3(256), 0, s, st, bu, bi, bj, B, l_w, l_b, lpt, loss_out, ctl);
}

// (Re)build the LOOP_GRAPH_STEPS-step graph when the launch parameters differ from the cached ones.
static int loop_graph_get(el_ctx* ctx, hipStream_t s, const LoopGraphKey& key, const AdamTriple& t, const int32_t* bi,
                          const int32_t* bj) {
    if (ctx->loop_graph_exec && ctx->loop_graph_key.size() == sizeof(key) && memcmp(ctx->loop_graph_key.data(), &key, sizeof(key)) == 0)
        return 0;
    if (ctx->loop_graph_exec) {
        (void)hipGraphExecDestroy((hipGraphExec_t)ctx->loop_graph_exec);
        ctx->loop_graph_exec = nullptr;
    }
    hipStream_t cs = nullptr;                                   // a capture stream of our own: `s` may be the null stream
    EL_CHECK_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        for (int k = 0; k < LOOP_GRAPH_STEPS; ++k) {
            if (key.vec) launch_fwd_g<4>(key.cpl, key.grid_f, cs, key.st, key.bu, bi, bj, key.B, key.l_w, key.l_b, key.lpt, key.loss_out, key.ctl);
            else launch_fwd_g<1>(key.cpl, key.grid_f, cs, key.st, key.bu, bi, bj, key.B, key.l_w, key.l_b, key.lpt, key.loss_out, key.ctl);
            hipLaunchKernelGGL(k_adam_dense3_g, dim3(key.grid_a), dim3(256), 0, cs, t, key.lr_tab, 0.9f, 0.999f, 1e-7f, key.ctl, key.B);
        }
        e = hipStreamEndCapture(cs, &graph);
    }
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(cs);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        el_set_error("el_bprmf_train_loop: graph capture failed: %s", hipGetErrorString(e));
        return 1;
    }
    ctx->loop_graph_exec = exec;
    ctx->loop_graph_key.assign((const unsigned char*)&key, (const unsigned char*)&key + sizeof(key));
    (void)s;
    return 0;
}

extern "C" int el_bprmf_train_loop(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int64_t* pos_indptr,
                                   const int32_t* pos_indices, uint64_t seed, uint64_t first_sample, int64_t events,
                                   int64_t B, float lr, float l_w, float l_b, int opt, int32_t first_step,
                                   const float* lr_t_host, double* loss_out, int algo, void* ws, size_t ws_bytes,
                                   void* loop_ws, size_t loop_ws_bytes, const void* sampler_meta) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp != nullptr && B >= 1 && events >= 0, "el_bprmf_train_loop: bad arguments");
    if (events == 0) return 0;
    const size_t need = el_bprmf_train_loop_ws_bytes(events, B);
    EL_REQUIRE(loop_ws != nullptr && loop_ws_bytes >= need, "el_bprmf_train_loop: loop workspace too small (%zu < %zu)", loop_ws_bytes, need);
    const bool adam = (opt == EL_OPT_ADAM_TF_DENSE || opt == EL_OPT_ADAM_LAZY);
    EL_REQUIRE(!adam || lr_t_host != nullptr, "el_bprmf_train_loop: lr_t_host is required for the Adam modes");
    hipStream_t s = (hipStream_t)stream;
    const int64_t cap = loop_cap(events, B), steps = (events + B - 1) / B;
    int32_t* bu = (int32_t*)loop_ws;
    int32_t* bi = bu + cap;
    int32_t* bj = bu + 2 * cap;
    float* lr_tab = (float*)((char*)loop_ws + loop_align((size_t)cap * 12));
    LoopCtl* ctl = (LoopCtl*)((char*)lr_tab + loop_align((size_t)steps * 4));

    // graph form: small batch (the atomic gradient kernel is what AUTO picks below 2048), TF-dense Adam on a model small
    // enough for the fused three-tensor pass, no per-kernel timing requested
    const bool graphs_on = ctx->opt.loop_graph != 0;
    const el_bprmf_state& st = *stp;
    bool use_graph = graphs_on && !ctx->timing && opt == EL_OPT_ADAM_TF_DENSE && steps >= 4 && !stp->uslot &&
                     (algo == EL_BPR_ATOMIC || (algo == EL_BPR_AUTO && B < 2048));
    const int64_t nu = st.U * (int64_t)st.F, ni = st.I * (int64_t)st.F;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    use_graph = use_graph && nu + ni <= (4LL << 20) && st.mGu && st.vGu && st.mGi && st.vGi && st.mBi && st.vBi && al16(st.Gu) &&
                al16(st.gGu) && al16(st.mGu) && al16(st.vGu) && al16(st.Gi) && al16(st.gGi) && al16(st.mGi) && al16(st.vGi) &&
                al16(st.Bi) && al16(st.gBi) && al16(st.mBi) && al16(st.vBi);
    if (use_graph) {
        bool vec = false, rows_mode = false;
        if (int rc = el_bprmf_check_state(stp, bu, bi, bj, loss_out, opt, first_step, &vec, &rows_mode)) return rc;
        LoopGraphKey key;
        memset(&key, 0, sizeof(key));
        key.st = st;
        key.st.tGu = key.st.tGi = key.st.tBi = nullptr;
        key.bu = bu, key.cap = cap, key.B = B, key.l_w = l_w, key.l_b = l_b, key.loss_out = loss_out, key.lr_tab = lr_tab, key.ctl = ctl;
        key.vec = vec ? 1 : 0;
        key.lpt = el_pick_lpt(st.F, vec ? 4 : 1, &key.cpl);
        EL_REQUIRE(key.cpl <= 4, "el_bprmf_train_loop: F=%d too large for this build", st.F);
        key.grid_f = (unsigned)((B * key.lpt + 255) / 256);
        const int64_t big = nu > ni ? nu : ni;
        key.grid_a = stream_grid(ctx, big / 4 + 1);
        AdamTriple t = {{st.Gu, st.Gi, st.Bi}, {st.gGu, st.gGi, st.gBi}, {st.mGu, st.mGi, st.mBi}, {st.vGu, st.vGi, st.vBi}, {nu, ni, st.I}};
        if (loop_graph_get(ctx, s, key, t, bi, bj)) use_graph = false;     // capture unavailable: eager path, same results
        else {
            // the table travels through a pinned buffer of the context: the caller's array need not outlive this call
            if (ctx->lr_copied) EL_CHECK_HIP(hipEventSynchronize(ctx->lr_copied));      // previous copy has left the buffer
            else EL_CHECK_HIP(hipEventCreateWithFlags(&ctx->lr_copied, hipEventDisableTiming));
            if (ctx->lr_pinned_cap < (size_t)steps) {
                if (ctx->lr_pinned) (void)hipHostFree(ctx->lr_pinned);
                ctx->lr_pinned = nullptr, ctx->lr_pinned_cap = 0;
                EL_CHECK_HIP(hipHostMalloc((void**)&ctx->lr_pinned, (size_t)steps * 4, hipHostMallocDefault));
                ctx->lr_pinned_cap = (size_t)steps;
            }
            memcpy(ctx->lr_pinned, lr_t_host, (size_t)steps * 4);
            EL_CHECK_HIP(hipMemcpyAsync(lr_tab, ctx->lr_pinned, (size_t)steps * 4, hipMemcpyHostToDevice, s));
            EL_CHECK_HIP(hipEventRecord(ctx->lr_copied, s));
        }
    }

    int64_t k = 0;
    el_bprmf_state cur = *stp;
    for (int64_t c0 = 0; c0 < events; c0 += cap) {
        const int64_t cn = (events - c0 < cap) ? events - c0 : cap;
        if (int rc = el_bpr_sample_meta(ctx, stream, pos_indptr, pos_indices, sampler_meta, st.U, st.I, 0, st.I, seed,
                                        first_sample + (uint64_t)c0, cn, bu, bi, bj))
            return rc;
        const int64_t csteps = (cn + B - 1) / B;
        if (use_graph) {
            hipLaunchKernelGGL(k_loop_ctl_set, dim3(1), dim3(1), 0, s, ctl, cn, (int32_t)k);
            for (int64_t g = 0; g < csteps; g += LOOP_GRAPH_STEPS) EL_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)ctx->loop_graph_exec, s));
            k += csteps;
            continue;
        }
        for (int64_t off = 0; off < cn; off += B, ++k) {
            const int64_t n = (cn - off < B) ? cn - off : B;
            if (int rc = el_bprmf_train_step(ctx, stream, &cur, bu + off, bi + off, bj + off, n, lr, l_w, l_b, opt,
                                             first_step + (int32_t)k, adam ? lr_t_host[k] : 0.f, loss_out, algo, ws, ws_bytes))
                return rc;
            if (cur.Gu_next && !cur.Gu_last && opt == EL_OPT_ADAM_TF_DENSE) {      // fused user side: the step left the current table in Gu_next
                float* t = cur.Gu;
                cur.Gu = cur.Gu_next;
                cur.Gu_next = t;
            }
        }
    }
    EL_CHECK_LAUNCH();
    return 0;
}
