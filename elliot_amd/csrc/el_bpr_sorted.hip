// BPRMF_batch train step, sort-based gradient reduction (no floating-point atomics on hot rows).
//
// Replaces BPRMF_batch_model.train_step (BPRMF_batch_model.py:58-80).  TensorFlow hands the
// optimiser IndexedSlices whose duplicate rows are segment-summed; here the batch is sorted by
// row once and every row's gradient is reduced in registers by the lane group that walks its
// segment:
//   1. k_bpr_prep          keys/payloads: (u_b, b) and (item, b | neg<<31) for i_b and j_b
//   2. rocprim radix sort  (stable -> segments keep batch order -> the sums are deterministic)
//   3. k_bpr_user_seg      per user segment: gather gamma_i, gamma_j of every occurrence, x_ui - x_uj,
//                          s_b = dloss/dd, batch loss, dGu row = sum s_b (gamma_i - gamma_j) + cnt l_w gamma_u
//   4. k_bpr_item_seg      per item segment: dGi row = sum +-s_b gamma_u(b) + cnt l_w gamma_item, dBi
//   5. optimiser           k_adam_dense (TF semantics) / k_rows_apply (el_bpr.hip)
// Segments longer than a chunk (popular items under Zipf: tens of thousands of occurrences) are
// split over several lane groups; segments that live inside one chunk are written with plain stores, a
// segment cut by chunk boundaries leaves one partial row per chunk and k_bpr_item_combine adds them IN
// CHUNK ORDER (round 6: no floating-point atomics anywhere in the sorted step -- two runs on the same
// batches give the same bits, and the fused / deferred forms equal the two-pass form bit for bit at any size).
#include "el_common.h"
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

template <int VW>
__device__ __forceinline__ void ldv(const float* p, float* dst) {
    if (VW == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        dst[0] = t.x;
        dst[1] = t.y;
        dst[2] = t.z;
        dst[3] = t.w;
    } else if (VW == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        dst[0] = t.x;
        dst[VW - 1] = t.y;
    } else {
        dst[0] = p[0];
    }
}
template <int VW>
__device__ __forceinline__ void stv(float* p, const float* src) {
    if (VW == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(src[0], src[1], src[2], src[3]);
    } else if (VW == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(src[0], src[VW - 1]);
    } else {
        p[0] = src[0];
    }
}

__device__ __forceinline__ float el_softplus_s(float x) {
    if (x > 15.0f) return x;
    if (x < -15.0f) return expf(x);
    return log1pf(expf(x));
}

__global__ __launch_bounds__(256) void k_bpr_prep(const int32_t* __restrict__ u, const int32_t* __restrict__ i,
                                                  const int32_t* __restrict__ j, int64_t B, u32* keyU, u32* valU,
                                                  u32* keyI, u32* valI, u32 item_key_off) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    keyU[t] = (u32)u[t];
    valU[t] = (u32)t;
    keyI[t] = (u32)i[t] + item_key_off;
    valI[t] = (u32)t;
    keyI[B + t] = (u32)j[t] + item_key_off;
    valI[B + t] = (u32)t | 0x80000000u;
}

#define BPR_USTG 16   // positions per LDS stage, user segments (6 words each)
#define BPR_ISTG 64   // item segments (4 words each)

struct SegParams {
    el_bprmf_state st;
    const int32_t* bi;   // item of positive per triplet
    const int32_t* bj;   // negative
    const int32_t* bu;   // user
    const u32* keys;     // sorted row ids (+ key_off)
    u32 key_off;         // items are sorted in ONE radix sort behind the users, as U + item
    const u32* vals;     // sorted payloads
    float* s;            // [B] dloss/dd per triplet (BPR: written by the user segments; CML: dloss/dD, given)
    const float* s2;     // CML only: dloss/dE per triplet (the bias-side coefficient)
    int cml;             // 0: BPR (the user segments compute the forward pass), 1: CML (coefficients given, el_cml.hip)
    int64_t n;           // number of sorted entries (B or 2B)
    int chunk;           // positions per lane group
    int lpt;
    float l_w, l_b;
    int32_t step;
    double* loss_out;
    // item segments: gamma_u of triplet b = ubase[uidx[b], :]  (NULL = st.Gu / bu: the user table itself, by user id; the deferred
    // decay points them at the pre-update rows the user-side kernel left in Gu_old, by segment head position)
    const float* ubase;
    const int32_t* uidx;
};

struct FusedParams {
    const int32_t* rowptr;    // [U + 1]
    float* Gu_new;            // [U, F]: the updated user rows (Gu keeps the pre-update values for the item segments)
    float lr_t, b1, b2, eps;
    // deferred decay (el_bprmf_state.Gu_last): only the rows with triplets in the batch are read; their missed gradient-free steps
    // (last, t - 1] are replayed in registers first, the caught-up row goes to old_rows[head position] for the item segments
    // (hpos[b] = that position, per triplet) and theta is updated IN PLACE
    int32_t* last;            // [U]
    float* old_rows;          // [>= B, F]
    int32_t* hpos;            // [B]
    float* hist;              // lr_t of step s at hist[s & hist_mask]
    int hist_mask;
    int32_t t;                // this optimiser step
};

// ---- user segments -----------------------------------------------------------------------
// DEFER (el_bprmf_state.Gu_last, the deferred decay): the step's whole user side on the batch's rows only.  A row's postponed
// gradient-free steps (last, t - 1] are taken in registers when its segment starts (SER: in closed form, el_adam_series_*;
// otherwise step by step, the same bits as the every-row pass); a group owns every segment whose head lies in its chunk (the compact
// rule below), and where the two-kernel form writes the gradient row, this form takes Keras' Adam step on the row right away:
// m, v (fetched when the segment starts) and theta updated IN PLACE, the pre-update theta left in old_rows[head] for the item
// segments, hpos[b] = head for every triplet of the segment, Gu_last[user] = t.  Work proportional to B, whatever U is.
// DEFER, PRE (round 5): the Adam slots m, v of a segment head and its Gu_last stamp are fetched TOGETHER with the row gathers of the
// SUB positions in flight (a position is a head when its key differs from the one in front of it: known from the staged keys before
// any row arrives) -- at 10 M users nearly every position of a 1 M-triplet batch starts a segment, and fetching m, v only once the
// walk reaches the head put a second full memory latency behind every position's gathers (5 L + 4 R per four positions instead of
// L + 4 R, at three waves per SIMD).
#ifndef EL_USEG_WAVES
#define EL_USEG_WAVES 4          // (A/B builds: scripts/exp/build_variants.sh)
#endif
template <int VW, int CPL, bool DEFER, int SUBD = 0, bool SER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((SER && CPL == 1) ? EL_USEG_WAVES : 1, 8))) void k_bpr_user_seg(SegParams p, FusedParams f) {
    const int F = p.st.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    if (DEFER && gid == 0) f.hist[f.t & f.hist_mask] = f.lr_t;     // lr_t of THIS step into the ring (replays here read steps < t)
    int64_t p0 = grp * p.chunk;
    int64_t p1 = (p0 + p.chunk < p.n) ? p0 + p.chunk : p.n;
    // Compact user-gradient rows (el_bprmf_state.uslot): a segment belongs, whole, to the group whose chunk holds its HEAD --
    // that group walks on past its chunk end, the next one skips the positions of a segment that began before its chunk
    // (user segments are short: Poisson(B/U)).  Every row is then one plain store into gGu_rows[head position].
    const bool compact = DEFER || p.st.uslot != nullptr;
    if (compact && p0 < p1) {
        while (p0 < p1 && p0 > 0 && p.keys[p0] == p.keys[p0 - 1]) ++p0;
        if (p0 < p1)
            while (p1 < p.n && p.keys[p1] == p.keys[p1 - 1]) ++p1;
    }
    float myloss = 0.f;
    if (p0 < p1) {
        int64_t cur = -1, head = 0;
        bool started_inside = false;
        float gu[CPL][VW], acc[CPL][VW];
        float mrow[DEFER ? CPL : 1][VW], vrow[DEFER ? CPL : 1][VW];
        int cnt = 0;
        auto flush = [&](bool ends_inside) {
            if (DEFER) {
                const float omb1 = 1.0f - f.b1, omb2 = 1.0f - f.b2;
                const float w = (float)cnt * p.l_w;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * VW;
                    if (e < F) {
                        float tn[VW], mn[VW], vn[VW];
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            const float g = acc[q][x] + w * gu[q][x];
                            tn[x] = gu[q][x], mn[x] = mrow[DEFER ? q : 0][x], vn[x] = vrow[DEFER ? q : 0][x];
                            el_adam_elem(tn[x], mn[x], vn[x], g, f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
                        }
                        stv<VW>(f.old_rows + head * F + e, gu[q]);          // the pre-update row, for the item segments
                        stv<VW>(p.st.Gu + cur * F + e, tn);
                        stv<VW>(p.st.mGu + cur * F + e, mn);
                        stv<VW>(p.st.vGu + cur * F + e, vn);
                    }
                }
                if (sub == 0) f.last[cur] = f.t;
                return;
            }
            float* g = compact ? p.st.gGu_rows + head * F : p.st.gGu + cur * F;
            const float w = (float)cnt * p.l_w;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
                if (e < F) {
                    float v[VW];
#pragma unroll
                    for (int x = 0; x < VW; ++x) v[x] = acc[q][x] + w * gu[q][x];
                    if (compact || (started_inside && ends_inside)) {
                        stv<VW>(g + e, v);
                    } else {
#pragma unroll
                        for (int x = 0; x < VW; ++x) atomicAdd(g + e + x, v[x]);
                    }
                }
            }
            if (sub == 0 && compact) p.st.uslot[cur] = ((int64_t)p.step << 32) | head;
            if (sub == 0 && p.st.tGu) p.st.tGu[cur] = p.step;
        };
        // The index chain of a position is three dependent loads deep (sorted key / triplet -> i, j -> Bi[i], Bi[j]); a group
        // stages BPR_USTG positions at once in LDS (one lane per position: all chains in flight together), then walks them
        // SUB at a time with only the row gathers left on the critical path (SUB*3 row loads in flight per lane).
        extern __shared__ unsigned char seg_lds[];
        const int gl = (int)(threadIdx.x / lpt), ngl = 256 / lpt;
        u32* s_key = reinterpret_cast<u32*>(seg_lds) + (0 * ngl + gl) * BPR_USTG;
        u32* s_b = reinterpret_cast<u32*>(seg_lds) + (1 * ngl + gl) * BPR_USTG;
        u32* s_i = reinterpret_cast<u32*>(seg_lds) + (2 * ngl + gl) * BPR_USTG;
        u32* s_j = reinterpret_cast<u32*>(seg_lds) + (3 * ngl + gl) * BPR_USTG;
        float* s_bi = reinterpret_cast<float*>(seg_lds) + (4 * ngl + gl) * BPR_USTG;
        float* s_bj = reinterpret_cast<float*>(seg_lds) + (5 * ngl + gl) * BPR_USTG;
        int32_t* s_last = reinterpret_cast<int32_t*>(seg_lds) + (6 * ngl + gl) * BPR_USTG;
        constexpr int SUB = SUBD > 0 ? SUBD : ((CPL == 1) ? 4 : (CPL == 2 ? 2 : 1));
        constexpr bool PRE = DEFER && SUBD > 0;                 // m, v, Gu_last of segment heads prefetched with the gathers
        for (int64_t sbase = p0; sbase < p1; sbase += BPR_USTG) {
            const int cs = (int)((p1 - sbase < BPR_USTG) ? p1 - sbase : BPR_USTG);
            for (int t = sub; t < cs; t += lpt) {
                const u32 b = p.vals[sbase + t];
                const int32_t ii = p.bi[b], jj = p.bj[b];
                const u32 kk = p.keys[sbase + t];
                s_key[t] = kk;
                s_b[t] = b;
                s_i[t] = (u32)ii;
                s_j[t] = (u32)jj;
                s_bi[t] = p.st.Bi[ii];
                s_bj[t] = p.st.Bi[jj];
                if (PRE) s_last[t] = f.last[kk];
            }
            el_wave_lds_sync();
            for (int base = 0; base < cs; base += SUB) {
                int64_t keyv[SUB];
                bool okv[SUB];
                float rgu[SUB][CPL][VW], rgi[SUB][CPL][VW], rgj[SUB][CPL][VW];
                float rm[PRE ? SUB : 1][CPL][VW], rv[PRE ? SUB : 1][CPL][VW];
                int rlast[PRE ? SUB : 1];
#pragma unroll
                for (int t = 0; t < SUB; ++t) {
                    okv[t] = base + t < cs;
                    const int tt = okv[t] ? base + t : base;
                    keyv[t] = (int64_t)s_key[tt];
                    const float* pu = p.st.Gu + keyv[t] * F;
                    const float* pi = p.st.Gi + (int64_t)s_i[tt] * F;
                    const float* pj = p.st.Gi + (int64_t)s_j[tt] * F;
                    // (group-uniform) this position starts a segment: what the walk below finds as key != cur
                    const bool hd = PRE && okv[t] && (t == 0 ? keyv[0] != cur : keyv[t] != keyv[t > 0 ? t - 1 : 0]);
                    if (PRE) rlast[PRE ? t : 0] = s_last[tt];
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        const int e = (sub + q * lpt) * VW;
#pragma unroll
                        for (int x = 0; x < VW; ++x) rgu[t][q][x] = rgi[t][q][x] = rgj[t][q][x] = 0.f;
                        if (PRE) {
#pragma unroll
                            for (int x = 0; x < VW; ++x) rm[PRE ? t : 0][q][x] = rv[PRE ? t : 0][q][x] = 0.f;
                        }
                        if (okv[t] && e < F) {
                            ldv<VW>(pu + e, rgu[t][q]);
                            ldv<VW>(pi + e, rgi[t][q]);
                            ldv<VW>(pj + e, rgj[t][q]);
                            if (hd) {
                                ldv<VW>(p.st.mGu + keyv[t] * F + e, rm[PRE ? t : 0][q]);
                                ldv<VW>(p.st.vGu + keyv[t] * F + e, rv[PRE ? t : 0][q]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < SUB; ++t) {
                    if (!okv[t]) continue;                       // group-uniform
                    const int64_t pos = sbase + base + t, key = keyv[t];
                    const int64_t b = (int64_t)s_b[base + t];
                    if (key != cur) {
                        if (cur >= 0) flush(true);
                        cur = key;
                        head = pos;
                        started_inside = (pos > p0) || (pos == 0) || ((int64_t)p.keys[pos - 1] != key);
                        cnt = 0;
#pragma unroll
                        for (int q = 0; q < CPL; ++q)
#pragma unroll
                            for (int x = 0; x < VW; ++x) {
                                gu[q][x] = rgu[t][q][x];
                                acc[q][x] = 0.f;
                            }
                        if (DEFER) {                             // the row's Adam slots: needed when the segment ends
                            int lastv;
                            if (PRE) {                           // they arrived with the gathers
                                lastv = rlast[PRE ? t : 0];
#pragma unroll
                                for (int q = 0; q < CPL; ++q)
#pragma unroll
                                    for (int x = 0; x < VW; ++x) {
                                        mrow[DEFER ? q : 0][x] = rm[PRE ? t : 0][q][x];
                                        vrow[DEFER ? q : 0][x] = rv[PRE ? t : 0][q][x];
                                    }
                            } else {
                                lastv = f.last[key];
#pragma unroll
                                for (int q = 0; q < CPL; ++q) {
                                    const int e = (sub + q * lpt) * VW;
#pragma unroll
                                    for (int x = 0; x < VW; ++x) mrow[DEFER ? q : 0][x] = vrow[DEFER ? q : 0][x] = 0.f;
                                    if (e < F) {
                                        ldv<VW>(p.st.mGu + key * F + e, mrow[DEFER ? q : 0]);
                                        ldv<VW>(p.st.vGu + key * F + e, vrow[DEFER ? q : 0]);
                                    }
                                }
                            }
                            // the row's postponed gradient-free steps (last, t - 1], in registers, before anything uses it
                            const int nsr = (f.t - 1) - lastv;
                            if (nsr > 0) {
                                bool nz = false;
#pragma unroll
                                for (int q = 0; q < CPL; ++q)
#pragma unroll
                                    for (int x = 0; x < VW; ++x) nz = nz || mrow[DEFER ? q : 0][x] != 0.f || vrow[DEFER ? q : 0][x] != 0.f;
                                if (el_group_any(nz, lpt)) {     // (m = v = 0: the fixed point of the step, whatever the gap)
                                    auto lrs = [&](int s2) { return f.hist[(lastv + 1 + s2) & f.hist_mask]; };
                                    if (SER) {                   // the row-level sums once, then O(1) per element
                                        const el_series sr = el_adam_series_sums(nsr, lrs);
#pragma unroll
                                        for (int q = 0; q < CPL; ++q) el_adam_series_apply<VW>(gu[q], mrow[DEFER ? q : 0], vrow[DEFER ? q : 0], sr);
                                    } else {
#pragma unroll
                                        for (int q = 0; q < CPL; ++q) el_adam_replay<VW>(gu[q], mrow[DEFER ? q : 0], vrow[DEFER ? q : 0], nsr, lrs);
                                    }
                                }
                            }
                        }
                    }
                    if (DEFER && sub == 0) f.hpos[b] = (int32_t)head;
                    float sb = 0.f;
                    if (p.cml) {
                        sb = 2.0f * p.s[b];                          // d|u-j|^2/du - d|u-i|^2/du = 2 (i - j), times dloss/dD_b
                    } else {
                        // only the two dot products have to be known to the whole group; the squared norms feed nothing but
                        // the batch loss, so every lane adds its own share and the loss reduction at the end sums them
                        float dpi = 0.f, dpj = 0.f, nsq = 0.f;
#pragma unroll
                        for (int q = 0; q < CPL; ++q)
#pragma unroll
                            for (int x = 0; x < VW; ++x) {
                                dpi += gu[q][x] * rgi[t][q][x];
                                dpj += gu[q][x] * rgj[t][q][x];
                                nsq += gu[q][x] * gu[q][x] + rgi[t][q][x] * rgi[t][q][x] + rgj[t][q][x] * rgj[t][q][x];
                            }
                        dpi = el_group_sum(dpi, lpt);
                        dpj = el_group_sum(dpj, lpt);
                        const float beta_i = s_bi[base + t], beta_j = s_bj[base + t];
                        const float d = (beta_i + dpi) - (beta_j + dpj);   // x_ui - x_uj  (BPRMF_batch_model.py:53,65)
                        const float dc = fminf(fmaxf(d, -80.0f), 1e8f);
                        if (d >= -80.0f) sb = -1.0f / (1.0f + expf(d));
                        myloss += p.l_w * 0.5f * nsq;
                        if (sub == 0) {
                            p.s[b] = sb;
                            myloss += el_softplus_s(-dc) + p.l_b * 0.5f * beta_i * beta_i + (p.l_b * 0.5f * beta_j * beta_j) / 10.0f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) acc[q][x] += sb * (rgi[t][q][x] - rgj[t][q][x]);
                    cnt++;
                }
            }
            el_wave_lds_sync();
        }
        flush(p1 == p.n || (int64_t)p.keys[p1] != cur);
    }
    __shared__ float wsum[4];
    float wl = el_group_sum(myloss, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = wl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0) atomicAdd(p.loss_out, tot);
    }
}

// ---- user segments fused with the optimiser pass (el_bprmf_state.Gu_next) -------------------------------------------------------
// rowptr[r] = first sorted position whose user key is >= r (rowptr[U] = B): one thread per position fills the rows of the gap
// in front of it.
__global__ __launch_bounds__(256) void k_bpr_rowptr(const u32* __restrict__ keys, int64_t B, int64_t U, int32_t* __restrict__ rowptr) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t > B) return;
    const int64_t key = t < B ? (int64_t)keys[t] : U;
    const int64_t prev = t > 0 ? (int64_t)keys[t - 1] : -1;
    for (int64_t r = prev + 1; r <= key; ++r) rowptr[r] = (int32_t)t;
}


// One lane group (lpt lanes x 16 B = a row) owns RPG consecutive user rows: it prefetches their theta / m / v, walks the sorted
// positions of those rows (a contiguous range, rowptr) exactly as k_bpr_user_seg does -- index chains staged in LDS, SUB
// triplets' row gathers in flight -- keeping the RPG gradient rows in registers (the row a triplet belongs to is picked by
// selects, not branches: two groups share a wave), then takes Keras' Adam step on every one of its rows (g = 0 for a row without
// triplets: m, v decay, theta moves -- the all-rows semantics of k_adam_rows) and stores theta to Gu_new, m and v in place.
// Same operations in the same order as k_bpr_user_seg + k_adam_rows: the tables come out bit-identical.
template <int CPL, int RPG>
__global__ __launch_bounds__(256) void k_bpr_user_adam(SegParams p, FusedParams f) {
    constexpr int VW = 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int F = p.st.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    const int64_t rowbase = grp * RPG;
    float myloss = 0.f;
    if (rowbase < p.st.U) {
        const int nrows = (int)((p.st.U - rowbase < RPG) ? p.st.U - rowbase : RPG);
        const int64_t p0 = f.rowptr[rowbase], p1 = f.rowptr[rowbase + nrows];
        float th[RPG][CPL][VW], mm[RPG][CPL][VW], vv[RPG][CPL][VW], acc[RPG][CPL][VW];
        int cnt[RPG];
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            cnt[r] = 0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
#pragma unroll
                for (int x = 0; x < VW; ++x) th[r][q][x] = mm[r][q][x] = vv[r][q][x] = acc[r][q][x] = 0.f;
                if (r < nrows && e < F) {
                    const int64_t o = (rowbase + r) * F + e;
                    ldv<VW>(p.st.Gu + o, th[r][q]);
                    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.st.mGu + o));
                    const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p.st.vGu + o));
#pragma unroll
                    for (int x = 0; x < VW; ++x) mm[r][q][x] = a[x], vv[r][q][x] = b[x];
                }
            }
        }
        extern __shared__ unsigned char seg_lds[];
        const int gl = (int)(threadIdx.x / lpt), ngl = 256 / lpt;
        u32* s_key = reinterpret_cast<u32*>(seg_lds) + (0 * ngl + gl) * BPR_USTG;
        u32* s_b = reinterpret_cast<u32*>(seg_lds) + (1 * ngl + gl) * BPR_USTG;
        u32* s_i = reinterpret_cast<u32*>(seg_lds) + (2 * ngl + gl) * BPR_USTG;
        u32* s_j = reinterpret_cast<u32*>(seg_lds) + (3 * ngl + gl) * BPR_USTG;
        float* s_bi = reinterpret_cast<float*>(seg_lds) + (4 * ngl + gl) * BPR_USTG;
        float* s_bj = reinterpret_cast<float*>(seg_lds) + (5 * ngl + gl) * BPR_USTG;
        constexpr int SUB = (CPL == 1) ? 2 : 1;
        for (int64_t sbase = p0; sbase < p1; sbase += BPR_USTG) {
            const int cs = (int)((p1 - sbase < BPR_USTG) ? p1 - sbase : BPR_USTG);
            for (int t = sub; t < cs; t += lpt) {
                const u32 b = p.vals[sbase + t];
                const int32_t ii = p.bi[b], jj = p.bj[b];
                s_key[t] = p.keys[sbase + t];
                s_b[t] = b;
                s_i[t] = (u32)ii;
                s_j[t] = (u32)jj;
                s_bi[t] = p.st.Bi[ii];
                s_bj[t] = p.st.Bi[jj];
            }
            el_wave_lds_sync();
            for (int base = 0; base < cs; base += SUB) {
                bool okv[SUB];
                float rgi[SUB][CPL][VW], rgj[SUB][CPL][VW];
#pragma unroll
                for (int t = 0; t < SUB; ++t) {
                    okv[t] = base + t < cs;
                    const int tt = okv[t] ? base + t : base;
                    const float* pi = p.st.Gi + (int64_t)s_i[tt] * F;
                    const float* pj = p.st.Gi + (int64_t)s_j[tt] * F;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        const int e = (sub + q * lpt) * VW;
#pragma unroll
                        for (int x = 0; x < VW; ++x) rgi[t][q][x] = rgj[t][q][x] = 0.f;
                        if (okv[t] && e < F) {
                            ldv<VW>(pi + e, rgi[t][q]);
                            ldv<VW>(pj + e, rgj[t][q]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < SUB; ++t) {
                    if (!okv[t]) continue;                       // group-uniform
                    const int ridx = (int)((int64_t)s_key[base + t] - rowbase);
                    const int64_t b = (int64_t)s_b[base + t];
                    float gu[CPL][VW];                              // gamma_u = the pre-update row of this triplet's user
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            float g = th[0][q][x];
#pragma unroll
                            for (int r = 1; r < RPG; ++r) g = (ridx == r) ? th[r][q][x] : g;
                            gu[q][x] = g;
                        }
                    float dpi = 0.f, dpj = 0.f, nsq = 0.f;
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            dpi += gu[q][x] * rgi[t][q][x];
                            dpj += gu[q][x] * rgj[t][q][x];
                            nsq += gu[q][x] * gu[q][x] + rgi[t][q][x] * rgi[t][q][x] + rgj[t][q][x] * rgj[t][q][x];
                        }
                    dpi = el_group_sum(dpi, lpt);
                    dpj = el_group_sum(dpj, lpt);
                    const float beta_i = s_bi[base + t], beta_j = s_bj[base + t];
                    const float d = (beta_i + dpi) - (beta_j + dpj);   // x_ui - x_uj  (BPRMF_batch_model.py:53,65)
                    const float dc = fminf(fmaxf(d, -80.0f), 1e8f);
                    float sb = 0.f;
                    if (d >= -80.0f) sb = -1.0f / (1.0f + expf(d));
                    myloss += p.l_w * 0.5f * nsq;
                    if (sub == 0) {
                        p.s[b] = sb;
                        myloss += el_softplus_s(-dc) + p.l_b * 0.5f * beta_i * beta_i + (p.l_b * 0.5f * beta_j * beta_j) / 10.0f;
                    }
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) {
                            const float c = sb * (rgi[t][q][x] - rgj[t][q][x]);
#pragma unroll
                            for (int r = 0; r < RPG; ++r) acc[r][q][x] += (ridx == r) ? c : 0.f;
                        }
#pragma unroll
                    for (int r = 0; r < RPG; ++r) cnt[r] += (ridx == r) ? 1 : 0;
                }
            }
            el_wave_lds_sync();
        }
        const float omb1 = 1.0f - f.b1, omb2 = 1.0f - f.b2;
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            if (r >= nrows) continue;
            const float w = (float)cnt[r] * p.l_w;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
                if (e < F) {
                    f4 a, m4, v4;
#pragma unroll
                    for (int x = 0; x < VW; ++x) {
                        const float g = cnt[r] ? acc[r][q][x] + w * th[r][q][x] : 0.f;
                        float tx = th[r][q][x], mx = mm[r][q][x], vx = vv[r][q][x];
                        el_adam_elem(tx, mx, vx, g, f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
                        a[x] = tx, m4[x] = mx, v4[x] = vx;
                    }
                    const int64_t o = (rowbase + r) * F + e;
                    __builtin_nontemporal_store(a, reinterpret_cast<f4*>(f.Gu_new + o));
                    __builtin_nontemporal_store(m4, reinterpret_cast<f4*>(p.st.mGu + o));
                    __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(p.st.vGu + o));
                }
            }
        }
    }
    __shared__ float wsum[4];
    float wl = el_group_sum(myloss, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = wl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0) atomicAdd(p.loss_out, tot);
    }
}

// One waiting row brought forward ns steps by a whole wave: elements e = (lane + 64 q) VW, all 64 lanes on the same gap (the item
// catch-up kernel: one wave per sorted position, the wave on a segment head owns the row; the flush walks for rows wider than one
// pass of the wave).  m = v = 0 (a row that never had a gradient) is a fixed point of the gradient-free step: nothing to do.
template <int VW, bool SER>
__device__ __forceinline__ void bpr_replay_row(float* __restrict__ tth, float* __restrict__ tm, float* __restrict__ tv, int F, int64_t row, int lane,
                                               int last, int ns, const float* __restrict__ hist, int hist_mask) {
    auto lrs = [&](int s) { return hist[(last + 1 + s) & hist_mask]; };
    el_series sr;
    if (SER) sr = el_adam_series_sums(ns, lrs);                  // (wave-uniform: once per row)
    for (int f0 = 0; f0 < F; f0 += 64 * VW) {
        const int e = f0 + lane * VW;
        float th[VW], mm[VW], vv[VW];
#pragma unroll
        for (int x = 0; x < VW; ++x) th[x] = mm[x] = vv[x] = 0.f;
        if (e < F) {
            ldv<VW>(tth + row * F + e, th);
            ldv<VW>(tm + row * F + e, mm);
            ldv<VW>(tv + row * F + e, vv);
        }
        bool nz = false;
#pragma unroll
        for (int x = 0; x < VW; ++x) nz = nz || mm[x] != 0.f || vv[x] != 0.f;
        if (__ballot(nz) == 0ull) continue;                     // this chunk of the row is at its fixed point
        if (SER) el_adam_series_apply<VW>(th, mm, vv, sr);
        else el_adam_replay<VW>(th, mm, vv, ns, lrs);
        if (e < F) {
            stv<VW>(tth + row * F + e, th);
            stv<VW>(tm + row * F + e, mm);
            stv<VW>(tv + row * F + e, vv);
        }
    }
}

// Flush walk shared by k_bpr_flush_users / k_bpr_flush_items (round 5).  The round-4 kernels gave a wave one row at a time: its stamp
// (a 4-byte load), then -- dependent on it -- the three rows, the replay, three stores: two memory round trips per 1.5 KB with one row
// in flight per wave (k_bpr_flush_items: 3.2 TB/s with three waves in four leaving right after the stamp).  Here a wave owns 64
// consecutive rows: their stamps in ONE coalesced load (lane = row), the pending rows by ballot, then the pending rows NR at a time --
// their theta / m / v in flight together -- each replayed for its own gap exactly as bpr_replay_row does (same element order, same
// arithmetic: the bit-for-bit tests of the deferred decay cover it).  Rows wider than one pass of the wave (F > 64 VW) keep the old walk.
template <int VW, int NR, bool SER>
__device__ __forceinline__ void bpr_flush_rows64(float* __restrict__ tth, float* __restrict__ tm, float* __restrict__ tv, int32_t* __restrict__ last_arr,
                                                 int F, int64_t n_rows, int64_t row0, int lane, int32_t t, const float* __restrict__ hist, int hist_mask) {
    const int64_t myrow = row0 + lane;
    const int mylast = myrow < n_rows ? last_arr[myrow] : t;
    unsigned long long pend = __ballot(t - mylast > 0);
    if (pend == 0ull) return;
    if (F > 64 * VW) {                                          // several passes per row: one row at a time
        while (pend) {
            const int l = __builtin_ctzll(pend);
            pend &= pend - 1;
            const int last = __shfl(mylast, l, 64);
            bpr_replay_row<VW, SER>(tth, tm, tv, F, row0 + l, lane, last, t - last, hist, hist_mask);
        }
        if (t - mylast > 0) last_arr[myrow] = t;
        return;
    }
    const int e = lane * VW;
    const bool live = e < F;
    while (pend) {
        int ls[NR];
        bool on[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            on[r] = pend != 0ull;
            ls[r] = on[r] ? __builtin_ctzll(pend) : ls[r > 0 ? r - 1 : 0];
            if (r == 0 && !on[r]) ls[r] = 0;
            if (on[r]) pend &= pend - 1;
        }
        float th[NR][VW], mm[NR][VW], vv[NR][VW];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int x = 0; x < VW; ++x) th[r][x] = mm[r][x] = vv[r][x] = 0.f;
            if (live && on[r]) {
                const int64_t row = row0 + ls[r];
                ldv<VW>(tth + row * F + e, th[r]);
                ldv<VW>(tm + row * F + e, mm[r]);
                ldv<VW>(tv + row * F + e, vv[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (!on[r]) continue;                              // (wave-uniform)
            const int lastr = __shfl(mylast, ls[r], 64);
            bool nz = false;
#pragma unroll
            for (int x = 0; x < VW; ++x) nz = nz || mm[r][x] != 0.f || vv[r][x] != 0.f;
            if (__ballot(nz) == 0ull) continue;                // (m = v = 0: the fixed point of the step, nothing to replay or write)
            el_adam_catchup<VW, SER>(th[r], mm[r], vv[r], t - lastr, [&](int s2) { return hist[(lastr + 1 + s2) & hist_mask]; });
            if (live) {
                const int64_t row = row0 + ls[r];
                stv<VW>(tth + row * F + e, th[r]);
                stv<VW>(tm + row * F + e, mm[r]);
                stv<VW>(tv + row * F + e, vv[r]);
            }
        }
    }
    if (t - mylast > 0) last_arr[myrow] = t;
}

// deferred decay: every user row up to step t (a wave per 64 consecutive rows, grid-stride)
template <int VW, int NR, bool SER>
__global__ __launch_bounds__(256) void k_bpr_flush_users(el_bprmf_state st, int32_t t, const float* __restrict__ hist, int hist_mask) {
    const int lane = threadIdx.x & 63;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; row0 < st.U; row0 += (int64_t)gridDim.x * 256)
        bpr_flush_rows64<VW, NR, SER>(st.Gu, st.mGu, st.vGu, st.Gu_last, st.F, st.U, row0, lane, t, hist, hist_mask);
}

// ---- fused item side (el_bprmf_state.Gi_last): replay kernels of the item table --------------------------------------------------
// The item rows are the user rows' case again (k_bpr_catchup / k_bpr_flush_users) plus one bias element per row.  A wave replays
// a row's factors; the biases are replayed by their own small kernel, one LANE per row (64 rows' gaps per wave instead of one
// whole wave walking a single element), launched BEFORE the row kernel, which is the one that advances Gi_last.
template <bool SER>
__device__ __forceinline__ void bpr_replay_bias(const el_bprmf_state& st, int64_t row, int last, int ns, const float* __restrict__ hist,
                                                int hist_mask) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    float th = st.Bi[row], mm = st.mBi[row], vv = st.vBi[row];
    if (mm == 0.f && vv == 0.f) return;                         // fixed point of the gradient-free step
    if (SER) {
        float t1[1] = {th}, m1[1] = {mm}, v1[1] = {vv};
        el_adam_series_apply<1>(t1, m1, v1, el_adam_series_sums(ns, [&](int s) { return hist[(last + 1 + s) & hist_mask]; }));
        th = t1[0], mm = m1[0], vv = v1[0];
    } else {
        for (int s = 0; s < ns; ++s) el_adam_elem(th, mm, vv, 0.0f, hist[(last + 1 + s) & hist_mask], b1, b2, omb1, omb2, eps);
    }
    st.Bi[row] = th, st.mBi[row] = mm, st.vBi[row] = vv;
}

// start of step t, deferred item decay: biases of the batch's distinct items to step t - 1 (one lane per sorted item position;
// heads work)
template <bool SER>
__global__ __launch_bounds__(256) void k_bpr_catchup_ibias(el_bprmf_state st, const u32* __restrict__ keys, u32 key_off, int64_t n, int32_t t,
                                                           const float* __restrict__ hist, int hist_mask) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const u32 key = keys[p];
    if (p > 0 && keys[p - 1] == key) return;
    const int64_t row = (int64_t)(key - key_off);
    const int last = st.Gi_last[row];
    const int ns = (t - 1) - last;
    if (ns > 0) bpr_replay_bias<SER>(st, row, last, ns, hist, hist_mask);
}

template <int VW, bool SER>
__global__ __launch_bounds__(256) void k_bpr_catchup_items(el_bprmf_state st, const u32* __restrict__ keys, u32 key_off, int64_t n, int32_t t,
                                                           const float* __restrict__ hist, int hist_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n) return;
    const u32 key = keys[p];
    if (p > 0 && keys[p - 1] == key) return;                   // not a segment head
    const int64_t row = (int64_t)(key - key_off);
    const int last = st.Gi_last[row];
    const int ns = (t - 1) - last;
    if (ns <= 0) return;
    bpr_replay_row<VW, SER>(st.Gi, st.mGi, st.vGi, st.F, row, lane, last, ns, hist, hist_mask);
    if (lane == 0) st.Gi_last[row] = t - 1;
}

// every item row (bias: one lane per row; factors: one wave per row, grid-stride) up to step t (lr_t of step t is in the ring: the
// fused item-segment kernel of that step put it there)
template <bool SER>
__global__ __launch_bounds__(256) void k_bpr_flush_ibias(el_bprmf_state st, int32_t t, const float* __restrict__ hist, int hist_mask) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= st.I) return;
    const int last = st.Gi_last[row];
    const int ns = t - last;
    if (ns > 0) bpr_replay_bias<SER>(st, row, last, ns, hist, hist_mask);
}

template <int VW, int NR, bool SER>
__global__ __launch_bounds__(256) void k_bpr_flush_items(el_bprmf_state st, int32_t t, const float* __restrict__ hist, int hist_mask) {
    const int lane = threadIdx.x & 63;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; row0 < st.I; row0 += (int64_t)gridDim.x * 256)
        bpr_flush_rows64<VW, NR, SER>(st.Gi, st.mGi, st.vGi, st.Gi_last, st.F, st.I, row0, lane, t, hist, hist_mask);
}

// ---- item segments -----------------------------------------------------------------------
// IFUSE (el_bprmf_state.Gi_last): the step's whole item side on the batch's rows.  Where the two-pass form writes the gradient row
// of a segment into gGi, this form takes Keras' Adam step on the item row right away -- theta (it is the L2 term's operand
// anyway), m, v and the bias with its slots, IN PLACE -- and stamps Gi_last[item] = t.  The three rows of a segment head are
// fetched together with the gamma_u gathers of its SUB-batch, so a walk over cold items (a new segment at nearly every position)
// keeps as many loads in flight as the two-pass walk did.
// Both forms: a segment cut by a chunk boundary (the popular items of a Zipf catalogue) leaves its partial rows in the step's partial
// buffer -- slot 2 g for the segment that reaches INTO lane group g's chunk from the left, slot 2 g + 1 for the one that starts
// inside it and runs on -- and the group that holds the segment's first position puts (item, g) on the step's split list;
// k_bpr_item_combine -- the next launch -- adds a listed item's partials in chunk order and takes the step (fused form) or stores the
// gradient row (two-pass form).  (No in-kernel "last partial" hand-over: an agent-scope fence on this part writes back the
// XCD's L2 -- measured: 0.25 -> 0.92 ms for the item segments at configs[1] with one fence pair per chunk.)
// (The user-side kernel of the step ran before: nothing else reads an item row here.)
struct ItemFuse {
    int32_t* last;       // [I]
    int32_t* split;      // [0] = number of listed rows (zeroed before the launch), [1 + 2 e], [2 + 2 e] = row and head group of entry e
    int32_t* split_long; // the same layout: listed rows with a whole lane group's width of continuations or more (k_bpr_item_combine fills it)
    float* part;         // [2 groups, F] partial rows of cut segments; part_b [2 groups] their bias parts
    float* part_b;
    float* hist;         // lr ring: this kernel records lr_t of step t for the replays that follow
    int hist_mask;
    float lr_t, b1, b2, eps;
    int32_t t;
};

template <int VW, int CPL, bool IFUSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_bpr_item_seg(SegParams p, ItemFuse f) {
    const int F = p.st.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    if (IFUSE && gid == 0) f.hist[f.t & f.hist_mask] = f.lr_t;
    const int64_t p0 = grp * p.chunk;
    if (p0 >= p.n) return;
    const int64_t p1 = (p0 + p.chunk < p.n) ? p0 + p.chunk : p.n;
    int64_t cur = -1;
    bool started_inside = false;
    float acc[CPL][VW];
    float rrow[IFUSE ? CPL : 1][VW], mrow[IFUSE ? CPL : 1][VW], vrow[IFUSE ? CPL : 1][VW];
    float bacc = 0.f, bacc2 = 0.f;
    int cpos = 0, cneg = 0;
    auto flush = [&](bool ends_inside) {
        const float* pr = p.st.Gi + cur * F;
        // CML: d/di of +-|u - i|^2 adds -(sum of the coefficients) times the row itself
        const float w = (float)(cpos + cneg) * p.l_w - (p.cml ? bacc : 0.f);
        const bool plain = started_inside && ends_inside;
        if (IFUSE && plain) {
            const float omb1 = 1.0f - f.b1, omb2 = 1.0f - f.b2;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
                if (e < F) {
                    float tn[VW], mn[VW], vn[VW];
#pragma unroll
                    for (int x = 0; x < VW; ++x) {
                        const float gg = acc[q][x] + w * rrow[IFUSE ? q : 0][x];
                        tn[x] = rrow[IFUSE ? q : 0][x], mn[x] = mrow[IFUSE ? q : 0][x], vn[x] = vrow[IFUSE ? q : 0][x];
                        el_adam_elem(tn[x], mn[x], vn[x], gg, f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
                    }
                    stv<VW>(p.st.Gi + cur * F + e, tn);
                    stv<VW>(p.st.mGi + cur * F + e, mn);
                    stv<VW>(p.st.vGi + cur * F + e, vn);
                }
            }
            if (sub == 0) {
                float beta = p.st.Bi[cur], mb = p.st.mBi[cur], vb = p.st.vBi[cur];
                const float gb = bacc + p.l_b * (float)cpos * beta + (p.l_b / 10.0f) * (float)cneg * beta;
                el_adam_elem(beta, mb, vb, gb, f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
                p.st.Bi[cur] = beta, p.st.mBi[cur] = mb, p.st.vBi[cur] = vb;
                f.last[cur] = f.t;
            }
            return;
        }
        // the gradient row of this walk: the whole segment's (plain, two-pass form: stored) or a partial of a cut segment
        float* g = plain ? p.st.gGi + cur * F : f.part + (2 * grp + (started_inside ? 1 : 0)) * (int64_t)F;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
                float r[VW], v[VW];
                if (IFUSE) {
#pragma unroll
                    for (int x = 0; x < VW; ++x) r[x] = rrow[IFUSE ? q : 0][x];
                } else {
                    ldv<VW>(pr + e, r);
                }
#pragma unroll
                for (int x = 0; x < VW; ++x) v[x] = acc[q][x] + w * r[x];
                stv<VW>(g + e, v);
            }
        }
        if (sub == 0) {
            const float beta = p.st.Bi[cur];
            const float gb = (p.cml ? bacc2 : bacc) + p.l_b * (float)cpos * beta + (p.l_b / 10.0f) * (float)cneg * beta;
            if (plain)
                p.st.gBi[cur] = gb;
            else
                f.part_b[2 * grp + (started_inside ? 1 : 0)] = gb;
            if (p.st.tGi) {
                p.st.tGi[cur] = p.step;
                p.st.tBi[cur] = p.step;
            }
            // (the head of a cut segment is not listed: k_bpr_item_combine finds it from the keys -- tens of thousands of appends to
            //  one counter cost more than three key reads per lane group)
        }
    };
    // staged index chains, as in k_bpr_user_seg: BPR_ISTG positions per stage (key, payload -> s_b, u_b)
    extern __shared__ unsigned char seg_lds[];
    const int gl = (int)(threadIdx.x / lpt), ngl = 256 / lpt;
    u32* s_key = reinterpret_cast<u32*>(seg_lds) + (0 * ngl + gl) * BPR_ISTG;
    u32* s_u = reinterpret_cast<u32*>(seg_lds) + (1 * ngl + gl) * BPR_ISTG;
    float* s_cf = reinterpret_cast<float*>(seg_lds) + (2 * ngl + gl) * BPR_ISTG;      // +s_b (positive item) / -s_b (negative)
    float* s_cf2 = reinterpret_cast<float*>(seg_lds) + (3 * ngl + gl) * BPR_ISTG;     // CML: +-dloss/dE_b
    // positions whose loads are in flight together (IFUSE: each may bring the three rows of a segment head along)
    constexpr int SUB = IFUSE ? (CPL == 1 ? 2 : 1) : ((CPL == 1) ? 4 : (CPL == 2 ? 2 : 1));
    for (int64_t sbase = p0; sbase < p1; sbase += BPR_ISTG) {
        const int cs = (int)((p1 - sbase < BPR_ISTG) ? p1 - sbase : BPR_ISTG);
        for (int t = sub; t < cs; t += lpt) {
            const u32 pay = p.vals[sbase + t];
            const int64_t b = (int64_t)(pay & 0x7fffffffu);
            const float sb = p.s[b];
            s_key[t] = (p.keys[sbase + t] - p.key_off) | (pay & 0x80000000u);   // item ids < 2^31: the top bit carries the role
            s_u[t] = (u32)(p.uidx ? p.uidx[b] : p.bu[b]);
            s_cf[t] = (pay >> 31) ? -sb : sb;
            if (p.cml) {
                const float e2 = p.s2[b];
                s_cf[t] = (pay >> 31) ? -2.0f * sb : 2.0f * sb;      // dD/di = 2 (u - i), dD/dj = -2 (u - j)
                s_cf2[t] = (pay >> 31) ? -e2 : e2;
            }
        }
        el_wave_lds_sync();
        for (int base = 0; base < cs; base += SUB) {
            int64_t keyv[SUB];
            float cfv[SUB];
            bool negv[SUB], okv[SUB];
            float rr[SUB][CPL][VW];
            float hr[IFUSE ? SUB : 1][CPL][VW], hm[IFUSE ? SUB : 1][CPL][VW], hv[IFUSE ? SUB : 1][CPL][VW];   // rows of the segment heads
#pragma unroll
            for (int t = 0; t < SUB; ++t) {
                okv[t] = base + t < cs;
                const int tt = okv[t] ? base + t : base;
                const u32 kk = s_key[tt];
                keyv[t] = (int64_t)(kk & 0x7fffffffu);
                negv[t] = (kk >> 31) != 0u;
                cfv[t] = s_cf[tt];
                const float* pu = (p.ubase ? p.ubase : p.st.Gu) + (int64_t)s_u[tt] * F;
                const bool headt = IFUSE && okv[t] && keyv[t] != (t == 0 ? cur : keyv[t > 0 ? t - 1 : 0]);
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * VW;
#pragma unroll
                    for (int x = 0; x < VW; ++x) rr[t][q][x] = 0.f;
                    if (okv[t] && e < F) ldv<VW>(pu + e, rr[t][q]);
                    if (IFUSE) {
#pragma unroll
                        for (int x = 0; x < VW; ++x) hr[IFUSE ? t : 0][q][x] = hm[IFUSE ? t : 0][q][x] = hv[IFUSE ? t : 0][q][x] = 0.f;
                        if (headt && e < F) {
                            ldv<VW>(p.st.Gi + keyv[t] * F + e, hr[IFUSE ? t : 0][q]);
                            ldv<VW>(p.st.mGi + keyv[t] * F + e, hm[IFUSE ? t : 0][q]);
                            ldv<VW>(p.st.vGi + keyv[t] * F + e, hv[IFUSE ? t : 0][q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < SUB; ++t) {
                if (!okv[t]) continue;
                const int64_t pos = sbase + base + t, key = keyv[t];
                if (key != cur) {
                    if (cur >= 0) flush(true);
                    cur = key;
                    started_inside = (pos > p0) || (pos == 0) || ((int64_t)(p.keys[pos - 1] - p.key_off) != key);
                    cpos = cneg = 0;
                    bacc = bacc2 = 0.f;
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
                    if (IFUSE) {                                 // the row and its Adam slots (fetched with the SUB-batch above)
#pragma unroll
                        for (int q = 0; q < CPL; ++q)
#pragma unroll
                            for (int x = 0; x < VW; ++x) {
                                rrow[IFUSE ? q : 0][x] = hr[IFUSE ? t : 0][q][x];
                                mrow[IFUSE ? q : 0][x] = hm[IFUSE ? t : 0][q][x];
                                vrow[IFUSE ? q : 0][x] = hv[IFUSE ? t : 0][q][x];
                            }
                    }
                }
                const float coef = cfv[t];
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) acc[q][x] += coef * rr[t][q][x];
                bacc += coef;
                if (p.cml) bacc2 += s_cf2[base + t];
                if (negv[t])
                    cneg++;
                else
                    cpos++;
            }
        }
        el_wave_lds_sync();
    }
    flush(p1 == p.n || (int64_t)(p.keys[p1] - p.key_off) != cur);
}

// ---- the rows on the split list (segments cut by chunk boundaries) ---------------------------------------------------------------------
// A listed item's partial rows are the head's (slot 2 g0 + 1) and one per following lane group whose chunk still begins inside the segment
// (slot 2 g, g = g0 + 1 ...: the sorted key at that chunk's first position is still this item).  They are added in a FIXED order -- no
// atomics, the same bits on every run and in both forms -- and then: fused form -- Keras' Adam step on the row in place, Gi_last stamped;
// two-pass form -- the gradient row stored into gGi / gBi for the dense pass.
//   k_bpr_item_combine       one LANE GROUP per listed row (most rows are cut once: two partials); lane k looks at lane group g0 + 1 + k:
//                            up to lpt - 1 continuations are summed here in ascending order, longer segments go on the long list
//   k_bpr_item_combine_long  one WORKGROUP per row of the long list (the popular items of a Zipf catalogue: ~1 100 partials for the
//                            hottest at B = 2^20): each lane group adds a contiguous share in ascending order, eight loads in flight, the
//                            shares meet in LDS in lane-group order
template <int VW, int CPL, bool IFUSE>
__device__ __forceinline__ void item_row_finish(const el_bprmf_state& st, const ItemFuse& f, int64_t row, int sub, int lpt, const float (&gg)[CPL][VW],
                                                float gb) {
    const int F = st.F;
    const float omb1 = 1.0f - f.b1, omb2 = 1.0f - f.b2;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
        if (e >= F) continue;
        if (IFUSE) {
            float th[VW], mm[VW], vv[VW];
            ldv<VW>(st.Gi + row * F + e, th);
            ldv<VW>(st.mGi + row * F + e, mm);
            ldv<VW>(st.vGi + row * F + e, vv);
#pragma unroll
            for (int x = 0; x < VW; ++x) el_adam_elem(th[x], mm[x], vv[x], gg[q][x], f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
            stv<VW>(st.Gi + row * F + e, th);
            stv<VW>(st.mGi + row * F + e, mm);
            stv<VW>(st.vGi + row * F + e, vv);
        } else {
            stv<VW>(st.gGi + row * F + e, gg[q]);
        }
    }
    if (sub == 0) {
        if (IFUSE) {
            float beta = st.Bi[row], mb = st.mBi[row], vb2 = st.vBi[row];
            el_adam_elem(beta, mb, vb2, gb, f.lr_t, f.b1, f.b2, omb1, omb2, f.eps);
            st.Bi[row] = beta, st.mBi[row] = mb, st.vBi[row] = vb2;
            f.last[row] = f.t;
        } else {
            st.gBi[row] = gb;
        }
    }
}

template <int VW, int CPL, bool IFUSE>
__global__ __launch_bounds__(256) void k_bpr_item_combine(el_bprmf_state st, ItemFuse f, const u32* __restrict__ keys, u32 key_off, int64_t n,
                                                          int chunk, int lpt) {
    const int F = st.F;
    const int64_t ent = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    // lane group g0 looks at ITS chunk: the chunk's last segment is the head of a cut row when it goes on into chunk g0 + 1 and started
    // inside chunk g0
    const int64_t g0 = ent;
    const int64_t p0 = g0 * (int64_t)chunk, p1 = p0 + chunk;
    if (p1 >= n) return;                                        // (the last chunk's last segment ends with the data)
    const u32 kl = keys[p1 - 1];
    if (keys[p1] != kl) return;                                 // ends inside
    if (keys[p0] == kl && p0 > 0 && keys[p0 - 1] == kl) return; // a middle piece: spans the whole chunk, started before it
    const int64_t row = (int64_t)(kl - key_off);
    const int64_t pos = (g0 + 1 + sub) * (int64_t)chunk;
    const bool cont = pos < n && (int64_t)(keys[pos] - key_off) == row;
    const unsigned long long bal = __ballot(cont);
    const int lane0 = (int)(threadIdx.x & 63) & ~(lpt - 1);
    const unsigned long long mine = lpt >= 64 ? bal : ((bal >> lane0) & ((1ull << lpt) - 1ull));
    const unsigned long long full = lpt >= 64 ? ~0ull : ((1ull << lpt) - 1ull);
    if (mine == full) {                                         // lpt or more continuations: the workgroup kernel's
        if (sub == 0) {
            const int e2 = atomicAdd(f.split_long, 1);
            f.split_long[1 + 2 * e2] = (int32_t)row;
            f.split_long[2 + 2 * e2] = (int32_t)g0;
        }
        return;
    }
    const int ncont = __builtin_ctzll(~mine);                   // consecutive continuations
    float acc[CPL][VW];
    float accb = f.part_b[2 * g0 + 1];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
#pragma unroll
        for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
        if (e < F) ldv<VW>(f.part + (2 * g0 + 1) * F + e, acc[q]);
    }
    for (int k = 1; k <= ncont; k += 4) {
        float v[4][CPL][VW], vb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t sl = 2 * (g0 + (k + t <= ncont ? k + t : k));
            vb[t] = f.part_b[sl];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
#pragma unroll
                for (int x = 0; x < VW; ++x) v[t][q][x] = 0.f;
                if (e < F) ldv<VW>(f.part + sl * F + e, v[t][q]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (k + t > ncont) continue;
            accb += vb[t];
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int x = 0; x < VW; ++x) acc[q][x] += v[t][q][x];
        }
    }
    item_row_finish<VW, CPL, IFUSE>(st, f, row, sub, lpt, acc, accb);
}

template <int VW, int CPL, bool IFUSE>
__global__ __launch_bounds__(256) void k_bpr_item_combine_long(el_bprmf_state st, ItemFuse f, const u32* __restrict__ keys, u32 key_off, int64_t n,
                                                               int chunk, int lpt) {
    const int F = st.F;
    const int nlist = f.split_long[0];
    __shared__ int s_np;
    __shared__ float s_red[4096];                             // (256 / lpt) lane groups x F floats: lpt * VW * CPL >= F, CPL <= 4 -> <= 4096
    __shared__ float s_rb[32];
    const int ngl = 256 / lpt, gl = threadIdx.x / lpt, sub = threadIdx.x & (lpt - 1);
    for (int ent = blockIdx.x; ent < nlist; ent += gridDim.x) {
        const int64_t row = (int64_t)f.split_long[1 + 2 * ent];
        const int64_t g0 = (int64_t)f.split_long[2 + 2 * ent];
        // how many lane groups after g0 continue the segment (consecutive): the first chunk whose first key is another item ends it
        if (threadIdx.x == 0) s_np = 0x7fffffff;
        __syncthreads();
        for (int base = 0; s_np == 0x7fffffff; base += 256) {
            const int64_t gq = g0 + 1 + base + threadIdx.x, pos = gq * (int64_t)chunk;
            const bool cont = pos < n && (int64_t)(keys[pos] - key_off) == row;
            if (!cont) atomicMin(&s_np, base + (int)threadIdx.x);
            __syncthreads();
        }
        const int ncont = s_np;                                // continuation partials: slots 2 (g0 + k), k = 1 .. ncont
        const int np = 1 + ncont;                              // + the head's
        const int per = (np + ngl - 1) / ngl;
        const int k0 = gl * per, k1 = (k0 + per < np) ? k0 + per : np;
        float acc[CPL][VW];
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
        float accb = 0.f;
        auto slot_of = [&](int k) { return k == 0 ? 2 * g0 + 1 : 2 * (g0 + k); };
        for (int k = k0; k < k1; k += 8) {
            float v[8][CPL][VW];
            float vb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int kk = k + t < k1 ? k + t : k;
                const int64_t sl = slot_of(kk);
                vb[t] = f.part_b[sl];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * VW;
#pragma unroll
                    for (int x = 0; x < VW; ++x) v[t][q][x] = 0.f;
                    if (e < F) ldv<VW>(f.part + sl * F + e, v[t][q]);
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (k + t >= k1) continue;
                accb += vb[t];
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) acc[q][x] += v[t][q][x];
            }
        }
        // lane-group sums -> LDS -> added in lane-group order by group 0
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) stv<VW>(s_red + gl * F + e, acc[q]);
        }
        if (sub == 0) s_rb[gl] = accb;
        __syncthreads();
        if (gl == 0) {
            const int used = (np + per - 1) / per;              // lane groups that held partials
            float gb = 0.f;
            for (int h = 0; h < used; ++h) gb += s_rb[h];
            float gg[CPL][VW];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
#pragma unroll
                for (int x = 0; x < VW; ++x) gg[q][x] = 0.f;
                if (e >= F) continue;
                for (int h = 0; h < used; ++h) {
                    float t4[VW];
                    ldv<VW>(s_red + h * F + e, t4);
#pragma unroll
                    for (int x = 0; x < VW; ++x) gg[q][x] += t4[x];
                }
            }
            item_row_finish<VW, CPL, IFUSE>(st, f, row, sub, lpt, gg, gb);
        }
        __syncthreads();
    }
}

// ---- host ---------------------------------------------------------------------------------
// Positions per lane group.  Popular items (Zipf) own segments of tens of thousands of occurrences; every chunk that
// does not contain a whole segment ends with an atomic flush onto the same few cache lines, so long chunks matter for
// the item side (16 -> 128 positions: 0.84 -> 0.35 ms at B = 1M) as long as enough groups remain to fill the chip.
static int item_chunk_for(int64_t B, int64_t I) {
    if (g_el_cur_ctx && g_el_cur_ctx->opt.ichunk > 0)           // el_ctx_set_option("ichunk"): tests pin the summation order with it
        return g_el_cur_ctx->opt.ichunk < 16 ? 16 : (int)g_el_cur_ctx->opt.ichunk;     // (>= 16: the split list is sized for it)
    int64_t c = (2 * B) / 8192;                          // (round 3, with the fused user side: 256 at B = 2^20 -- 1.275 -> 1.25 ms per step;
    c = c < 16 ? 16 : (c > 256 ? 256 : c);               //  128: 0.259, 256: 0.247, 384: 0.256, 512: 0.277 ms for the item segments at 100 K items)
    // Round 5: that optimum belongs to LONG segments (100 K items under 2^21 positions: 20 per item).  With a catalogue the batch
    // barely covers -- 1 M items: 2.7 positions per distinct item, 5 M: 1.2 -- nearly every position starts a segment with three row
    // fetches of its own, and what counts is lane groups in flight, not rows reused inside a group: a quarter of the chunk.  Measured
    // at 10 M x 1 M x 128 (one box, ms per step / item segments): 16: 2.98 / 0.95, 32: 2.67 / 0.73, 48: 2.63 / 0.68, 64: 2.62-2.66 / 0.69,
    // 96: 2.63 / 0.70, 128: 2.65 / 0.71, 192: 2.80 / 0.81, 256: 2.75 / 0.73, 384: 2.90 / 0.88; at 6.25 M x 5 M x 256: 16: 5.67, 32: 5.80,
    // 64: 5.84, 256: 5.93.
    const double cover = I > 0 ? (double)I * (1.0 - exp(-2.0 * (double)B / (double)I)) : 1.0;     // expected distinct items of 2 B draws
    if (cover > 0.0 && 2.0 * (double)B / cover < 8.0) c = c / 4 < 16 ? 16 : c / 4;
    return (int)c;
}
static int user_chunk_for(int64_t B) {
    if (g_el_cur_ctx && g_el_cur_ctx->opt.uchunk > 0) return (int)g_el_cur_ctx->opt.uchunk;
    int64_t c = B / 65536;
    return (int)(c < 4 ? 4 : (c > 16 ? 16 : c));
}

static int bits_for(int64_t n) {
    int b = 1;
    while ((1LL << b) < n && b < 32) ++b;
    return b;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SortedWs {
    u32 *keyU_in, *valU_in, *keyU, *valU, *keyI_in, *valI_in, *keyI, *valI;
    float* s;
    void* tmp;
    size_t tmp_bytes;
    int32_t* rowptr;       // [U + 1] first sorted position of every user row (fused user-side kernel)
    int32_t* hpos;         // [B] per triplet: sorted head position of its user's segment (deferred decay)
    int32_t* split;        // (unused since the combine pass finds the cut rows from the keys; kept so that the workspace layout stays put)
    int32_t* split_long;   // the same, for the rows k_bpr_item_combine hands on to k_bpr_item_combine_long
    float* part;           // [2 (B / 8 + 1), F] + [2 (B / 8 + 1)]: their partial rows and bias parts (NULL when carved without F)
    float* part_b;
    size_t total;
};

// F = 0: the part of the layout the sort needs (el_bprmf_presort does not know the factor count); F > 0: + the partial rows of the cut
// item segments at the END (the arrays in front of them sit at the same offsets either way)
static int carve_ws(int64_t B, int64_t U, int64_t I, char* base, SortedWs* w, int F = 0) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    // users [0, B) and items [B, 3B) share one array each, so that one radix sort of 3B keys (users as u, items as
    // U + item) orders both sides: half the launches of two sorts, and these small sorts are launch-bound
    u32** slots[4] = {&w->keyU_in, &w->valU_in, &w->keyU, &w->valU};
    u32** islots[4] = {&w->keyI_in, &w->valI_in, &w->keyI, &w->valI};
    for (int k = 0; k < 4; ++k) {
        *slots[k] = (u32*)take((size_t)B * 12);
        *islots[k] = base ? *slots[k] + B : nullptr;
    }
    w->s = (float*)take((size_t)B * 4);
    size_t t1 = 0, t2 = 0;
    u32* np = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, t1, np, np, np, np, (unsigned)(3 * B), 0, bits_for(U + I), (hipStream_t)0) != hipSuccess) return 1;
    if (rocprim::radix_sort_pairs(nullptr, t2, np, np, np, np, (unsigned)(2 * B), 0, bits_for(I), (hipStream_t)0) != hipSuccess) return 1;
    w->tmp_bytes = t1 > t2 ? t1 : t2;
    w->tmp = take(w->tmp_bytes);
    w->rowptr = (int32_t*)take((size_t)(U + 1) * 4);
    w->hpos = (int32_t*)take((size_t)B * 4);
    w->split = (int32_t*)take((size_t)(2 * (B / 8 + 4) + 4) * 4);
    w->split_long = (int32_t*)take((size_t)(2 * (B / 8 + 4) + 4) * 4);
    w->part = w->part_b = nullptr;
    if (F > 0) {                                         // (lane groups: at most 2 B / 16 -- item_chunk_for never goes below 16 positions)
        w->part = (float*)take((size_t)2 * (B / 8 + 1) * (size_t)F * 4);
        w->part_b = (float*)take((size_t)2 * (B / 8 + 1) * 4);
    }
    w->total = off;
    return 0;
}

// (u, b), (U + i, b), (U + j, b | neg) -> one stable radix sort: the first B sorted entries are the user side, the next 2B
// the item side (every user key is below every item key)
// with_rowptr: also rowptr[U + 1] of the sorted user keys (the every-row fused user kernel walks rows, not positions) -- here, not
// in front of that kernel, so that a pipelined step prepares it on the side stream together with the sort
static int sort_batch(hipStream_t s, const SortedWs& w, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                      int64_t U, int64_t I, bool with_rowptr) {
    EL_REQUIRE(U + I < (1LL << 32) && 3 * B < (1LL << 32), "sorted gradient path: U + I and 3B must fit 32-bit sort keys");
    EL_LAUNCH("k_bpr_prep", k_bpr_prep, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, u, i, j, B, w.keyU_in, w.valU_in,
              w.keyI_in, w.valI_in, (u32)U);
    ElKernelTimer t("rocprim_radix_sort_pairs", s);
    size_t tb = w.tmp_bytes;
    EL_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.keyU_in, w.keyU, w.valU_in, w.valU, (unsigned)(3 * B), 0, bits_for(U + I), s));
    if (with_rowptr) EL_LAUNCH("k_bpr_rowptr", k_bpr_rowptr, dim3((unsigned)((B + 1 + 255) / 256)), dim3(256), 0, s, w.keyU, B, U, w.rowptr);
    return 0;
}

extern "C" size_t el_bprmf_ws_bytes(int64_t B, int64_t U, int64_t I, int32_t F) {
    if (B <= 0 || F <= 0) return 0;
    SortedWs w;
    if (carve_ws(B, U, I, nullptr, &w, F)) return 0;
    return w.total;
}

// 1: the sorted step sums every gradient row in a fixed order (no floating-point atomics on rows): same batches, same bits
extern "C" int el_bprmf_deterministic(void) { return 1; }

// defined in el_bpr.hip
int el_bprmf_apply_optimizer(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, const int32_t* u, const int32_t* i,
                             const int32_t* j, int64_t B, float lr, int opt, int32_t step, float lr_t);
int el_bprmf_check_state(const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j,
                         double* loss_out, int opt, int32_t step, bool* vec, bool* rows_mode);
int el_pick_lpt(int F, int vw, int* cpl);

int el_bprmf_apply_items_adam(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, float lr_t);      // el_bpr.hip

// user side of the step as ONE kernel (el_bprmf_state.Gu_next): rowptr, then segments + Adam over every user row
static int launch_flush_users(const el_bprmf_state& st, hipStream_t s, int32_t t) {
    EL_REQUIRE(st.F % 4 == 0 && st.F <= 512, "el_bprmf_sync_users: F=%d outside the deferred decay's range", st.F);
    int64_t grid = (st.U + 255) / 256;                       // a wave owns 64 consecutive rows, two pending rows in flight per wave
    if (grid > (1 << 18)) grid = 1 << 18;
    const int mask = st.lr_hist_cap - 1;
#define EL_FU(VW_, SER_) EL_LAUNCH("k_bpr_flush_users", (k_bpr_flush_users<VW_, 2, SER_>), dim3((unsigned)grid), dim3(256), 0, s, st, t, st.lr_hist, mask)
    if (st.replay_series) {
        if (st.F >= 256) EL_FU(4, true); else if (st.F >= 128) EL_FU(2, true); else EL_FU(1, true);
    } else {
        if (st.F >= 256) EL_FU(4, false); else if (st.F >= 128) EL_FU(2, false); else EL_FU(1, false);
    }
#undef EL_FU
    EL_CHECK_LAUNCH();
    return 0;
}

static int check_deferred(const el_bprmf_state& st) {
    EL_REQUIRE(st.Gu_last && st.lr_hist && st.lr_hist_cap >= 4 && (st.lr_hist_cap & (st.lr_hist_cap - 1)) == 0,
               "el_bprmf: deferred decay needs Gu_last, lr_hist and a power-of-two lr_hist_cap >= 4");
    EL_REQUIRE(st.mGu && st.vGu && st.F % 4 == 0 && (((uintptr_t)st.Gu | (uintptr_t)st.mGu | (uintptr_t)st.vGu) & 15) == 0,
               "el_bprmf: deferred decay needs Adam slots, F %% 4 == 0 and 16-byte aligned tables");
    return 0;
}

extern "C" int el_bprmf_sync_users(el_ctx* ctx, void* stream, const el_bprmf_state* stp, int32_t step) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp != nullptr && step >= 0, "el_bprmf_sync_users: bad arguments");
    if (stp->Gu_last == nullptr || step == 0) return 0;
    if (int rc = check_deferred(*stp)) return rc;
    return launch_flush_users(*stp, (hipStream_t)stream, step);
}

// ---- self-test of the packed replay arithmetic (el_common.h: el_pk_sqrt, el_pk_div, el_adam_replay2) ----------------------------
// out[0] = floats v in [2^-96, 2^96] (ALL of them: 1 610 612 736) with el_pk_sqrt(v) != sqrtf(v)
// out[1] = of n_pairs pseudo-random (num, den) pairs spanning the guard range -- half of them with neighbouring mantissas, the
//          near-tie quotients -- those with el_pk_div != num / den
// out[2] = of n_pairs (theta, m, v, lr) tuples inside the guard, those where EL_REPLAY_CHUNK packed steps differ from el_adam_elem
__global__ __launch_bounds__(256) void k_selftest_sqrt(u64* out) {
    const u32 lo = (127u - 96u) << 23, hi = (127u + 96u) << 23;
    u64 bad = 0;
    for (u64 b = lo + ((u64)blockIdx.x * 256 + threadIdx.x) * 2; b < hi; b += (u64)gridDim.x * 512) {
        el_f2 v;
        v.x = __uint_as_float((u32)b), v.y = __uint_as_float((u32)b + 1u);
        const el_f2 s = el_pk_sqrt(v);
        bad += (__float_as_uint(s.x) != __float_as_uint(sqrtf(v.x))) + (__float_as_uint(s.y) != __float_as_uint(sqrtf(v.y)));
    }
    if (bad) atomicAdd((unsigned long long*)out, (unsigned long long)bad);
}
__device__ __forceinline__ u64 selftest_mix(u64& s) {
    u64 z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_selftest_div(int iters, u64* out) {
    u64 s = 0x1234ull + ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E37ull, bad = 0;
    for (int it = 0; it < iters; ++it) {
        el_f2 a, b;
        for (int h = 0; h < 2; ++h) {
            const u64 r = selftest_mix(s);
            const u32 ma = (u32)r & 0x7fffffu;
            const u32 mb = (it & 1) ? ((ma + (u32)(r >> 60)) & 0x7fffffu) : ((u32)(r >> 23) & 0x7fffffu);
            const u32 ea = 127 - 90 + (u32)((r >> 46) % 120), eb = 127 - 24 + (u32)((r >> 53) % 54);     // |num| in [2^-90, 2^30), den in [2^-24, 2^30)
            const float av = __uint_as_float(((u32)(r >> 63) << 31) | (ea << 23) | ma), bv = __uint_as_float((eb << 23) | mb);
            if (h == 0) a.x = av, b.x = bv; else a.y = av, b.y = bv;
        }
        const el_f2 q = el_pk_div(a, b);
        bad += (__float_as_uint(q.x) != __float_as_uint(a.x / b.x)) + (__float_as_uint(q.y) != __float_as_uint(a.y / b.y));
    }
    if (bad) atomicAdd((unsigned long long*)(out + 1), (unsigned long long)bad);
}
__global__ __launch_bounds__(256) void k_selftest_step(int iters, u64* out) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    u64 s = 0x77ull + ((u64)blockIdx.x * 256 + threadIdx.x) * 0x51ull, bad = 0;
    for (int it = 0; it < iters; ++it) {
        float th[2], mm[2], vv[2];
        for (int h = 0; h < 2; ++h) {
            const u64 r = selftest_mix(s);
            th[h] = __uint_as_float(((u32)(r >> 62) << 31) | ((127u - 20u + (u32)(r % 24)) << 23) | ((u32)(r >> 8) & 0x7fffffu));
            mm[h] = __uint_as_float(((u32)(r >> 61) << 31) | ((127u - 58u + (u32)((r >> 32) % 88)) << 23) | ((u32)(r >> 9) & 0x7fffffu));
            vv[h] = __uint_as_float(((127u - 89u + (u32)((r >> 40) % 149)) << 23) | ((u32)(r >> 17) & 0x7fffffu));
        }
        const float lr = __uint_as_float(((127u - 30u + (u32)(selftest_mix(s) % 30)) << 23) | ((u32)s & 0x7fffffu));
        if (!(el_replay_ok(mm[0], vv[0]) && el_replay_ok(mm[1], vv[1]) && el_replay_lr_ok(lr))) continue;
        el_f2 T = {th[0], th[1]}, M = {mm[0], mm[1]}, V = {vv[0], vv[1]};
        // (the production form: the first step of a chunk seeds its square root with v_rsq, the following ones with the root of the
        //  step before -- el_adam_replay2s; the step size moves from step to step as it does in a run: lr_t of Adam's bias correction)
        el_f2 SG, SH;
        for (int k = 0; k < EL_REPLAY_CHUNK; ++k) {
            const float lrk = lr * (1.0f - 0.01f * (float)k);
            if (k == 0) el_adam_replay2s<true>(T, M, V, lrk, SG, SH);
            else el_adam_replay2s<false>(T, M, V, lrk, SG, SH);
            for (int h = 0; h < 2; ++h) el_adam_elem(th[h], mm[h], vv[h], 0.0f, lrk, b1, b2, omb1, omb2, eps);
        }
        bad += (__float_as_uint(T.x) != __float_as_uint(th[0])) + (__float_as_uint(T.y) != __float_as_uint(th[1])) +
               (__float_as_uint(M.x) != __float_as_uint(mm[0])) + (__float_as_uint(M.y) != __float_as_uint(mm[1])) +
               (__float_as_uint(V.x) != __float_as_uint(vv[0])) + (__float_as_uint(V.y) != __float_as_uint(vv[1])) +
               (__float_as_uint(SG.x) != __float_as_uint(sqrtf(vv[0]))) + (__float_as_uint(SG.y) != __float_as_uint(sqrtf(vv[1])));
    }
    if (bad) atomicAdd((unsigned long long*)(out + 2), (unsigned long long)bad);
}

extern "C" int el_selftest_replay_math(el_ctx* ctx, void* stream, int64_t n_pairs, uint64_t* out3) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(out3 != nullptr && n_pairs >= 0, "el_selftest_replay_math: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    EL_CHECK_HIP(hipMemsetAsync(out3, 0, 24, s));
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(8192), dim3(256), 0, s, (u64*)out3);
    const int iters = (int)((n_pairs + 2 * 4096 * 256 - 1) / (2 * 4096 * 256));
    if (iters > 0) {
        hipLaunchKernelGGL(k_selftest_div, dim3(4096), dim3(256), 0, s, iters, (u64*)out3);
        hipLaunchKernelGGL(k_selftest_step, dim3(4096), dim3(256), 0, s, iters, (u64*)out3);
    }
    EL_CHECK_LAUNCH();
    return 0;
}

// ---- fused item side: replay launches ----------------------------------------------------------------------------------------
static int check_item_fuse(const el_bprmf_state& st) {
    EL_REQUIRE(st.Gi_last && st.lr_hist && st.lr_hist_cap >= 4 && (st.lr_hist_cap & (st.lr_hist_cap - 1)) == 0,
               "el_bprmf: the fused item side needs Gi_last, lr_hist and a power-of-two lr_hist_cap >= 4");
    EL_REQUIRE(st.mGi && st.vGi && st.mBi && st.vBi && st.gGi && st.gBi && st.F % 4 == 0 &&
               (((uintptr_t)st.Gi | (uintptr_t)st.mGi | (uintptr_t)st.vGi | (uintptr_t)st.gGi) & 15) == 0,
               "el_bprmf: the fused item side needs Adam slots, the accumulators gGi / gBi, F %% 4 == 0 and 16-byte aligned tables");
    return 0;
}

static int launch_flush_items(const el_bprmf_state& st, hipStream_t s, int32_t t) {
    const int mask = st.lr_hist_cap - 1;
    int64_t grid = (st.I + 255) / 256;                       // a wave owns 64 consecutive rows
    if (grid > (1 << 18)) grid = 1 << 18;
    const dim3 gb((unsigned)((st.I + 255) / 256));
#define EL_FI(VW_, SER_) EL_LAUNCH("k_bpr_flush_items", (k_bpr_flush_items<VW_, 2, SER_>), dim3((unsigned)grid), dim3(256), 0, s, st, t, st.lr_hist, mask)
    if (st.replay_series) {
        EL_LAUNCH("k_bpr_flush_ibias", k_bpr_flush_ibias<true>, gb, dim3(256), 0, s, st, t, st.lr_hist, mask);
        if (st.F >= 256) EL_FI(4, true); else if (st.F >= 128) EL_FI(2, true); else EL_FI(1, true);
    } else {
        EL_LAUNCH("k_bpr_flush_ibias", k_bpr_flush_ibias<false>, gb, dim3(256), 0, s, st, t, st.lr_hist, mask);
        if (st.F >= 256) EL_FI(4, false); else if (st.F >= 128) EL_FI(2, false); else EL_FI(1, false);
    }
#undef EL_FI
    EL_CHECK_LAUNCH();
    return 0;
}

// start of step t with Gi_defer: the rows of the batch's distinct items (biases first: the row kernel advances Gi_last) to t - 1
static int launch_catchup_items(const el_bprmf_state& st, hipStream_t s, const SortedWs& w, int64_t B, int32_t t) {
    const int mask = st.lr_hist_cap - 1;
    const int64_t n = 2 * B;
    const u32 off = (u32)st.U;
    const dim3 gb((unsigned)((n + 255) / 256)), gc((unsigned)((n + 3) / 4));
#define EL_CI(VW_, SER_) EL_LAUNCH("k_bpr_catchup_items", (k_bpr_catchup_items<VW_, SER_>), gc, dim3(256), 0, s, st, w.keyI, off, n, t, st.lr_hist, mask)
    if (st.replay_series) {
        EL_LAUNCH("k_bpr_catchup_ibias", k_bpr_catchup_ibias<true>, gb, dim3(256), 0, s, st, w.keyI, off, n, t, st.lr_hist, mask);
        if (st.F >= 256) EL_CI(4, true); else if (st.F >= 128) EL_CI(2, true); else EL_CI(1, true);
    } else {
        EL_LAUNCH("k_bpr_catchup_ibias", k_bpr_catchup_ibias<false>, gb, dim3(256), 0, s, st, w.keyI, off, n, t, st.lr_hist, mask);
        if (st.F >= 256) EL_CI(4, false); else if (st.F >= 128) EL_CI(2, false); else EL_CI(1, false);
    }
#undef EL_CI
    return 0;
}

extern "C" int el_bprmf_sync_items(el_ctx* ctx, void* stream, const el_bprmf_state* stp, int32_t step) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp != nullptr && step >= 0, "el_bprmf_sync_items: bad arguments");
    if (stp->Gi_last == nullptr || step == 0) return 0;
    if (int rc = check_item_fuse(*stp)) return rc;
    return launch_flush_items(*stp, (hipStream_t)stream, step);
}

static int launch_user_adam(const SegParams& pu, hipStream_t s, int64_t B, const SortedWs& w, int lpt, int cpl, float lr_t) {
    FusedParams f;
    memset(&f, 0, sizeof(f));
    f.rowptr = w.rowptr;
    f.Gu_new = pu.st.Gu_next;
    f.lr_t = lr_t, f.b1 = 0.9f, f.b2 = 0.999f, f.eps = 1e-7f;
    // (rowptr was filled behind the sort: sort_batch)
    const int rpg = 4;                                          // rows per lane group (2 and 8 measured slower: round 3)
    const int64_t groups = (pu.st.U + rpg - 1) / rpg;
    const unsigned grid = (unsigned)((groups * lpt + 255) / 256);
    const size_t lds = (size_t)(256 / lpt) * BPR_USTG * 6 * 4;
    if (cpl == 1) EL_LAUNCH("k_bpr_user_adam", (k_bpr_user_adam<1, 4>), dim3(grid), dim3(256), lds, s, pu, f);
    else EL_LAUNCH("k_bpr_user_adam", (k_bpr_user_adam<2, 4>), dim3(grid), dim3(256), lds, s, pu, f);
    return 0;
}

// deferred decay: parameters of the user-segment kernel that catches the batch's rows up and takes the Adam step on each of them
static void defer_params(const SegParams& pu, const SortedWs& w, float lr_t, FusedParams* f) {
    memset(f, 0, sizeof(*f));
    f->lr_t = lr_t, f->b1 = 0.9f, f->b2 = 0.999f, f->eps = 1e-7f;
    f->last = pu.st.Gu_last, f->old_rows = pu.st.Gu_old, f->hpos = w.hpos;
    f->hist = pu.st.lr_hist, f->hist_mask = pu.st.lr_hist_cap - 1, f->t = pu.step;
}

template <int VW>
static int launch_segs(const SegParams& base, hipStream_t s, int64_t B, const SortedWs& w, bool fused = false, float lr_t = 0.f, bool defer = false,
                       bool ifuse = false) {
    int cpl = 1;
    const int lpt = el_pick_lpt(base.st.F, VW, &cpl);
    EL_REQUIRE(cpl <= 4, "el_bprmf_train_step: F=%d too large for this build", base.st.F);
    SegParams pu = base;
    pu.keys = w.keyU;
    pu.vals = w.valU;
    pu.n = B;
    pu.chunk = user_chunk_for(B);
    pu.lpt = lpt;
    SegParams pi = base;
    pi.keys = w.keyI;
    pi.vals = w.valI;
    pi.key_off = (u32)base.st.U;
    pi.n = 2 * B;
    pi.chunk = item_chunk_for(B, base.st.I);
    pi.lpt = lpt;
    if (defer) pi.ubase = base.st.Gu_old, pi.uidx = w.hpos;      // the pre-update user rows, one per distinct user of the batch
    const int64_t gu = (B + pu.chunk - 1) / pu.chunk, gi = (2 * B + pi.chunk - 1) / pi.chunk;
    const unsigned gridU = (unsigned)((gu * lpt + 255) / 256), gridI = (unsigned)((gi * lpt + 255) / 256);
    // (7 words per staged position where the heads' Gu_last stamps ride along: the PRE instantiations)
    const size_t ldsU = (size_t)(256 / lpt) * BPR_USTG * ((defer && VW == 4 && cpl <= 2) ? 7 : 6) * 4, ldsI = (size_t)(256 / lpt) * BPR_ISTG * 4 * 4;
    FusedParams fz;
    memset(&fz, 0, sizeof(fz));
    const bool ser = base.st.replay_series != 0;
    ItemFuse fi;
    memset(&fi, 0, sizeof(fi));
    fi.split = w.split, fi.split_long = w.split_long, fi.part = w.part, fi.part_b = w.part_b;
    if (ifuse) {
        fi.last = base.st.Gi_last, fi.hist = base.st.lr_hist, fi.hist_mask = base.st.lr_hist_cap - 1;
        fi.lr_t = lr_t, fi.b1 = 0.9f, fi.b2 = 0.999f, fi.eps = 1e-7f, fi.t = base.step;
    }
    const unsigned gridC = gridI;                                 // a lane group per listed row: at most one row per item lane group
    const unsigned gridL = (unsigned)(gi < 1024 ? (gi < 1 ? 1 : gi) : 1024);      // long rows: one workgroup each, grid-stride
    // deferred user side: two positions in flight per lane group with the heads' m / v / stamp prefetched (rows of <= 512 B per lane
    // pass: VW == 4, CPL <= 2; measured against 3, 4, 8 in flight and against a row per whole wave in round 5: profiles/r05_*)
#define EL_SEG(CPL_)                                                                                      \
    do {                                                                                                  \
        if (VW == 4 && defer) {                                                                           \
            if constexpr (VW == 4) {                                                                      \
                defer_params(pu, w, lr_t, &fz);                                                           \
                constexpr int PRE_ = CPL_ <= 2 ? 2 : 0;                                                   \
                if (ser) EL_LAUNCH("k_bpr_user_seg", (k_bpr_user_seg<4, CPL_, true, PRE_, true>), dim3(gridU), dim3(256), ldsU, s, pu, fz);   \
                else EL_LAUNCH("k_bpr_user_seg", (k_bpr_user_seg<4, CPL_, true, PRE_, false>), dim3(gridU), dim3(256), ldsU, s, pu, fz);    \
            }                                                                                             \
        } else if (fused) {                                                                               \
            if (int rc = launch_user_adam(pu, s, B, w, lpt, cpl, lr_t)) return rc;                        \
        } else {                                                                                          \
            EL_LAUNCH("k_bpr_user_seg", (k_bpr_user_seg<VW, CPL_, false>), dim3(gridU), dim3(256), ldsU, s, pu, fz);  \
        }                                                                                                 \
        EL_CHECK_HIP(hipMemsetAsync(w.split_long, 0, 4, s));                                              \
        if (ifuse && VW == 4 && CPL_ <= 2) {                                                              \
            EL_LAUNCH("k_bpr_item_seg", (k_bpr_item_seg<VW, CPL_, (VW == 4 && CPL_ <= 2)>), dim3(gridI), dim3(256), ldsI, s, pi, fi);   \
            EL_LAUNCH("k_bpr_item_combine", (k_bpr_item_combine<VW, CPL_, (VW == 4 && CPL_ <= 2)>), dim3(gridC), dim3(256), 0, s, pi.st, fi, pi.keys, pi.key_off, pi.n, pi.chunk, lpt);  \
            EL_LAUNCH("k_bpr_item_combine_long", (k_bpr_item_combine_long<VW, CPL_, (VW == 4 && CPL_ <= 2)>), dim3(gridL), dim3(256), 0, s, pi.st, fi, pi.keys, pi.key_off, pi.n, pi.chunk, lpt);  \
        } else {                                                                                          \
            EL_LAUNCH("k_bpr_item_seg", (k_bpr_item_seg<VW, CPL_, false>), dim3(gridI), dim3(256), ldsI, s, pi, fi);    \
            EL_LAUNCH("k_bpr_item_combine", (k_bpr_item_combine<VW, CPL_, false>), dim3(gridC), dim3(256), 0, s, pi.st, fi, pi.keys, pi.key_off, pi.n, pi.chunk, lpt);  \
            EL_LAUNCH("k_bpr_item_combine_long", (k_bpr_item_combine_long<VW, CPL_, false>), dim3(gridL), dim3(256), 0, s, pi.st, fi, pi.keys, pi.key_off, pi.n, pi.chunk, lpt);  \
        }                                                                                                 \
    } while (0)
    if (cpl == 1) EL_SEG(1);
    else if (cpl == 2) EL_SEG(2);
    else EL_SEG(4);
#undef EL_SEG
    EL_CHECK_LAUNCH();
    return 0;
}

// opt: the optimiser, or -1 = gradients only (el_bprmf_grads); presorted: el_bprmf_presort already ordered this batch in `ws`
static int sorted_step(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i, const int32_t* j,
                       int64_t B, float lr, float l_w, float l_b, int opt, bool presorted, int32_t step, float lr_t, double* loss_out,
                       void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    bool vec = false, rows_mode = false;
    if (int rc = el_bprmf_check_state(stp, u, i, j, loss_out, opt < 0 ? EL_OPT_SGD : opt, step, &vec, &rows_mode)) return rc;
    if (opt < 0) rows_mode = false;
    if (B <= 0) return 0;
    EL_REQUIRE(B < (1LL << 30), "el_bprmf_train_step_sorted: batch too large");
    const el_bprmf_state st = *stp;
    SortedWs w;
    EL_REQUIRE(carve_ws(B, st.U, st.I, (char*)ws, &w, st.F) == 0, "el_bprmf_train_step_sorted: rocprim size query failed");
    EL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "el_bprmf_train_step_sorted: workspace too small (%zu < %zu)",
               ws_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    if (!presorted)
        if (int rc = sort_batch(s, w, u, i, j, B, st.U, st.I, st.Gu_next != nullptr && st.Gu_last == nullptr)) return rc;
    if (st.uslot) {
        EL_REQUIRE(vec && ((uintptr_t)st.gGu_rows & 15) == 0 && !rows_mode && (opt == EL_OPT_ADAM_TF_DENSE || opt < 0),
                   "el_bprmf_train_step: compact user-gradient rows need F %% 4 == 0, 16-byte aligned tables and the TF-dense Adam");
    }
    SegParams base;
    memset(&base, 0, sizeof(base));
    base.st = st;
    if (!rows_mode) base.st.tGu = base.st.tGi = base.st.tBi = nullptr;
    base.bi = i;
    base.bj = j;
    base.bu = u;
    base.s = w.s;
    base.l_w = l_w;
    base.l_b = l_b;
    base.step = step;
    base.loss_out = loss_out;
    // fused user side (el_bprmf_state.Gu_next): segments + Keras Adam over every user row in one kernel, new rows to Gu_next
    int cplq = 1;
    el_pick_lpt(st.F, 4, &cplq);
    // deferred decay (el_bprmf_state.Gu_last): the fused user-side kernel on the batch's rows only, theta updated in place
    const bool defer = st.Gu_last != nullptr && opt == EL_OPT_ADAM_TF_DENSE;
    if (st.Gu_last != nullptr) {
        EL_REQUIRE(opt == EL_OPT_ADAM_TF_DENSE, "el_bprmf: with the deferred decay on (Gu_last) the user rows are only current after "
                   "el_bprmf_sync_users; gradient-only / other-optimiser calls take a state with Gu_last = NULL after that sync");
        if (int rc = check_deferred(st)) return rc;
        EL_REQUIRE(vec && !rows_mode && cplq <= 2, "el_bprmf_train_step: deferred decay needs F %% 4 == 0, F <= 512 and 16-byte aligned tables");
        EL_REQUIRE(st.Gu_old != nullptr && st.Gu_old_cap >= B && ((uintptr_t)st.Gu_old & 15) == 0,
                   "el_bprmf_train_step: deferred decay needs Gu_old with >= B rows (%lld < %lld)", (long long)st.Gu_old_cap, (long long)B);
        // the lr history is a ring: no row may fall more than half of it behind
        if (step > 1 && (step - 1) % (st.lr_hist_cap / 2) == 0)
            if (int rc = launch_flush_users(st, s, step - 1)) return rc;
    }
    const bool fused = defer || (st.Gu_next != nullptr && opt == EL_OPT_ADAM_TF_DENSE && vec && !rows_mode && cplq <= 2 &&
                                 (((uintptr_t)st.Gu_next | (uintptr_t)st.mGu | (uintptr_t)st.vGu) & 15) == 0);
    if (st.Gu_next != nullptr && opt == EL_OPT_ADAM_TF_DENSE && !defer)
        EL_REQUIRE(fused, "el_bprmf_train_step: Gu_next needs F %% 4 == 0, F <= 512 and 16-byte aligned tables");
    if (st.uslot && !fused) {
        EL_REQUIRE(st.gGu_rows != nullptr && st.gGu_cap >= B, "el_bprmf_train_step: compact user-gradient rows need gGu_rows with >= B rows (%lld < %lld)",
                   (long long)st.gGu_cap, (long long)B);
    }
    // fused item side (el_bprmf_state.Gi_last): the item segments take the Adam step on their rows; rows outside the batch are
    // replayed at the end of the step (Gi_defer == 0) or when next needed (Gi_defer != 0: caught up here, before the user side
    // gathers them)
    const bool ifuse = st.Gi_last != nullptr;
    if (ifuse) {
        EL_REQUIRE(opt == EL_OPT_ADAM_TF_DENSE && fused && vec && !rows_mode,
                   "el_bprmf: the fused item side (Gi_last) runs with EL_OPT_ADAM_TF_DENSE and the fused user side (Gu_next or Gu_last) only; "
                   "gradient-only / other-optimiser calls take a state with Gi_last = NULL after el_bprmf_sync_items");
        if (int rc = check_item_fuse(st)) return rc;
        if (st.Gi_defer) {
            if (step > 1 && (step - 1) % (st.lr_hist_cap / 2) == 0)          // the lr ring: no row may fall more than half of it behind
                if (int rc = launch_flush_items(st, s, step - 1)) return rc;
            if (int rc = launch_catchup_items(st, s, w, B, step)) return rc;
        }
    }
    int rc = vec ? launch_segs<4>(base, s, B, w, fused, lr_t, defer, ifuse) : launch_segs<1>(base, s, B, w);
    if (rc) return rc;
    if (opt < 0) return 0;                           // gradients only (el_bprmf_grads)
    if (ifuse) return st.Gi_defer ? 0 : launch_flush_items(st, s, step);
    if (fused) return el_bprmf_apply_items_adam(ctx, s, st, lr_t);
    return el_bprmf_apply_optimizer(ctx, s, st, u, i, j, B, lr, opt, step, lr_t);
}

extern "C" __attribute__((visibility("hidden"))) int el_bprmf_train_step_sorted(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u,
                                          const int32_t* i, const int32_t* j, int64_t B, float lr, float l_w,
                                          float l_b, int opt, int32_t step, float lr_t, double* loss_out, void* ws,
                                          size_t ws_bytes) {
    return sorted_step(ctx, stream, stp, u, i, j, B, lr, l_w, l_b, opt, false, step, lr_t, loss_out, ws, ws_bytes);
}

// CML (el_cml.hip): the same sort + segment walk with the per-triplet coefficients given (cD = dloss/dD, cE = dloss/dE)
// instead of computed by the user segments.  ws: el_bprmf_ws_bytes(B, U, I).
int el_bpr_sorted_cml_grads(el_ctx* ctx, hipStream_t s, const el_bprmf_state& st, const int32_t* u, const int32_t* i,
                            const int32_t* j, int64_t B, float l_w, float l_b, float* cD, const float* cE, void* ws,
                            size_t ws_bytes) {
    SortedWs w;
    EL_REQUIRE(carve_ws(B, st.U, st.I, (char*)ws, &w, st.F) == 0, "el_cml_train_step: rocprim size query failed");
    EL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "el_cml_train_step: segment workspace too small (%zu < %zu)", ws_bytes, w.total);
    if (int rc = sort_batch(s, w, u, i, j, B, st.U, st.I, false)) return rc;
    SegParams base;
    memset(&base, 0, sizeof(base));
    base.st = st;
    base.st.tGu = base.st.tGi = base.st.tBi = nullptr;
    base.st.uslot = nullptr;                                  // CML writes the dense accumulator
    base.bi = i, base.bj = j, base.bu = u;
    base.s = cD, base.s2 = cE, base.cml = 1;
    base.l_w = l_w, base.l_b = l_b;
    const bool vec = st.F % 4 == 0 && (((uintptr_t)st.Gu | (uintptr_t)st.Gi | (uintptr_t)st.gGu | (uintptr_t)st.gGi) & 15) == 0;
    return vec ? launch_segs<4>(base, s, B, w) : launch_segs<1>(base, s, B, w);
}

// Gradients of one batch into the dense accumulators gGu / gGi / gBi (+ the loss), no optimiser: the first half of the
// step above.  Used by the multi-GPU "dense" mode: reduce-scatter of gGu over the ranks, optimiser on the owned rows.
extern "C" int el_bprmf_grads(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                              const int32_t* j, int64_t B, float l_w, float l_b, int32_t step, double* loss_out, void* ws,
                              size_t ws_bytes) {
    return sorted_step(ctx, stream, stp, u, i, j, B, 0.f, l_w, l_b, -1, false, step, 0.f, loss_out, ws, ws_bytes);
}

// The sort of a batch reads nothing but the triplets, so a multi-GPU step can order the NEXT batch while its collective is in
// flight: el_bprmf_presort fills `ws` (prep + the one radix sort), el_bprmf_grads_presorted runs the segment kernels on it.
extern "C" int el_bprmf_presort(el_ctx* ctx, void* stream, const int32_t* u, const int32_t* i, const int32_t* j, int64_t B,
                                int64_t U, int64_t I, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(u && i && j && B >= 1 && B < (1LL << 30) && U >= 1 && I >= 1, "el_bprmf_presort: bad arguments");
    SortedWs w;
    EL_REQUIRE(carve_ws(B, U, I, (char*)ws, &w) == 0, "el_bprmf_presort: rocprim size query failed");
    EL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "el_bprmf_presort: workspace too small (%zu < %zu)", ws_bytes, w.total);
    return sort_batch((hipStream_t)stream, w, u, i, j, B, U, I, true);
}

extern "C" int el_bprmf_grads_presorted(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                                        const int32_t* j, int64_t B, float l_w, float l_b, int32_t step, double* loss_out,
                                        void* ws, size_t ws_bytes) {
    return sorted_step(ctx, stream, stp, u, i, j, B, 0.f, l_w, l_b, -1, true, step, 0.f, loss_out, ws, ws_bytes);
}

// The whole step on a batch el_bprmf_presort ordered into `ws` (software pipeline of a single GPU: the NEXT batch is drawn and
// ordered on a side stream meanwhile): segment kernels + loss + optimiser; the fused user-side kernel when st->Gu_next is set.
extern "C" int el_bprmf_train_step_presorted(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u, const int32_t* i,
                                             const int32_t* j, int64_t B, float lr, float l_w, float l_b, int opt, int32_t step,
                                             float lr_t, double* loss_out, void* ws, size_t ws_bytes) {
    EL_REQUIRE(opt == EL_OPT_ADAM_TF_DENSE || opt == EL_OPT_SGD, "el_bprmf_train_step_presorted: the dense optimisers only (adam_tf_dense, sgd)");
    return sorted_step(ctx, stream, stp, u, i, j, B, lr, l_w, l_b, opt, true, step, lr_t, loss_out, ws, ws_bytes);
}

// =====================================================================================================
// Item-sharded training (new design -- the reference is single-device).
// Rank r holds the item rows [lo_r, hi_r) (+ their optimiser state) and a replica of the user table.  One step:
//   1. el_bprmf_shard_grads   local triplets (positives and negatives inside the shard): s_b, loss, the item-row
//                             gradients (sorted segments, as above) and ONE user-gradient row per triplet
//                             dU[b,:] = s_b (gamma_i - gamma_j) + l_w gamma_u          (no reduction yet)
//   2. RCCL all-gather of (u[b], dU[b,:]) over the ranks                      (torch.distributed, parallel.py)
//   3. el_rows_segment_sum    every rank reduces the gathered rows by user id into its dense dGu -- the same
//                             rows in the same order on every rank, so the user-table replicas stay bit-identical
//   4. el_bprmf_apply         optimiser on the replicated user table and the local item shard
// The gradient of the global batch is the sum over triplets, so G ranks x B triplets equal one rank with the
// concatenated batch (tests/test_dist_gloo.py checks exactly that).
// =====================================================================================================
template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_bpr_triplet_rows(el_bprmf_state st, const int32_t* __restrict__ bu,
                                                          const int32_t* __restrict__ bi_, const int32_t* __restrict__ bj,
                                                          int64_t B, float l_w, float l_b, int lpt, float* __restrict__ s_out,
                                                          float* __restrict__ dU, double* loss_out) {
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
#pragma unroll
        for (int x = 0; x < VW; ++x) gu[q][x] = gi[q][x] = gj[q][x] = 0.f;
        if (active && e < F) {
            ldv<VW>(pu + e, gu[q]);
            ldv<VW>(pi + e, gi[q]);
            ldv<VW>(pj + e, gj[q]);
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
    const float d = (beta_i + dpi) - (beta_j + dpj);
    const float dc = fminf(fmaxf(d, -80.0f), 1e8f);
    float sb = 0.f;
    if (d >= -80.0f) sb = -1.0f / (1.0f + expf(d));
    float myloss = 0.f;
    if (active) {
        if (sub == 0) {
            s_out[b] = sb;
            myloss = el_softplus_s(-dc) + l_w * 0.5f * (nu + ni + nj) + l_b * 0.5f * beta_i * beta_i +
                     (l_b * 0.5f * beta_j * beta_j) / 10.0f;
        }
        float* o = dU + b * F;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
                float v[VW];
#pragma unroll
                for (int x = 0; x < VW; ++x) v[x] = sb * (gi[q][x] - gj[q][x]) + l_w * gu[q][x];
                stv<VW>(o + e, v);
            }
        }
    }
    __shared__ float wsum[4];
    float wl = el_group_sum(myloss, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = wl;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0) atomicAdd(loss_out, tot);
    }
}

// out[id,:] = sum of rows[pos,:] over the sorted (id, pos) pairs.  One lane group per sorted position; only the
// group sitting on a segment HEAD works and walks its whole segment in order -> plain stores, no atomics, the same
// summation order on every rank (user segments are short: Poisson(B/U)).
template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_rows_segsum(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                     const float* __restrict__ rows, int64_t n, int F, int chunk, int lpt,
                                                     float* __restrict__ out) {
    (void)chunk;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pos0 = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    if (pos0 >= n) return;
    const u32 key = keys[pos0];
    if (pos0 > 0 && keys[pos0 - 1] == key) return;      // not a head
    float acc[CPL][VW];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
    for (int64_t pos = pos0; pos < n && keys[pos] == key; ++pos) {
        const float* r = rows + (int64_t)vals[pos] * F;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
                float t[VW];
                ldv<VW>(r + e, t);
#pragma unroll
                for (int x = 0; x < VW; ++x) acc[q][x] += t[x];
            }
        }
    }
    float* g = out + (int64_t)key * F;
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
        if (e < F) stv<VW>(g + e, acc[q]);
    }
}

__global__ __launch_bounds__(256) void k_iota_keys(const int32_t* __restrict__ ids, int64_t n, u32* key, u32* val) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    key[t] = (u32)ids[t];
    val[t] = (u32)t;
}

extern "C" int el_bprmf_shard_grads(el_ctx* ctx, void* stream, const el_bprmf_state* stp, const int32_t* u,
                                    const int32_t* i, const int32_t* j, int64_t B, float l_w, float l_b, int32_t step,
                                    float* dU, double* loss_out, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    bool vec = false, rows_mode = false;
    if (int rc = el_bprmf_check_state(stp, u, i, j, loss_out, EL_OPT_SGD, step, &vec, &rows_mode)) return rc;
    EL_REQUIRE(dU != nullptr, "el_bprmf_shard_grads: null dU");
    if (B <= 0) return 0;
    const el_bprmf_state st = *stp;
    vec = vec && (((uintptr_t)dU) % 16 == 0);
    SortedWs w;
    EL_REQUIRE(carve_ws(B, st.U, st.I, (char*)ws, &w, st.F) == 0, "el_bprmf_shard_grads: rocprim size query failed");
    EL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "el_bprmf_shard_grads: workspace too small (%zu < %zu)", ws_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    int cpl = 1;
    const int lpt = el_pick_lpt(st.F, vec ? 4 : 1, &cpl);
    EL_REQUIRE(cpl <= 4, "el_bprmf_shard_grads: F=%d too large for this build", st.F);
    const unsigned gridT = (unsigned)((B * lpt + 255) / 256);
#define EL_TR(VW_, CPL_) \
    EL_LAUNCH("k_bpr_triplet_rows", (k_bpr_triplet_rows<VW_, CPL_>), dim3(gridT), dim3(256), 0, s, st, u, i, j, B, l_w, l_b, lpt, w.s, dU, loss_out)
    if (vec) {
        if (cpl == 1) EL_TR(4, 1); else if (cpl == 2) EL_TR(4, 2); else EL_TR(4, 4);
    } else {
        if (cpl == 1) EL_TR(1, 1); else if (cpl == 2) EL_TR(1, 2); else EL_TR(1, 4);
    }
#undef EL_TR
    // item side: sorted segments over the local shard
    EL_LAUNCH("k_bpr_prep", k_bpr_prep, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, u, i, j, B, w.keyU_in, w.valU_in,
              w.keyI_in, w.valI_in, 0u);
    {
        ElKernelTimer t("rocprim_radix_sort_pairs", s);
        size_t tb = w.tmp_bytes;
        EL_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.keyI_in, w.keyI, w.valI_in, w.valI, (unsigned)(2 * B), 0,
                                               bits_for(st.I), s));
    }
    SegParams pi;
    memset(&pi, 0, sizeof(pi));
    pi.st = st;
    pi.st.tGu = pi.st.tGi = pi.st.tBi = nullptr;
    pi.bi = i;
    pi.bj = j;
    pi.bu = u;
    pi.s = w.s;
    pi.l_w = l_w;
    pi.l_b = l_b;
    pi.step = step;
    pi.loss_out = loss_out;
    pi.keys = w.keyI;
    pi.vals = w.valI;
    pi.n = 2 * B;
    pi.chunk = item_chunk_for(B, st.I);
    pi.lpt = lpt;
    const int64_t gi = (2 * B + pi.chunk - 1) / pi.chunk;
    const unsigned gridI = (unsigned)((gi * lpt + 255) / 256);
    const size_t ldsI = (size_t)(256 / lpt) * BPR_ISTG * 4 * 4;
    ItemFuse fi;
    memset(&fi, 0, sizeof(fi));
    fi.split = w.split, fi.split_long = w.split_long, fi.part = w.part, fi.part_b = w.part_b;
    EL_CHECK_HIP(hipMemsetAsync(w.split_long, 0, 4, s));
    const unsigned gridC = gridI, gridL = (unsigned)(gi < 1024 ? (gi < 1 ? 1 : gi) : 1024);
#define EL_IS(VW_, CPL_)                                                                                                                     \
    do {                                                                                                                                     \
        EL_LAUNCH("k_bpr_item_seg", (k_bpr_item_seg<VW_, CPL_, false>), dim3(gridI), dim3(256), ldsI, s, pi, fi);                            \
        EL_LAUNCH("k_bpr_item_combine", (k_bpr_item_combine<VW_, CPL_, false>), dim3(gridC), dim3(256), 0, s, pi.st, fi, pi.keys, pi.key_off, \
                  pi.n, pi.chunk, lpt);                                                                                                      \
        EL_LAUNCH("k_bpr_item_combine_long", (k_bpr_item_combine_long<VW_, CPL_, false>), dim3(gridL), dim3(256), 0, s, pi.st, fi, pi.keys,       \
                  pi.key_off, pi.n, pi.chunk, lpt);                                                                                          \
    } while (0)
    if (vec) {
        if (cpl == 1) EL_IS(4, 1); else if (cpl == 2) EL_IS(4, 2); else EL_IS(4, 4);
    } else {
        if (cpl == 1) EL_IS(1, 1); else if (cpl == 2) EL_IS(1, 2); else EL_IS(1, 4);
    }
#undef EL_IS
    EL_CHECK_LAUNCH();
    return 0;
}

static int carve_rows_ws(int64_t n, int64_t n_ids, char* base, u32** kin, u32** vin, u32** kout, u32** vout, void** tmp,
                         size_t* tmp_bytes, size_t* total) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    *kin = (u32*)take((size_t)n * 4);
    *vin = (u32*)take((size_t)n * 4);
    *kout = (u32*)take((size_t)n * 4);
    *vout = (u32*)take((size_t)n * 4);
    size_t t1 = 0;
    u32* np = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, t1, np, np, np, np, (unsigned)n, 0, bits_for(n_ids), (hipStream_t)0) != hipSuccess) return 1;
    *tmp_bytes = t1;
    *tmp = take(t1);
    *total = off;
    return 0;
}

extern "C" size_t el_rows_segment_sum_ws_bytes(int64_t n, int64_t n_ids) {
    if (n <= 0) return 0;
    u32 *a, *b, *c, *d;
    void* t;
    size_t tb, total;
    if (carve_rows_ws(n, n_ids, nullptr, &a, &b, &c, &d, &t, &tb, &total)) return 0;
    return total;
}

// out[ids[p], :] = sum over p of rows[p, :] per id (touched rows are overwritten; stable order, deterministic)
extern "C" int el_rows_segment_sum(el_ctx* ctx, void* stream, const int32_t* ids, const float* rows, int64_t n, int32_t F,
                                   int64_t n_ids, float* out, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(ids && rows && out && F >= 1 && n_ids >= 1, "el_rows_segment_sum: bad arguments");
    if (n <= 0) return 0;
    EL_REQUIRE(n < (1LL << 31), "el_rows_segment_sum: too many rows");
    u32 *kin, *vin, *kout, *vout;
    void* tmp;
    size_t tb, total;
    EL_REQUIRE(carve_rows_ws(n, n_ids, (char*)ws, &kin, &vin, &kout, &vout, &tmp, &tb, &total) == 0, "el_rows_segment_sum: size query failed");
    EL_REQUIRE(ws != nullptr && ws_bytes >= total, "el_rows_segment_sum: workspace too small (%zu < %zu)", ws_bytes, total);
    hipStream_t s = (hipStream_t)stream;
    EL_LAUNCH("k_iota_keys", k_iota_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, n, kin, vin);
    {
        ElKernelTimer t("rocprim_radix_sort_pairs", s);
        EL_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, (unsigned)n, 0, bits_for(n_ids), s));
    }
    const bool vec = (F % 4 == 0) && (((uintptr_t)rows) % 16 == 0) && (((uintptr_t)out) % 16 == 0);
    int cpl = 1;
    const int lpt = el_pick_lpt(F, vec ? 4 : 1, &cpl);
    EL_REQUIRE(cpl <= 4, "el_rows_segment_sum: F=%d too large for this build", F);
    const int chunk = 1;
    const int64_t groups = n;
    const unsigned grid = (unsigned)((groups * lpt + 255) / 256);
#define EL_RS(VW_, CPL_) EL_LAUNCH("k_rows_segsum", (k_rows_segsum<VW_, CPL_>), dim3(grid), dim3(256), 0, s, kout, vout, rows, n, (int)F, chunk, lpt, out)
    if (vec) {
        if (cpl == 1) EL_RS(4, 1); else if (cpl == 2) EL_RS(4, 2); else EL_RS(4, 4);
    } else {
        if (cpl == 1) EL_RS(1, 1); else if (cpl == 2) EL_RS(1, 2); else EL_RS(1, 4);
    }
#undef EL_RS
    EL_CHECK_LAUNCH();
    return 0;
}

// optimiser phase alone (gradients already in gGu / gGi / gBi): TF-dense Adam or dense SGD
extern "C" int el_bprmf_apply(el_ctx* ctx, void* stream, const el_bprmf_state* stp, float lr, int opt, int32_t step, float lr_t) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp != nullptr && stp->Gu && stp->Gi && stp->Bi && (stp->gGu || stp->uslot) && stp->gGi && stp->gBi, "el_bprmf_apply: null state");
    EL_REQUIRE(opt == EL_OPT_ADAM_TF_DENSE || opt == EL_OPT_SGD, "el_bprmf_apply: only the dense optimisers (adam_tf_dense, sgd) are available here");
    if (opt == EL_OPT_ADAM_TF_DENSE) EL_REQUIRE(stp->mGu && stp->vGu && stp->mGi && stp->vGi && stp->mBi && stp->vBi, "el_bprmf_apply: Adam slots missing");
    el_bprmf_state st = *stp;
    st.tGu = st.tGi = st.tBi = nullptr;
    return el_bprmf_apply_optimizer(ctx, (hipStream_t)stream, st, nullptr, nullptr, nullptr, 0, lr, opt, step, lr_t);
}
