// Full-catalog scoring + masked top-k.
//
// Replaces BPRMF_batch_model.predict/get_top_k (BPRMF_batch_model.py:83-88),
// MFModel.get_user_predictions (BPRMF_model.py:70-85) and the get_top_k bodies of the
// VAE / NeuMF / GMF models (multi_vae_model.py:158-159, ...).
//
// Numerics contract (pinned by oracle/c/el_oracle.c):
//   dot(u,i)  = fma chain over f = 0..F-1 in that order, starting from +0
//               (v_mfma_f32_32x32x2_f32 is bitwise this chain, MI355X_MICROARCH.md)
//   score     = (Bi ? dot + Bi[i] : dot) + 0.0f            (-0 canonicalised)
//   order     = score descending, item index ascending     (tf.nn.top_k tie rule)
//   NaN scores are never selected.
// The [users, items] score matrix is never written to HBM: selection is fused behind the
// MFMA accumulators (threshold filter in registers, candidate lists in LDS).
#include <stdlib.h>
#include "el_common.h"

#include "el_topk_common.h"

// =====================================================================================
// Wave-per-user kernel (VALU fma chain).  General k, general F, candidate protocol,
// dense-preds source.  Also the cross-check for the MFMA kernel (bitwise-equal scores).
// =====================================================================================
template <bool DENSE>
__global__ __launch_bounds__(64) void k_topk_wave(TopkParams p, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);              // [cap]
    int* cnt_s = reinterpret_cast<int*>(smem + (size_t)cap * 8);  // [1] (+pad), used by el_wave_compact
    const int lane = threadIdx.x;
    int64_t urel = blockIdx.x;                             // row of out_idx / out_val
    if (p.ulist) {                                         // (dense rows stay indexed by blockIdx.x)
        int64_t n = (int64_t)*p.ulist_n;
        if (p.ulist_max > 0 && n > p.ulist_max) n = p.ulist_max;
        if ((int64_t)blockIdx.x + p.ulist_skip >= n) return;
        urel = p.ulist[blockIdx.x + p.ulist_skip];
    }
    const int64_t user = p.u_start + urel;
    const int F = p.F;
    const float* gu = DENSE ? nullptr : p.Gu + user * (int64_t)F;
    int64_t e0 = 0, e1 = 0, c0 = 0, c1 = 0;
    if (p.excl_indptr) {
        e0 = p.excl_indptr[user];
        e1 = p.excl_indptr[user + 1];
    }
    int64_t ncand = p.I_local;
    if (p.cand_indptr) {
        c0 = p.cand_indptr[user];
        c1 = p.cand_indptr[user + 1];
        ncand = c1 - c0;
    }
    const bool use_excl = (p.excl_indptr != nullptr) && (p.cand_indptr == nullptr);
    // item-split mode (gridDim.y = nsplit, full-catalogue scan only): this wave takes one slice, writes a partial list
    int64_t pos_lo = 0, pos_hi = ncand;
    const bool split = p.nsplit > 1 && !p.cand_indptr;
    if (split) {
        pos_lo = ncand * blockIdx.y / p.nsplit;
        pos_hi = ncand * (blockIdx.y + 1) / p.nsplit;
    }
    int cnt = 0;
    float tau = -INFINITY;
    for (int64_t base = pos_lo; base < pos_hi; base += 64) {
        int64_t pos = base + lane;
        bool valid = pos < pos_hi;
        int32_t gitem = -1;
        int64_t il = 0;
        if (valid) {
            if (p.cand_indptr) {
                gitem = p.cand_indices[c0 + pos];
                il = (int64_t)gitem - p.item_offset;
                valid = (il >= 0 && il < p.I_local);
            } else {
                il = pos;
                gitem = (int32_t)(p.item_offset + pos);
            }
        }
        float s = 0.f;
        if (valid) {
            if (DENSE) {
                s = p.preds[(int64_t)blockIdx.x * p.ld + il] + 0.0f;
            } else {
                const float* gi = p.Gi + il * (int64_t)F;
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc = __builtin_fmaf(gi[f], gu[f], acc);
                s = (p.Bi ? acc + p.Bi[il] : acc) + 0.0f;
            }
        }
        bool hit = valid && (s >= tau);
        if (hit && use_excl) hit = !el_row_contains(p.excl_indices, e0, e1, gitem);
        u64 bal = __ballot(hit);
        if (bal) {
            int offp = __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) keys[cnt + offp] = el_make_key(s, gitem);
            cnt += __popcll(bal);
        }
        if (cnt > cap - 64) {
            if (lane == 0) *cnt_s = cnt;
            tau = el_wave_compact(keys, cnt_s, cap, p.k, lane);
            cnt = cnt < p.k ? cnt : p.k;
        }
    }
    if (lane == 0) *cnt_s = cnt;
    el_wave_compact(keys, cnt_s, cap, p.k, lane);
    const int nv = cnt < p.k ? cnt : p.k;
    const int64_t orow = split ? ((int64_t)blockIdx.y * p.part_stride + blockIdx.x) * p.k : urel * p.k;
    for (int t = lane; t < p.k; t += 64) {
        int32_t oi;
        float ov;
        if (t < nv) {
            u64 key = keys[t];
            oi = el_key_item(key);
            ov = el_key_score(key);
        } else {
            oi = split ? el_fill_masked_range(p, p.item_offset + pos_lo, p.item_offset + pos_hi, e0, e1, c0, c1, t - nv)
                       : el_fill_masked(p, e0, e1, c0, c1, t - nv);
            ov = -INFINITY;
        }
        p.out_idx[orow + t] = oi;
        p.out_val[orow + t] = ov;
    }
}

// =====================================================================================
// MFMA kernel: 128 users per workgroup (4 waves x 32 users), items streamed in tiles of
// 32*NIB rows through LDS in K-chunks of 32, user fragments resident in VGPRs for the
// whole kernel, D[item][user] accumulated by v_mfma_f32_32x32x2_f32, selection fused.
//   A operand (items): lane l supplies A[row = l&31][k = l>>5]    <- LDS (stride 33)
//   B operand (users): lane l supplies B[k = l>>5][col = l&31]    <- registers
//   D: lane l holds col (user) l&31, rows (items) (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
// Lanes l and l+32 therefore own the same user; every user belongs to exactly one wave,
// so all top-k state is wave-private (no cross-wave synchronisation outside staging).
// =====================================================================================
template <int FP, int NIB, int CAP, int KC, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void k_score_topk_mfma(TopkParams p, int vec) {
    constexpr int LDA = KC + 1, BI = 32 * NIB, NCH = (FP + KC - 1) / KC, UPB = NW * 32, NT = NW * 64;
    constexpr int C4 = KC / 4;                      // float4 pieces per staged row
    constexpr int NLD = (BI * C4 + NT - 1) / NT;    // float4 loads per thread per chunk
    static_assert((BI * C4) % NT == 0, "staging must divide evenly");
    constexpr int A_FLOATS = 2 * BI * LDA + 2 * BI + 8;
    static_assert(CAP <= 64, "el_wave_compact_excl handles one key per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                // [2][BI][LDA]
    float* Bs = As + 2 * BI * LDA;                             // [2][BI] bias
    float* Bm = Bs + 2 * BI;                                   // [2][4] per-wave max of the staged bias tile
    u64* keys = reinterpret_cast<u64*>(smem + A_FLOATS * 4);   // [UPB][CAP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    const int uslot = wave * 32 + col;
    // users of this workgroup: a slice of [u_start, u_stop), or of the caller's list (exact fallback of the screened path)
    int64_t nlist = p.u_stop - p.u_start;
    if (p.ulist) {
        nlist = (int64_t)*p.ulist_n;
        if (p.ulist_max > 0 && nlist > p.ulist_max) nlist = p.ulist_max;
        nlist -= p.ulist_skip;
    }
    if ((int64_t)blockIdx.x * UPB >= nlist) return;
    const int64_t uidx = (int64_t)blockIdx.x * UPB + uslot;
    const bool uvalid = uidx < nlist;
    const int64_t user = p.u_start + (uvalid ? (p.ulist ? (int64_t)p.ulist[p.ulist_skip + uidx] : uidx) : 0);
    const int F = p.F;
    const int64_t I = p.I_local;

    // resident user fragments
    float bfrag[FP / 2];
    {
        const float* gu = p.Gu + (uvalid ? user : p.u_start) * (int64_t)F;
#pragma unroll
        for (int s = 0; s < FP / 2; ++s) {
            int kk = 2 * s + hi;
            bfrag[s] = (uvalid && kk < F) ? gu[kk] : 0.f;
        }
    }
    int64_t e0 = 0, e1 = 0;
    if (p.excl_indptr && uvalid) {
        e0 = p.excl_indptr[user];
        e1 = p.excl_indptr[user + 1];
    }
    int ucnt = 0;            // keys in this user's list (identical in lanes l and l+32)
    float tau = -INFINITY;

    const int ntiles_all = (int)((I + BI - 1) / BI);
    const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int tile0 = (int)((int64_t)ntiles_all * blockIdx.y / nsplit);         // this workgroup's slice of the item tiles
    const int ntiles = (int)((int64_t)ntiles_all * (blockIdx.y + 1) / nsplit);
    const int nch = (F + KC - 1) / KC;

    float4 pre[NLD];
    float pre_bias = 0.f;

    auto gload = [&](int tile, int ch) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            int f4 = q * NT + tid;
            int row = f4 / C4, c4 = f4 % C4;
            int64_t item = (int64_t)tile * BI + row;
            int kk = ch * KC + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < I && kk < F) {
                const float* src = p.Gi + item * (int64_t)F + kk;
                if (vec) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (kk + 1 < F) v.y = src[1];
                    if (kk + 2 < F) v.z = src[2];
                    if (kk + 3 < F) v.w = src[3];
                }
            }
            pre[q] = v;
        }
        if (ch == 0 && tid < BI) {
            int64_t item = (int64_t)tile * BI + tid;
            pre_bias = (p.Bi && item < I) ? p.Bi[item] : 0.f;
        }
    };
    auto lstore = [&](int buf, int bias_buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            int f4 = q * NT + tid;
            int row = f4 / C4, c4 = f4 % C4;
            float* dst = As + (buf * BI + row) * LDA + c4 * 4;
            dst[0] = pre[q].x;
            dst[1] = pre[q].y;
            dst[2] = pre[q].z;
            dst[3] = pre[q].w;
        }
        if (bias_buf >= 0 && tid < ((BI + 63) & ~63)) {   // whole waves: tid < BI rounded up to 64
            if (tid < BI) Bs[bias_buf * BI + tid] = pre_bias;
            float wm = (tid < BI) ? pre_bias : -INFINITY;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
            if (lane == 0) Bm[bias_buf * 4 + wave] = wm;
        }
    };

    u64* wkeys = keys + (size_t)wave * 32 * CAP;

    int buf = 0;
    if (ntiles > tile0) gload(tile0, 0);
    for (int tile = tile0; tile < ntiles; ++tile) {
        floatx16 acc[NIB];
#pragma unroll
        for (int b = 0; b < NIB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch < nch) {
                lstore(buf, ch == 0 ? (tile & 1) : -1);
                __syncthreads();
                int nt = tile, nc = ch + 1;
                if (nc >= nch) {
                    nt = tile + 1;
                    nc = 0;
                }
                if (nt < ntiles) gload(nt, nc);
                const float* Ab = As + buf * BI * LDA;
#pragma unroll
                for (int s = 0; s < KC / 2; ++s) {
                    if (ch * (KC / 2) + s >= FP / 2) break;   // compile-time (FP < KC * NCH)
                    float a[NIB];
#pragma unroll
                    for (int b = 0; b < NIB; ++b) a[b] = Ab[(b * 32 + col) * LDA + 2 * s + hi];
#pragma unroll
                    for (int b = 0; b < NIB; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bfrag[ch * (KC / 2) + s], acc[b], 0, 0, 0);
                }
                buf ^= 1;
            }
        }

        // ---- fused selection ------------------------------------------------------
        const float* bb = Bs + (tile & 1) * BI;
        if (p.dbg & 1) {
            float sink = 0.f;
#pragma unroll
            for (int b = 0; b < NIB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[b][r];
            if (sink == 123.456f) tau = sink;   // keeps the accumulators live
            continue;
        }
        float bmax = Bm[(tile & 1) * 4];
        if (BI > 64) bmax = fmaxf(bmax, Bm[(tile & 1) * 4 + 1]);
#pragma unroll
        for (int b = 0; b < NIB; ++b) {
            // cheap conservative filter: max over this lane's 16 rows plus the tile's largest bias.
            // fl(acc + bias) <= fl(max acc + max bias) (rounding is monotonic) => no false negatives.
            float m = acc[b][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[b][r]);
            if (__ballot(m + bmax >= tau) != 0ull) {
                // rare path: exact scores, bitmask of this lane's passing rows, then one insertion per
                // lane per round (so a user -- lanes l and l+32 -- gains at most 2 keys per round)
                float sc[16];
                u32 hm = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    sc[r] = (acc[b][r] + bb[row]) + 0.0f;
                    hm |= (sc[r] >= tau) ? (1u << r) : 0u;
                }
                while (__ballot(hm != 0u) != 0ull) {
                    const bool pend = hm != 0u;
                    const int r = pend ? (__ffs((int)hm) - 1) : 0;
                    hm &= hm - 1u;
                    float sv = sc[0];
#pragma unroll
                    for (int q = 1; q < 16; ++q) sv = (r == q) ? sc[q] : sv;
                    int64_t il = (int64_t)tile * BI + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool v = pend && uvalid && il < I && (sv >= tau);
                    const int32_t g = (int32_t)(p.item_offset + il);
                    // slot bookkeeping in registers: the two lanes of a user exchange their insert flags with one
                    // v_permlane32_swap (no LDS atomic, no wait); lane l writes first, lane l+32 after it
                    const u32 pv = el_partner32(v ? 1u : 0u, hi);
                    if (v) keys[(size_t)uslot * CAP + ucnt + (hi ? (int)pv : 0)] = el_make_key(sv, g);
                    ucnt += (v ? 1 : 0) + (int)pv;
                    u64 full = __ballot(hi == 0 && ucnt > CAP - 2);
                    while (full) {
                        const int ul = __ffsll((long long)full) - 1;
                        full &= full - 1ull;
                        const int64_t ue0 = __shfl(e0, ul, 64), ue1 = __shfl(e1, ul, 64);
                        const int un = __shfl(ucnt, ul, 64);
                        int nn = 0;
                        const float ntau = el_wave_compact_excl(wkeys + (size_t)ul * CAP, un, nn, CAP, p.k, lane,
                                                                p.excl_indices, ue0, ue1);
                        if (col == ul) {
                            tau = ntau;
                            ucnt = nn;
                        }
                    }
                }
            }
        }
    }

    // ---- final sort + write-out ---------------------------------------------------
    for (int ul = 0; ul < 32; ++ul) {
        if ((int64_t)blockIdx.x * UPB + wave * 32 + ul >= nlist) break;
        const int64_t uu = __shfl(user, ul, 64);
        u64* kb = wkeys + (size_t)ul * CAP;
        const int64_t ue0 = __shfl(e0, ul, 64), ue1 = __shfl(e1, ul, 64);
        const int un = __shfl(ucnt, ul, 64);
        int nv = 0;
        el_wave_compact_excl(kb, un, nv, CAP, p.k, lane, p.excl_indices, ue0, ue1);
        int64_t orow = (uu - p.u_start) * (int64_t)p.k;
        int64_t fill_lo = p.item_offset, fill_hi = p.item_offset + I;
        if (nsplit > 1) {                            // partial list of this item slice; pads with the slice's own masked items
            orow = ((int64_t)blockIdx.y * p.part_stride + (int64_t)blockIdx.x * UPB + wave * 32 + ul) * (int64_t)p.k;
            fill_lo = p.item_offset + (int64_t)tile0 * BI;
            const int64_t hi_raw = p.item_offset + (int64_t)ntiles * BI;
            fill_hi = hi_raw < fill_hi ? hi_raw : fill_hi;
        }
        for (int t = lane; t < p.k; t += 64) {
            int32_t oi;
            float ov;
            if (t < nv) {
                u64 key = kb[t];
                oi = el_key_item(key);
                ov = el_key_score(key);
            } else {
                oi = el_fill_masked_range(p, fill_lo, fill_hi, ue0, ue1, 0, 0, t - nv);
                ov = -INFINITY;
            }
            p.out_idx[orow + t] = oi;
            p.out_val[orow + t] = ov;
        }
        el_wave_lds_sync();
    }
}

// =====================================================================================
// merge of G partial lists per user
// =====================================================================================
// row_map / n_rows / row_skip (optional): partial row u belongs to list entry u + row_skip (absent if >= *n_rows) and is
// written to out row row_map[u + row_skip]
__global__ __launch_bounds__(64) void k_topk_merge(const int32_t* parts_idx, const float* parts_val, int G,
                                                   int64_t n_users, int k, int cap, int32_t* out_idx,
                                                   float* out_val, const int32_t* row_map, const int32_t* n_rows, int row_skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    const int lane = threadIdx.x;
    const int64_t u = blockIdx.x;
    if (n_rows && u + row_skip >= (int64_t)*n_rows) return;
    const int64_t orow = row_map ? (int64_t)row_map[u + row_skip] : u;
    const int total = G * k;
    for (int t = lane; t < cap; t += 64) {
        u64 key = 0ull;
        if (t < total) {
            int g = t / k, r = t - g * k;
            int64_t src = ((int64_t)g * n_users + u) * k + r;
            int32_t idx = parts_idx[src];
            float v = parts_val[src];
            if (idx >= 0 && v == v) key = el_make_key(v, idx);
        }
        keys[t] = key;
    }
    el_wave_lds_sync();
    el_wave_bitonic_desc(keys, cap, lane);
    for (int t = lane; t < k; t += 64) {
        u64 key = keys[t];
        int32_t oi = -1;
        float ov = -INFINITY;
        if (key != 0ull) {
            oi = el_key_item(key);
            ov = el_key_score(key);
        }
        out_idx[orow * k + t] = oi;
        out_val[orow * k + t] = ov;
    }
}

// =====================================================================================
// fp64 variant (BPRMF NumPy model): wave-per-user, (ord64(score), ~item) pairs
// =====================================================================================
struct TopkParams64 {
    const double* P;
    const double* Q;
    const double* b;
    int64_t u_start, u_stop, item_offset, I_local;
    int F;
    const int64_t* excl_indptr;
    const int32_t* excl_indices;
    const int64_t* cand_indptr;
    const int32_t* cand_indices;
    int k;
    int32_t* out_idx;
    double* out_val;
};

__device__ __forceinline__ bool el_pair_less(u64 pa, u32 sa, u64 pb, u32 sb) {
    return (pa < pb) || (pa == pb && sa < sb);
}

__device__ __forceinline__ void el_wave_bitonic_desc_pair(u64* a, u32* s, int n, int lane) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                int i = 2 * t - (t & (stride - 1));
                int j = i + stride;
                bool desc = ((i & size) == 0);
                u64 x = a[i], y = a[j];
                u32 sx = s[i], sy = s[j];
                bool sw = desc ? el_pair_less(x, sx, y, sy) : el_pair_less(y, sy, x, sx);
                if (sw) {
                    a[i] = y;
                    a[j] = x;
                    s[i] = sy;
                    s[j] = sx;
                }
            }
            el_wave_lds_sync();
        }
    }
}

__global__ __launch_bounds__(64) void k_topk_wave_f64(TopkParams64 p, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* prim = reinterpret_cast<u64*>(smem);        // [cap]
    u32* sec = reinterpret_cast<u32*>(prim + cap);   // [cap]
    const int lane = threadIdx.x;
    const int64_t user = p.u_start + blockIdx.x;
    const int F = p.F;
    const double* pu = p.P + user * (int64_t)F;
    int64_t e0 = 0, e1 = 0, c0 = 0, c1 = 0;
    if (p.excl_indptr) {
        e0 = p.excl_indptr[user];
        e1 = p.excl_indptr[user + 1];
    }
    int64_t ncand = p.I_local;
    if (p.cand_indptr) {
        c0 = p.cand_indptr[user];
        c1 = p.cand_indptr[user + 1];
        ncand = c1 - c0;
    }
    const bool use_excl = (p.excl_indptr != nullptr) && (p.cand_indptr == nullptr);
    int cnt = 0;
    double tau = -INFINITY;
    auto compact = [&]() {
        el_wave_lds_sync();
        for (int t = cnt + lane; t < cap; t += 64) {
            prim[t] = 0ull;
            sec[t] = 0u;
        }
        el_wave_lds_sync();
        el_wave_bitonic_desc_pair(prim, sec, cap, lane);
        cnt = cnt < p.k ? cnt : p.k;
        tau = (cnt >= p.k) ? el_ord2d(prim[p.k - 1]) : -INFINITY;
        el_wave_lds_sync();
    };
    for (int64_t base = 0; base < ncand; base += 64) {
        int64_t pos = base + lane;
        bool valid = pos < ncand;
        int32_t gitem = -1;
        int64_t il = 0;
        if (valid) {
            if (p.cand_indptr) {
                gitem = p.cand_indices[c0 + pos];
                il = (int64_t)gitem - p.item_offset;
                valid = (il >= 0 && il < p.I_local);
            } else {
                il = pos;
                gitem = (int32_t)(p.item_offset + pos);
            }
        }
        double s = 0.0;
        if (valid) {
            const double* qi = p.Q + il * (int64_t)F;
            double acc = 0.0;
            for (int f = 0; f < F; ++f) acc = __builtin_fma(qi[f], pu[f], acc);
            s = (p.b ? acc + p.b[il] : acc) + 0.0;
        }
        bool hit = valid && (s >= tau);
        if (hit && use_excl) hit = !el_row_contains(p.excl_indices, e0, e1, gitem);
        u64 bal = __ballot(hit);
        if (bal) {
            int offp = __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) {
                prim[cnt + offp] = el_d2ord(s);
                sec[cnt + offp] = 0xffffffffu - (u32)gitem;
            }
            cnt += __popcll(bal);
        }
        if (cnt > cap - 64) compact();
    }
    compact();
    const int nv = cnt;
    const int64_t orow = (int64_t)blockIdx.x * p.k;
    TopkParams pf = {};
    pf.item_offset = p.item_offset;
    pf.I_local = p.I_local;
    pf.excl_indptr = p.excl_indptr;
    pf.excl_indices = p.excl_indices;
    pf.cand_indptr = p.cand_indptr;
    pf.cand_indices = p.cand_indices;
    for (int t = lane; t < p.k; t += 64) {
        int32_t oi;
        double ov;
        if (t < nv) {
            oi = (int32_t)(0xffffffffu - sec[t]);
            ov = el_ord2d(prim[t]);
        } else {
            oi = el_fill_masked(pf, e0, e1, c0, c1, t - nv);
            ov = -INFINITY;
        }
        p.out_idx[orow + t] = oi;
        p.out_val[orow + t] = ov;
    }
}

// =====================================================================================
// host side
// =====================================================================================
static int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

static int wave_cap_for_k(int k) {
    int cap = next_pow2(k + 64);
    return cap < 128 ? 128 : cap;
}


static bool mfma_eligible(int F, int k, const void* cand) { return cand == nullptr && F >= 1 && F <= 256 && k >= 1 && k <= 40; }

// el_topk_screen.hip
bool el_topk_screen_eligible(int F, int k, const void* cand);
size_t el_topk_screen_ws_bytes(int64_t n_users, int64_t I_local, int F, int k, int64_t excl_nnz);
int el_topk_screen_run(const TopkParams& p, void* ws, size_t ws_bytes, hipStream_t st, bool items_unchanged);

extern "C" size_t el_score_topk_ws_bytes(int64_t n_users, int64_t I_local, int32_t F, int32_t k, int64_t excl_nnz, int algo) {
    algo &= 0xff;
    if ((algo == EL_TOPK_AUTO || algo == EL_TOPK_SCREEN) && el_topk_screen_eligible(F, k, nullptr) && n_users > 0)
        return el_topk_screen_ws_bytes(n_users, I_local, F, k, excl_nnz);
    return 0;
}

template <int FP, int NIB, int CAP, int KC, int NW, int OCC>
static int launch_mfma(const TopkParams& p, int vec, hipStream_t st) {
    constexpr int LDA = KC + 1, BI = 32 * NIB, UPB = NW * 32;
    constexpr size_t lds = (size_t)(2 * BI * LDA + 2 * BI + 8) * 4 + (size_t)UPB * CAP * 8;
    auto kern = k_score_topk_mfma<FP, NIB, CAP, KC, NW, OCC>;
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t n_users = p.u_stop - p.u_start;
    if (p.ulist && p.ulist_max > 0 && n_users > p.ulist_max) n_users = p.ulist_max;
    unsigned grid = (unsigned)((n_users + UPB - 1) / UPB);
    EL_LAUNCH("k_score_topk_mfma", kern, dim3(grid, p.nsplit > 1 ? p.nsplit : 1), dim3(NW * 64), lds, st, p, vec);
    EL_CHECK_LAUNCH();
    return 0;
}

template <int CAP>
static int dispatch_mfma_fp(const TopkParams& p, int vec, hipStream_t st) {
    if (p.F <= 32) return launch_mfma<32, 4, CAP, 32, 4, 2>(p, vec, st);
    if (p.F <= 64) return launch_mfma<64, 4, CAP, 32, 4, 2>(p, vec, st);
    // (128-item tiles, K chunk 32, 128 users per workgroup, two workgroups per CU: the geometry sweep of round 2 -- 64-item tiles,
    //  K chunk 64, 256 users per workgroup, one workgroup per CU -- is in profiles/r02_*)
    if (p.F <= 128) return launch_mfma<128, 4, CAP, 32, 4, 2>(p, vec, st);
    return launch_mfma<256, 2, CAP, 32, 4, 1>(p, vec, st);
}

static int check_topk_args(const char* fn, int64_t u_start, int64_t u_stop, int64_t I_local, int F, int k,
                           const void* out_idx, const void* out_val) {
    EL_REQUIRE(u_stop >= u_start, "%s: u_stop < u_start", fn);
    EL_REQUIRE(I_local >= 0 && I_local < 0x7fffffffLL, "%s: I_local out of range", fn);
    EL_REQUIRE(F >= 1, "%s: F must be >= 1", fn);
    EL_REQUIRE(k >= 1 && k <= 4032, "%s: k=%d unsupported (1..4032)", fn, k);
    EL_REQUIRE(u_stop == u_start || (out_idx && out_val), "%s: null output", fn);
    return 0;
}

int el_topk_launch_mfma(const TopkParams& p, hipStream_t st) {
    const int vec = (p.F % 4 == 0) && (((uintptr_t)p.Gi) % 16 == 0);
    if (p.k <= 14) return dispatch_mfma_fp<32>(p, vec, st);
    return dispatch_mfma_fp<64>(p, vec, st);
}

// ---- exact top-k of a device-side user list (fallback of the screened path) -----------------------------------------
// Three tiers, all launched unconditionally (the list length lives on the device; surplus workgroups exit at once):
//   entries [0, 64)        exact scores of the whole catalogue into a dense [64][I] buffer (one thread per (item, entry),
//                          bandwidth-bound, microseconds) + the dense wave top-k -- a handful of users is a LATENCY problem
//   entries [64, 64+cap)   fp32 MFMA kernel over LIST_SPLIT item slices + merge
//   entries beyond         fp32 MFMA kernel, one workgroup per 128 users over the whole catalogue
static const int LIST_SPLIT = 64;
static const int LIST_DENSE = 64;

// exact scores of the first min(*ulist_n, LIST_DENSE) listed users against the whole shard: one thread per item keeps (half of)
// its row in registers and walks the users (rows staged in LDS), so the item table is read once.  F <= 256: factors 128..255 are
// a second sweep that CONTINUES each chain from the partial sum the first sweep left in `preds` -- the same k-ordered fma chain
// (the one-thread-per-(item, user) kernel below re-read the 5 M x 256 table 64 times: 108 ms per call at BASELINE configs[4]).
__global__ __launch_bounds__(256) void k_list_scores(TopkParams p, float* __restrict__ preds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* us = reinterpret_cast<float*>(smem);            // [n][F]
    int n = *p.ulist_n;
    if (n > LIST_DENSE) n = LIST_DENSE;
    if (n <= 0) return;
    const int F = p.F;
    for (int t = threadIdx.x; t < n * F; t += 256) {
        const int slot = t / F, f = t - slot * F;
        us[t] = p.Gu[(p.u_start + p.ulist[slot]) * (int64_t)F + f];
    }
    __syncthreads();
    const int64_t il = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (il >= p.I_local) return;
    const float* row = p.Gi + il * (int64_t)F;
    const float bias = p.Bi ? p.Bi[il] : 0.f;
    for (int f0 = 0; f0 < F; f0 += 128) {
        float gi[128];
#pragma unroll
        for (int f = 0; f < 128; ++f) gi[f] = f0 + f < F ? row[f0 + f] : 0.f;
        const bool last = f0 + 128 >= F;
        for (int slot = 0; slot < n; ++slot) {
            const float* gu = us + slot * F + f0;
            float a = f0 ? preds[(int64_t)slot * p.I_local + il] : 0.f;
#pragma unroll
            for (int f = 0; f < 128; ++f)
                if (f0 + f < F) a = __builtin_fmaf(gi[f], gu[f], a);
            preds[(int64_t)slot * p.I_local + il] = last ? (p.Bi ? a + bias : a) + 0.0f : a;
        }
    }
}

// same result for any F: one thread per (item, listed user), the item row is re-read per user (L2)
__global__ __launch_bounds__(256) void k_list_scores_wide(TopkParams p, float* __restrict__ preds) {
    const int slot = blockIdx.y;
    int n = *p.ulist_n;
    if (n > LIST_DENSE) n = LIST_DENSE;
    if (slot >= n) return;
    const int64_t il = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (il >= p.I_local) return;
    const float* gi = p.Gi + il * (int64_t)p.F;
    const float* gu = p.Gu + (p.u_start + p.ulist[slot]) * (int64_t)p.F;
    float a = 0.f;
    for (int f = 0; f < p.F; ++f) a = __builtin_fmaf(gi[f], gu[f], a);
    preds[(int64_t)slot * p.I_local + il] = (p.Bi ? a + p.Bi[il] : a) + 0.0f;
}

static int64_t list_cap_for(int64_t n_users) {
    int64_t c = n_users / 16;
    if (c < 512) c = 512;
    if (c > n_users) c = n_users;
    return c < LIST_DENSE ? LIST_DENSE : c;       // (the dense tier borrows the split scratch for its partial lists)
}

static size_t a256_(size_t x) { return (x + 255) & ~(size_t)255; }

size_t el_topk_list_scratch_bytes(int64_t n_users, int64_t I_local, int k) {
    return 2 * a256_((size_t)LIST_SPLIT * (size_t)list_cap_for(n_users) * (size_t)k * 4) + a256_((size_t)LIST_DENSE * (size_t)I_local * 4);
}

int el_topk_run_list(const TopkParams& p0, void* scratch, size_t scratch_bytes, hipStream_t st) {
    const int64_t n_users = p0.u_stop - p0.u_start;
    const int64_t cap = list_cap_for(n_users);
    EL_REQUIRE(scratch && scratch_bytes >= el_topk_list_scratch_bytes(n_users, p0.I_local, p0.k), "el_topk_run_list: scratch too small");
    int32_t* part_idx = (int32_t*)scratch;
    float* part_val = (float*)((char*)scratch + a256_((size_t)LIST_SPLIT * (size_t)cap * (size_t)p0.k * 4));
    float* preds = (float*)((char*)scratch + 2 * a256_((size_t)LIST_SPLIT * (size_t)cap * (size_t)p0.k * 4));
    if (p0.I_local > 0) {                            // tier 1: entries [0, LIST_DENSE)
        TopkParams d = p0;
        d.ulist_skip = 0;
        d.ulist_max = LIST_DENSE;
        d.nsplit = 0;
        if (p0.F <= 256)
            EL_LAUNCH("k_list_scores", k_list_scores, dim3((unsigned)((p0.I_local + 255) / 256)), dim3(256), (size_t)LIST_DENSE * p0.F * 4, st,
                      d, preds);
        else
            EL_LAUNCH("k_list_scores", k_list_scores_wide, dim3((unsigned)((p0.I_local + 255) / 256), LIST_DENSE), dim3(256), 0, st, d, preds);
        d.preds = preds;
        d.ld = p0.I_local;
        // one wave per (entry, item slice) -> partial lists in the split scratch (free until tier 2 runs) -> merge
        int DS = (int)((p0.I_local + 2047) / 2048);
        if (DS > LIST_SPLIT) DS = LIST_SPLIT;
        d.nsplit = DS > 1 ? DS : 0;
        if (DS > 1) {
            d.part_stride = LIST_DENSE;
            d.out_idx = part_idx;
            d.out_val = part_val;
        }
        const int wcap = wave_cap_for_k(p0.k);
        EL_LAUNCH("k_topk_wave", k_topk_wave<true>, dim3(LIST_DENSE, DS > 1 ? DS : 1), dim3(64), (size_t)wcap * 8 + 16, st, d, wcap);
        if (DS > 1) {
            int mcap = next_pow2(DS * p0.k);
            if (mcap < 64) mcap = 64;
            EL_LAUNCH("k_topk_merge", k_topk_merge, dim3(LIST_DENSE), dim3(64), (size_t)mcap * 8, st, (const int32_t*)part_idx,
                      (const float*)part_val, DS, (int64_t)LIST_DENSE, p0.k, mcap, p0.out_idx, p0.out_val, p0.ulist, p0.ulist_n, 0);
        }
        EL_CHECK_LAUNCH();
    }
    const int skip1 = p0.I_local > 0 ? LIST_DENSE : 0;
    if (!mfma_eligible(p0.F, p0.k, p0.cand_indptr)) {       // large k: the wave kernel takes the rest of the list
        if (n_users > skip1) {
            const int wcap = wave_cap_for_k(p0.k);
            int WS = (int)((p0.I_local + 1023) / 1024);             // item slices: one wave per (entry, slice) + merge
            if (WS > LIST_SPLIT) WS = LIST_SPLIT;
            while (WS > 1 && WS * p0.k > 8192) --WS;                // merge buffer limit
            TopkParams q = p0;
            q.ulist_skip = skip1;
            q.ulist_max = (int)(skip1 + cap);
            q.nsplit = WS > 1 ? WS : 0;
            if (WS > 1) {
                q.part_stride = cap;
                q.out_idx = part_idx;
                q.out_val = part_val;
            }
            const int64_t rows = (n_users - skip1) < cap ? (n_users - skip1) : cap;
            EL_LAUNCH("k_topk_wave", k_topk_wave<false>, dim3((unsigned)rows, WS > 1 ? WS : 1), dim3(64), (size_t)wcap * 8 + 16, st, q, wcap);
            if (WS > 1) {
                int mcap = next_pow2(WS * p0.k);
                if (mcap < 64) mcap = 64;
                EL_LAUNCH("k_topk_merge", k_topk_merge, dim3((unsigned)rows), dim3(64), (size_t)mcap * 8, st, (const int32_t*)part_idx,
                          (const float*)part_val, WS, cap, p0.k, mcap, p0.out_idx, p0.out_val, p0.ulist, p0.ulist_n, skip1);
            }
            if (n_users > skip1 + cap) {                             // beyond the split scratch: unsplit
                TopkParams r = p0;
                r.ulist_skip = (int)(skip1 + cap);
                r.ulist_max = 0;
                r.nsplit = 0;
                EL_LAUNCH("k_topk_wave", k_topk_wave<false>, dim3((unsigned)(n_users - skip1 - cap)), dim3(64), (size_t)wcap * 8 + 16, st, r, wcap);
            }
            EL_CHECK_LAUNCH();
        }
        return 0;
    }
    // item slices of >= 4 MFMA tiles (128 items each), at most LIST_SPLIT of them
    int64_t S = (p0.I_local + 511) / 512;
    if (S > LIST_SPLIT) S = LIST_SPLIT;
    if (S < 1) S = 1;
    TopkParams p = p0;
    p.ulist_skip = skip1;
    p.ulist_max = (int)(skip1 + cap);
    p.nsplit = (int)(S > 1 ? S : 0);
    if (S > 1) {
        p.part_stride = cap;
        p.out_idx = part_idx;
        p.out_val = part_val;
    }
    if (int rc = el_topk_launch_mfma(p, st)) return rc;
    if (S > 1) {
        int mcap = next_pow2((int)S * p0.k);
        if (mcap < 64) mcap = 64;
        EL_LAUNCH("k_topk_merge", k_topk_merge, dim3((unsigned)cap), dim3(64), (size_t)mcap * 8, st, (const int32_t*)part_idx,
                  (const float*)part_val, (int)S, cap, p0.k, mcap, p0.out_idx, p0.out_val, p0.ulist, p0.ulist_n, skip1);
        EL_CHECK_LAUNCH();
    }
    if (n_users > skip1 + cap) {                     // more flagged users than the split scratch holds: plain kernel for the rest
        TopkParams q = p0;
        q.ulist_skip = (int)(skip1 + cap);
        q.ulist_max = 0;
        q.nsplit = 0;
        if (int rc = el_topk_launch_mfma(q, st)) return rc;
    }
    return 0;
}

extern "C" int el_score_topk(el_ctx* ctx, void* stream, const float* Gu, const float* Gi, const float* Bi,
                             int64_t u_start, int64_t u_stop, int64_t item_offset, int64_t I_local, int32_t F,
                             const int64_t* excl_indptr, const int32_t* excl_indices, const int64_t* cand_indptr,
                             const int32_t* cand_indices, int32_t k, int32_t* out_idx, float* out_val, int algo,
                             void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    // el_topk_screen_stats describes the LAST el_score_topk call: cleared here, set again only by a screened run that completed (a call
    // on another route, a failed call or a freed workspace must not leave pointers of an earlier call behind)
    ctx->scr_cnt = nullptr, ctx->scr_flagged = nullptr, ctx->scr_users = 0;
    if (int rc = check_topk_args("el_score_topk", u_start, u_stop, I_local, F, k, out_idx, out_val)) return rc;
    const bool items_unchanged = (algo & EL_TOPK_ITEMS_UNCHANGED) != 0;
    algo &= 0xff;
    EL_REQUIRE(Gu && Gi, "el_score_topk: null factor table");
    EL_REQUIRE((excl_indptr == nullptr) == (excl_indices == nullptr) || excl_indptr != nullptr,
               "el_score_topk: excl_indices without excl_indptr");
    EL_REQUIRE((cand_indptr == nullptr) == (cand_indices == nullptr), "el_score_topk: cand CSR needs both arrays");
    if (u_stop == u_start) return 0;
    hipStream_t st = (hipStream_t)stream;
    TopkParams p;
    memset(&p, 0, sizeof(p));
    p.Gu = Gu;
    p.Gi = Gi;
    p.Bi = Bi;
    p.u_start = u_start;
    p.u_stop = u_stop;
    p.item_offset = item_offset;
    p.I_local = I_local;
    p.F = F;
    p.excl_indptr = excl_indptr;
    p.excl_indices = excl_indices;
    p.cand_indptr = cand_indptr;
    p.cand_indices = cand_indices;
    p.k = k;
    p.out_idx = out_idx;
    p.out_val = out_val;
    p.dbg = 0;
    const bool selig = el_topk_screen_eligible(F, k, cand_indptr);
    if (algo == EL_TOPK_SCREEN) EL_REQUIRE(selig, "el_score_topk: screened kernel needs F<=256, k<=128 and no candidate list");
    if (algo == EL_TOPK_SCREEN ||
        (algo == EL_TOPK_AUTO && selig && ws != nullptr && ws_bytes >= el_topk_screen_ws_bytes(u_stop - u_start, I_local, F, k, 0)))
        return el_topk_screen_run(p, ws, ws_bytes, st, items_unchanged);
    bool elig = mfma_eligible(F, k, cand_indptr);
    if (algo == EL_TOPK_MFMA) EL_REQUIRE(elig, "el_score_topk: MFMA kernel needs F<=256, k<=40 and no candidate list");
    bool use_mfma = (algo == EL_TOPK_MFMA) || (algo == EL_TOPK_AUTO && elig);
    if (use_mfma) return el_topk_launch_mfma(p, st);
    int cap = wave_cap_for_k(k);
    EL_LAUNCH("k_topk_wave", k_topk_wave<false>, dim3((unsigned)(u_stop - u_start)), dim3(64), (size_t)cap * 8 + 16, st, p, cap);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_dense_topk(el_ctx* ctx, void* stream, const float* preds, int64_t ld, int64_t u_start,
                             int64_t u_stop, int64_t I, const int64_t* excl_indptr, const int32_t* excl_indices,
                             const int64_t* cand_indptr, const int32_t* cand_indices, int32_t k, int32_t* out_idx,
                             float* out_val) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_topk_args("el_dense_topk", u_start, u_stop, I, 1, k, out_idx, out_val)) return rc;
    EL_REQUIRE(preds && ld >= I, "el_dense_topk: bad preds/ld");
    if (u_stop == u_start) return 0;
    TopkParams p;
    memset(&p, 0, sizeof(p));
    p.u_start = u_start;
    p.u_stop = u_stop;
    p.item_offset = 0;
    p.I_local = I;
    p.F = 1;
    p.excl_indptr = excl_indptr;
    p.excl_indices = excl_indices;
    p.cand_indptr = cand_indptr;
    p.cand_indices = cand_indices;
    p.k = k;
    p.out_idx = out_idx;
    p.out_val = out_val;
    p.preds = preds;
    p.ld = ld;
    int cap = wave_cap_for_k(k);
    EL_LAUNCH("k_topk_wave", k_topk_wave<true>, dim3((unsigned)(u_stop - u_start)), dim3(64), (size_t)cap * 8 + 16,
                       (hipStream_t)stream, p, cap);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_score_topk_f64(el_ctx* ctx, void* stream, const double* P, const double* Q, const double* b,
                                 int64_t u_start, int64_t u_stop, int64_t item_offset, int64_t I_local, int32_t F,
                                 const int64_t* excl_indptr, const int32_t* excl_indices, const int64_t* cand_indptr,
                                 const int32_t* cand_indices, int32_t k, int32_t* out_idx, double* out_val) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_topk_args("el_score_topk_f64", u_start, u_stop, I_local, F, k, out_idx, out_val)) return rc;
    EL_REQUIRE(P && Q, "el_score_topk_f64: null factor table");
    if (u_stop == u_start) return 0;
    TopkParams64 p;
    memset(&p, 0, sizeof(p));
    p.P = P;
    p.Q = Q;
    p.b = b;
    p.u_start = u_start;
    p.u_stop = u_stop;
    p.item_offset = item_offset;
    p.I_local = I_local;
    p.F = F;
    p.excl_indptr = excl_indptr;
    p.excl_indices = excl_indices;
    p.cand_indptr = cand_indptr;
    p.cand_indices = cand_indices;
    p.k = k;
    p.out_idx = out_idx;
    p.out_val = out_val;
    int cap = wave_cap_for_k(k);
    EL_LAUNCH("k_topk_wave_f64", k_topk_wave_f64, dim3((unsigned)(u_stop - u_start)), dim3(64), (size_t)cap * 12,
                       (hipStream_t)stream, p, cap);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_topk_merge(el_ctx* ctx, void* stream, const int32_t* parts_idx, const float* parts_val, int32_t G,
                             int64_t n_users, int32_t k, int32_t* out_idx, float* out_val) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(parts_idx && parts_val && out_idx && out_val, "el_topk_merge: null pointer");
    EL_REQUIRE(G >= 1 && k >= 1 && (int64_t)G * k <= 8192, "el_topk_merge: G*k=%lld unsupported (<=8192)",
               (long long)G * k);
    if (n_users <= 0) return 0;
    int cap = next_pow2(G * k);
    if (cap < 64) cap = 64;
    EL_LAUNCH("k_topk_merge", k_topk_merge, dim3((unsigned)n_users), dim3(64), (size_t)cap * 8, (hipStream_t)stream,
                       parts_idx, parts_val, G, n_users, k, cap, out_idx, out_val, (const int32_t*)nullptr, (const int32_t*)nullptr, 0);
    EL_CHECK_LAUNCH();
    return 0;
}
