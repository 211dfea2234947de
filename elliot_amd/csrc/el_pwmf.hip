// Point-wise factor models (MF, PMF, FunkSVD, LogisticMF) -- SURVEY 8f N3; contract in include/elliot_hip.h.
//
// One step = five launches + the optimiser passes:
//   1. k_pw_fwd        one lane group per sample: gather gamma_u, gamma_i (+ biases), x = <gamma_u, gamma_i> + b,
//                      link, loss, c_b = dloss/dx; also emits the (row id, sample) sort pairs
//   2. ONE rocprim radix sort of (user, b) and (U + item, b)   (stable -> a segment is summed in batch order)
//   3. k_pw_seg (users) dGu[u] = sum_{b in segment} c_b gamma_i(b) (+ cnt l_w gamma_u),  dBu[u] = sum c_b
//      k_pw_seg (items) dGi[i] = sum_{b in segment} c_b gamma_u(b) (+ cnt l_w gamma_i),  dBi[i] = sum c_b
//      Segments are cut into chunks of `chunk` sorted positions per lane group; a segment that lives inside one chunk is
//      written with plain stores, the pieces of one that crosses chunk borders (popular items) go to partial slots and are added
//      in a fixed order by k_seg_combine / k_seg_combine_long (el_segcombine.h; round 6: no floating-point atomics -- the same
//      gradient bits on every run).
//   4. k_adam_dense / k_adagrad_dense over every variable of the updated side (TF 2.3 sparse-apply semantics).
// The layout of the work (lane group per row, 16-byte lanes, chunks sized so that >= 16K groups are in flight) is the one
// measured for the BPR step (el_bpr_sorted.hip); the BPR kernels fuse the forward into the user segments, which a
// point-wise sample cannot do (its c_b is needed by two different segment owners).
#include "el_common.h"
#include "el_segcombine.h"
#include "el_topk_common.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

int el_pick_lpt(int F, int vw, int* cpl);                                       // el_bpr.hip
__global__ void k_adam_dense(float* th, float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps);

namespace {

template <int VW>
__device__ __forceinline__ void pw_ld(const float* p, float* dst) {
    if (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        dst[0] = t.x, dst[1] = t.y, dst[2] = t.z, dst[3] = t.w;
    } else {
        dst[0] = p[0];
    }
}
template <int VW>
__device__ __forceinline__ void pw_st(float* p, const float* src) {
    if (VW == 4)
        *reinterpret_cast<float4*>(p) = make_float4(src[0], src[1], src[2], src[3]);
    else
        p[0] = src[0];
}

__device__ __forceinline__ float pw_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float pw_softplus(float x) {           // log(1 + exp(x)) without overflow
    if (x > 15.0f) return x + expf(-x);
    return log1pf(expf(x));
}

struct PwFwd {
    el_pwmf_state st;
    const int32_t* bu;
    const int32_t* bi;
    const float* label;
    float* coef;          // [n] dloss/dx
    float* out;           // forward-only output
    u32 *keyU, *valU, *keyI, *valI;
    u32 item_key_off;     // both sides are ordered by ONE radix sort: users as u, items as U + i
    int64_t n;
    int lpt;
    float inv_n;
    double* loss_out;
};

template <int VW, int CPL, bool TRAIN>
__global__ __launch_bounds__(256) void k_pw_fwd(PwFwd p) {
    const int F = p.st.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t groups = (int64_t)gridDim.x * blockDim.x / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    double lossv = 0.0;
    // persistent lane groups: the batch loss leaves each workgroup as ONE double atomic (131 072 single-sample workgroups
    // queueing on that address cost 1.3 ms at B = 1M)
    for (int64_t b = gid / lpt; b < p.n; b += groups) {
        const int32_t u = p.bu[b], i = p.bi[b];
        const float* pu = p.st.Gu + (int64_t)u * F;
        const float* pi = p.st.Gi + (int64_t)i * F;
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
                float a[VW], c[VW];
                pw_ld<VW>(pu + e, a);
                pw_ld<VW>(pi + e, c);
#pragma unroll
                for (int x = 0; x < VW; ++x) dot = fmaf(a[x], c[x], dot);
            }
        }
        float sq = 0.f;
        if (TRAIN && p.st.kind == EL_PW_LOGISTIC && p.st.l_w != 0.f) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
                if (e < F) {
                    float a[VW], c[VW];
                    pw_ld<VW>(pu + e, a);
                    pw_ld<VW>(pi + e, c);
#pragma unroll
                    for (int x = 0; x < VW; ++x) sq += a[x] * a[x] + c[x] * c[x];
                }
            }
            sq = el_group_sum(sq, lpt);
        }
        dot = el_group_sum(dot, lpt);
        if (sub == 0) {
            float x = dot;
            if (p.st.Bu) x += p.st.Bu[u] + p.st.Bi[i];
            const int kind = p.st.kind;
            const float o = (kind == EL_PW_MSE_SIGMOID) ? pw_sigmoid(x) : x;
            if (!TRAIN) {
                p.out[b] = o;
            } else {
                const float y = p.label[b];
                float c;
                if (kind == EL_PW_LOGISTIC) {
                    const float w = 1.0f + p.st.alpha * y;
                    lossv += w * pw_softplus(x) - p.st.alpha * y * x + 0.5f * p.st.l_w * sq;
                    c = w * pw_sigmoid(x) - p.st.alpha * y;
                } else {
                    const float r = o - y;
                    lossv += r * r * p.inv_n;
                    c = 2.0f * r * p.inv_n;
                    if (kind == EL_PW_MSE_SIGMOID) c *= o * (1.0f - o);
                }
                p.coef[b] = c;
                p.keyU[b] = (u32)u, p.valU[b] = (u32)b;
                p.keyI[b] = (u32)i + p.item_key_off, p.valI[b] = (u32)b;
            }
        }
    }
    if (TRAIN) {
        __shared__ double part[4];
        double l = lossv;
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) part[wv] = l;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = (part[0] + part[1]) + (part[2] + part[3]);
            if (t != 0.0) atomicAdd(p.loss_out, t);
        }
    }
}

#define PW_STG 64

struct PwSeg {
    const u32* keys;          // sorted row ids of THIS side (+ key_off)
    u32 key_off;
    const u32* vals;          // sample of each sorted position
    const float* coef;        // [n]
    const int32_t* other_ids; // the sample's row in the other table
    const float* other;       // the other table
    const float* own;         // this side's table (L2 term), used when l_w != 0
    float* g;                 // this side's gradient accumulator
    float* gb;                // bias gradient accumulator or NULL
    int64_t n;
    int F, chunk, lpt;
    float l_w;
    // pieces of segments cut by chunk boundaries (el_segcombine.h)
    float* part;
    float* part_b;
    int32_t* split;
};

// finish of a combined row: the gradient row and its bias part, stored
struct PwFinish {
    float* g;
    float* gb;
    int F;
    template <int CPL, int VW>
    __device__ __forceinline__ void operator()(int64_t row, int sub, int lpt, const float (&gg)[CPL][VW], float b) const {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) sc_st<VW>(g + row * F + e, gg[q]);
        }
        if (sub == 0 && gb) gb[row] = b;
    }
};

template <int VW, int CPL>
__global__ __launch_bounds__(256) void k_pw_seg(PwSeg p) {
    const int F = p.F, lpt = p.lpt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    const int64_t p0 = grp * p.chunk;
    if (p0 >= p.n) return;
    const int64_t p1 = (p0 + p.chunk < p.n) ? p0 + p.chunk : p.n;
    int64_t cur = -1;
    bool started_inside = false;
    float acc[CPL][VW];
    float bacc = 0.f;
    int cnt = 0;
    auto flush = [&](bool ends_inside) {
        const bool plain = started_inside && ends_inside;
        // a cut piece: slot 2 grp + 1 when the segment starts in this chunk (the head piece: it lists the row), 2 grp otherwise
        const int64_t slot = 2 * grp + (started_inside ? 1 : 0);
        float* g = plain ? p.g + cur * F : p.part + slot * F;
        const float w = (float)cnt * p.l_w;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) {
                float v[VW];
#pragma unroll
                for (int x = 0; x < VW; ++x) v[x] = acc[q][x];
                if (w != 0.f) {
                    float r[VW];
                    pw_ld<VW>(p.own + cur * F + e, r);
#pragma unroll
                    for (int x = 0; x < VW; ++x) v[x] += w * r[x];
                }
                pw_st<VW>(g + e, v);
            }
        }
        if (sub == 0) {
            if (plain) {
                if (p.gb) p.gb[cur] = bacc;
            } else {
                p.part_b[slot] = bacc;                         // (k_seg_combine finds the head pieces from the keys: no list)
            }
        }
    };
    // The index chain of a position (sorted key, sample -> coefficient and the other table's row id) is two dependent
    // global loads deep; walking it position by position in front of every row gather made this kernel latency-bound.
    // A group therefore stages PW_STG positions at once (all chains in flight together, coalesced) in LDS and walks them
    // from there, so that only the row gathers remain on the critical path.
    __shared__ u32 s_key[32][PW_STG];
    __shared__ u32 s_oth[32][PW_STG];
    __shared__ float s_cf[32][PW_STG];
    const int gl = (int)(threadIdx.x / lpt);
    constexpr int SUB = (CPL == 1) ? 4 : (CPL == 2 ? 2 : 1);     // positions whose row loads are in flight together
    for (int64_t sbase = p0; sbase < p1; sbase += PW_STG) {
        const int cs = (int)((p1 - sbase < PW_STG) ? p1 - sbase : PW_STG);
        for (int t = sub; t < cs; t += lpt) {
            const int64_t b = (int64_t)p.vals[sbase + t];
            s_key[gl][t] = p.keys[sbase + t] - p.key_off;
            s_cf[gl][t] = p.coef[b];
            s_oth[gl][t] = (u32)p.other_ids[b];
        }
        el_wave_lds_sync();
        for (int base = 0; base < cs; base += SUB) {
            int64_t keyv[SUB];
            float cv[SUB];
            bool okv[SUB];
            float rr[SUB][CPL][VW];
#pragma unroll
            for (int t = 0; t < SUB; ++t) {
                okv[t] = base + t < cs;
                const int tt = okv[t] ? base + t : base;
                keyv[t] = (int64_t)s_key[gl][tt];
                cv[t] = s_cf[gl][tt];
                const float* po = p.other + (int64_t)s_oth[gl][tt] * F;
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * VW;
#pragma unroll
                    for (int x = 0; x < VW; ++x) rr[t][q][x] = 0.f;
                    if (okv[t] && e < F) pw_ld<VW>(po + e, rr[t][q]);
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
                    cnt = 0;
                    bacc = 0.f;
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) acc[q][x] = fmaf(cv[t], rr[t][q][x], acc[q][x]);
                bacc += cv[t];
                ++cnt;
            }
        }
        el_wave_lds_sync();
    }
    flush(p1 == p.n || (int64_t)(p.keys[p1] - p.key_off) != cur);
}

// Keras Adagrad over a whole variable; rows without a gradient are left as they are (g = 0 changes nothing), which is
// what the sparse apply of the reference does.  Clears the accumulator it consumed.
__global__ __launch_bounds__(256) void k_adagrad_dense(float* __restrict__ th, float* __restrict__ g, float* __restrict__ acc,
                                                       int64_t n, float lr, float eps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float gg = g[e];
        if (gg != 0.f) {
            const float a = acc[e] + gg * gg;
            acc[e] = a;
            th[e] -= lr * gg / (sqrtf(a) + eps);
            g[e] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void k_pw_link(float* __restrict__ vals, int64_t n_rows, int64_t ld, int k, int kind,
                                                 const float* __restrict__ Bu, int64_t u_start) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * k) return;
    const int64_t r = t / k;
    const int c = (int)(t - r * k);
    float v = vals[r * ld + c];
    if (v == -INFINITY) return;
    if (Bu) v += Bu[u_start + r];
    if (kind == EL_PW_MSE_SIGMOID) v = pw_sigmoid(v);
    vals[r * ld + c] = v;
}

// rows of (idx, vals) re-ordered in place by (value desc, index asc) -- the order tf.nn.top_k gives -- over their first kk <= 64
// entries: one wave per row, the pair packed into the order-preserving key of the top-k kernels, bitonic sort in LDS
__global__ __launch_bounds__(256) void k_rerank_rows(int32_t* __restrict__ idx, float* __restrict__ vals, int64_t n_rows, int64_t ld,
                                                     int kk) {
    __shared__ u64 keys[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= n_rows) return;
    u64* kb = keys[wv];
    kb[lane] = lane < kk ? el_make_key(vals[r * ld + lane], idx[r * ld + lane]) : 0ull;
    el_wave_lds_sync();
    el_wave_bitonic_desc(kb, 64, lane);
    if (lane < kk) {
        const u64 k = kb[lane];
        idx[r * ld + lane] = el_key_item(k);
        vals[r * ld + lane] = el_key_score(k);
    }
}

// the same for lists of up to 4096 entries (the link / re-score margins of point-wise models and CML grow the lists up to
// 4032): one wave per row and workgroup, key buffer of cap = 2^m >= kk slots in dynamic LDS
__global__ __launch_bounds__(64) void k_rerank_rows_long(int32_t* __restrict__ idx, float* __restrict__ vals, int64_t n_rows, int64_t ld,
                                                         int kk, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rr_lds[];
    u64* kb = reinterpret_cast<u64*>(rr_lds);
    const int lane = threadIdx.x;
    const int64_t r = blockIdx.x;
    for (int t = lane; t < cap; t += 64) kb[t] = t < kk ? el_make_key(vals[r * ld + t], idx[r * ld + t]) : 0ull;
    el_wave_lds_sync();
    el_wave_bitonic_desc(kb, cap, lane);
    for (int t = lane; t < kk; t += 64) {
        const u64 k = kb[t];
        idx[r * ld + t] = el_key_item(k);
        vals[r * ld + t] = el_key_score(k);
    }
}

int bits_for(int64_t n) {
    int b = 1;
    while ((1LL << b) < n && b < 32) ++b;
    return b;
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct PwWs {
    u32 *keyU_in, *valU_in, *keyU, *valU, *keyI_in, *valI_in, *keyI, *valI;
    float* coef;
    void* tmp;
    size_t tmp_bytes, total;
    float *part, *part_b;          // partial rows of cut segments: 2 per lane group of the finer side
    int32_t *split, *split_long;
};

int item_chunk(int64_t n);
int user_chunk(int64_t n);

int carve(int64_t n, int64_t U, int64_t I, int F, char* base, PwWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    u32** slots[4] = {&w->keyU_in, &w->valU_in, &w->keyU, &w->valU};      // users [0, n), items [n, 2n) of one array each
    u32** islots[4] = {&w->keyI_in, &w->valI_in, &w->keyI, &w->valI};
    for (int k = 0; k < 4; ++k) {
        *slots[k] = (u32*)take((size_t)n * 8);
        *islots[k] = base ? *slots[k] + n : nullptr;
    }
    w->coef = (float*)take((size_t)n * 4);
    size_t t1 = 0, t2 = 0, t3 = 0;
    u32* np = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, t1, np, np, np, np, (unsigned)(2 * n), 0, bits_for(U + I), (hipStream_t)0) != hipSuccess) return 1;
    if (rocprim::radix_sort_pairs(nullptr, t2, np, np, np, np, (unsigned)n, 0, bits_for(U), (hipStream_t)0) != hipSuccess) return 1;
    if (rocprim::radix_sort_pairs(nullptr, t3, np, np, np, np, (unsigned)n, 0, bits_for(U + I), (hipStream_t)0) != hipSuccess) return 1;
    if (t2 > t1) t1 = t2;
    if (t3 > t1) t1 = t3;
    w->tmp_bytes = t1;
    w->tmp = take(w->tmp_bytes);
    const int cu = user_chunk(n), ci = item_chunk(n);
    const int64_t groups = (n + (cu < ci ? cu : ci) - 1) / (cu < ci ? cu : ci) + 1;
    w->part = (float*)take((size_t)2 * groups * (size_t)(F > 0 ? F : 1) * 4);
    w->part_b = (float*)take((size_t)2 * groups * 4);
    w->split = (int32_t*)take((size_t)(2 * groups + 4) * 4);
    w->split_long = (int32_t*)take((size_t)(2 * groups + 4) * 4);
    w->total = off;
    return 0;
}

// positions per lane group: long enough that a popular item's segment ends in few atomic flushes, short enough that
// >= 16K groups exist (el_bpr_sorted.hip measured 16 -> 128 positions: 0.84 -> 0.35 ms on the item side at B = 1M)
int item_chunk(int64_t n) {
    const int64_t c = n / 16384;
    return (int)(c < 16 ? 16 : (c > 256 ? 256 : c));
}
int user_chunk(int64_t n) {
    const int64_t c = n / 65536;
    return (int)(c < 4 ? 4 : (c > 16 ? 16 : c));
}

unsigned stream_grid(el_ctx* ctx, int64_t n) {
    const int64_t blocks = (n + 255) / 256, cap = (int64_t)ctx->cus * 8;
    return (unsigned)(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

int check_state(const el_pwmf_state* st, const char* who, bool train, int opt) {
    EL_REQUIRE(st != nullptr, "%s: null state", who);
    EL_REQUIRE(st->U > 0 && st->I > 0 && st->F > 0, "%s: bad shape U=%lld I=%lld F=%d", who, (long long)st->U, (long long)st->I, st->F);
    EL_REQUIRE(st->kind >= EL_PW_MSE && st->kind <= EL_PW_LOGISTIC, "%s: unknown kind %d", who, st->kind);
    EL_REQUIRE(st->Gu && st->Gi, "%s: null tables", who);
    EL_REQUIRE((st->Bu == nullptr) == (st->Bi == nullptr), "%s: Bu and Bi must both be given or both be NULL", who);
    if (train) {
        EL_REQUIRE(st->gGu && st->gGi && st->mGu && st->mGi, "%s: null gradient / optimiser buffers", who);
        EL_REQUIRE(opt == EL_PW_ADAGRAD || (st->vGu && st->vGi), "%s: Adam needs the v slots", who);
        if (st->Bu) {
            EL_REQUIRE(st->gBu && st->gBi && st->mBu && st->mBi, "%s: null bias gradient / optimiser buffers", who);
            EL_REQUIRE(opt == EL_PW_ADAGRAD || (st->vBu && st->vBi), "%s: Adam needs the bias v slots", who);
        }
    }
    return 0;
}

bool rows_vec4(const el_pwmf_state& st) {
    auto ok = [&](const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; };
    return st.F % 4 == 0 && ok(st.Gu) && ok(st.Gi) && ok(st.gGu) && ok(st.gGi);
}

template <int VW, bool TRAIN>
int launch_fwd(PwFwd p, hipStream_t s) {
    int cpl = 1;
    p.lpt = el_pick_lpt(p.st.F, VW, &cpl);
    EL_REQUIRE(cpl <= 4, "el_pwmf: F=%d too large for this build (max %d)", p.st.F, 64 * 4 * VW);
    const int64_t want = (p.n * p.lpt + 255) / 256, cap = (int64_t)g_el_cur_ctx->cus * 16;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    const char* nm = TRAIN ? "k_pw_fwd" : "k_pw_predict";
    if (cpl == 1) EL_LAUNCH(nm, (k_pw_fwd<VW, 1, TRAIN>), dim3(grid), dim3(256), 0, s, p);
    else if (cpl == 2) EL_LAUNCH(nm, (k_pw_fwd<VW, 2, TRAIN>), dim3(grid), dim3(256), 0, s, p);
    else EL_LAUNCH(nm, (k_pw_fwd<VW, 4, TRAIN>), dim3(grid), dim3(256), 0, s, p);
    EL_CHECK_LAUNCH();
    return 0;
}

template <int VW>
int launch_seg(PwSeg p, const PwWs& w, const char* nm, hipStream_t s) {
    int cpl = 1;
    p.lpt = el_pick_lpt(p.F, VW, &cpl);
    p.part = w.part, p.part_b = w.part_b, p.split = w.split;
    const int64_t groups = (p.n + p.chunk - 1) / p.chunk;
    const unsigned grid = (unsigned)((groups * p.lpt + 255) / 256);
    EL_CHECK_HIP(hipMemsetAsync(w.split_long, 0, 4, s));
    SegParts sp = {p.keys, p.key_off, p.n, p.chunk, p.lpt, p.F, nullptr, w.split_long, w.part, w.part_b};
    PwFinish fin = {p.g, p.gb, p.F};
    const unsigned gl = (unsigned)(groups < 1024 ? (groups < 1 ? 1 : groups) : 1024);
#define EL_PW_SEG(CPL_)                                                                                                \
    do {                                                                                                               \
        EL_LAUNCH(nm, (k_pw_seg<VW, CPL_>), dim3(grid), dim3(256), 0, s, p);                                           \
        EL_LAUNCH("k_seg_combine", (k_seg_combine<VW, CPL_, PwFinish>), dim3(grid), dim3(256), 0, s, sp, fin);         \
        EL_LAUNCH("k_seg_combine_long", (k_seg_combine_long<VW, CPL_, PwFinish>), dim3(gl), dim3(256), 0, s, sp, fin); \
    } while (0)
    if (cpl == 1) EL_PW_SEG(1);
    else if (cpl == 2) EL_PW_SEG(2);
    else EL_PW_SEG(4);
#undef EL_PW_SEG
    EL_CHECK_LAUNCH();
    return 0;
}

int apply(el_ctx* ctx, hipStream_t s, int opt, const char* nm, float* th, float* g, float* m, float* v, int64_t n, float lr_t) {
    if (opt == EL_PW_ADAM)
        EL_LAUNCH(nm, k_adam_dense, dim3(stream_grid(ctx, n / 4 + 1)), dim3(256), 0, s, th, g, m, v, n, lr_t, 0.9f, 0.999f, 1e-7f);
    else
        EL_LAUNCH(nm, k_adagrad_dense, dim3(stream_grid(ctx, n)), dim3(256), 0, s, th, g, m, n, lr_t, 1e-7f);
    EL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" size_t el_pwmf_ws_bytes(int64_t n, int64_t U, int64_t I, int32_t F) {
    if (n <= 0) return 0;
    PwWs w;
    if (carve(n, U, I, F, nullptr, &w)) return 0;
    return w.total;
}

extern "C" int el_pwmf_forward(el_ctx* ctx, void* stream, const el_pwmf_state* st, const int32_t* u, const int32_t* i,
                               int64_t n, float* out) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_state(st, "el_pwmf_forward", false, 0)) return rc;
    if (n <= 0) return 0;
    EL_REQUIRE(u && i && out, "el_pwmf_forward: null argument");
    PwFwd p;
    memset(&p, 0, sizeof(p));
    p.st = *st, p.bu = u, p.bi = i, p.out = out, p.n = n;
    return rows_vec4(*st) ? launch_fwd<4, false>(p, (hipStream_t)stream) : launch_fwd<1, false>(p, (hipStream_t)stream);
}

// forward + loss + c_b + sorted-segment gradient sums into the accumulators of `side` (mean losses divide by n_div)
static int pw_grads(el_ctx* ctx, hipStream_t s, const el_pwmf_state* stp, const int32_t* u, const int32_t* i, const float* label,
                    int64_t n, int64_t n_div, int side, double* loss_out, void* ws, size_t ws_bytes, const char* who) {
    EL_REQUIRE(side >= EL_PW_BOTH && side <= EL_PW_USERS, "%s: unknown side %d", who, side);
    EL_REQUIRE(u && i && label && loss_out, "%s: null argument", who);
    EL_REQUIRE(n < (1LL << 31) && n_div >= n, "%s: bad batch sizes", who);
    EL_REQUIRE(stp->U + stp->I < (1LL << 32), "%s: U + I must fit a 32-bit sort key", who);
    const el_pwmf_state st = *stp;
    PwWs w;
    EL_REQUIRE(carve(n, st.U, st.I, st.F, (char*)ws, &w) == 0, "%s: rocprim size query failed", who);
    EL_REQUIRE(ws != nullptr && ws_bytes >= w.total, "%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total);
    const bool vec = rows_vec4(st);

    PwFwd f;
    memset(&f, 0, sizeof(f));
    f.st = st, f.bu = u, f.bi = i, f.label = label, f.coef = w.coef, f.n = n;
    f.keyU = w.keyU_in, f.valU = w.valU_in, f.keyI = w.keyI_in, f.valI = w.valI_in;
    f.item_key_off = (u32)st.U;
    f.inv_n = 1.0f / (float)n_div;
    f.loss_out = loss_out;
    if (int rc = vec ? launch_fwd<4, true>(f, s) : launch_fwd<1, true>(f, s)) return rc;

    const bool do_users = side != EL_PW_ITEMS, do_items = side != EL_PW_USERS;
    {
        ElKernelTimer t("rocprim_radix_sort_pairs", s);
        size_t tb = w.tmp_bytes;
        if (do_users && do_items)                                  // one sort orders both sides (every user key < every item key)
            EL_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.keyU_in, w.keyU, w.valU_in, w.valU, (unsigned)(2 * n), 0, bits_for(st.U + st.I), s));
        else if (do_users)
            EL_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.keyU_in, w.keyU, w.valU_in, w.valU, (unsigned)n, 0, bits_for(st.U), s));
        else
            EL_CHECK_HIP(rocprim::radix_sort_pairs(w.tmp, tb, w.keyI_in, w.keyI, w.valI_in, w.valI, (unsigned)n, 0, bits_for(st.U + st.I), s));
    }
    const float l_w = st.kind == EL_PW_LOGISTIC ? st.l_w : 0.f;
    if (do_users) {
        PwSeg p = {w.keyU, 0u, w.valU, w.coef, i, st.Gi, st.Gu, st.gGu, st.gBu, n, st.F, user_chunk(n), 0, l_w};
        if (int rc = vec ? launch_seg<4>(p, w, "k_pw_seg_users", s) : launch_seg<1>(p, w, "k_pw_seg_users", s)) return rc;
    }
    if (do_items) {
        PwSeg p = {w.keyI, (u32)st.U, w.valI, w.coef, u, st.Gu, st.Gi, st.gGi, st.gBi, n, st.F, item_chunk(n), 0, l_w};
        if (int rc = vec ? launch_seg<4>(p, w, "k_pw_seg_items", s) : launch_seg<1>(p, w, "k_pw_seg_items", s)) return rc;
    }
    (void)ctx;
    return 0;
}

static int pw_apply(el_ctx* ctx, hipStream_t s, const el_pwmf_state& st, int opt, int side, float lr_t) {
    const bool do_users = side != EL_PW_ITEMS, do_items = side != EL_PW_USERS;
    if (do_users && st.U > 0) {
        if (int rc = apply(ctx, s, opt, "k_pw_opt_Gu", st.Gu, st.gGu, st.mGu, st.vGu, st.U * st.F, lr_t)) return rc;
        if (st.Bu)
            if (int rc = apply(ctx, s, opt, "k_pw_opt_Bu", st.Bu, st.gBu, st.mBu, st.vBu, st.U, lr_t)) return rc;
    }
    if (do_items && st.I > 0) {
        if (int rc = apply(ctx, s, opt, "k_pw_opt_Gi", st.Gi, st.gGi, st.mGi, st.vGi, st.I * st.F, lr_t)) return rc;
        if (st.Bi)
            if (int rc = apply(ctx, s, opt, "k_pw_opt_Bi", st.Bi, st.gBi, st.mBi, st.vBi, st.I, lr_t)) return rc;
    }
    return 0;
}

extern "C" int el_pwmf_train_step(el_ctx* ctx, void* stream, const el_pwmf_state* stp, const int32_t* u, const int32_t* i,
                                  const float* label, int64_t n, int opt, int side, int32_t step, float lr_t,
                                  double* loss_out, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(opt == EL_PW_ADAM || opt == EL_PW_ADAGRAD, "el_pwmf_train_step: unknown optimiser %d", opt);
    if (int rc = check_state(stp, "el_pwmf_train_step", true, opt)) return rc;
    if (n <= 0) return 0;
    EL_REQUIRE(step >= 1, "el_pwmf_train_step: step must be >= 1");
    if (int rc = pw_grads(ctx, (hipStream_t)stream, stp, u, i, label, n, n, side, loss_out, ws, ws_bytes, "el_pwmf_train_step")) return rc;
    return pw_apply(ctx, (hipStream_t)stream, *stp, opt, side, lr_t);
}

extern "C" int el_pwmf_grads(el_ctx* ctx, void* stream, const el_pwmf_state* stp, const int32_t* u, const int32_t* i,
                             const float* label, int64_t n, int64_t n_global, int side, double* loss_out, void* ws,
                             size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = check_state(stp, "el_pwmf_grads", true, EL_PW_ADAGRAD)) return rc;       // (slot layout is checked by el_pwmf_apply)
    if (n <= 0) return 0;
    return pw_grads(ctx, (hipStream_t)stream, stp, u, i, label, n, n_global, side, loss_out, ws, ws_bytes, "el_pwmf_grads");
}

extern "C" int el_pwmf_apply(el_ctx* ctx, void* stream, const el_pwmf_state* stp, int opt, int side, int32_t step, float lr_t) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(opt == EL_PW_ADAM || opt == EL_PW_ADAGRAD, "el_pwmf_apply: unknown optimiser %d", opt);
    EL_REQUIRE(side >= EL_PW_BOTH && side <= EL_PW_USERS && step >= 1, "el_pwmf_apply: bad side / step");
    if (int rc = check_state(stp, "el_pwmf_apply", true, opt)) return rc;
    return pw_apply(ctx, (hipStream_t)stream, *stp, opt, side, lr_t);
}

extern "C" int el_pwmf_link_values(el_ctx* ctx, void* stream, float* vals, int64_t n_rows, int64_t ld, int32_t k, int kind,
                                   const float* Bu, int64_t u_start) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(kind >= EL_PW_MSE && kind <= EL_PW_LOGISTIC, "el_pwmf_link_values: unknown kind %d", kind);
    if (n_rows <= 0 || k <= 0) return 0;
    EL_REQUIRE(vals != nullptr && ld >= k, "el_pwmf_link_values: bad arguments");
    if (Bu == nullptr && kind != EL_PW_MSE_SIGMOID) return 0;
    EL_LAUNCH("k_pw_link", k_pw_link, dim3((unsigned)((n_rows * k + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vals,
              n_rows, ld, (int)k, kind, Bu, u_start);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_topk_rerank(el_ctx* ctx, void* stream, int32_t* idx, float* vals, int64_t n_rows, int64_t ld, int32_t kk) {
    if (int rc = el_bind(ctx)) return rc;
    if (n_rows <= 0 || kk <= 1) return 0;
    EL_REQUIRE(idx && vals && ld >= kk && kk <= 4096, "el_topk_rerank: needs kk <= 4096 and ld >= kk (kk=%d, ld=%lld)", kk, (long long)ld);
    if (kk <= 64) {
        EL_LAUNCH("k_rerank_rows", k_rerank_rows, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, idx, vals, n_rows, ld,
                  (int)kk);
    } else {
        int cap = 128;
        while (cap < kk) cap <<= 1;
        EL_LAUNCH("k_rerank_rows", k_rerank_rows_long, dim3((unsigned)n_rows), dim3(64), (size_t)cap * 8, (hipStream_t)stream, idx, vals, n_rows,
                  ld, (int)kk, cap);
    }
    EL_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The epoch loop of the point-wise plugins (matrix_factorization.py:85-97 and siblings) from one call, as el_bprmf_train_loop does
// for BPRMF_batch: the samples of up to PW_LOOP_CHUNK draws come from ONE sampler launch (the sampler does not read the model),
// the batches are consecutive slices of it.  Same Philox stream and kernels as the per-batch calls.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int el_pointwise_sample_meta(el_ctx* ctx, void* stream, const int64_t* pos_indptr, const int32_t* pos_indices,
                                        const void* meta, int64_t U, int64_t I, uint64_t seed, uint64_t first_sample, int64_t n,
                                        int32_t* out_u, int32_t* out_i, float* out_label);

static const int64_t PW_LOOP_CHUNK = 4 << 20;

static int64_t pw_loop_cap(int64_t events, int64_t B) {
    int64_t most = (PW_LOOP_CHUNK / B) * B;
    if (most < B) most = B;
    const int64_t want = ((events + B - 1) / B) * B;
    return want < most ? (want < B ? B : want) : most;
}

extern "C" size_t el_pwmf_train_loop_ws_bytes(int64_t events, int64_t B) {
    if (events <= 0 || B <= 0) return 0;
    return align256((size_t)pw_loop_cap(events, B) * 12);
}

extern "C" int el_pwmf_train_loop(el_ctx* ctx, void* stream, const el_pwmf_state* stp, const int64_t* pos_indptr,
                                  const int32_t* pos_indices, const void* sampler_meta, uint64_t seed, uint64_t first_sample,
                                  int64_t events, int64_t B, int opt, int side, int32_t first_step, const float* lr_t_host,
                                  double* loss_out, void* ws, size_t ws_bytes, void* loop_ws, size_t loop_ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp != nullptr && B >= 1 && events >= 0 && lr_t_host != nullptr, "el_pwmf_train_loop: bad arguments");
    if (events == 0) return 0;
    const size_t need = el_pwmf_train_loop_ws_bytes(events, B);
    EL_REQUIRE(loop_ws != nullptr && loop_ws_bytes >= need, "el_pwmf_train_loop: loop workspace too small (%zu < %zu)", loop_ws_bytes, need);
    const int64_t cap = pw_loop_cap(events, B);
    int32_t* bu = (int32_t*)loop_ws;
    int32_t* bi = bu + cap;
    float* by = (float*)(bu + 2 * cap);
    int64_t k = 0;
    for (int64_t c0 = 0; c0 < events; c0 += cap) {
        const int64_t cn = (events - c0 < cap) ? events - c0 : cap;
        if (int rc = el_pointwise_sample_meta(ctx, stream, pos_indptr, pos_indices, sampler_meta, stp->U, stp->I, seed,
                                              first_sample + (uint64_t)c0, cn, bu, bi, by))
            return rc;
        for (int64_t off = 0; off < cn; off += B, ++k) {
            const int64_t n = (cn - off < B) ? cn - off : B;
            if (int rc = el_pwmf_train_step(ctx, stream, stp, bu + off, bi + off, by + off, n, opt, side, first_step + (int32_t)k,
                                            lr_t_host[k], loss_out, ws, ws_bytes))
                return rc;
        }
    }
    return 0;
}
