// NeuMF / GMF full-catalogue scoring fused with the masked top-k.
//
// Replaces NeuMF.get_recommendations' index grids (neural/NeuMF/neural_matrix_factorization.py:111-119) +
// NeuralMatrixFactorizationModel.get_recs (neural_matrix_factorization_model.py:119-144) + get_top_k (:146-148), and GMF's
// (neural/GeneralizedMF/generalized_matrix_factorization.py:98-104, generalized_matrix_factorization_model.py:81-93).
// The reference evaluates the whole network on a [Ub, I] grid of (user, item) pairs: 36 F^2 flop per pair and a dense
// [Ub, I, 4F] activation.  Here
//   * layer 1 is separable: W1^T [u ; i] = W1[:E]^T u + W1[E:]^T i.  PI = Imlp W1[E:] ([I, H1], once per evaluation) and
//     PU = Umlp[users] W1[:E] ([users, H1], once per block) are plain projections (k_nmf_proj, fp32 MFMA); a pair's layer-1
//     activation is formed in registers as relu((PU[u] + PI[i]) + b1)                    -- 16 F^2 flop per pair gone
//   * layers 2 and 3 run on v_mfma_f32_32x32x2_f32 with the PAIRS along the MFMA columns (one lane column = one item of the
//     workgroup's user): D2^T[feature][pair] = W2^T a1, D3^T = W3^T a2.  The C/D layout of the 32x32 MFMA puts rows {0-3,
//     8-11, ...} of a column into lane n and rows {4-7, 12-15, ...} into lane n+32; the weight rows are stored PERMUTED so that
//     lane n's accumulator t holds feature 2t and lane n+32's holds feature 2t+1 -- exactly what the next layer's B operand
//     (k = 2t from the low half-wave, 2t+1 from the high one) wants: the activations never leave the registers
//   * the head w . [mf ; mlp] + b is two interleaved fma chains (even / odd positions) on the VALU, summed
//   * the running top-k of the user lives in LDS (the wave-per-user selection of el_topk.hip); no [users, I] block in HBM
//
// Numerics contract (pinned by oracle/c/el_oracle.c, orc_nmf_*): every dot product is the k-ordered fp32 fma chain from +0
// (what the MFMA computes), then `+ bias`, relu; layer 1 = (chain_u + chain_i) + b1; mf term = fl(u_f * i_f);
// head: acc_even / acc_odd over the positions of [mf ; mlp] with that parity, logit = ((acc_even + acc_odd) + b) + 0.0f.
// The kernels rank by the LOGIT (sigmoid is monotone); the host applies the link to the k' = k + margin survivors and re-ranks
// (el_pwmf_link_values / el_topk_rerank), like the point-wise factor models do.
#include <stdlib.h>
#include "el_common.h"

#include "el_topk_common.h"

extern "C" int el_topk_merge(el_ctx* ctx, void* stream, const int32_t* parts_idx, const float* parts_val, int32_t G,
                             int64_t n_users, int32_t k, int32_t* out_idx, float* out_val);

#define NS_WAVES 8
#define NS_THREADS (NS_WAVES * 64)

// position p of a 16-wide k chunk <-> k offset: the low half-wave consumes positions 0..7 (= even k), the high one 8..15 (odd k)
__host__ __device__ inline int ns_perm16(int p) { return p < 8 ? 2 * p : 2 * (p - 8) + 1; }
// accumulator row m (0..31) of a 32-row output tile <-> feature offset inside the tile (see the header)
__host__ __device__ inline int ns_row_feature(int m) {
    const int g = m >> 3, r = m & 7;
    return 8 * g + (r < 4 ? 2 * r : 2 * (r - 4) + 1);
}

static inline int64_t ns_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ---- workspace ----------------------------------------------------------------------------------------------------------
struct NsLayout {
    size_t ctl, W1a, W1b, b1P, W2P, W3P, b2D, b3D, hwD, hwmf, PI, PU, pidx, pval, total;
    int H1P, H2P, H3P, FP, NC1, NC2, S;
};

static bool ns_shape_ok(const el_nmf_state* st) {
    if (!st->use_mlp || st->n_layers != 3) return false;
    const int H1 = st->units[0], H2 = st->units[1], H3 = st->units[2];
    if (H1 < 1 || H1 > 1024 || H2 < 1 || H2 > 256 || H3 < 1 || H3 > 128) return false;
    if (st->E < 1 || st->E > 256) return false;
    if (st->use_mf && (st->F < 1 || st->F > 256)) return false;
    const int H2P = (int)ns_up(H2, 32);
    const int H3P = H2P >= 256 ? 128 : (H2P >= 128 ? 64 : 32);
    return H3 <= H3P;
}

static int ns_cap_for_k(int k);
// dynamic LDS of k_nmf_score for a list capacity `cap` (the weights' staging area + one candidate list per wave, as ns_launch asks
// for it); the one architecture this library is built for (gfx950) gives a workgroup 160 KB
static const size_t NS_LDS_LIMIT = 160u * 1024u;
static size_t ns_lds_bytes(const el_nmf_state* st, int cap) {
    const int H1P = (int)ns_up(st->units[0], 16);
    int H2P = (int)ns_up(st->units[1], 32);
    if (H2P == 96) H2P = 128;
    if (H2P > 128 && H2P < 256) H2P = 256;
    const int H3P = H2P >= 256 ? 128 : (H2P >= 128 ? 64 : 32);
    const int FP = st->use_mf ? (int)ns_up(st->F, 8) : 0;
    return (size_t)(2 * H2P * 16 + 2 * H1P + H2P + 2 * H3P + 2 * (FP > 0 ? FP : 8)) * 4 + (size_t)NS_WAVES * cap * 8 + (size_t)NS_WAVES * 16;
}

static int ns_pick_split(el_ctx* ctx, int64_t n_users, int64_t n_items, int k, bool cand) {
    if (cand) return 1;
    int64_t want = ((int64_t)ctx->cus * 3 + n_users - 1) / n_users;      // ~3 workgroups per CU in all
    const int64_t by_items = n_items / (NS_WAVES * 256);                 // >= 8 tiles of 32 items per wave
    if (want > by_items) want = by_items;
    const int64_t by_merge = 8192 / ((int64_t)NS_WAVES * k);             // el_topk_merge: G * k <= 8192
    if (want > by_merge) want = by_merge;
    return (int)(want < 1 ? 1 : want);
}

static NsLayout ns_layout(el_ctx* ctx, const el_nmf_state* st, int64_t n_users, int64_t I_local, int k, bool cand) {
    NsLayout L;
    memset(&L, 0, sizeof(L));
    L.H1P = (int)ns_up(st->units[0], 16);
    L.H2P = (int)ns_up(st->units[1], 32);
    if (L.H2P == 96) L.H2P = 128;
    if (L.H2P > 128 && L.H2P < 256) L.H2P = 256;
    L.H3P = L.H2P >= 256 ? 128 : (L.H2P >= 128 ? 64 : 32);
    L.FP = st->use_mf ? (int)ns_up(st->F, 8) : 0;
    L.NC1 = L.H1P / 16;
    L.NC2 = L.H2P / 16;
    L.S = ns_pick_split(ctx, n_users > 0 ? n_users : 1, I_local, k, cand) * NS_WAVES;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.ctl = take(64);
    L.W1a = take((size_t)st->E * L.H1P * 4);
    L.W1b = take((size_t)st->E * L.H1P * 4);
    L.b1P = take((size_t)L.H1P * 4);
    L.W2P = take((size_t)L.NC1 * L.H2P * 16 * 4);
    L.W3P = take((size_t)L.NC2 * L.H3P * 16 * 4);
    L.b2D = take((size_t)L.H2P * 4);
    L.b3D = take((size_t)L.H3P * 4);
    L.hwD = take((size_t)L.H3P * 4);
    L.hwmf = take((size_t)(L.FP > 0 ? L.FP : 8) * 4);
    L.PI = take((size_t)(I_local > 0 ? I_local : 1) * L.H1P * 4);
    L.PU = take((size_t)(n_users > 0 ? n_users : 1) * L.H1P * 4);
    L.pidx = take((size_t)L.S * (n_users > 0 ? n_users : 1) * k * 4);
    L.pval = take((size_t)L.S * (n_users > 0 ? n_users : 1) * k * 4);
    L.total = o;
    return L;
}

// ---- item-side image: packed / permuted weights -----------------------------------------------------------------------
struct NsPack {
    const float *W1, *b1, *W2, *b2, *W3, *b3, *hw;
    float *W1a, *W1b, *b1P, *W2P, *W3P, *b2D, *b3D, *hwD, *hwmf;
    int E, H1, H2, H3, F, H1P, H2P, H3P, FP;
};

// chunk image of a Dense kernel W [K][N] (Keras layout): [chunk][row = 32 mt + m][16 k positions, 16-byte slots XOR-swizzled by
// (m >> 2) & 3]; row -> feature ns_row_feature, position -> k ns_perm16; zero outside [K) x [N)
__device__ __forceinline__ void ns_pack_w(const float* __restrict__ W, int K, int N, int NP, float* __restrict__ out, int64_t e) {
    const int p_phys = (int)(e & 15);
    const int64_t rowg = e >> 4;
    const int row = (int)(rowg % NP), c = (int)(rowg / NP);
    const int m = row & 31, mt = row >> 5;
    const int slot = (p_phys >> 2) ^ ((m >> 2) & 3);
    const int p = slot * 4 + (p_phys & 3);
    const int k = c * 16 + ns_perm16(p);
    const int feat = mt * 32 + ns_row_feature(m);
    out[e] = (k < K && feat < N) ? W[(int64_t)k * N + feat] : 0.f;
}

__global__ __launch_bounds__(256) void k_nmf_pack(NsPack q) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n1 = (int64_t)q.E * q.H1P;
    if (t < n1) {                                        // layer-1 kernels, columns in chunk order (even k's, odd k's)
        const int e = (int)(t / q.H1P), cp = (int)(t % q.H1P);
        const int k = (cp & ~15) + ns_perm16(cp & 15);
        q.W1a[t] = k < q.H1 ? q.W1[(int64_t)e * q.H1 + k] : 0.f;
        q.W1b[t] = k < q.H1 ? q.W1[(int64_t)(q.E + e) * q.H1 + k] : 0.f;
    }
    if (t < q.H1P) {
        const int k = ((int)t & ~15) + ns_perm16((int)t & 15);
        q.b1P[t] = k < q.H1 ? q.b1[k] : 0.f;
    }
    const int64_t n2 = (int64_t)(q.H1P / 16) * q.H2P * 16;
    if (t < n2) ns_pack_w(q.W2, q.H1, q.H2, q.H2P, q.W2P, t);
    const int64_t n3 = (int64_t)(q.H2P / 16) * q.H3P * 16;
    if (t < n3) ns_pack_w(q.W3, q.H2, q.H3, q.H3P, q.W3P, t);
    // per-feature vectors in accumulator order: [tile][half][t] <- feature 32 tile + 2 t + half
    if (t < q.H2P) {
        const int tile = (int)t >> 5, half = ((int)t >> 4) & 1, tt = (int)t & 15, f = 32 * tile + 2 * tt + half;
        q.b2D[t] = f < q.H2 ? q.b2[f] : 0.f;
    }
    if (t < q.H3P) {
        const int tile = (int)t >> 5, half = ((int)t >> 4) & 1, tt = (int)t & 15, f = 32 * tile + 2 * tt + half;
        q.b3D[t] = f < q.H3 ? q.b3[f] : 0.f;
        q.hwD[t] = f < q.H3 ? q.hw[q.F + f] : 0.f;
    }
    if (t < q.FP) {                                      // mf head weights: [half][f / 2]
        const int half = (int)t / (q.FP / 2), qq = (int)t % (q.FP / 2), f = 2 * qq + half;
        q.hwmf[t] = f < q.F ? q.hw[f] : 0.f;
    }
}

// ---- projection P[r][:] = X[rows[r] or r][0:E] . Wp[0:E][0:NP]  (pure k-ordered chains from +0) -------------------------
// 128 rows x 128 columns per workgroup: wave w owns rows 32 w .. 32 w + 31 and all four 32-column tiles.  X tile staged in
// LDS ([row][E + 1]); the B operand (Wp, 2 k-rows x 32 columns per MFMA) comes straight from L2.
// rebuild: optional device flag -- 0 = the image is current, return at once (EL_TOPK_ITEMS_UNCHANGED, verified by hash).
__global__ __launch_bounds__(256) void k_nmf_proj(const float* __restrict__ X, int64_t row0, int64_t n_rows, int E,
                                                  const float* __restrict__ Wp, int NP, float* __restrict__ P,
                                                  const unsigned long long* __restrict__ rebuild) {
    if (rebuild && *rebuild == 0ull) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);          // [128][E + 1]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int ldx = E + 1;
    const int64_t r0 = (int64_t)blockIdx.x * 128;
    for (int t = threadIdx.x; t < 128 * E; t += 256) {
        const int r = t / E, e = t - r * E;
        const int64_t row = r0 + r;
        xs[r * ldx + e] = row < n_rows ? X[(row0 + row) * (int64_t)E + e] : 0.f;
    }
    __syncthreads();
    const int c0 = blockIdx.y * 128;
    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* xrow = xs + (w * 32 + n) * ldx;
    for (int e = 0; e < E; e += 2) {
        const int ek = e + h;
        const float a = ek < E ? xrow[ek] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c0 + 32 * j + n;
            const float b = (ek < E && col < NP) ? Wp[(int64_t)ek * NP + col] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = c0 + 32 * j + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = r0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < n_rows && col < NP) P[row * (int64_t)NP + col] = acc[j][r];
        }
    }
}

// ---- hash of the tables the PI image is derived from (Imlp, W1) -----------------------------------------------------------
__global__ void k_items_hash(const float* __restrict__ Gi, const float* __restrict__ Bi, int64_t n_g, int64_t n_b, u64* ctl);   // el_topk_screen.hip
__global__ void k_nmf_decide(u64* ctl, int force) {
    const bool stale = force || ctl[0] != ctl[1];
    ctl[0] = ctl[1];
    ctl[2] = stale ? 1ull : 0ull;
}

// ---- the fused kernel -----------------------------------------------------------------------------------------------------
struct NsParams {
    TopkParams t;                 // users, item shard, masks, k (out_* unused: partial lists go to part_*)
    const float *Umf, *Imf;       // [U][F], [I_local][F] or NULL
    const float *hwmf;            // [FP] mf head weights, [half][f / 2]
    const float* hb;              // Dense(1) bias (device scalar) or NULL
    int F, FP;
    const float *PU, *PI, *b1P;   // [n_users][H1P], [I_local][H1P], [H1P]
    const float *W2P, *W3P, *b2D, *b3D, *hwD;
    int H1P, NC1;
    int32_t* part_idx;
    float* part_val;
    int S, cap;
};

// 16 bytes at (wave-uniform base) + (32-bit per-thread byte offset): the `global_load_dwordx4 v, v_off, s[base]` form -- one offset
// register serves every chunk (64-bit per-thread addresses of all chunks, precomputed by the compiler, cost 70 spilled registers)
__device__ __forceinline__ float4 ns_ldg16(const float* ubase, u32 boff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ubase) + boff);
}

template <int H2P, int H3P>
__global__ __launch_bounds__(NS_THREADS) void k_nmf_score(NsParams p) {
    constexpr int MT2 = H2P / 32, MT3 = H3P / 32, NC2 = H2P / 16;
    constexpr int G3 = MT3 > 2 ? 2 : 1, MT3G = MT3 / G3;  // layer-3 output tiles per pass
    constexpr int WCH = H2P * 16;                        // floats of one W2 chunk image (a W3 chunk is H3P * 16 <= that)
    constexpr int NST2 = (H2P * 4 + NS_THREADS - 1) / NS_THREADS;   // float4 staging registers per thread and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wbuf = reinterpret_cast<float*>(smem);        // [2][WCH]
    float* pus = wbuf + 2 * WCH;                         // [H1P]  PU row of this user (chunk order)
    float* b1s = pus + p.H1P;                            // [H1P]
    float* b2s = b1s + p.H1P;                            // [H2P]
    float* b3s = b2s + H2P;                              // [H3P]
    float* hws = b3s + H3P;                              // [H3P]
    float* ums = hws + H3P;                              // [FP]  user mf row, [half][f / 2]
    float* hms = ums + (p.FP > 0 ? p.FP : 8);            // [FP]
    u64* keys_all = reinterpret_cast<u64*>(hms + (p.FP > 0 ? p.FP : 8));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, h = lane >> 5;
    u64* keys = keys_all + (size_t)w * p.cap;
    int* cnt_s = reinterpret_cast<int*>(keys_all + (size_t)NS_WAVES * p.cap) + w * 4;

    const int64_t urel = blockIdx.x;
    const int64_t user = p.t.u_start + urel;
    int64_t e0 = 0, e1 = 0, c0 = 0, c1 = 0;
    if (p.t.excl_indptr) {
        e0 = p.t.excl_indptr[user];
        e1 = p.t.excl_indptr[user + 1];
    }
    int64_t ncand = p.t.I_local;
    if (p.t.cand_indptr) {
        c0 = p.t.cand_indptr[user];
        c1 = p.t.cand_indptr[user + 1];
        ncand = c1 - c0;
    }
    const bool use_excl = (p.t.excl_indptr != nullptr) && (p.t.cand_indptr == nullptr);
    const int s = blockIdx.y * NS_WAVES + w;             // this wave's slice of the positions
    const int64_t pos_lo = ncand * s / p.S, pos_hi = ncand * (s + 1) / p.S;
    const int64_t max_len = (ncand + p.S - 1) / p.S;
    const int T = (int)((max_len + 31) / 32);            // every wave of the grid row walks the same number of tiles (barriers)

    // per-user vectors -> LDS
    for (int t = tid; t < p.H1P; t += NS_THREADS) {
        pus[t] = p.PU[urel * p.H1P + t];
        b1s[t] = p.b1P[t];
    }
    for (int t = tid; t < H2P; t += NS_THREADS) b2s[t] = p.b2D[t];
    for (int t = tid; t < H3P; t += NS_THREADS) {
        b3s[t] = p.b3D[t];
        hws[t] = p.hwD[t];
    }
    for (int t = tid; t < p.FP; t += NS_THREADS) {
        const int half = t / (p.FP / 2), f = 2 * (t % (p.FP / 2)) + half;
        ums[t] = f < p.F ? p.Umf[user * (int64_t)p.F + f] : 0.f;
        hms[t] = p.hwmf[t];
    }
    // chunk 0 of W2 -> buffer 0
    float4 stg[NST2];
#pragma unroll
    for (int q = 0; q < NST2; ++q) {
        const int e4 = tid + q * NS_THREADS;
        if (e4 < H2P * 4) reinterpret_cast<float4*>(wbuf)[e4] = reinterpret_cast<const float4*>(p.W2P)[e4];
    }
    __syncthreads();
    const float hbias = p.hb ? *p.hb : 0.f;
    int par = 0;                                          // buffer that holds the chunk about to be consumed
    int cnt = 0;
    float tau = -INFINITY;
    const int swz = (n >> 2) & 3;
    const int aoff0 = n * 16 + (((2 * h) ^ swz) << 2), aoff1 = n * 16 + (((2 * h + 1) ^ swz) << 2);

    for (int tile = 0; tile < T; ++tile) {
        const int64_t pos = pos_lo + (int64_t)tile * 32 + n;
        bool valid = pos < pos_hi;
        int32_t gitem = -1;
        int64_t il = 0;
        if (valid) {
            if (p.t.cand_indptr) {
                gitem = p.t.cand_indices[c0 + pos];
                il = (int64_t)gitem - p.t.item_offset;
                valid = il >= 0 && il < p.t.I_local;
                if (!valid) il = 0;
            } else {
                il = pos;
                gitem = (int32_t)(p.t.item_offset + pos);
            }
        }
        // ---------------- layer 2: acc2[feature][pair] = sum_k W2[k][feature] a1[k][pair] ---------------------------
        floatx16 acc2[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
        const float4* pirow = reinterpret_cast<const float4*>(p.PI + il * (int64_t)p.H1P) + 2 * h;
        float4 pi0 = pirow[0], pi1 = pirow[1];
        for (int c = 0; c < p.NC1; ++c) {
            // the next chunk's image leaves L2 now and lands in the other buffer after this chunk's MFMAs
            const bool more2 = c + 1 < p.NC1;
            const float* src = more2 ? p.W2P + (size_t)(c + 1) * WCH : p.W3P;
            const int lim = more2 ? H2P * 4 : H3P * 4;
#pragma unroll
            for (int q = 0; q < NST2; ++q) {
                const int e4 = tid + q * NS_THREADS;
                if (e4 < lim) stg[q] = ns_ldg16(src, (u32)e4 * 16u);
            }
            float4 nx0 = pi0, nx1 = pi1;
            if (more2) {
                nx0 = pirow[4 * (c + 1)];
                nx1 = pirow[4 * (c + 1) + 1];
            }
            const float4* pu4 = reinterpret_cast<const float4*>(pus + c * 16 + 8 * h);
            const float4* b14 = reinterpret_cast<const float4*>(b1s + c * 16 + 8 * h);
            const float4 u0 = pu4[0], u1 = pu4[1], d0 = b14[0], d1 = b14[1];
            float bv[8];
            bv[0] = fmaxf((u0.x + pi0.x) + d0.x, 0.f);
            bv[1] = fmaxf((u0.y + pi0.y) + d0.y, 0.f);
            bv[2] = fmaxf((u0.z + pi0.z) + d0.z, 0.f);
            bv[3] = fmaxf((u0.w + pi0.w) + d0.w, 0.f);
            bv[4] = fmaxf((u1.x + pi1.x) + d1.x, 0.f);
            bv[5] = fmaxf((u1.y + pi1.y) + d1.y, 0.f);
            bv[6] = fmaxf((u1.z + pi1.z) + d1.z, 0.f);
            bv[7] = fmaxf((u1.w + pi1.w) + d1.w, 0.f);
            const float* wb = wbuf + par * WCH;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const float4 a0 = *reinterpret_cast<const float4*>(wb + mt * 512 + aoff0);
                const float4 a1 = *reinterpret_cast<const float4*>(wb + mt * 512 + aoff1);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv[0], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv[1], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bv[2], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bv[3], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bv[4], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bv[5], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bv[6], acc2[mt], 0, 0, 0);
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bv[7], acc2[mt], 0, 0, 0);
            }
            pi0 = nx0;
            pi1 = nx1;
            float4* dst = reinterpret_cast<float4*>(wbuf + (par ^ 1) * WCH);
#pragma unroll
            for (int q = 0; q < NST2; ++q) {
                const int e4 = tid + q * NS_THREADS;
                if (e4 < lim) dst[e4] = stg[q];
            }
            __syncthreads();
            par ^= 1;
        }
        // a2 = relu(acc2 + b2), in place: lane (n, h) register t of tile mt = feature 32 mt + 2 t + h
        // (the offset goes through an empty asm: these LDS vectors do not change over the tiles, and hoisting their loads out of the
        //  tile loop -- 4 x H2P / 2 + H3P registers -- is what the compiler otherwise does, at the price of 136 spilled registers)
        int hoff = h * 16;
        asm volatile("" : "+v"(hoff));
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const float4* bb = reinterpret_cast<const float4*>(b2s + mt * 32 + hoff);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = bb[q];
                acc2[mt][4 * q + 0] = fmaxf(acc2[mt][4 * q + 0] + b.x, 0.f);
                acc2[mt][4 * q + 1] = fmaxf(acc2[mt][4 * q + 1] + b.y, 0.f);
                acc2[mt][4 * q + 2] = fmaxf(acc2[mt][4 * q + 2] + b.z, 0.f);
                acc2[mt][4 * q + 3] = fmaxf(acc2[mt][4 * q + 3] + b.w, 0.f);
            }
            asm volatile("" : "+v"(acc2[mt]));            // (the sums happen HERE, not sunk to their uses in layer 3 with 128
            __builtin_amdgcn_sched_barrier(0);           //  bias registers kept alive until then)
        }
        // ---------------- head, mf part: two interleaved chains over [mf ; mlp] (this half-wave's parity) ----------------------
        float acc = 0.f;
        if (p.FP > 0) {
            int moff = h * (p.FP / 2);
            asm volatile("" : "+v"(moff));
            const float* irow = p.Imf + il * (int64_t)p.F;
            const bool vec = (p.F & 7) == 0;
            for (int j = 0; j < p.FP / 8; ++j) {
                float v[8];
                if (vec) {
                    const float4 x0 = reinterpret_cast<const float4*>(irow)[2 * j], x1 = reinterpret_cast<const float4*>(irow)[2 * j + 1];
                    v[0] = x0.x, v[1] = x0.y, v[2] = x0.z, v[3] = x0.w, v[4] = x1.x, v[5] = x1.y, v[6] = x1.z, v[7] = x1.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (8 * j + q) < p.F ? irow[8 * j + q] : 0.f;
                }
                const float4 uu = *reinterpret_cast<const float4*>(ums + moff + 4 * j);
                const float4 ww = *reinterpret_cast<const float4*>(hms + moff + 4 * j);
                acc = __builtin_fmaf(ww.x, uu.x * (h ? v[1] : v[0]), acc);
                acc = __builtin_fmaf(ww.y, uu.y * (h ? v[3] : v[2]), acc);
                acc = __builtin_fmaf(ww.z, uu.z * (h ? v[5] : v[4]), acc);
                acc = __builtin_fmaf(ww.w, uu.w * (h ? v[7] : v[6]), acc);
            }
        }
        // ---------------- layer 3: acc3[feature][pair] = sum_k W3[k][feature] a2[k][pair], a2 straight from acc2; the output
        // tiles in groups of two (acc2 + four accumulator tiles + staging would not fit 256 registers), each followed by its
        // share of the head chain (features ascending, so the chain order is that of one pass)
#pragma unroll
        for (int g3 = 0; g3 < G3; ++g3) {
            floatx16 acc3[MT3G];
#pragma unroll
            for (int mt = 0; mt < MT3G; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mt][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC2; ++c) {
                const bool more3 = c + 1 < NC2, again = g3 + 1 < G3;
                const float* src = more3 ? p.W3P + (size_t)(c + 1) * (H3P * 16) : (again ? p.W3P : p.W2P);
                const int lim = (more3 || again) ? H3P * 4 : H2P * 4;   // (after the last W3 chunk: chunk 0 of W2 for the next tile)
#pragma unroll
                for (int q = 0; q < NST2; ++q) {
                    const int e4 = tid + q * NS_THREADS;
                    if (e4 < lim) stg[q] = ns_ldg16(src, (u32)e4 * 16u);
                }
                const float* wb = wbuf + par * WCH;
#pragma unroll
                for (int mt = 0; mt < MT3G; ++mt) {
                    const float4 a0 = *reinterpret_cast<const float4*>(wb + (g3 * MT3G + mt) * 512 + aoff0);
                    const float4 a1 = *reinterpret_cast<const float4*>(wb + (g3 * MT3G + mt) * 512 + aoff1);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, acc2[c >> 1][(c & 1) * 8 + 0], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, acc2[c >> 1][(c & 1) * 8 + 1], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, acc2[c >> 1][(c & 1) * 8 + 2], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, acc2[c >> 1][(c & 1) * 8 + 3], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, acc2[c >> 1][(c & 1) * 8 + 4], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, acc2[c >> 1][(c & 1) * 8 + 5], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, acc2[c >> 1][(c & 1) * 8 + 6], acc3[mt], 0, 0, 0);
                    acc3[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, acc2[c >> 1][(c & 1) * 8 + 7], acc3[mt], 0, 0, 0);
                }
                float4* dst = reinterpret_cast<float4*>(wbuf + (par ^ 1) * WCH);
#pragma unroll
                for (int q = 0; q < NST2; ++q) {
                    const int e4 = tid + q * NS_THREADS;
                    if (e4 < lim) dst[e4] = stg[q];
                }
                __syncthreads();
                par ^= 1;
            }
            // head, mlp part of this group: a3 = relu(acc3 + b3), acc = fma(hw, a3, acc)
            int hoff3 = h * 16;
            asm volatile("" : "+v"(hoff3));
#pragma unroll
            for (int mt = 0; mt < MT3G; ++mt) {
                const float4* bb = reinterpret_cast<const float4*>(b3s + (g3 * MT3G + mt) * 32 + hoff3);
                const float4* hh = reinterpret_cast<const float4*>(hws + (g3 * MT3G + mt) * 32 + hoff3);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = bb[q], g = hh[q];
                    acc = __builtin_fmaf(g.x, fmaxf(acc3[mt][4 * q + 0] + b.x, 0.f), acc);
                    acc = __builtin_fmaf(g.y, fmaxf(acc3[mt][4 * q + 1] + b.y, 0.f), acc);
                    acc = __builtin_fmaf(g.z, fmaxf(acc3[mt][4 * q + 2] + b.z, 0.f), acc);
                    acc = __builtin_fmaf(g.w, fmaxf(acc3[mt][4 * q + 3] + b.w, 0.f), acc);
                    asm volatile("" : "+v"(acc));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const float other = __uint_as_float(el_partner32(__float_as_uint(acc), h));
        const float logit = (((h ? other : acc) + (h ? acc : other)) + hbias) + 0.0f;     // (even + odd) + b
        // ---------------- selection (lanes 0..31 carry the tile's 32 pairs) ------------------------------------------------
        bool hit = valid && h == 0 && (logit >= tau);
        if (hit && use_excl) hit = !el_row_contains(p.t.excl_indices, e0, e1, gitem);
        const u64 bal = __ballot(hit);
        if (bal) {
            const int offp = __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) keys[cnt + offp] = el_make_key(logit, gitem);
            cnt += __popcll(bal);
        }
        if (cnt > p.cap - 32) {
            if (lane == 0) *cnt_s = cnt;
            tau = el_wave_compact(keys, cnt_s, p.cap, p.t.k, lane);
            cnt = cnt < p.t.k ? cnt : p.t.k;
        }
    }
    if (lane == 0) *cnt_s = cnt;
    el_wave_compact(keys, cnt_s, p.cap, p.t.k, lane);
    const int nv = cnt < p.t.k ? cnt : p.t.k;
    const int64_t orow = ((int64_t)s * (p.t.u_stop - p.t.u_start) + urel) * p.t.k;
    for (int t = lane; t < p.t.k; t += 64) {
        int32_t oi;
        float ov;
        if (t < nv) {
            const u64 key = keys[t];
            oi = el_key_item(key);
            ov = el_key_score(key);
        } else {
            // padding = the lowest masked items: of this wave's own item range (disjoint ranges -> no duplicates after the merge); with
            // a candidate list the positions are not item ranges: slice 0 alone pads, the other slices leave empty entries
            oi = p.t.cand_indptr ? (s == 0 ? el_fill_masked(p.t, e0, e1, c0, c1, t - nv) : -1)
                                 : el_fill_masked_range(p.t, p.t.item_offset + pos_lo, p.t.item_offset + pos_hi, e0, e1, c0, c1, t - nv);
            ov = -INFINITY;
        }
        p.part_idx[orow + t] = oi;
        p.part_val[orow + t] = ov;
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static int ns_cap_for_k(int k) {
    int cap = 128;
    while (cap < k + 64) cap <<= 1;
    return cap;
}

extern "C" int el_nmf_score_supported(const el_nmf_state* st, int32_t k) {
    if (st == nullptr || !ns_shape_ok(st) || k < 1 || k > 448) return 0;
    return ns_lds_bytes(st, ns_cap_for_k(k)) <= NS_LDS_LIMIT ? 1 : 0;       // (callers fall back to the pair route otherwise)
}

extern "C" size_t el_nmf_score_ws_bytes(el_ctx* ctx, const el_nmf_state* st, int64_t n_users, int64_t I_local, int32_t k, int with_cand) {
    if (!ctx || !st || !ns_shape_ok(st) || k < 1) return 0;
    return ns_layout(ctx, st, n_users, I_local, k, with_cand != 0).total;
}

template <int H2P, int H3P>
static int ns_launch(const NsParams& p, int64_t n_users, int nsplit, hipStream_t s) {
    const size_t lds = (size_t)(2 * H2P * 16 + 2 * p.H1P + H2P + 2 * H3P + 2 * (p.FP > 0 ? p.FP : 8)) * 4 + (size_t)NS_WAVES * p.cap * 8 +
                       (size_t)NS_WAVES * 16;
    auto kern = k_nmf_score<H2P, H3P>;
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EL_LAUNCH("k_nmf_score", kern, dim3((unsigned)n_users, (unsigned)nsplit), dim3(NS_THREADS), lds, s, p);
    EL_CHECK_LAUNCH();
    return 0;
}

extern "C" int el_nmf_sync_tables(el_ctx* ctx, void* stream, el_nmf_state* st);      // el_neural.hip

extern "C" int el_nmf_score_topk(el_ctx* ctx, void* stream, el_nmf_state* st, int64_t u_start, int64_t u_stop,
                                 int64_t item_offset, int64_t I_local, const int64_t* excl_indptr, const int32_t* excl_indices,
                                 const int64_t* cand_indptr, const int32_t* cand_indices, int32_t k, int32_t* out_idx,
                                 float* out_val, int flags, void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(st != nullptr, "el_nmf_score_topk: null state");
    EL_REQUIRE(ns_shape_ok(st), "el_nmf_score_topk: needs an MLP tower of three Dense layers with units <= (1024, 256, 128), "
                                "units[2] <= half the padded units[1], E <= 256, F <= 256 (use el_nmf_forward + el_dense_topk otherwise)");
    EL_REQUIRE(k >= 1 && k <= 448, "el_nmf_score_topk: k=%d unsupported (1..448)", k);
    EL_REQUIRE(u_stop >= u_start && u_start >= 0 && u_stop <= st->U, "el_nmf_score_topk: bad user range");
    EL_REQUIRE(I_local >= 0 && item_offset >= 0 && item_offset + I_local <= st->I, "el_nmf_score_topk: item shard outside the tables");
    EL_REQUIRE((cand_indptr == nullptr) == (cand_indices == nullptr), "el_nmf_score_topk: cand CSR needs both arrays");
    EL_REQUIRE(excl_indptr == nullptr || excl_indices != nullptr, "el_nmf_score_topk: excl_indptr without excl_indices");
    const int64_t n_users = u_stop - u_start;
    if (n_users == 0) return 0;
    EL_REQUIRE(out_idx && out_val, "el_nmf_score_topk: null output");
    EL_REQUIRE(n_users <= 0x7fffffffLL / 2, "el_nmf_score_topk: too many users per call");
    const bool cand = cand_indptr != nullptr;
    const NsLayout L = ns_layout(ctx, st, n_users, I_local, k, cand);
    EL_REQUIRE(ws != nullptr && ws_bytes >= L.total, "el_nmf_score_topk: workspace too small (%zu < %zu; el_nmf_score_ws_bytes)", ws_bytes, L.total);
    EL_REQUIRE(((uintptr_t)ws & 15) == 0, "el_nmf_score_topk: workspace must be 16-byte aligned");
    EL_REQUIRE(!st->use_mf || (((uintptr_t)st->tab[1] & 15) == 0), "el_nmf_score_topk: item MF table must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = el_nmf_sync_tables(ctx, stream, st)) return rc;            // deferred decay: every row current before it is read
    char* base = (char*)ws;
    u64* ctl = (u64*)(base + L.ctl);
    // ---- item side.  The small packed images are rebuilt every call (microseconds); PI (I x H1 projection) only when the caller
    // does not assert EL_TOPK_ITEMS_UNCHANGED for this workspace / table / shape -- and the assertion is verified by a hash of
    // Imlp and W1 on the device
    NsPack q;
    q.W1 = st->W[0], q.b1 = st->b[0], q.W2 = st->W[1], q.b2 = st->b[1], q.W3 = st->W[2], q.b3 = st->b[2], q.hw = st->hw;
    q.W1a = (float*)(base + L.W1a), q.W1b = (float*)(base + L.W1b), q.b1P = (float*)(base + L.b1P);
    q.W2P = (float*)(base + L.W2P), q.W3P = (float*)(base + L.W3P), q.b2D = (float*)(base + L.b2D), q.b3D = (float*)(base + L.b3D);
    q.hwD = (float*)(base + L.hwD), q.hwmf = (float*)(base + L.hwmf);
    q.E = st->E, q.H1 = st->units[0], q.H2 = st->units[1], q.H3 = st->units[2], q.F = st->use_mf ? st->F : 0;
    q.H1P = L.H1P, q.H2P = L.H2P, q.H3P = L.H3P, q.FP = L.FP;
    {
        int64_t nmax = (int64_t)st->E * L.H1P;
        const int64_t n2 = (int64_t)L.NC1 * L.H2P * 16, n3 = (int64_t)L.NC2 * L.H3P * 16;
        if (n2 > nmax) nmax = n2;
        if (n3 > nmax) nmax = n3;
        if (nmax < 1024) nmax = 1024;
        EL_LAUNCH("k_nmf_pack", k_nmf_pack, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, s, q);
    }
    const float* Imlp = st->tab[3] + item_offset * (int64_t)st->E;
    const bool claim = (flags & EL_TOPK_ITEMS_UNCHANGED) != 0 && ctx->nmf_ws == ws && ctx->nmf_Imlp == Imlp && ctx->nmf_W1 == st->W[0] &&
                       ctx->nmf_I == I_local && ctx->nmf_E == st->E && ctx->nmf_H1 == st->units[0];
    if (I_local > 0) {
        EL_CHECK_HIP(hipMemsetAsync(ctl + 1, 0, 8, s));
        const int64_t n_g = I_local * (int64_t)st->E;
        int64_t hb = (n_g / 4 + 255) / 256;
        if (hb > (int64_t)ctx->cus * 8) hb = (int64_t)ctx->cus * 8;
        if (hb < 1) hb = 1;
        EL_LAUNCH("k_items_hash", k_items_hash, dim3((unsigned)hb), dim3(256), 0, s, Imlp, (const float*)st->W[0], n_g,
                  (int64_t)2 * st->E * st->units[0], ctl);
        EL_LAUNCH("k_nmf_decide", k_nmf_decide, dim3(1), dim3(1), 0, s, ctl, claim ? 0 : 1);
        const size_t lds = (size_t)128 * (st->E + 1) * 4;
        EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nmf_proj), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        EL_LAUNCH("k_nmf_proj", k_nmf_proj, dim3((unsigned)((I_local + 127) / 128), (unsigned)((L.H1P + 127) / 128)), dim3(256), lds, s, Imlp,
                  (int64_t)0, I_local, (int)st->E, (const float*)q.W1b, L.H1P, (float*)(base + L.PI), (const unsigned long long*)(ctl + 2));
    }
    ctx->nmf_ws = ws, ctx->nmf_Imlp = Imlp, ctx->nmf_W1 = st->W[0], ctx->nmf_I = I_local, ctx->nmf_E = st->E, ctx->nmf_H1 = st->units[0];
    // ---- user side: PU = Umlp[u_start .. u_stop) W1[:E]
    {
        const size_t lds = (size_t)128 * (st->E + 1) * 4;
        EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nmf_proj), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        EL_LAUNCH("k_nmf_proj", k_nmf_proj, dim3((unsigned)((n_users + 127) / 128), (unsigned)((L.H1P + 127) / 128)), dim3(256), lds, s,
                  (const float*)st->tab[2], u_start, n_users, (int)st->E, (const float*)q.W1a, L.H1P, (float*)(base + L.PU),
                  (const unsigned long long*)nullptr);
    }
    // ---- fused scoring + selection
    NsParams p;
    memset(&p, 0, sizeof(p));
    p.t.u_start = u_start, p.t.u_stop = u_stop, p.t.item_offset = item_offset, p.t.I_local = I_local;
    p.t.excl_indptr = excl_indptr, p.t.excl_indices = excl_indices, p.t.cand_indptr = cand_indptr, p.t.cand_indices = cand_indices;
    p.t.k = k;
    p.Umf = st->use_mf ? st->tab[0] : nullptr;
    p.Imf = st->use_mf ? st->tab[1] + item_offset * (int64_t)st->F : nullptr;
    p.hwmf = q.hwmf;
    p.F = q.F, p.FP = L.FP;
    p.PU = (const float*)(base + L.PU), p.PI = (const float*)(base + L.PI), p.b1P = q.b1P;
    p.W2P = q.W2P, p.W3P = q.W3P, p.b2D = q.b2D, p.b3D = q.b3D, p.hwD = q.hwD;
    p.H1P = L.H1P, p.NC1 = L.NC1;
    p.part_idx = (int32_t*)(base + L.pidx), p.part_val = (float*)(base + L.pval);
    p.S = L.S, p.cap = ns_cap_for_k(k);
    p.hb = st->head_bias ? st->hb : nullptr;
    const int nsplit = L.S / NS_WAVES;
    int rc;
    if (L.H2P == 256) rc = ns_launch<256, 128>(p, n_users, nsplit, s);
    else if (L.H2P == 128) rc = ns_launch<128, 64>(p, n_users, nsplit, s);
    else if (L.H2P == 64) rc = ns_launch<64, 32>(p, n_users, nsplit, s);
    else rc = ns_launch<32, 32>(p, n_users, nsplit, s);
    if (rc) return rc;
    return el_topk_merge(ctx, stream, p.part_idx, p.part_val, L.S, n_users, k, out_idx, out_val);
}

// ---- GMF: score = sigmoid(sum_f h_f u_f i_f) = sigmoid(<u, i * h>): the item image for the fused dot-product kernels ------------
__global__ __launch_bounds__(256) void k_scale_cols(const float* __restrict__ X, const float* __restrict__ h, int64_t n, int F,
                                                    float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) out[t] = X[t] * h[t % F];
}

extern "C" int el_gmf_item_image(el_ctx* ctx, void* stream, const float* Imf, const float* hw, int64_t I, int32_t F, float* out) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(Imf && hw && out && I >= 0 && F >= 1, "el_gmf_item_image: bad arguments");
    if (I == 0) return 0;
    const int64_t n = I * (int64_t)F;
    int64_t g = (n + 255) / 256;
    if (g > (int64_t)ctx->cus * 16) g = (int64_t)ctx->cus * 16;
    EL_LAUNCH("k_scale_cols", k_scale_cols, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, Imf, hw, n, (int)F, out);
    EL_CHECK_LAUNCH();
    return 0;
}
