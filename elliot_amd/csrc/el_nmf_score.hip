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
    // screened route (el_nmf_score_topk with EL_NMF_SCREEN): half-precision stage images of W2 / W3, accumulator-order vectors, half-precision image of
    // PI + the residual norm of every row, the bound's constants, one candidate region per (user, wave slice)
    size_t W2B, W3B, b2E, b3E, hwE, PIB, Rn, cst, gram, upb, thr, tidx, regi, regc, sflag;
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
    return (size_t)(2 * H2P * 16 + 2 * H1P + H2P + 2 * H3P + 2 * (FP > 0 ? FP : 8)) * 4 + (size_t)NS_WAVES * cap * 8 + (size_t)NS_WAVES * 16 + 16;
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

static NsLayout ns_layout(el_ctx* ctx, const el_nmf_state* st, int64_t n_users, int64_t I_local, int k, bool cand, bool screen = false) {
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
    // the persistent item-side images come first: nothing before this line depends on the user range, k or the split, so a later call on
    // the same workspace with a shorter user block (the last block of an evaluation, EL_TOPK_ITEMS_UNCHANGED) finds them where they were built
    if (screen) {
        L.PIB = take((size_t)(I_local > 0 ? I_local : 1) * ns_up(L.H1P, 256) * 2);
        L.Rn = take((size_t)(I_local > 0 ? I_local : 1) * 4);
    }
    L.PU = take((size_t)(n_users > 0 ? n_users : 1) * L.H1P * 4);
    L.pidx = take((size_t)L.S * (n_users > 0 ? n_users : 1) * k * 4);
    L.pval = take((size_t)L.S * (n_users > 0 ? n_users : 1) * k * 4);
    if (screen) {
        L.W2B = take((size_t)ns_up(L.H1P, 256) * L.H2P * 2);
        L.W3B = take((size_t)L.H2P * L.H3P * 2);
        L.b2E = take((size_t)L.H2P * 4);
        L.b3E = take((size_t)L.H3P * 4);
        L.hwE = take((size_t)L.H3P * 4);
        L.cst = take(256);                               // 8 constants, then the traces of the squarings (k_nmf_gram_sq)
        L.gram = take((size_t)4 * 2 * 65536 * 4);
        const size_t nu = (size_t)(n_users > 0 ? n_users : 1), ni = (size_t)(I_local > 0 ? I_local : 1);
        L.upb = take(nu * ni * 4);                       // upper bound logit' + E of every pair
        L.thr = take(nu * (size_t)k * 4);                // merged top-k of the lower bounds (its last column = the user's threshold)
        L.tidx = take(nu * (size_t)k * 4);
        L.regi = take(nu * ni * 4);                      // candidates of slice s of user u: compacted at the start of the slice's own range
        L.regc = take((size_t)L.S * nu * 4);
        L.sflag = take(64);
    }
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
// ctl[3]: the half-precision image of PI (screened route) is older than PI -- set whenever PI is rebuilt, by screened and
// unscreened calls alike, cleared by k_nmf_pib_done after the image was rebuilt (a call that forces the rebuild -- a workspace the
// library has not seen, other tables -- initialises it)
__global__ void k_nmf_decide(u64* ctl, int force) {
    const bool stale = force || ctl[0] != ctl[1];
    ctl[0] = ctl[1];
    ctl[2] = stale ? 1ull : 0ull;
    if (stale) ctl[3] = 1ull;
}
__global__ void k_nmf_pib_done(u64* ctl) { ctl[3] = 0ull; }

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
    // region mode (second stage of the screened route): wave slice s of user u scores the reg_cnt[u * S + s] candidates stored at
    // the start of the slice's own position range, reg_idx[u * I_local + pos_lo(s) ..] (already free of masked items)
    const int32_t* reg_idx;
    const int32_t* reg_cnt;
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
    const bool region = p.reg_cnt != nullptr;
    const bool use_excl = (p.t.excl_indptr != nullptr) && (p.t.cand_indptr == nullptr) && !region;
    const int s = blockIdx.y * NS_WAVES + w;             // this wave's slice of the positions
    int64_t pos_lo = ncand * s / p.S, pos_hi = ncand * (s + 1) / p.S;
    int64_t max_len = (ncand + p.S - 1) / p.S;
    const int32_t* rlist = nullptr;
    if (region) {
        const int64_t rg = urel * p.S + s;
        rlist = p.reg_idx + urel * p.t.I_local + pos_lo;
        pos_lo = 0, pos_hi = p.reg_cnt[rg];
        // the workgroup's waves walk the same number of tiles (barriers): the longest of their eight lists
        int* tmax = reinterpret_cast<int*>(keys_all + (size_t)NS_WAVES * p.cap) + NS_WAVES * 4;
        if (tid == 0) *tmax = 0;
        __syncthreads();
        if (lane == 0) atomicMax(tmax, (int)pos_hi);
        __syncthreads();
        max_len = *tmax;
    }
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
            if (region) {
                gitem = rlist[pos];
                il = (int64_t)gitem - p.t.item_offset;
            } else if (p.t.cand_indptr) {
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
            oi = region ? -1
                 : p.t.cand_indptr ? (s == 0 ? el_fill_masked(p.t, e0, e1, c0, c1, t - nv) : -1)
                                   : el_fill_masked_range(p.t, p.t.item_offset + pos_lo, p.t.item_offset + pos_hi, e0, e1, c0, c1, t - nv);
            ov = -INFINITY;
        }
        p.part_idx[orow + t] = oi;
        p.part_val[orow + t] = ov;
    }
}

// =========================================================================================================================
// Screened route (EL_NMF_SCREEN): layers 2-3 on the half-precision matrix instruction with a per-pair error bound, the exact fp32 kernel
// above only on the pairs that can still belong to the answer.
//
// The fp32 kernel sits at 0.82 of the fp32 MFMA peak and still needs 0.33 s per 128 users against 1 M items: 36 F^2 flop per pair
// are what they are.  v_mfma_f32_32x32x16_f16 is 16x faster, and its result need not be right, only boundedly wrong (half precision
// keeps 11 bits where bf16 keeps 8, and the bound decides everything: the survivors are a Gaussian tail in E / sigma(logit)):
//   x   = relu(PU_u + b1 + PI_i)                      the exact kernel's layer-1 activation (fp32)
//   x'  = relu_h(h(PU_u + b1) +_h h(PI_i))            h = round to half, +_h / relu_h = packed half arithmetic (one rounding);
//   ||x' - x|| <= dh1 = R_i + R_u + uh ||x'|| + 4 u32 ||x'||,   R_i = ||PI_i - h(PI_i)|| per item (k_nmf_pib), R_u the same for the
//                                                     user's vector (per workgroup), uh = 2^-11 (1 + 2^-11)
//   z2' - z2 = W2h^T (x' - x) + (W2h - W2)^T x + (fp32 accumulation of both kernels)        W2h = h(W2)
//   ||z2' - z2|| <= s2 dh1 + (d2 + g2) (||x'|| + dh1),    s2 = ||W2h||_2, d2 = ||W2h - W2||_2 (SPECTRAL norms: a ReLU network's layers
//   are Lipschitz in them; Frobenius or column-sum bounds accumulate sqrt(width) per layer and exceed the spread of the logits),
//   g2 = 3.4 K 2^-24 ||W2h||_F (k_nmf_spec_finish)
//   y' = h(relu(z2' + b2)):   dh2 = ||y' - h2|| <= dz2 + uh ||y'|| + 4 u32 ||y'||      (ReLU is 1-Lipschitz)
//   dz3 <= s3 dh2 + (d3 + g3) (||y'|| + dh2),   h3' = relu(z3' + b3) in fp32
//   |logit' - logit| <= E = 1.02 ||hw_mlp||_2 dz3 + 4e-5 sum |head terms|   (mf part: the same fp32 products, another order)
// ||x'|| and ||y'|| are the pair's own (summed in the kernel), the norms of the four matrices are computed on the device per
// call (k_nmf_gram ...: upper bounds from the trace of (W^T W)^16).  Selection: k_nmf_screen writes the UPPER bound logit' + E of
// every pair and keeps, per workgroup, the k largest LOWER bounds logit' - E of unmasked items; the merge of the lists gives the user's
// threshold T = the k-th largest lower bound of the whole catalogue (k items are certainly at or above T, so nothing whose upper bound
// is below T can be in the exact top-k); k_nmf_compact collects the unmasked items with upper bound >= T slice by slice, the exact
// kernel scores those (NsParams.reg_*) and the usual merge returns the lists -- the same index lists and logit bits as the unscreened
// call, because the exact kernel computes them.  When more than half of the pairs survive (the bound is a worst case over
// directions; a network whose logits barely move between items leaves it no room) or a user has fewer than k candidates, the
// call takes the unscreened route.
typedef unsigned short u16;
typedef _Float16 ns_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ns_h2 __attribute__((ext_vector_type(2)));
typedef float ns_f8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32 ns_f2h(float x) { return (u32)__builtin_bit_cast(unsigned short, (_Float16)x); }     // v_cvt_f16_f32: nearest even
__device__ __forceinline__ float ns_h2f(u32 h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }

struct NsPackB {
    const float *W2, *b2, *W3, *b3, *hw;
    u16 *W2B, *W3B;
    float *b2E, *b3E, *hwE;
    int H1, H2, H3, F, H1Q, H2P, H3P;                   // H1Q = 16 x the layer-2 k-steps of the kernel (256 or 512)
};

// fragment images [k-step][mt][m][g][8 halves]: what lane (m, g) feeds the matrix instruction as its A fragment of the k-step
// (row m of output tile mt, k positions 8 g .. 8 g + 7).  Layer 2: position p of a 16-wide chunk is k = chunk + ns_perm16(p), the
// order of the PI / PU / b1P images.  Layer 3: its k index runs over the layer-2 ACCUMULATORS -- lane (n, g) holds rows
// 4 g + 8 q + t of tile mt in registers r = 4 q + t, and hands registers 0..7 to k-step 2 mt, 8..15 to k-step 2 mt + 1.
__global__ __launch_bounds__(256) void k_nmf_pack_h16(NsPackB q) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n2 = (int64_t)q.H1Q * q.H2P;
    if (e < n2) {
        const int MT = q.H2P / 32;
        const int j = (int)(e & 7), g = (int)((e >> 3) & 1), m = (int)((e >> 4) & 31);
        const int64_t rest = e >> 9;
        const int mt = (int)(rest % MT), c2 = (int)(rest / MT);
        const int kpos = 16 * c2 + 8 * g + j;
        const int k = (kpos & ~15) + ns_perm16(kpos & 15), feat = 32 * mt + m;
        q.W2B[e] = (u16)ns_f2h((k < q.H1 && feat < q.H2) ? q.W2[(int64_t)k * q.H2 + feat] : 0.f);
    }
    const int64_t n3 = (int64_t)q.H2P * q.H3P;
    if (e < n3) {
        const int MT = q.H3P / 32;
        const int j = (int)(e & 7), g = (int)((e >> 3) & 1), m = (int)((e >> 4) & 31);
        const int64_t rest = e >> 9;
        const int mt = (int)(rest % MT), c2 = (int)(rest / MT);
        const int src_mt = c2 >> 1, qq = 2 * (c2 & 1) + (j >> 2), t = j & 3;
        const int f2 = 32 * src_mt + 4 * g + 8 * qq + t, feat = 32 * mt + m;
        q.W3B[e] = (u16)ns_f2h((f2 < q.H2 && feat < q.H3) ? q.W3[(int64_t)f2 * q.H3 + feat] : 0.f);
    }
    // per-feature vectors in accumulator order [tile][g][r]: feature 32 tile + 4 g + 8 (r / 4) + r % 4
    if (e < q.H2P) {
        const int tile = (int)e >> 5, g = ((int)e >> 4) & 1, r = (int)e & 15, f = 32 * tile + 4 * g + 8 * (r >> 2) + (r & 3);
        q.b2E[e] = f < q.H2 ? q.b2[f] : 0.f;
    }
    if (e < q.H3P) {
        const int tile = (int)e >> 5, g = ((int)e >> 4) & 1, r = (int)e & 15, f = 32 * tile + 4 * g + 8 * (r >> 2) + (r & 3);
        q.b3E[e] = f < q.H3 ? q.b3[f] : 0.f;
        q.hwE[e] = f < q.H3 ? q.hw[q.F + f] : 0.f;
    }
}

// half-precision image of the PI rows + R_i = ||PI_i - h(PI_i)||_2 (one wave per item; skipped while *rebuild == 0: ctl[3] above)
__global__ __launch_bounds__(256) void k_nmf_pib(const float* __restrict__ PI, int64_t I, int H1P, int KP, u16* __restrict__ PIB, float* __restrict__ Rn,
                                                 const unsigned long long* __restrict__ rebuild) {
    if (rebuild && *rebuild == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= I) return;
    float ss = 0.f;
    for (int c = lane; c < KP; c += 64) {                // rows of KP = 16 x the screen kernel's k-steps, zero past H1P
        const float v = c < H1P ? PI[row * H1P + c] : 0.f;
        const u32 h = ns_f2h(v);
        PIB[row * KP + c] = (u16)h;
        const float d = v - ns_h2f(h);
        ss = __builtin_fmaf(d, d, ss);
    }
    ss = el_group_sum(ss, 64);
    if (lane == 0) Rn[row] = sqrtf(ss) * 1.0001f;
}

// cst[0..7] = s2, d2, g2, s3, d3, g3, ||hw_mlp||_2, 0 -- the constants of the bound above.  The four spectral norms (of h(W2), h(W2) - W2,
// h(W3), h(W3) - W3) are UPPER bounds, not estimates: with G = V^T V (N x N, N <= 256) and M_0 = G / tr G, M_{j+1} = (M_j / tr M_j)^2,
//   sigma_max(V)^2 = tr G * lambda_max(M_0),   lambda_max(M_j / tr M_j) <= sqrt(tr M_{j+1} * lambda_max(M_{j+1} / tr M_{j+1})),   lambda_max(.) <= 1
// (the Schatten-32 norm after four squarings: (sum sigma_i^32)^(1/32), a few percent above sigma_max for these spectra, where a power
// iteration approaches it from below).  Matrix m = 0..3 as listed; one workgroup per 16 rows of the N x N product.
struct NsSpec {
    const float *W2, *W3, *hw;
    int K2, N2, K3, N3, F;
    float* gram;                                         // [4 matrices][2][256 * 256]
    float* cst;
};
__device__ __forceinline__ float ns_spec_val(const NsSpec& q, int m, int k, int n) {
    const float w = m < 2 ? q.W2[(int64_t)k * q.N2 + n] : q.W3[(int64_t)k * q.N3 + n];
    const float wh = ns_h2f(ns_f2h(w));
    return (m & 1) ? wh - w : wh;
}
// G = V^T V: thread = column j, 16 rows i per workgroup
__global__ __launch_bounds__(256) void k_nmf_gram(NsSpec q) {
    const int m = blockIdx.y, K = m < 2 ? q.K2 : q.K3, N = m < 2 ? q.N2 : q.N3;
    const int i0 = blockIdx.x * 16, j = threadIdx.x;
    if (i0 >= N) return;
    __shared__ float vi[16];
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; ++k) {
        __syncthreads();
        if (j < 16) vi[j] = (i0 + j) < N ? ns_spec_val(q, m, k, i0 + j) : 0.f;
        __syncthreads();
        const float vj = j < N ? ns_spec_val(q, m, k, j) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(vi[r], vj, acc[r]);
    }
    float* G = q.gram + (size_t)m * 2 * 65536;
    if (j < N)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (i0 + r < N) G[(i0 + r) * 256 + j] = acc[r];
}
// trace of an N x N matrix held with row stride 256, the same order of additions in every workgroup
__device__ __forceinline__ float ns_trace(const float* __restrict__ A, int N, float* red) {
    const int tid = threadIdx.x;
    float v = tid < N ? A[tid * 256 + tid] : 0.f;
    v = el_group_sum(v, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// out = (in / tr in)^2; trs[m][step] = tr in
__global__ __launch_bounds__(256) void k_nmf_gram_sq(NsSpec q, int step, float* __restrict__ trs) {
    const int m = blockIdx.y, N = m < 2 ? q.N2 : q.N3;
    const int i0 = blockIdx.x * 16, j = threadIdx.x;
    if (i0 >= N) return;
    __shared__ float red[4], ai[16];
    const float* in = q.gram + ((size_t)m * 2 + (step & 1)) * 65536;
    float* out = q.gram + ((size_t)m * 2 + ((step + 1) & 1)) * 65536;
    const float tr = ns_trace(in, N, red);
    if (blockIdx.x == 0 && j == 0) trs[m * 8 + step] = tr;
    const float inv = tr > 0.f ? 1.0f / tr : 0.f;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < N; ++k) {
        __syncthreads();
        if (j < 16) ai[j] = (i0 + j) < N ? in[(i0 + j) * 256 + k] * inv : 0.f;
        __syncthreads();
        const float bj = j < N ? in[k * 256 + j] * inv : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(ai[r], bj, acc[r]);
    }
    if (j < N)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (i0 + r < N) out[(i0 + r) * 256 + j] = acc[r];
}
#define NS_SPEC_SQ 4
__global__ __launch_bounds__(256) void k_nmf_spec_finish(NsSpec q, const float* __restrict__ trs) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    if (m == 4) {                                        // the head weights of the MLP part
        float ss = 0.f;
        for (int f = tid; f < q.N3; f += 256) ss += q.hw[q.F + f] * q.hw[q.F + f];
        ss = el_group_sum(ss, 64);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        if (tid == 0) q.cst[6] = sqrtf((red[0] + red[1]) + (red[2] + red[3])) * 1.0001f, q.cst[7] = 0.f;
        return;
    }
    const int K = m < 2 ? q.K2 : q.K3, N = m < 2 ? q.N2 : q.N3;
    const float trl = ns_trace(q.gram + ((size_t)m * 2 + (NS_SPEC_SQ & 1)) * 65536, N, red);     // tr M_p
    if (tid != 0) return;
    float lam = 1.0f;                                    // lambda_max(M_p / tr M_p) <= 1
    for (int jstep = NS_SPEC_SQ; jstep >= 1; --jstep) {  // lambda_max(M_{j-1} / tr M_{j-1}) = sqrt(tr M_j * lambda_max(M_j / tr M_j))
        const float tj = jstep == NS_SPEC_SQ ? trl : trs[m * 8 + jstep];
        lam = sqrtf(tj * lam) * 1.0001f;
    }
    const float trG = trs[m * 8 + 0];                    // tr G = ||V||_F^2
    const float fro = sqrtf(trG) * 1.0001f;
    // + 1 %: the fp32 sums of the squarings (N <= 256 terms; worst for a flat spectrum, where the cap below takes over anyway);
    // never more than the Frobenius norm
    const float sig = fminf(sqrtf(trG * lam) * 1.01f, fro) + 1e-30f;
    q.cst[m < 2 ? m : m + 1] = sig;
    // g = 3.4 K 2^-24 ||W||_F: the exact kernel's fma chain (K u) + this route's matrix instruction (taken as two roundings per
    // product and per accumulate: (K + K / 16) 2 u), both against sum |w| |x| <= || |W| ||_2 ||x|| <= ||W||_F ||x||
    if ((m & 1) == 0) q.cst[m < 2 ? 2 : 5] = 3.4f * (float)K * 5.96e-8f * fro * 1.01f;
}

// sum over the aligned 16 lanes of a row, in every lane of the row (four v_add_f32 with row_ror: no LDS traffic)
__device__ __forceinline__ float ns_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// workgroup barrier for LDS hand-overs only: waits for this wave's LDS operations, NOT for its global loads in flight (__syncthreads()
// carries a fence that drains vmcnt as well -- the next tile's prefetched rows would be waited for at every barrier)
__device__ __forceinline__ void ns_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct NsScreenParams {
    const u16 *W2B, *W3B, *PIB;
    const float *Rn, *cst, *b2E, *b3E, *hwE, *hw;        // hw: the head weights as the model holds them ([F mf ; H3 mlp])
    float* up;                                           // [n_users][I_local] upper bound logit' + E
    int spb;                                             // slices (of the exact kernel's S) per workgroup: 1, 2, 4 or 8
};

// One workgroup of 8 waves = NSC_UB users x 8 consecutive slices of the item range, one user x 32 items (one matrix-instruction column
// tile of pairs) at a time; an item tile's rows are fetched once and serve the workgroup's users in turn:
//   P1  all 512 threads build the tile's x' in LDS as B fragments (16 threads per pair, 16-byte pieces of the PI image and the mf row
//       prefetched one tile ahead; v_pk_add_f16 / v_pk_max_f16 / v_dot2_f32_f16: 1.5 instructions per element), and the mf part of the head
//   P2  wave w owns output features 32 w .. 32 w + 31 of layer 2: its slice of W2 lives in REGISTERS for the whole kernel (4 VGPRs
//       per 16-wide k-step: nothing is streamed, nothing waits on a barrier inside the k loop), x' comes from LDS, one matrix
//       instruction per k-step; relu(. + b2) -> half -> LDS as the layer-3 B fragments
//   P3  waves 0 .. H3P / 32 - 1: layer 3 (W3 image resident in LDS), relu(. + b3) . hw -> the pair's partial logits
//   P4  wave 4, one tile behind, while layer 3 of the next tile runs: error bound, upper bound out, the k largest lower bounds of
//       the range (one list per workgroup)
// NSTM = 16-wide k-steps of layer 2 (units[0] padded to 256 or 512).
#define NSC_WAVES 8
#define NSC_P4 4                                         // the wave that finishes tiles (no layer-3 tile: H3P <= 128)
#ifndef NSC_UB
#define NSC_UB 4                                         // users per workgroup: an item tile's rows are fetched once for all of them
#endif
template <int H2P, int H3P, int NSTM>
__global__ __launch_bounds__(NSC_WAVES * 64) void k_nmf_screen(NsParams p, NsScreenParams q) {
    constexpr int NTH = NSC_WAVES * 64;
    constexpr int NT2 = H2P / 32, NT3 = H3P / 32, NK3 = H2P / 16;
    constexpr int NCH = 2 * NSTM, CPT = NCH / 16;       // 8-position pieces of a PI row; pieces per thread (16 threads per pair)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                                     // [NSTM][lane = 32 g + pair, swizzled][8 halves]   B fragments of layer 2
    char* ys = xs + NSTM * 1024;                         // [NK3][lane][8]                         B fragments of layer 3
    char* w3s = ys + NK3 * 1024;                         // [NK3][NT3][row m][g][8]                A fragments of layer 3
    _Float16* aus = reinterpret_cast<_Float16*>(w3s + (size_t)NK3 * NT3 * 1024);   // [UB][NSTM * 16]  h(PU_u + b1) (image order), zero past H1P
    float* b2s = reinterpret_cast<float*>(aus + NSC_UB * NSTM * 16);    // [H2P] accumulator order
    float* b3s = b2s + H2P;                              // [H3P]
    float* hws = b3s + H3P;                              // [H3P]
    float* ums = hws + H3P;                              // [UB][128]  (F <= 128 on this route, zero past F)
    float* hms = ums + NSC_UB * 128;                     // [128]
    float* cs = hms + 128;                               // [8]
    float* rus = cs + 8;                                 // [UB][8] per-wave parts of ||(PU_u + b1) - h(PU_u + b1)||^2
    float* pq1 = rus + NSC_UB * 8;                       // [2 parity][4: nrm1, mf, |mf|, R_i][32]
    float* pq2 = pq1 + 2 * 4 * 32;                       // [2][8 waves][32]  nrm2 parts
    float* pq3 = pq2 + 2 * 8 * 32;                       // [2][4 tiles][2: logit, |.|][32]
    int64_t* es = reinterpret_cast<int64_t*>(pq3 + 2 * 4 * 2 * 32);      // [UB][2] the users' exclusion rows
    float* taus = reinterpret_cast<float*>(es + NSC_UB * 2);              // [UB] list thresholds ...
    int* cnts = reinterpret_cast<int*>(taus + NSC_UB);                   // [UB] ... and fills (kept by the finishing wave between its turns)
    u64* keys = reinterpret_cast<u64*>(cnts + NSC_UB);                   // [UB][cap]
    int* cnt_s = reinterpret_cast<int*>(keys + (size_t)NSC_UB * p.cap);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, h = lane >> 5;

    const int64_t nu = p.t.u_stop - p.t.u_start;
    const int64_t urel0 = (int64_t)blockIdx.x * NSC_UB;
    const int nub = (int)((nu - urel0) < NSC_UB ? (nu - urel0) : NSC_UB);      // users of this workgroup
    if (tid < nub) {
        const int64_t user = p.t.u_start + urel0 + tid;
        es[2 * tid] = p.t.excl_indptr ? p.t.excl_indptr[user] : 0;
        es[2 * tid + 1] = p.t.excl_indptr ? p.t.excl_indptr[user + 1] : 0;
        taus[tid] = -INFINITY;
        cnts[tid] = 0;
    }
    const bool use_excl = p.t.excl_indptr != nullptr;
    const int64_t ncand = p.t.I_local;
    const int s0 = blockIdx.y * q.spb;                   // this workgroup's slices s0 .. s0 + spb - 1 of the exact kernel's S
    const int64_t pos_lo = ncand * s0 / p.S, pos_hi = ncand * (s0 + q.spb) / p.S;
    const int T = (int)((pos_hi - pos_lo + 31) / 32);

    for (int uu = 0; uu < nub; ++uu) {
        float rs = 0.f;
        for (int t = tid; t < NSTM * 16; t += NTH) {
            const float a = t < p.H1P ? p.PU[(urel0 + uu) * p.H1P + t] + p.b1P[t] : 0.f;
            const _Float16 ah = (_Float16)a;
            aus[uu * NSTM * 16 + t] = ah;
            const float d = a - (float)ah;
            rs = __builtin_fmaf(d, d, rs);
        }
        rs = el_group_sum(rs, 64);
        if (lane == 0) rus[uu * 8 + w] = rs;
    }
    for (int t = tid; t < H2P; t += NTH) b2s[t] = q.b2E[t];
    for (int t = tid; t < H3P; t += NTH) {
        b3s[t] = q.b3E[t];
        hws[t] = q.hwE[t];
    }
    for (int t = tid; t < 128 * nub; t += NTH) {
        const int uu = t >> 7, f = t & 127;
        ums[t] = f < p.F ? p.Umf[(p.t.u_start + urel0 + uu) * (int64_t)p.F + f] : 0.f;      // (p.F = 0 without an mf part)
    }
    for (int t = tid; t < 128; t += NTH) hms[t] = t < p.F ? q.hw[t] : 0.f;
    if (tid < 8) cs[tid] = q.cst[tid];
    for (int e4 = tid; e4 < NK3 * NT3 * 64; e4 += NTH) reinterpret_cast<float4*>(w3s)[e4] = reinterpret_cast<const float4*>(q.W3B)[e4];
    // this wave's slice of the W2 image: A fragment of k-step ks = 16 bytes at [ks][tile w][lane]
    ns_h8 wreg[NSTM];
    if (w < NT2) {
#pragma unroll
        for (int ks = 0; ks < NSTM; ++ks)
            wreg[ks] = *reinterpret_cast<const ns_h8*>(reinterpret_cast<const char*>(q.W2B) + (size_t)(ks * NT2 + w) * 1024 + n * 32 + h * 16);
    }
    const float hbias = p.hb ? *p.hb : 0.f;
    // P1 roles: pair pp of the tile, piece lane pj of 16
    const int pp = tid >> 4, pj = tid & 15;
    uint4 pre[CPT];
    float4 mpre[2];                                      // this thread's two 4-feature pieces of the item's mf row (F <= 128, F % 4 == 0)
    float rpre = 0.f;                                    // R_i (thread 0 of the pair)
    const bool mf0 = 4 * pj < p.F, mf1 = 4 * pj + 64 < p.F;
    auto prefetch = [&](int tile) {
        const int64_t pos = pos_lo + (int64_t)tile * 32 + pp;
        const int64_t il = pos < pos_hi ? pos : pos_lo;
        const uint4* row = reinterpret_cast<const uint4*>(q.PIB + il * (int64_t)(NSTM * 16)) + pj;   // (rows padded with zeros to NSTM k-steps)
#pragma unroll
        for (int i = 0; i < CPT; ++i) pre[i] = row[16 * i];
        if (pj == 0) rpre = q.Rn[il];
        const float* irow = p.Imf + il * (int64_t)p.F + 4 * pj;
        mpre[0] = mf0 ? *reinterpret_cast<const float4*>(irow) : make_float4(0.f, 0.f, 0.f, 0.f);
        mpre[1] = mf1 ? *reinterpret_cast<const float4*>(irow + 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (T > 0) prefetch(0);
    __syncthreads();

    // P4 of one tile: error bound, upper bound out, the range's k largest lower bounds.  One wave (the first without a layer-3 tile),
    // one tile behind the others, while layer 3 of the next tile runs.
    auto finish_tile = [&](int tile, int uu, int par) {
        const float* ruu = rus + uu * 8;
        const float ru = sqrtf(((ruu[0] + ruu[1]) + (ruu[2] + ruu[3])) + ((ruu[4] + ruu[5]) + (ruu[6] + ruu[7]))) * 1.0001f;
        const int64_t e0 = es[2 * uu], e1 = es[2 * uu + 1];
        float tau = taus[uu];
        int cnt = cnts[uu];
        u64* ukeys = keys + (size_t)uu * p.cap;
        float* uprow = q.up + (urel0 + uu) * p.t.I_local;
        const int64_t pos = pos_lo + (int64_t)tile * 32 + n;
        const bool valid = pos < pos_hi && h == 0;
        const int32_t gitem = valid ? (int32_t)(p.t.item_offset + pos) : -1;
        const float* o1 = pq1 + par * 128 + n;
        float nrm1 = o1[0], acc = o1[32], aabs = o1[64], nrm2 = 0.f;
        const float rni = o1[96];
#pragma unroll
        for (int x = 0; x < NT2; ++x) nrm2 += pq2[(par * 8 + x) * 32 + n];
#pragma unroll
        for (int x = 0; x < NT3; ++x) {
            const float* o3 = pq3 + (par * 4 + x) * 64 + n;
            acc += o3[0], aabs += o3[32];
        }
        const float logit = acc + hbias;
        const float s2 = cs[0], d2 = cs[1], g2 = cs[2], s3 = cs[3], d3 = cs[4], g3 = cs[5], hwn = cs[6];
        const float uh = 4.886e-4f;                        // > 2^-11 (1 + 2^-11): one rounding to half, against the rounded value
        // (+ 1.5e-3: v_dot2_f32_f16 may drop subnormal halves, 512 squares below 2^-28 each)
        const float r1 = sqrtf(nrm1) * 1.0001f + 1.5e-3f, r2 = sqrtf(nrm2) * 1.0001f;
        const float dh1 = rni + ru + uh * r1 + 2.4e-7f * r1;
        const float dz2 = s2 * dh1 + d2 * (r1 + dh1) + g2 * (r1 + dh1);
        const float dh2 = dz2 + uh * r2 + 2.4e-7f * r2;
        const float dz3 = s3 * dh2 + d3 * (r2 + dh2) + g3 * (r2 + dh2);
        const float E = 1.02f * hwn * dz3 + 4e-5f * (aabs + fabsf(hbias)) + 1e-30f;
        // an activation past the half-precision range (inf, then NaN) leaves nothing to bound: the pair goes to the exact kernel
        const bool bounded = E < INFINITY && fabsf(logit) < INFINITY;
        const float lo = bounded ? logit - E : -INFINITY, up = bounded ? logit + E : INFINITY;
        if (valid) uprow[pos] = up;
        bool hit = valid && (lo >= tau);
        if (hit && use_excl) hit = !el_row_contains(p.t.excl_indices, e0, e1, gitem);
        const u64 bal = __ballot(hit);
        if (bal) {
            const int offp = __popcll(bal & ((1ull << lane) - 1ull));
            if (hit) ukeys[cnt + offp] = el_make_key(lo, gitem);
            cnt += __popcll(bal);
        }
        if (cnt > p.cap - 32) {
            if (lane == 0) *cnt_s = cnt;
            tau = el_wave_compact(ukeys, cnt_s, p.cap, p.t.k, lane);
            cnt = cnt < p.t.k ? cnt : p.t.k;
        }
        if (lane == 0) taus[uu] = tau, cnts[uu] = cnt;
    };

    // iterations = (tile, user): the tile's item rows sit in the prefetch registers for all of the workgroup's users
    int par = 0, ptile = 0, puu = 0;
    bool first = true;
    for (int tile = 0; tile < T; ++tile)
    for (int uu = 0; uu < nub; ++uu, par ^= 1) {
        const _Float16* uaus = aus + uu * NSTM * 16;
        const float* uums = ums + uu * 128;
        // ---------------- P1: x' tile in packed half arithmetic, ||x'||^2, mf part -------------------------------------------------
        {
            float nrm1 = 0.f;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = pj + 16 * i;
                union {
                    uint4 u;
                    ns_h2 v[4];
                } pi, au, xq;
                pi.u = pre[i];                            // (zero past the live pieces, like aus: x' = 0 there)
                au.u = *reinterpret_cast<const uint4*>(uaus + 8 * c);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    xq.v[x] = __builtin_elementwise_max(au.v[x] + pi.v[x], (ns_h2)(_Float16)0);      // v_pk_add_f16, v_pk_max_f16
                    nrm1 = __builtin_amdgcn_fdot2(xq.v[x], xq.v[x], nrm1, false);
                }
                // row ks = c / 2, slot (32 g + pair) ^ (2 (ks & 3) + g): the 8 lanes of a store group (one pair, 4 k-steps x 2 halves)
                // land in 8 different 16-byte columns instead of one
                *reinterpret_cast<uint4*>(xs + (size_t)(c >> 1) * 1024 + (size_t)((((c & 1) * 32 + pp) ^ (2 * ((c >> 1) & 3) + (c & 1))) * 16)) = xq.u;
            }
            float acc = 0.f, aabs = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {                  // (all zero without an mf part)
                const float4 uv = *reinterpret_cast<const float4*>(uums + 4 * pj + 64 * i), ww = *reinterpret_cast<const float4*>(hms + 4 * pj + 64 * i);
                const float t0 = uv.x * mpre[i].x, t1 = uv.y * mpre[i].y, t2 = uv.z * mpre[i].z, t3 = uv.w * mpre[i].w;
                acc = __builtin_fmaf(ww.x, t0, acc), aabs = __builtin_fmaf(fabsf(ww.x), fabsf(t0), aabs);
                acc = __builtin_fmaf(ww.y, t1, acc), aabs = __builtin_fmaf(fabsf(ww.y), fabsf(t1), aabs);
                acc = __builtin_fmaf(ww.z, t2, acc), aabs = __builtin_fmaf(fabsf(ww.z), fabsf(t2), aabs);
                acc = __builtin_fmaf(ww.w, t3, acc), aabs = __builtin_fmaf(fabsf(ww.w), fabsf(t3), aabs);
            }
            nrm1 = ns_row16_sum(nrm1), acc = ns_row16_sum(acc), aabs = ns_row16_sum(aabs);
            if (pj == 0) {
                float* o = pq1 + par * 128 + pp;
                o[0] = nrm1, o[32] = acc, o[64] = aabs, o[96] = rpre;
            }
            if (uu + 1 == nub && tile + 1 < T) prefetch(tile + 1);       // (the registers are free after the last user's turn)
        }
        ns_lds_barrier();
        // ---------------- P2: layer 2, this wave's 32 features x the tile's 32 pairs ----------------------------------------------
        if (w < NT2) {
            floatx16 ca;                                   // (one chain: a matrix instruction that accumulates onto its predecessor's
#pragma unroll                                           //  result issues back to back, and the other wave of the SIMD fills what is left)
            for (int r = 0; r < 16; ++r) ca[r] = 0.f;
            const char* xl[4];                              // (the swizzle of P1: slot ^ (2 (ks & 3) + h))
#pragma unroll
            for (int x = 0; x < 4; ++x) xl[x] = xs + (size_t)((lane ^ h) ^ (2 * x)) * 16;
            // B fragments four k-steps ahead of the matrix instructions that use them (left alone the compiler hoists all NSTM loads)
            ns_h8 bq[2][4];
#pragma unroll
            for (int x = 0; x < 4; ++x) bq[0][x] = *reinterpret_cast<const ns_h8*>(xl[x] + x * 1024);
#pragma unroll
            for (int g4 = 0; g4 < NSTM / 4; ++g4) {
                if (g4 + 1 < NSTM / 4) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) bq[(g4 + 1) & 1][x] = *reinterpret_cast<const ns_h8*>(xl[x] + (4 * (g4 + 1) + x) * 1024);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[4 * g4 + x], bq[g4 & 1][x], ca, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // y' = h(relu(z2' + b2)): registers 0..7 are this lane's B fragment of layer-3 k-step 2 w, 8..15 of 2 w + 1 (W3 image order)
            const float4* bb = reinterpret_cast<const float4*>(b2s + w * 32 + h * 16);
            float nrm2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float4 q0 = bb[2 * kk], q1 = bb[2 * kk + 1];
                ns_f8 yv;
                yv[0] = fmaxf(ca[8 * kk + 0] + q0.x, 0.f), yv[1] = fmaxf(ca[8 * kk + 1] + q0.y, 0.f);
                yv[2] = fmaxf(ca[8 * kk + 2] + q0.z, 0.f), yv[3] = fmaxf(ca[8 * kk + 3] + q0.w, 0.f);
                yv[4] = fmaxf(ca[8 * kk + 4] + q1.x, 0.f), yv[5] = fmaxf(ca[8 * kk + 5] + q1.y, 0.f);
                yv[6] = fmaxf(ca[8 * kk + 6] + q1.z, 0.f), yv[7] = fmaxf(ca[8 * kk + 7] + q1.w, 0.f);
                union {
                    ns_h8 v;
                    ns_h2 d[4];
                } yq;
                yq.v = __builtin_convertvector(yv, ns_h8);               // v_cvt_pk_f16_f32, round to nearest even
#pragma unroll
                for (int x = 0; x < 4; ++x) nrm2 = __builtin_amdgcn_fdot2(yq.d[x], yq.d[x], nrm2, false);
                *reinterpret_cast<ns_h8*>(ys + (size_t)(2 * w + kk) * 1024 + (size_t)lane * 16) = yq.v;
            }
            nrm2 += __uint_as_float(el_partner32(__float_as_uint(nrm2), h));
            if (h == 0) pq2[(par * 8 + w) * 32 + n] = nrm2;
        }
        ns_lds_barrier();
        // ---------------- P3: layer 3 + head, one 32-feature tile per wave; P4 of the previous tile on the first wave without one ----
        if (w < NT3) {
            floatx16 ca;
#pragma unroll
            for (int r = 0; r < 16; ++r) ca[r] = 0.f;
            const char* yl = ys + (size_t)lane * 16;
            const char* wl = w3s + (size_t)w * 1024 + n * 32 + h * 16;
#pragma unroll
            for (int ks = 0; ks < NK3; ++ks)
                ca = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const ns_h8*>(wl + (size_t)ks * NT3 * 1024),
                                                            *reinterpret_cast<const ns_h8*>(yl + ks * 1024), ca, 0, 0, 0);
            const float4* bb = reinterpret_cast<const float4*>(b3s + w * 32 + h * 16);
            const float4* hh = reinterpret_cast<const float4*>(hws + w * 32 + h * 16);
            float acc = 0.f, aabs = 0.f;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float4 b = bb[x], gq = hh[x];
                const float y0 = fmaxf(ca[4 * x + 0] + b.x, 0.f), y1 = fmaxf(ca[4 * x + 1] + b.y, 0.f);
                const float y2 = fmaxf(ca[4 * x + 2] + b.z, 0.f), y3 = fmaxf(ca[4 * x + 3] + b.w, 0.f);
                acc = __builtin_fmaf(gq.x, y0, acc), aabs = __builtin_fmaf(fabsf(gq.x), y0, aabs);
                acc = __builtin_fmaf(gq.y, y1, acc), aabs = __builtin_fmaf(fabsf(gq.y), y1, aabs);
                acc = __builtin_fmaf(gq.z, y2, acc), aabs = __builtin_fmaf(fabsf(gq.z), y2, aabs);
                acc = __builtin_fmaf(gq.w, y3, acc), aabs = __builtin_fmaf(fabsf(gq.w), y3, aabs);
            }
            acc += __uint_as_float(el_partner32(__float_as_uint(acc), h));
            aabs += __uint_as_float(el_partner32(__float_as_uint(aabs), h));
            if (h == 0) {
                float* o = pq3 + (par * 4 + w) * 64 + n;
                o[0] = acc, o[32] = aabs;
            }
        } else if (w == NSC_P4 && !first) {
            finish_tile(ptile, puu, par ^ 1);              // (its parity's sums are not rewritten before the next iteration's phases, all past the barrier below)
        }
        ns_lds_barrier();
        ptile = tile, puu = uu, first = false;
    }
    if (w == NSC_P4 && !first) finish_tile(ptile, puu, par ^ 1);
    // one list per user and workgroup, filed under the workgroup's first slice; its other slices stay empty for the merge
    for (int uu = 0; uu < nub; ++uu) {
        if (w == NSC_P4) {
            el_wave_lds_sync();
            const int cnt = cnts[uu];
            u64* ukeys = keys + (size_t)uu * p.cap;
            if (lane == 0) *cnt_s = cnt;
            el_wave_compact(ukeys, cnt_s, p.cap, p.t.k, lane);
            const int nv = cnt < p.t.k ? cnt : p.t.k;
            const int64_t orow = ((int64_t)s0 * nu + urel0 + uu) * p.t.k;
            for (int t = lane; t < p.t.k; t += 64) {
                p.part_idx[orow + t] = t < nv ? el_key_item(ukeys[t]) : -1;
                p.part_val[orow + t] = t < nv ? el_key_score(ukeys[t]) : -INFINITY;
            }
        } else {
            for (int sl = 1 + (w < NSC_P4 ? w : w - 1); sl < q.spb; sl += NSC_WAVES - 1) {
                const int64_t orow = ((int64_t)(s0 + sl) * nu + urel0 + uu) * p.t.k;
                for (int t = lane; t < p.t.k; t += 64) {
                    p.part_idx[orow + t] = -1;
                    p.part_val[orow + t] = -INFINITY;
                }
            }
        }
    }
}

// the unmasked items of slice s of user u whose upper bound reaches the user's threshold T (= the k-th largest lower bound over the
// catalogue; -inf when fewer than k unmasked items exist), compacted in ascending order at the start of the slice's own position
// range.  One wave per (user, slice).
__global__ __launch_bounds__(256) void k_nmf_compact(TopkParams t, const float* __restrict__ up, const float* __restrict__ thr, int S,
                                                     int32_t* __restrict__ reg_idx, int32_t* __restrict__ reg_cnt, int32_t* sflag) {
    const int lane = threadIdx.x & 63;
    const int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_users = t.u_stop - t.u_start;
    if (rg >= n_users * S) return;
    const int64_t urel = rg / S;
    const int s = (int)(rg % S);
    const int64_t user = t.u_start + urel;
    const float T = thr[urel * t.k + (t.k - 1)];
    int64_t e0 = 0, e1 = 0;
    if (t.excl_indptr) {
        e0 = t.excl_indptr[user];
        e1 = t.excl_indptr[user + 1];
    }
    const int64_t pos_lo = t.I_local * s / S, pos_hi = t.I_local * (s + 1) / S;
    const float* urow = up + urel * t.I_local;
    int32_t* out = reg_idx + urel * t.I_local + pos_lo;
    int cnt = 0;
    for (int64_t base = pos_lo; base < pos_hi; base += 64) {
        const int64_t pos = base + lane;
        bool c = pos < pos_hi && urow[pos] >= T;
        const int32_t gitem = (int32_t)(t.item_offset + pos);
        if (c && t.excl_indptr) c = !el_row_contains(t.excl_indices, e0, e1, gitem);
        const u64 b = __ballot(c);
        if (c) out[cnt + __popcll(b & ((1ull << lane) - 1ull))] = gitem;       // (cnt + prefix <= pos - pos_lo: never past the reader)
        cnt += __popcll(b);
    }
    if (lane == 0) {
        reg_cnt[rg] = cnt;
        atomicAdd(reinterpret_cast<unsigned long long*>(sflag + 2), (unsigned long long)cnt);
    }
}

// a user whose regions hold fewer than k candidates (a nearly empty catalogue after masking, non-finite logits) sends the call
// through the unscreened route, which pads such rows the way top_k(where(mask, preds, -inf)) does
__global__ __launch_bounds__(256) void k_nmf_reg_check(const int32_t* __restrict__ reg_cnt, int64_t n_users, int S, int k, int32_t* sflag) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_users) return;
    int64_t tot = 0;
    for (int s = 0; s < S; ++s) tot += reg_cnt[u * S + s];
    if (tot < k) atomicOr(sflag, 2);
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

// with_cand: 0 = dense scoring, 1 = + a candidate-list scratch, 2 = + the screened route's buffers (EL_NMF_SCREEN)
extern "C" size_t el_nmf_score_ws_bytes(el_ctx* ctx, const el_nmf_state* st, int64_t n_users, int64_t I_local, int32_t k, int with_cand) {
    if (!ctx || !st || !ns_shape_ok(st) || k < 1) return 0;
    return ns_layout(ctx, st, n_users, I_local, k, with_cand == 1, with_cand == 2).total;
}

extern "C" int el_nmf_screen_stats(el_ctx* ctx, int64_t* exact_pairs, int* fell_back) {
    EL_REQUIRE(ctx != nullptr, "el_nmf_screen_stats: null context");
    if (exact_pairs) *exact_pairs = ctx->nmf_screen_cands;
    if (fell_back) *fell_back = ctx->nmf_screen_fallback ? 1 : 0;
    return 0;
}

template <int H2P, int H3P>
static int ns_launch(const NsParams& p, int64_t n_users, int nsplit, hipStream_t s) {
    const size_t lds = (size_t)(2 * H2P * 16 + 2 * p.H1P + H2P + 2 * H3P + 2 * (p.FP > 0 ? p.FP : 8)) * 4 + (size_t)NS_WAVES * p.cap * 8 +
                       (size_t)NS_WAVES * 16 + 16;
    auto kern = k_nmf_score<H2P, H3P>;
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EL_LAUNCH("k_nmf_score", kern, dim3((unsigned)n_users, (unsigned)nsplit), dim3(NS_THREADS), lds, s, p);
    EL_CHECK_LAUNCH();
    return 0;
}

template <int H2P, int H3P, int NSTM>
static int ns_launch_screen(const NsParams& p, const NsScreenParams& q, int64_t n_users, int nsplit, hipStream_t s) {
    const size_t lds = (size_t)(NSTM + H2P / 16 + (H2P / 16) * (H3P / 32)) * 1024 +
                       (size_t)(NSC_UB * NSTM * 8 + H2P + 2 * H3P + (NSC_UB + 1) * 128 + 8 + NSC_UB * 8 + 2 * 4 * 32 + 2 * 8 * 32 + 2 * 4 * 2 * 32) * 4 +
                       (size_t)NSC_UB * (16 + 8) + (size_t)NSC_UB * p.cap * 8 + 16;
    EL_REQUIRE(lds <= NS_LDS_LIMIT, "el_nmf_score_topk: the screened kernel needs %zu bytes of LDS", lds);
    auto kern = k_nmf_screen<H2P, H3P, NSTM>;
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EL_LAUNCH("k_nmf_screen", kern, dim3((unsigned)((n_users + NSC_UB - 1) / NSC_UB), (unsigned)(nsplit * NS_WAVES / q.spb)), dim3(NSC_WAVES * 64), lds, s, p, q);
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
    // screened route: half-precision matrix instruction + per-pair error bound first, the exact kernel on the surviving pairs (same lists, same
    // logit bits); a catalogue too small to be worth it, or a candidate list, takes the exact kernel alone
    const bool screen = (flags & EL_NMF_SCREEN) != 0 && !cand && I_local >= 4096 && k <= 256 && ns_up(st->units[0], 16) <= 512 &&
                        (!st->use_mf || ((st->F & 3) == 0 && st->F <= 128));
    // (el_nmf_screen_stats: what the exact kernel scores, and whether a call that asked for the screen goes without it)
    ctx->nmf_screen_cands = cand ? -1 : n_users * I_local;
    ctx->nmf_screen_fallback = (flags & EL_NMF_SCREEN) != 0 && !screen;
    if (screen) {
        // the screened route reads its survivor count back (one stream synchronisation per call): it cannot be recorded into a graph
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        EL_CHECK_HIP(hipStreamIsCapturing((hipStream_t)stream, &cs));
        EL_REQUIRE(cs == hipStreamCaptureStatusNone, "el_nmf_score_topk: EL_NMF_SCREEN synchronises the stream and cannot be captured in a "
                                                     "graph (call without the flag under capture)");
    }
    const NsLayout L = ns_layout(ctx, st, n_users, I_local, k, cand, screen);
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
    // (the claim also needs PI where the last call on this workspace left it: the regions in front of it are sized by the layer widths)
    const bool claim = (flags & EL_TOPK_ITEMS_UNCHANGED) != 0 && ctx->nmf_ws == ws && ctx->nmf_Imlp == Imlp && ctx->nmf_W1 == st->W[0] &&
                       ctx->nmf_I == I_local && ctx->nmf_E == st->E && ctx->nmf_H1 == st->units[0] && ctx->nmf_PI_off == L.PI;
    // the half-precision image of PI (screened route) is trusted only where the LAST call on this workspace was a screened one that put
    // it at the same offsets: an unscreened call in between lays its user-side regions over it
    const bool pib_in_place = claim && ctx->nmf_PIB_off == L.PIB && ctx->nmf_Rn_off == L.Rn && L.PIB != 0;
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
    ctx->nmf_PI_off = L.PI, ctx->nmf_PIB_off = screen ? L.PIB : 0, ctx->nmf_Rn_off = screen ? L.Rn : 0;
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
    if (screen) {
        NsPackB qb;
        qb.W2 = st->W[1], qb.b2 = st->b[1], qb.W3 = st->W[2], qb.b3 = st->b[2], qb.hw = st->hw;
        qb.W2B = (u16*)(base + L.W2B), qb.W3B = (u16*)(base + L.W3B);
        qb.b2E = (float*)(base + L.b2E), qb.b3E = (float*)(base + L.b3E), qb.hwE = (float*)(base + L.hwE);
        qb.H1 = st->units[0], qb.H2 = st->units[1], qb.H3 = st->units[2], qb.F = q.F;
        qb.H1Q = L.H1P <= 256 ? 256 : 512, qb.H2P = L.H2P, qb.H3P = L.H3P;
        int64_t nb = (int64_t)qb.H1Q * L.H2P;
        if ((int64_t)L.H2P * L.H3P > nb) nb = (int64_t)L.H2P * L.H3P;
        EL_LAUNCH("k_nmf_pack_h16", k_nmf_pack_h16, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, qb);
        {
            NsSpec sp;
            sp.W2 = st->W[1], sp.W3 = st->W[2], sp.hw = st->hw;
            sp.K2 = st->units[0], sp.N2 = st->units[1], sp.K3 = st->units[1], sp.N3 = st->units[2], sp.F = q.F;
            sp.gram = (float*)(base + L.gram), sp.cst = (float*)(base + L.cst);
            float* trs = sp.cst + 16;
            EL_LAUNCH("k_nmf_gram", k_nmf_gram, dim3(16, 4), dim3(256), 0, s, sp);
            for (int step = 0; step < NS_SPEC_SQ; ++step) EL_LAUNCH("k_nmf_gram_sq", k_nmf_gram_sq, dim3(16, 4), dim3(256), 0, s, sp, step, trs);
            EL_LAUNCH("k_nmf_spec_finish", k_nmf_spec_finish, dim3(5), dim3(256), 0, s, sp, (const float*)trs);
        }
        // the half-precision image of PI follows PI: rebuilt when ctl[3] says PI changed since it was last built (an unscreened call in
        // between may have rebuilt PI for new weights without touching it)
        EL_LAUNCH("k_nmf_pib", k_nmf_pib, dim3((unsigned)((I_local + 3) / 4)), dim3(256), 0, s, (const float*)(base + L.PI), I_local, L.H1P,
                  (int)(L.H1P <= 256 ? 256 : 512), (u16*)(base + L.PIB), (float*)(base + L.Rn),
                  pib_in_place ? (const unsigned long long*)(ctl + 3) : (const unsigned long long*)nullptr);
        EL_LAUNCH("k_nmf_pib_done", k_nmf_pib_done, dim3(1), dim3(1), 0, s, ctl);
        EL_CHECK_HIP(hipMemsetAsync(base + L.sflag, 0, 16, s));
        NsScreenParams sq;
        sq.W2B = qb.W2B, sq.W3B = qb.W3B, sq.PIB = (const u16*)(base + L.PIB), sq.Rn = (const float*)(base + L.Rn);
        sq.cst = (const float*)(base + L.cst), sq.b2E = qb.b2E, sq.b3E = qb.b3E, sq.hwE = qb.hwE;
        sq.up = (float*)(base + L.upb), sq.hw = st->hw;
        // slices per workgroup: ~3 workgroups per CU over (user groups) x (S / spb) ranges
        {
            const int64_t groups = (n_users + NSC_UB - 1) / NSC_UB;
            int spb = NS_WAVES;
            while (spb > 1 && groups * (L.S / spb) < (int64_t)ctx->cus * 3) spb >>= 1;
            sq.spb = spb;
        }
        const bool k16 = L.H1P <= 256;
        if (L.H2P == 256) rc = k16 ? ns_launch_screen<256, 128, 16>(p, sq, n_users, nsplit, s) : ns_launch_screen<256, 128, 32>(p, sq, n_users, nsplit, s);
        else if (L.H2P == 128) rc = k16 ? ns_launch_screen<128, 64, 16>(p, sq, n_users, nsplit, s) : ns_launch_screen<128, 64, 32>(p, sq, n_users, nsplit, s);
        else if (L.H2P == 64) rc = k16 ? ns_launch_screen<64, 32, 16>(p, sq, n_users, nsplit, s) : ns_launch_screen<64, 32, 32>(p, sq, n_users, nsplit, s);
        else rc = k16 ? ns_launch_screen<32, 32, 16>(p, sq, n_users, nsplit, s) : ns_launch_screen<32, 32, 32>(p, sq, n_users, nsplit, s);
        if (rc) return rc;
        // the user's threshold: k-th largest lower bound over the slices' lists
        float* thr = (float*)(base + L.thr);
        if (int rc2 = el_topk_merge(ctx, stream, p.part_idx, p.part_val, L.S, n_users, k, (int32_t*)(base + L.tidx), thr)) return rc2;
        int32_t* regi = (int32_t*)(base + L.regi);
        int32_t* regc = (int32_t*)(base + L.regc);
        int32_t* sflag = (int32_t*)(base + L.sflag);
        EL_LAUNCH("k_nmf_compact", k_nmf_compact, dim3((unsigned)((n_users * L.S + 3) / 4)), dim3(256), 0, s, p.t, (const float*)sq.up, (const float*)thr,
                  L.S, regi, regc, sflag);
        EL_LAUNCH("k_nmf_reg_check", k_nmf_reg_check, dim3((unsigned)((n_users + 255) / 256)), dim3(256), 0, s, (const int32_t*)regc, n_users, L.S, (int)k,
                  sflag);
        int32_t hflag[4] = {0, 0, 0, 0};
        EL_CHECK_HIP(hipMemcpyAsync(hflag, sflag, sizeof(hflag), hipMemcpyDeviceToHost, s));
        EL_CHECK_HIP(hipStreamSynchronize(s));              // (an evaluation call: its results are read by the host next anyway)
        unsigned long long ncands = 0;
        memcpy(&ncands, hflag + 2, 8);
        // worth it when the exact kernel is left with less than half of the pairs (the screen costs a sixth of it; EL_NMF_SCREEN_MAXFRAC overrides)
        const double maxfrac = ctx->opt.nmf_screen_maxfrac;   // (el_ctx_set_option: the tests move it between calls)
        const bool use = hflag[0] == 0 && (double)ncands <= maxfrac * (double)n_users * (double)I_local;
        if (use) ctx->nmf_screen_cands = (int64_t)ncands;
        ctx->nmf_screen_fallback = !use;
        if (use) {
            p.reg_idx = regi, p.reg_cnt = regc;
            p.t.excl_indptr = nullptr, p.t.excl_indices = nullptr;     // the regions are free of masked items already
        }
    }
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
