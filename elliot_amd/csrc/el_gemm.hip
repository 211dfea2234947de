// fp32 GEMM on v_mfma_f32_32x32x2_f32 with fused bias + activation epilogue.
//
// The dense layers of the reference's neural latent-factor models are Keras `Dense` calls
// (multi_vae_model.py:44-53,72-78; neural_matrix_factorization_model.py:59-64): y = act(x W + b) forward and
// the two transposed products backward.  One tiled kernel family serves all of them:
//   C[M,N] = act( op(A)[M,K] * op(B)[K,N] + bias[N] )
// op() selects the storage order of each operand (K-contiguous or X-contiguous), so x W, dY W^T and X^T dY
// run without materialising a transpose.
//
// k_gemm_f32_v<TA, TB, TM, TN> (the aligned fast path: every leading dimension and contiguous extent a multiple of 4
// floats, 16-byte aligned bases -- all Dense shapes of the models):
//   * 256 threads = 2 x 2 waves, wave tile (32 TM) x (32 TN), block tile (64 TM) x (64 TN), BK = 32; the host picks
//     TM, TN in {1, 2} per shape so that the tile count fills the 256 CUs evenly (M = 512 x N = 26 744 is 836 tiles of
//     128 x 128 = 3.27 per CU, i.e. a 4th round at 27 % occupancy; 1672 tiles of 128 x 64 waste 7 %).
//   * global -> registers -> LDS staging with one 16-byte load / ds_write_b128 per 4 floats and NO divergent code in
//     the k loop: a float4 is whole inside or whole outside the matrix (alignment contract), so the bounds test is a
//     select on the address and on the loaded value.  (The first version tested every float; hipcc turned the tile
//     fetch into ~9 000 instructions of branches per k tile.)
//   * LDS images and fragment reads, chosen so that reads are wide AND conflict-free:
//       K-contiguous operand  -> tile[X][36]: lane (col, hi) reads ONE ds_read_b128 = k 8t+4hi .. 8t+4hi+3 of its row and
//                                feeds four MFMA k-steps from it (row stride 36 = 4 * 9: the 16 lanes of every b128 lane
//                                group fall on 16 distinct bank quads);
//       X-contiguous operand  -> tile[32][BX+4]: lane (col, hi) reads one ds_read_b64 at k = 8t+4hi+j, x = 2 col, 2 col+1
//                                -- the two values go to the wave's two MFMA tiles, i.e. tile t holds the rows / columns
//                                x = 2 c + t (a permutation the epilogue undoes; adjacent columns of one lane are then
//                                stored as one float2).
//     Both operands use the same k <-> (step, hi) assignment, so every product a_mk b_kn is formed exactly once; inside a
//     block of 8 k values the accumulation order is (0,4,1,5,2,6,3,7) instead of ascending -- fp32 round-off level, the
//     parity tests compare against fp64 with a sqrt(K) bound.
//   * one __syncthreads per k tile (two LDS buffers), next tile's global loads in flight under the MFMAs.
// k_gemm_f32 (generic path, any alignment): 128 x 128 x 32 tiles, scalar-capable loads, ds_read_b32 fragments.
// Small-MN / huge-K products (dH = dLogits * W4^T: 512 x 600 x 26744) are split along K over gridDim.z; the
// partial tiles go to a workspace and k_gemm_reduce sums them in a fixed order (deterministic) and applies the
// epilogue.  Numerics: MFMA f32 is exact fp32 fma arithmetic (MI355X_MICROARCH.md).
#include "el_common.h"
#include <cmath>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum { EL_ACT_NONE = 0, EL_ACT_TANH = 1, EL_ACT_RELU = 2, EL_ACT_SIGMOID = 3 };

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int64_t M, N, K, lda, ldb, ldc;
    int act;
    int64_t kchunk;  // K range per blockIdx.z
    float* ws;       // split-K partials [gridDim.z][M][N] or NULL; stream-K: partial-tile slots [2 * P][BM * BN]
    int vecA, vecB;
    // stream-K fast path: the (tile, k tile) units of the whole problem are dealt to P persistent workgroups in contiguous ranges
    int64_t units, nkt, tiles_n;
    int P, whole_tiles;
    const float* zeros;   // >= 16 bytes of zeros in device memory
    // k_gemm_b3, XCD-aware workgroup order (1-D grid): tiles that read the same operand strip form a GROUP of `grp` consecutive
    // workgroups of one XCD (workgroup b runs on XCD b % 8: observed, used for speed only -- any placement is correct)
    // k_gemm_b3 epilogue of a Dense layer's backward pass (el_gemm_f32_x): C <- C where rmask > 0 else 0 (the ReLU derivative of the layer
    // below, taken from its OUTPUT); the tile's column sums go to colsum_part[row tile, n] (plain stores: el_colsum_finish adds the row
    // tiles in order -- that layer's bias gradient, the same bits on every run)
    const float* rmask;
    int64_t ldy;
    float* colsum_part;
    int gx, gy, gz;       // tiles along N, M, K-splits
    int grp_mode;         // 0: 3-D grid as launched; 1: group = the gy row tiles of one column strip; 2: group = the gx column tiles of one
    //                       row strip; 3: group = every tile of one K split
    int ngroups;
};

__device__ __forceinline__ float el_act(float v, int act) {
    if (act == EL_ACT_TANH) return tanhf(v);
    if (act == EL_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == EL_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int G_KC_LD = GBK + 1;     // K-contiguous image: [X][33]
constexpr int G_XC_LD = GBM + 4;     // X-contiguous image: [BK][132]
constexpr int G_TILE = 4224;         // floats per operand buffer = max(128*33, 32*132)

// Fetch one operand tile (X0..X0+127, k0..k0+31) into 4 float4 registers per thread.
//   XC = false: storage is [X][K] (K contiguous);  XC = true: storage is [K][X] (X contiguous)
template <bool XC>
__device__ __forceinline__ void g_fetch(const float* __restrict__ base, int64_t ld, int64_t X, int64_t x0,
                                        int64_t k0, int64_t kend, int vec, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f4 = q * 256 + tid;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!XC) {
            const int x = f4 >> 3, c4 = f4 & 7;
            const int64_t gx = x0 + x, gk = k0 + c4 * 4;
            if (gx < X && gk < kend) {
                const float* src = base + gx * ld + gk;
                if (vec && gk + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gk + 1 < kend) v.y = src[1];
                    if (gk + 2 < kend) v.z = src[2];
                    if (gk + 3 < kend) v.w = src[3];
                }
            }
        } else {
            const int kk = f4 >> 5, x4 = f4 & 31;
            const int64_t gk = k0 + kk, gx = x0 + x4 * 4;
            if (gk < kend && gx < X) {
                const float* src = base + gk * ld + gx;
                if (vec && gx + 3 < X) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gx + 1 < X) v.y = src[1];
                    if (gx + 2 < X) v.z = src[2];
                    if (gx + 3 < X) v.w = src[3];
                }
            }
        }
        r[q] = v;
    }
}

template <bool XC>
__device__ __forceinline__ void g_store(float* tile, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f4 = q * 256 + tid;
        if (!XC) {
            const int x = f4 >> 3, c4 = f4 & 7;
            float* d = tile + x * G_KC_LD + c4 * 4;
            d[0] = r[q].x;
            d[1] = r[q].y;
            d[2] = r[q].z;
            d[3] = r[q].w;
        } else {
            const int kk = f4 >> 5, x4 = f4 & 31;
            *reinterpret_cast<float4*>(tile + kk * G_XC_LD + x4 * 4) = r[q];
        }
    }
}

template <bool XC>
__device__ __forceinline__ float g_read(const float* tile, int x, int k) {
    return XC ? tile[k * G_XC_LD + x] : tile[x * G_KC_LD + k];
}

// TA: A stored [K][M] (M contiguous).  TB: B stored [N][K] (K contiguous).  The natural (non-transposed)
// storages are A [M][K] (K contiguous) and B [K][N] (N contiguous).
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);   // [2][G_TILE]
    float* Bs = As + 2 * G_TILE;                  // [2][G_TILE]
    constexpr bool AXC = TA;       // A image X-contiguous iff stored transposed
    constexpr bool BXC = !TB;      // B image X(N)-contiguous iff stored naturally
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GBM, n0 = (int64_t)blockIdx.x * GBN;
    const int64_t kbeg = (int64_t)blockIdx.z * p.kchunk;
    const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];
    const int nkt = (int)((kend - kbeg + GBK - 1) / GBK);
    if (nkt > 0) {
        g_fetch<AXC>(p.A, p.lda, p.M, m0, kbeg, kend, p.vecA, tid, ra);
        g_fetch<BXC>(p.B, p.ldb, p.N, n0, kbeg, kend, p.vecB, tid, rb);
    }
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        g_store<AXC>(As + buf * G_TILE, tid, ra);
        g_store<BXC>(Bs + buf * G_TILE, tid, rb);
        __syncthreads();
        if (kt + 1 < nkt) {
            const int64_t k0 = kbeg + (int64_t)(kt + 1) * GBK;
            g_fetch<AXC>(p.A, p.lda, p.M, m0, k0, kend, p.vecA, tid, ra);
            g_fetch<BXC>(p.B, p.ldb, p.N, n0, k0, kend, p.vecB, tid, rb);
        }
        const float* At = As + buf * G_TILE;
        const float* Bt = Bs + buf * G_TILE;
#pragma unroll
        for (int s = 0; s < GBK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = g_read<AXC>(At, wr * 64 + t * 32 + col, 2 * s + hi);
                b[t] = g_read<BXC>(Bt, wc * 64 + t * 32 + col, 2 * s + hi);
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        buf ^= 1;
    }

    // epilogue: lane holds column n = .. + col, rows (r&3) + 8*(r>>2) + 4*hi
    float* out = p.ws ? p.ws + (int64_t)blockIdx.z * p.M * p.N : p.C;
    const int64_t ldo = p.ws ? p.N : p.ldc;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int64_t n = n0 + wc * 64 + tb * 32 + col;
        if (n >= p.N) continue;
        const float bv = (!p.ws && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wr * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.M) {
                    float v = acc[ta][tb][r];
                    if (!p.ws) v = el_act(v + bv, p.act);
                    out[m * ldo + n] = v;
                }
            }
        }
    }
}

// ---- aligned fast path --------------------------------------------------------------------------------------------
// LDS images of one k tile (KB = 32 k values), both UNPADDED so that LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base +
// lane * 16 bytes, no VGPRs, no ds_write) can fill them:
//   K-contiguous operand  tile[BX][32]: the 16-byte chunk c4 of row x sits at chunk position c4 ^ ((x >> 1) & 7) -- the
//                         swizzle is applied to the SOURCE address of each lane (lane -> row = lane/8, position = lane%8), the
//                         LDS side stays lane-linear.  Lane (col, hi) reads ONE ds_read_b128 = k 8t+4hi .. +3 of its row: the
//                         16 lanes of every b128 lane group hold 16 distinct (x&1, (x>>1)&7) pairs = 16 distinct bank quads.
//   X-contiguous operand  tile[32][BX]: lane (col, hi) reads one ds_read_b64 at k = 8t+4hi+j, x = 2 col, 2 col+1 (32 lanes =
//                         64 consecutive banks); the two values feed the wave's two MFMA tiles, i.e. tile t holds the rows /
//                         columns x = 2 c + t (a permutation the epilogue undoes; adjacent columns are stored as float2).
// Both operands use the same k <-> (MFMA step, hi) assignment; inside a block of 8 k values the accumulation order is
// (0,4,1,5,2,6,3,7) instead of ascending -- fp32 round-off level.
// Lanes whose float4 lies outside the matrix (M / N edge, K tail) read 16 bytes of zeros instead (el_ctx.zeros): the bounds
// test is a select on the source address, the k loop has no divergent code.
constexpr int KB = GBK;

template <bool XC, int BX>
struct VTile {
    static constexpr int FLOATS = BX * KB;
    static constexpr int NDMA = BX * KB / 256 / 4;      // LDS-DMA instructions per wave and k tile (1 KiB each, 4 waves)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// one operand tile (x0 .. x0+BX-1, k0 .. k0+31) -> LDS, NDMA instructions per wave
template <bool XC, int BX>
__device__ __forceinline__ void v_dma(const float* __restrict__ base, int64_t ld, int X, int x0, int K, int k0,
                                      const float* __restrict__ zeros, float* tile, int wave, int lane, bool live = true) {
#pragma unroll
    for (int q = 0; q < VTile<XC, BX>::NDMA; ++q) {
        const int i = q * 4 + wave;                      // 1 KiB piece of the image
        int gx, gk;
        if (!XC) {
            const int x = 8 * i + (lane >> 3);
            gx = x0 + x;
            gk = k0 + 4 * ((lane & 7) ^ ((x >> 1) & 7));
        } else {
            const int flat = 64 * i + lane;
            gk = k0 + flat / (BX / 4);
            gx = x0 + 4 * (flat % (BX / 4));
        }
        const bool ok = live && gx < X && gk < K;
        const float* src = XC ? base + (int64_t)gk * ld + gx : base + (int64_t)gx * ld + gk;
        src = ok ? src : zeros;
        float* dst = tile + 256 * i;                     // wave-uniform; the hardware adds lane * 16 bytes
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
}

// fragments of one operand for the 8 k values of block t8: f[j][t] = value of MFMA k-step j (0..3) for the wave's tile t.
//   xw = first x of the wave inside the block tile; lane = (col, hi)
template <bool XC, int BX, int T>
__device__ __forceinline__ void v_frags(const float* tile, int xw, int col, int hi, int t8, float (&f)[4][T]) {
    if (!XC) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int x = xw + 32 * t + col;
            const float4 v = *reinterpret_cast<const float4*>(tile + x * KB + (((2 * t8 + hi) ^ ((x >> 1) & 7)) << 2));
            f[0][t] = v.x;
            f[1][t] = v.y;
            f[2][t] = v.z;
            f[3][t] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* src = tile + (8 * t8 + 4 * hi + j) * BX + xw + T * col;
            if (T == 2) {
                const float2 v = *reinterpret_cast<const float2*>(src);
                f[j][0] = v.x;
                f[j][T - 1] = v.y;
            } else {
                f[j][0] = src[0];
            }
        }
    }
}

// TA: A stored [K][M] (M contiguous).  TB: B stored [N][K] (K contiguous).  Block tile (64 TM) x (64 TN), 2 x 2 waves.
//
// Persistent schedule.  The work is the list of units (tile t, k tile kt), t-major; workgroup w of P (P = two per CU, all
// resident at once) owns the contiguous range [w U / P, (w+1) U / P) -- rounded to whole tiles when there are many tiles
// (p.whole_tiles), to single k tiles ("stream-K") when there are few -- and runs ONE pipeline through it: the LDS-DMA of
// unit u+1 is issued before the MFMAs of unit u whichever tile u+1 belongs to, so a tile boundary costs an epilogue, not a
// pipeline restart.  A tile whose k range lies inside one workgroup's range is finished there (bias + activation, stored
// to C); a tile cut by a range boundary leaves one partial accumulator tile per contributing workgroup in `ws` (slot 2w: the
// segment that starts w's range, 2w+1: the one that ends it) and k_gemm_sk_fix adds them in workgroup order --
// deterministic, no spinning on other workgroups.
template <bool TA, bool TB, int TM, int TN>
__global__ __launch_bounds__(256, 2) void k_gemm_f32_v(GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr bool AXC = TA, BXC = !TB;
    typedef VTile<AXC, BM> TlA;
    typedef VTile<BXC, BN> TlB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);        // [2][TlA::FLOATS]
    float* Bs = As + 2 * TlA::FLOATS;                  // [2][TlB::FLOATS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, col = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int w = blockIdx.x, nkt = (int)p.nkt, tiles_n = (int)p.tiles_n;
    int u0, u1;
    if (p.whole_tiles) {
        const int T = (int)(p.units / p.nkt);
        u0 = (int)(((int64_t)w * T) / p.P) * nkt;
        u1 = (int)(((int64_t)(w + 1) * T) / p.P) * nkt;
    } else {
        u0 = (int)(((int64_t)w * p.units) / p.P);
        u1 = (int)(((int64_t)(w + 1) * p.units) / p.P);
    }
    if (u0 >= u1) return;
    const int M = (int)p.M, N = (int)p.N, K = (int)p.K;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int tile = u0 / nkt;
    int kt = u0 - tile * nkt;
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    v_dma<AXC, BM>(p.A, p.lda, M, m0, K, kt * KB, p.zeros, As, wave, lane);
    v_dma<BXC, BN>(p.B, p.ldb, N, n0, K, kt * KB, p.zeros, Bs, wave, lane);
    int buf = 0;
    for (int u = u0; u < u1; ++u) {
        __syncthreads();                                // unit u has landed (vmcnt(0) + barrier); buffer buf^1 is free again
        // next unit (possibly the first k tile of the next output tile)
        int ntile = tile, nk = kt + 1;
        if (nk == nkt) {
            nk = 0;
            ntile = tile + 1;
        }
        const int nm0 = (ntile / tiles_n) * BM, nn0 = (ntile % tiles_n) * BN;
        const bool more = u + 1 < u1;                   // (the last unit "prefetches" zeros: no branch in the loop body)
        const float* At = As + buf * TlA::FLOATS;
        const float* Bt = Bs + buf * TlB::FLOATS;
#pragma unroll
        for (int t8 = 0; t8 < KB / 8; ++t8) {
            float fa[4][TM], fb[4][TN];
            v_frags<AXC, BM, TM>(At, wr * 32 * TM, col, hi, t8, fa);
            v_frags<BXC, BN, TN>(Bt, wc * 32 * TN, col, hi, t8, fb);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TN; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][ta], fb[j][tb], acc[ta][tb], 0, 0, 0);
            if (t8 == 0) {                              // the next unit's LDS-DMA goes out behind the first MFMAs of this one
                v_dma<AXC, BM>(p.A, p.lda, M, nm0, K, nk * KB, p.zeros, As + (buf ^ 1) * TlA::FLOATS, wave, lane, more);
                v_dma<BXC, BN>(p.B, p.ldb, N, nn0, K, nk * KB, p.zeros, Bs + (buf ^ 1) * TlB::FLOATS, wave, lane, more);
            }
        }
        buf ^= 1;
        if (kt == nkt - 1 || u == u1 - 1) {
            // the segment of `tile` inside this range ends here.  MFMA D: lane holds column c = col of its tile, rows
            // rr = (r&3) + 8*(r>>2) + 4*hi.  Tile t of an X-contiguous operand holds x = T c + t (v_frags), of a K-contiguous
            // one x = 32 t + c.
            const int seg0 = tile * nkt > u0 ? tile * nkt : u0;
            const bool whole = seg0 == tile * nkt && kt == nkt - 1;
            if (!whole) {
                // partial accumulators in REGISTER order (k_gemm_sk_fix knows the layout): 16 fully coalesced 16-byte stores
                float* slot = p.ws + (int64_t)(2 * w + (seg0 == u0 ? 0 : 1)) * (BM * BN);
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TN; ++tb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int q = (ta * TN + tb) * 4 + g;
                            *reinterpret_cast<float4*>(slot + ((int64_t)q * 256 + tid) * 4) =
                                make_float4(acc[ta][tb][4 * g], acc[ta][tb][4 * g + 1], acc[ta][tb][4 * g + 2], acc[ta][tb][4 * g + 3]);
                        }
            } else {
                float* out = p.C + (int64_t)m0 * p.ldc + n0;
                const int64_t ldo = p.ldc;
                const int mlim = M - m0, nlim = N - n0;
                if (BXC && TN == 2 && (ldo % 4 == 0) && (((uintptr_t)out & 15) == 0)) {
                    // a lane holds columns 2c, 2c+1 of rows 8g+4hi+{0..3}: lanes c, c^1 swap half of their rows (DPP quad_perm) so
                    // that each ends up with FOUR adjacent columns of two rows -> 16-byte stores, half the store instructions
                    // (the store tail of a 64-value-per-lane epilogue is issue-bound: cdna_hip_programming.md T21)
                    const bool odd = (col & 1) != 0;
                    const int n = wc * 64 + 2 * (col & ~1);
                    float b0 = 0.f, b1 = 0.f;
                    if (p.bias) {
                        const int nn = wc * 64 + 2 * col;
                        b0 = nn < nlim ? p.bias[n0 + nn] : 0.f;
                        b1 = nn + 1 < nlim ? p.bias[n0 + nn + 1] : 0.f;
                    }
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float o0[4], o1[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                o0[q] = el_act(acc[ta][0][4 * g + q] + b0, p.act);
                                o1[q] = el_act(acc[ta][TN - 1][4 * g + q] + b1, p.act);
                            }
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float k0v = odd ? o0[2 + j] : o0[j], k1v = odd ? o1[2 + j] : o1[j];
                                const float s0v = odd ? o0[j] : o0[2 + j], s1v = odd ? o1[j] : o1[2 + j];
                                const float r0v = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0v), 0xB1, 0xF, 0xF, true));
                                const float r1v = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1v), 0xB1, 0xF, 0xF, true));
                                const int rr = 8 * g + 4 * hi + (odd ? 2 : 0) + j;
                                const int m = wr * 32 * TM + (AXC ? TM * rr + ta : 32 * ta + rr);
                                const float4 v = odd ? make_float4(r0v, r1v, k0v, k1v) : make_float4(k0v, k1v, r0v, r1v);
                                if (m < mlim) {
                                    float* dst = out + m * ldo + n;
                                    if (n + 3 < nlim) {
                                        *reinterpret_cast<float4*>(dst) = v;
                                    } else {
                                        if (n < nlim) dst[0] = v.x;
                                        if (n + 1 < nlim) dst[1] = v.y;
                                        if (n + 2 < nlim) dst[2] = v.z;
                                    }
                                }
                            }
                        }
                } else {
                    const bool pair = BXC && TN == 2 && (ldo % 2 == 0) && (((uintptr_t)out & 7) == 0);
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
                            const int m = wr * 32 * TM + (AXC ? TM * rr + ta : 32 * ta + rr);
                            if (m < mlim) {
                                float* orow = out + m * ldo;
                                if (BXC && TN == 2) {
                                    const int n = wc * 64 + 2 * col;
                                    const float v0 = el_act(acc[ta][0][r] + ((p.bias && n < nlim) ? p.bias[n0 + n] : 0.f), p.act);
                                    const float v1 = el_act(acc[ta][TN - 1][r] + ((p.bias && n + 1 < nlim) ? p.bias[n0 + n + 1] : 0.f), p.act);
                                    if (pair && n + 1 < nlim) {
                                        *reinterpret_cast<float2*>(orow + n) = make_float2(v0, v1);
                                    } else {
                                        if (n < nlim) orow[n] = v0;
                                        if (n + 1 < nlim) orow[n + 1] = v1;
                                    }
                                } else {
#pragma unroll
                                    for (int tb = 0; tb < TN; ++tb) {
                                        const int n = wc * 32 * TN + (BXC ? TN * col + tb : 32 * tb + col);
                                        if (n < nlim) orow[n] = el_act(acc[ta][tb][r] + (p.bias ? p.bias[n0 + n] : 0.f), p.act);
                                    }
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;
        }
        tile = ntile;
        kt = nk;
        m0 = nm0;
        n0 = nn0;
    }
}

// Tiles cut by a range boundary: sum the partial accumulator tiles of the contributing workgroups (ascending), epilogue, store.
// Partials are in register order: float4 number q * 256 + tid of a slot = accumulator registers 4g .. 4g+3 of MFMA tile
// (ta, tb) (q = (ta TN + tb) 4 + g) of thread tid.  grid = (tiles, TM 4); tiles finished inside one range return at once.
template <bool AXC, bool BXC, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_sk_fix(GemmParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const int64_t tile = blockIdx.x;
    const int64_t ua = tile * p.nkt, ub = ua + p.nkt - 1;
    const int64_t wa = ((ua + 1) * p.P - 1) / p.units, wb = ((ub + 1) * p.P - 1) / p.units;   // owner of a unit x: floor(((x+1) P - 1) / U)
    if (wa == wb) return;
    const int64_t m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int64_t sa = (wa * p.units) / p.P;                      // first unit of wa's range
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, hi = lane >> 5, col = lane & 31, wr = wave >> 1, wc = wave & 1;
    const int g = blockIdx.y & 3, ta = blockIdx.y >> 2;           // grid.y = TM * 4: one (ta, g) register group, every tb
    const float* first = p.ws + (2 * wa + (sa == ua ? 0 : 1)) * (int64_t)(BM * BN);
    float4 acc[TN];
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int64_t e = ((int64_t)((ta * TN + tb) * 4 + g) * 256 + tid) * 4;
        float4 a = *reinterpret_cast<const float4*>(first + e);
        // the partial tiles are added in workgroup order (deterministic), but FETCHED eight at a time: a product with a huge K and
        // few tiles (dW = X^T dY, K = the batch) has ~60 contributors per tile, and one dependent 16-byte load per addition made
        // this kernel a chain of ~60 memory latencies (66 us per launch at the NeuMF shapes)
        int64_t w0 = wa + 1;
        if (wb - wa <= 2) {                                     // a tile cut once or twice (many-tile products): nothing to batch
            for (; w0 <= wb; ++w0) {
                const float4 b = *reinterpret_cast<const float4*>(p.ws + (2 * w0) * (int64_t)(BM * BN) + e);
                a.x += b.x;
                a.y += b.y;
                a.z += b.z;
                a.w += b.w;
            }
        }
        for (; w0 <= wb; w0 += 8) {
            float4 b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t w = (w0 + j <= wb) ? w0 + j : wb;
                b[j] = *reinterpret_cast<const float4*>(p.ws + (2 * w) * (int64_t)(BM * BN) + e);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (w0 + j <= wb) {
                    a.x += b[j].x;
                    a.y += b[j].y;
                    a.z += b[j].z;
                    a.w += b[j].w;
                }
            }
        }
        acc[tb] = a;
    }
    const bool pair = BXC && TN == 2 && (p.ldc % 2 == 0) && (((uintptr_t)p.C & 7) == 0);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int rr = 8 * g + 4 * hi + x;
        const int64_t m = m0 + wr * 32 * TM + (AXC ? TM * rr + ta : 32 * ta + rr);
        if (m >= p.M) continue;
        float* orow = p.C + m * p.ldc;
        float v[TN];
        int64_t n[TN];
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            n[tb] = n0 + wc * 32 * TN + (BXC ? TN * col + tb : 32 * tb + col);
            const float raw = x == 0 ? acc[tb].x : (x == 1 ? acc[tb].y : (x == 2 ? acc[tb].z : acc[tb].w));
            v[tb] = el_act(raw + ((p.bias && n[tb] < p.N) ? p.bias[n[tb]] : 0.f), p.act);
        }
        if (pair && n[TN - 1] < p.N) {
            *reinterpret_cast<float2*>(orow + n[0]) = make_float2(v[0], v[TN - 1]);
        } else {
#pragma unroll
            for (int tb = 0; tb < TN; ++tb)
                if (n[tb] < p.N) orow[n[tb]] = v[tb];
        }
    }
}

__global__ __launch_bounds__(256) void k_gemm_reduce(const float* __restrict__ ws, int splits, int64_t M, int64_t N,
                                                     float* C, int64_t ldc, const float* __restrict__ bias, int act) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    const int64_t m = t / N, n = t - m * N;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(int64_t)z * M * N + t];
    if (bias) s += bias[n];
    C[m * ldc + n] = el_act(s, act);
}

// ---- fp32 GEMM on the bf16 matrix instruction: three-way split operands ---------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of v_mfma_f32_32x32x16_bf16.  An fp32 number is the exact sum of three bf16 numbers
// (a = a0 + a1 + a2 + eps, |eps| <= 2^-24 |a|: a0 = the upper 16 bits of a, a1 = the upper 16 bits of a - a0, a2 = those of
// a - a0 - a1 -- both differences are exact in fp32), and a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + O(2^-24 |a b|):
// six bf16 products per fp32 product, each exact in the instruction's fp32 accumulator arithmetic, 16 / 6 = 2.7x the fp32
// instruction's rate at the rounding level of fp32 itself (the terms left out, a1 b2 + a2 b1 + a2 b2, are below 3 * 2^-24 |a b|;
// a different summation order of the same fp32 products moves a dot product by more).  The parity tests of the Dense layers
// (tests/test_gpu_dense.py, test_gpu_neumf.py, test_gpu_gemm.py) hold at their fp32 tolerances; the option gemm_split = 0 selects the
// fp32 instruction (k_gemm_f32_v above).
//   * block tile 128 x 128, BK = 32 (two k-steps), 256 threads, two workgroups per CU
//   * staging: global -> registers (8 float4 per thread, the next tile in flight under the matrix instructions) -> split -> LDS as
//     FRAGMENT images, one per plane (below)
//   * per k-step and consumer wave 15 fragment loads (ds_read_b128) and 24 matrix instructions (lowest-order products first)
//   * gridDim.z splits K (few tiles, long K: the dW products over the batch); partials go to the workspace, k_gemm_reduce sums them
constexpr int B3_BM = 128, B3_BN = 128, B3_BK = 32;
typedef __bf16 b3_h8 __attribute__((ext_vector_type(8)));

// 8 consecutive k values of one row -> the three planes' 16-byte fragment pieces
__device__ __forceinline__ void b3_split8(const float (&v)[8], uint4& p0, uint4& p1, uint4& p2) {
    unsigned int h0[8], h1[8], h2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned int u = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u & 0xffff0000u);
        const unsigned int u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
        h0[e] = u, h1[e] = u1, h2[e] = __float_as_uint(r2);
    }
    // upper halves of two words into one: bytes (hi.3, hi.2, lo.3, lo.2)
    p0 = make_uint4(__builtin_amdgcn_perm(h0[1], h0[0], 0x07060302u), __builtin_amdgcn_perm(h0[3], h0[2], 0x07060302u),
                    __builtin_amdgcn_perm(h0[5], h0[4], 0x07060302u), __builtin_amdgcn_perm(h0[7], h0[6], 0x07060302u));
    p1 = make_uint4(__builtin_amdgcn_perm(h1[1], h1[0], 0x07060302u), __builtin_amdgcn_perm(h1[3], h1[2], 0x07060302u),
                    __builtin_amdgcn_perm(h1[5], h1[4], 0x07060302u), __builtin_amdgcn_perm(h1[7], h1[6], 0x07060302u));
    p2 = make_uint4(__builtin_amdgcn_perm(h2[1], h2[0], 0x07060302u), __builtin_amdgcn_perm(h2[3], h2[2], 0x07060302u),
                    __builtin_amdgcn_perm(h2[5], h2[4], 0x07060302u), __builtin_amdgcn_perm(h2[7], h2[6], 0x07060302u));
}

// One operand tile (128 rows x 32 k) from global memory into 8 float4 registers of each of 128 threads (t = 0..127), then split and
// stored as the planes' fragment images img[plane][q = k / 8][pos] (16 bytes each: 8 consecutive k of one row), where image position
// pos <-> row 4 (pos & 31) + (pos >> 5): the 32-row tile T = pos / 32 of the matrix instruction holds the rows = T (mod 4) -- a
// permutation the epilogue undoes, chosen so that BOTH storage orders stage with conflict-free 16-byte LDS stores (the 8 lanes of a
// store group write 8 consecutive positions) and the fragment reads are one contiguous kilobyte per wave-instruction:
//   KC (operand stored with k contiguous, row stride ld): thread t = position t, the 32 k values of its row (8 float4)
//   XC (row index contiguous, k stride ld): thread t = rows 4 r4 .. 4 r4 + 3 (r4 = t & 31: one float4) x the 8 k values of q = t / 32
// Out of range (row >= nX, k >= k_end): zeros, fetched from a block of zeros -- a select on the address, no branch (a float4 is
// whole inside or whole outside: alignment contract of the fast path).
template <bool KC>
__device__ __forceinline__ void b3_load(const float* __restrict__ X, int64_t ld, int64_t nX, int64_t x0, int64_t k_end, int64_t k0, int t,
                                        const float* __restrict__ zeros, float4 (&rg)[8]) {
    if (KC) {
        const int64_t row = x0 + 4 * (t & 31) + (t >> 5);
        const float* src = X + row * ld + k0;
#pragma unroll
        for (int j = 0; j < 8; ++j) rg[j] = *reinterpret_cast<const float4*>((row < nX && k0 + 4 * j < k_end) ? src + 4 * j : zeros);
    } else {
        const int64_t row = x0 + 4 * (t & 31);
        const int q = t >> 5;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int64_t k = k0 + 8 * q + kk;
            rg[kk] = *reinterpret_cast<const float4*>((row < nX && k < k_end) ? X + k * ld + row : zeros);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void b3_store(const float4 (&rg)[8], char* img, int t) {
    constexpr int PLANE = 4 * 128 * 16;                 // bytes of one plane: four 8-k groups x 128 positions x 16
    if (KC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[8] = {rg[2 * q].x, rg[2 * q].y, rg[2 * q].z, rg[2 * q].w, rg[2 * q + 1].x, rg[2 * q + 1].y, rg[2 * q + 1].z, rg[2 * q + 1].w};
            uint4 p0, p1, p2;
            b3_split8(v, p0, p1, p2);
            char* dst = img + (q * 128 + t) * 16;
            *reinterpret_cast<uint4*>(dst) = p0;
            *reinterpret_cast<uint4*>(dst + PLANE) = p1;
            *reinterpret_cast<uint4*>(dst + 2 * PLANE) = p2;
        }
    } else {
        const int r4 = t & 31, q = t >> 5;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float v[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) v[kk] = mi == 0 ? rg[kk].x : (mi == 1 ? rg[kk].y : (mi == 2 ? rg[kk].z : rg[kk].w));
            uint4 p0, p1, p2;
            b3_split8(v, p0, p1, p2);
            char* dst = img + (q * 128 + mi * 32 + r4) * 16;
            *reinterpret_cast<uint4*>(dst) = p0;
            *reinterpret_cast<uint4*>(dst + PLANE) = p1;
            *reinterpret_cast<uint4*>(dst + 2 * PLANE) = p2;
        }
    }
}

// workgroup barrier for the LDS hand-over only: __syncthreads() carries a fence that also drains the global loads in flight --
// every k tile would wait out a trip to memory
__device__ __forceinline__ void b3_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 256 threads = 4 waves, two workgroups per CU (one stages while the other runs its matrix instructions): threads 0..127 stage the
// A tile, 128..255 the B tile; wave w computes the row tile w (rows = w mod 4 of the block's 128) against all four column tiles, so
// that a lane ends up with four ADJACENT columns (4 n + 0..3) of each of its rows: the epilogue stores float4, rows of 512
// contiguous bytes per wave-instruction.
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void k_gemm_b3(GemmParams p) {
    constexpr int PLANE = 4 * 128 * 16, IMG = 3 * PLANE;
    __shared__ __attribute__((aligned(16))) char lds[2 * IMG];   // A image, B image (48 KB)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, g = lane >> 5;
    // tile of this workgroup.  Launched 3-D (grp_mode 0) the linear workgroup id walks N first, so the four or five row tiles that
    // share a 26 744-wide column strip of the weights land on different XCDs at different times and every L2 fetches the strip again
    // (round-4 PMC: 461 MB per launch against 120 MB of operands).  Launched 1-D, the tiles that share a strip are dealt to
    // consecutive workgroups of ONE XCD: group gi -> XCD gi % 8, its tiles q % grp run side by side on that XCD's CUs.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.grp_mode != 0) {
        const unsigned b = blockIdx.x, xcd = b & 7u, q = b >> 3;
        const unsigned grp = p.grp_mode == 1 ? (unsigned)p.gy : (p.grp_mode == 2 ? (unsigned)p.gx : (unsigned)(p.gx * p.gy));
        const unsigned gi = (q / grp) * 8u + xcd, ti = q % grp;
        if (gi >= (unsigned)p.ngroups) return;
        if (p.grp_mode == 1) by = ti, bx = gi % (unsigned)p.gx, bz = gi / (unsigned)p.gx;
        else if (p.grp_mode == 2) bx = ti, by = gi % (unsigned)p.gy, bz = gi / (unsigned)p.gy;
        else bx = ti % (unsigned)p.gx, by = ti / (unsigned)p.gx, bz = gi;
    }
    const int64_t m0 = (int64_t)by * B3_BM, n0 = (int64_t)bx * B3_BN;
    const int64_t kbeg = (int64_t)bz * p.kchunk;
    const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
    const int t = tid & 127;
    const bool stA = tid < 128;
    float4 rg[8];
    auto fetch = [&](int64_t k0) {
        if (stA) b3_load<AKC>(p.A, p.lda, p.M, m0, kend, k0, t, p.zeros, rg);
        else b3_load<BKC>(p.B, p.ldb, p.N, n0, kend, k0, t, p.zeros, rg);
    };
    auto stash = [&]() {
        if (stA) b3_store<AKC>(rg, lds, t);
        else b3_store<BKC>(rg, lds + IMG, t);
    };
    floatx16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    fetch(kbeg);
    // fragment offsets: k-step s, lane (n, g) -> group q = 2 s + g, position 32 T + n
    const int fa = (g * 128 + 32 * w + n) * 16, fb = IMG + (g * 128 + n) * 16;
    for (int64_t k0 = kbeg; k0 < kend; k0 += B3_BK) {
        b3_lds_barrier();                                 // the previous tile's fragments are read
        stash();
        b3_lds_barrier();
        fetch(k0 + B3_BK);                                // (past kend: zeros, never stored)
        // both k-steps' fragments first (the second step's loads complete under the first step's matrix instructions); per product
        // term the four column tiles in turn: consecutive matrix instructions never wait for each other's accumulator
        b3_h8 a[2][3], b[2][4][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[s][pl] = *reinterpret_cast<const b3_h8*>(lds + fa + pl * PLANE + s * 2 * 128 * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[s][j][pl] = *reinterpret_cast<const b3_h8*>(lds + fb + pl * PLANE + (s * 2 * 128 + 32 * j) * 16);
            }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], b[s][j][0], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][j][1], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][j][2], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[s][j][0], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][j][1], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[s][j][0], acc[j], 0, 0, 0);
        }
    }
    // epilogue: register r of column tile j = row 4 (8 (r / 4) + 4 g + r % 4) + w of the block, column 4 n + j
    float* out = p.ws ? p.ws + (int64_t)bz * p.M * p.N : p.C;
    const int64_t ldo = p.ws ? p.N : p.ldc;
    const int64_t col = n0 + 4 * n;
    const int act = p.ws ? EL_ACT_NONE : p.act;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p.ws && p.bias && col < p.N) bv = *reinterpret_cast<const float4*>(p.bias + col);     // (N % 4 == 0 on this path; bias 16-byte aligned: checked by the host)
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + 4 * (8 * (r >> 2) + 4 * g + (r & 3)) + w;
            float4 v = make_float4(acc[0][r] + bv.x, acc[1][r] + bv.y, acc[2][r] + bv.z, acc[3][r] + bv.w);
            if (act == EL_ACT_RELU) v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
            else if (act != EL_ACT_NONE) v = make_float4(el_act(v.x, act), el_act(v.y, act), el_act(v.z, act), el_act(v.w, act));
            if (row < p.M) {
                if (p.rmask) {                                          // (workgroup-uniform; the host never sets it with split K)
                    const float4 y = *reinterpret_cast<const float4*>(p.rmask + row * p.ldy + col);
                    v = make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
                    cs.x += v.x, cs.y += v.y, cs.z += v.z, cs.w += v.w;
                }
                *reinterpret_cast<float4*>(out + row * ldo + col) = v;
            }
        }
    }
    if (p.rmask) {
        // column sums of the tile: the two lane halves hold different rows of the same four columns, the four waves different rows
        // again -- one store per column and tile into the tile's row of the partial sums
        cs.x += __shfl_xor(cs.x, 32, 64), cs.y += __shfl_xor(cs.y, 32, 64), cs.z += __shfl_xor(cs.z, 32, 64), cs.w += __shfl_xor(cs.w, 32, 64);
        __syncthreads();                                                // every wave is done with the fragment images
        float* red = reinterpret_cast<float*>(lds);
        if (g == 0) *reinterpret_cast<float4*>(red + w * 128 + 4 * n) = cs;
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            const float t4 = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            p.colsum_part[(m0 / B3_BM) * p.N + n0 + tid] = t4;
        }
    }
}

static int gemm_splits(el_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    const int wg_per_cu = 2;
    const int64_t tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
    const int64_t target = (int64_t)ctx->cus * wg_per_cu;
    if (tiles >= target / 2 || K < 4 * GBK) return 1;
    // as many splits as keep tiles x splits within the resident workgroups (rounding UP left 520 workgroups for 512 places at
    // 512 x 600 x 26 744: eight of them ran a second round alone)
    int64_t s = target / tiles;
    const int64_t maxs = K / (2 * GBK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    if (s >= 8) s &= ~(int64_t)7;                        // equal shares for the 8 XCDs (the XCD-aware order groups the tiles of a split)
    return s < 1 ? 1 : (int)s;
}

// Tile shape of the stream-K fast path: 128 x 128 unless the padding of a narrow M or N to multiples of 128 wastes more than
// the smaller tile's lower efficiency costs; P = persistent workgroups (two per CU; fewer when there is less work than that).
struct GemmPlan {
    int tm, tn, P, whole_tiles;
    int64_t nkt, tiles_n, units;
};

static GemmPlan gemm_plan(el_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    GemmPlan pl;
    auto waste = [](int64_t X, int64_t b) { return (double)((X + b - 1) / b * b) / (double)X; };
    pl.tm = (waste(M, 128) > 1.12 * waste(M, 64)) ? 1 : 2;
    pl.tn = (waste(N, 128) > 1.12 * waste(N, 64)) ? 1 : 2;
    int mode = -1;
    // small products (the 512 x 400 x 600 class of the Mult-VAE heads: ~0.25 GFLOP, latency-bound): whole 64 x 64 tiles, one
    // launch, no partial tiles to combine (measured 13-24 us against 21-29 us for stream-K + fix-up)
    if (2.0 * (double)M * (double)N * (double)K < 2.0e9 && ((M + 63) / 64) * ((N + 63) / 64) >= 16) pl.tm = pl.tn = 1, mode = 1;
    const int64_t bm = 64 * pl.tm, bn = 64 * pl.tn;
    pl.nkt = (K + KB - 1) / KB;
    if (pl.nkt < 1) pl.nkt = 1;
    pl.tiles_n = (N + bn - 1) / bn;
    const int64_t tiles = ((M + bm - 1) / bm) * pl.tiles_n;
    pl.units = tiles * pl.nkt;
    const int per_cu = 2, min_units = 4;        // persistent workgroups per CU; at least this many k tiles per workgroup
    int64_t P = (int64_t)ctx->cus * per_cu;
    // whole tiles per workgroup (no partial tiles to combine) when the rounding costs little: >= 8 tiles per workgroup
    pl.whole_tiles = mode >= 0 ? mode : (tiles >= 8 * P ? 1 : 0);
    if (pl.whole_tiles) {
        if (P > tiles) P = tiles;
    } else {
        const int64_t cap = (pl.units + min_units - 1) / min_units;        // at least min_units k tiles per workgroup
        if (P > cap) P = cap;
    }
    if (P < 1) P = 1;
    pl.P = (int)P;
    return pl;
}

extern "C" size_t el_gemm_ws_bytes(el_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    if (!ctx) return 0;
    const int s1 = gemm_splits(ctx, M, N, K);
    const size_t generic = s1 > 1 ? (size_t)s1 * (size_t)M * (size_t)N * 4 : 0;
    const size_t sk = (size_t)2 * (size_t)ctx->cus * 2 * 128 * 128 * 4;            // two partial-tile slots per persistent workgroup
    return generic > sk ? generic : sk;
}

template <bool TA, bool TB, int TM, int TN>
static int gemm_launch_v(el_ctx* ctx, const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const size_t lds = (size_t)2 * (VTile<TA, BM>::FLOATS + VTile<!TB, BN>::FLOATS) * 4;
    auto kern = k_gemm_f32_v<TA, TB, TM, TN>;
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EL_LAUNCH("k_gemm_f32", kern, dim3((unsigned)p.P), dim3(256), lds, s, p);
    const int64_t tiles = p.units / p.nkt;
    if (!p.whole_tiles && (p.units % p.P != 0 || (p.units / p.P) % p.nkt != 0))       // some tile is cut by a range boundary
        EL_LAUNCH("k_gemm_reduce", (k_gemm_sk_fix<TA, !TB, TM, TN>), dim3((unsigned)tiles, (unsigned)(TM * 4)), dim3(256), 0, s, p);
    (void)ctx;
    return 0;
}

template <bool TA, bool TB>
static int gemm_launch_tile(el_ctx* ctx, const GemmParams& p, const GemmPlan& pl, hipStream_t s) {
    if (pl.tm == 2 && pl.tn == 2) return gemm_launch_v<TA, TB, 2, 2>(ctx, p, s);
    if (pl.tm == 2 && pl.tn == 1) return gemm_launch_v<TA, TB, 2, 1>(ctx, p, s);
    if (pl.tm == 1 && pl.tn == 2) return gemm_launch_v<TA, TB, 1, 2>(ctx, p, s);
    return gemm_launch_v<TA, TB, 1, 1>(ctx, p, s);
}

// C[M,N] = act(op(A) op(B) + bias).  transA = 0: A is [M,K] (lda >= K); 1: A is stored [K,M] (lda >= M).
// transB = 0: B is [K,N] (ldb >= N); 1: B is stored [N,K] (ldb >= K).
extern "C" int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           const float* bias, int act, void* ws, size_t ws_bytes) {
    return el_gemm_f32_x(ctx, stream, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, nullptr, 0, nullptr, nullptr, ws, ws_bytes, nullptr);
}

// out[c] = part[0, c] + part[1, c] + ... + part[P - 1, c], in a FIXED order: a workgroup owns 16 columns, its 64 thread rows a contiguous
// share of the P partial rows each (ascending), the 64 shares are added in ascending order by the first thread row.  Used wherever a
// reduction over the batch used to end in float atomics (bias gradients, head-weight gradients).
__global__ __launch_bounds__(1024) void k_colsum_finish(const float* __restrict__ part, int P, int64_t C, float* __restrict__ out) {
    __shared__ float red[64][17];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + tc;
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
        out[c] = t;
    }
}

int el_colsum_finish(void* stream, const float* part, int P, int64_t C, float* out) {
    if (C <= 0) return 0;
    EL_REQUIRE(part && out && P >= 1, "el_colsum_finish: bad arguments");
    EL_LAUNCH("k_colsum_finish", k_colsum_finish, dim3((unsigned)((C + 15) / 16)), dim3(1024), 0, (hipStream_t)stream, part, P, C, out);
    return 0;
}

// el_gemm_f32 + the optional backward epilogue of a Dense layer (library-internal: el_neural.hip).  rmask / colsum: C is masked with
// the ReLU derivative of rmask [M, ldy] and colsum[N] receives its column sums (SET, summed in a fixed order through colsum_part
// [ceil(M / 128), N]) -- when the product runs on k_gemm_b3 without a K split; *fused tells (0: C holds the plain product, the caller
// applies mask and sums itself).
int el_gemm_f32_x(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, const float* rmask, int64_t ldy,
                  float* colsum, float* colsum_part, void* ws, size_t ws_bytes, int* fused) {
    if (fused) *fused = 0;
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(A && B && C, "el_gemm_f32: null matrix");
    EL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "el_gemm_f32: negative dimension");
    EL_REQUIRE(act >= EL_ACT_NONE && act <= EL_ACT_SIGMOID, "el_gemm_f32: bad activation %d", act);
    EL_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "el_gemm_f32: leading dimension too small");
    if (M == 0 || N == 0) return 0;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A;
    p.B = B;
    p.C = C;
    p.bias = bias;
    p.M = M;
    p.N = N;
    p.K = K;
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = ldc;
    p.act = act;
    p.vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    p.vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    // aligned fast path: a float4 of either operand never straddles the matrix edge
    const bool fast0 = p.vecA && p.vecB && K >= 1 && (transA ? M : K) % 4 == 0 && (transB ? K : N) % 4 == 0 &&
                       M < (1LL << 30) && N < (1LL << 30) && K < (1LL << 30) && ctx->zeros != nullptr;
    int splits = 1;
    GemmPlan pl = gemm_plan(ctx, M, N, K);
    const bool fast = fast0 && ws != nullptr && ws_bytes >= (size_t)2 * pl.P * (64 * pl.tm) * (64 * pl.tn) * 4;
    // three-way split on the bf16 matrix instruction (header of k_gemm_b3): the default wherever the fast path's alignment holds
    const bool split_on = ctx->opt.gemm_split != 0;          // (el_ctx_set_option("gemm_split"): the tests time and compare both forms)
    // (products under 2 GFLOP -- the 512 x 400 x 600 class -- are latency-bound: the one-launch small-tile path below stays faster)
    const bool outvec = N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16 == 0) && (bias == nullptr || (uintptr_t)bias % 16 == 0);
    if (split_on && fast0 && outvec && 2.0 * (double)M * (double)N * (double)K >= 2.0e9) {
        splits = gemm_splits(ctx, M, N, K);
        if (splits > 1 && (ws == nullptr || ws_bytes < (size_t)splits * M * N * 4)) splits = 1;
        p.kchunk = ((K + splits - 1) / splits + B3_BK - 1) / B3_BK * B3_BK;
        if (p.kchunk < B3_BK) p.kchunk = B3_BK;
        splits = (int)((K + p.kchunk - 1) / p.kchunk);
        if (splits < 1) splits = 1;
        p.ws = splits > 1 ? (float*)ws : nullptr;
        p.zeros = ctx->zeros;
        const bool want_mask = rmask != nullptr && colsum != nullptr && colsum_part != nullptr;
        const bool fuse_mask = want_mask && splits == 1 && ldy % 4 == 0 && ((uintptr_t)rmask % 16 == 0) && ldy >= N;
        if (fuse_mask) {
            p.rmask = rmask, p.ldy = ldy, p.colsum_part = colsum_part;
            if (fused) *fused = 1;
        }
        dim3 grid((unsigned)((N + B3_BN - 1) / B3_BN), (unsigned)((M + B3_BM - 1) / B3_BM), (unsigned)splits);
        // XCD-aware order (option gemm_xcd = 0: the plain 3-D grid): with K splits every tile of a split shares its two K chunks (up to
        // 64 tiles per group); without, the tiles along the SHORTER grid edge share the strip of the longer operand
        const bool xcd_on = ctx->opt.gemm_xcd != 0;
        p.gx = (int)grid.x, p.gy = (int)grid.y, p.gz = (int)grid.z;
        if (xcd_on && (int64_t)grid.x * grid.y * grid.z >= 16) {
            int64_t grp = 0;
            // (K splits: one group = every tile of a split -- they read the same two K chunks.  Only with a split count that deals
            //  the XCDs equal shares: 26 groups over 8 XCDs left two of them a third more work, 0.152 -> 0.176 ms at 512 x 600 x 26 744;
            //  gemm_splits rounds to a multiple of 8 for that reason)
            if (splits > 1) {
                if (splits % 8 == 0 && (int64_t)grid.x * grid.y <= 64) p.grp_mode = 3, grp = (int64_t)grid.x * grid.y, p.ngroups = splits;
            } else if (grid.y <= grid.x && grid.y <= 64) p.grp_mode = 1, grp = grid.y, p.ngroups = (int)((int64_t)grid.x * grid.z);
            else if (grid.x <= 64) p.grp_mode = 2, grp = grid.x, p.ngroups = (int)((int64_t)grid.y * grid.z);
            if (p.grp_mode != 0) {
                const int64_t rounds = ((int64_t)p.ngroups + 7) / 8;
                if (rounds * 8 * grp < (1LL << 31)) grid = dim3((unsigned)(rounds * 8 * grp), 1, 1);
                else p.grp_mode = 0;
            }
        }
        // A is k-contiguous unless transposed ([K, M]); B ([K, N]) is k-contiguous when transposed ([N, K])
        if (!transA && !transB) EL_LAUNCH("k_gemm_b3", (k_gemm_b3<true, false>), grid, dim3(256), 0, s, p);
        else if (!transA && transB) EL_LAUNCH("k_gemm_b3", (k_gemm_b3<true, true>), grid, dim3(256), 0, s, p);
        else if (transA && !transB) EL_LAUNCH("k_gemm_b3", (k_gemm_b3<false, false>), grid, dim3(256), 0, s, p);
        else EL_LAUNCH("k_gemm_b3", (k_gemm_b3<false, true>), grid, dim3(256), 0, s, p);
        if (fuse_mask)
            if (int rc = el_colsum_finish(s, colsum_part, p.gy, N, colsum)) return rc;
    } else if (fast) {
        p.ws = (float*)ws;
        p.units = pl.units;
        p.nkt = pl.nkt;
        p.tiles_n = pl.tiles_n;
        p.P = pl.P;
        p.whole_tiles = pl.whole_tiles;
        p.zeros = ctx->zeros;
        int rc;
        if (!transA && !transB) rc = gemm_launch_tile<false, false>(ctx, p, pl, s);
        else if (!transA && transB) rc = gemm_launch_tile<false, true>(ctx, p, pl, s);
        else if (transA && !transB) rc = gemm_launch_tile<true, false>(ctx, p, pl, s);
        else rc = gemm_launch_tile<true, true>(ctx, p, pl, s);
        if (rc) return rc;
    } else {
        splits = gemm_splits(ctx, M, N, K);
        if (splits > 1 && (ws == nullptr || ws_bytes < (size_t)splits * M * N * 4)) splits = 1;
        p.kchunk = ((K + splits - 1) / splits + GBK - 1) / GBK * GBK;
        if (p.kchunk < GBK) p.kchunk = GBK;
        splits = (int)((K + p.kchunk - 1) / p.kchunk);
        if (splits < 1) splits = 1;
        p.ws = splits > 1 ? (float*)ws : nullptr;
        dim3 grid((unsigned)((N + GBN - 1) / GBN), (unsigned)((M + GBM - 1) / GBM), (unsigned)splits);
        const size_t lds = (size_t)4 * G_TILE * 4;
#define EL_GEMM_GO(TA_, TB_)                                                                                          \
    do {                                                                                                              \
        auto kern = k_gemm_f32<TA_, TB_>;                                                                             \
        EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        EL_LAUNCH("k_gemm_f32", kern, grid, dim3(256), lds, s, p);                                                    \
    } while (0)
        if (!transA && !transB) EL_GEMM_GO(false, false);
        else if (!transA && transB) EL_GEMM_GO(false, true);
        else if (transA && !transB) EL_GEMM_GO(true, false);
        else EL_GEMM_GO(true, true);
#undef EL_GEMM_GO
    }
    EL_CHECK_LAUNCH();
    if (splits > 1) {
        const int64_t n = M * N;
        EL_LAUNCH("k_gemm_reduce", k_gemm_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.ws, splits, M, N, C,
                  ldc, bias, act);
        EL_CHECK_LAUNCH();
    }
    return 0;
}
