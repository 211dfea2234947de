// fp32 GEMM on v_mfma_f32_32x32x2_f32 with fused bias + activation epilogue.
//
// The dense layers of the reference's neural latent-factor models are Keras `Dense` calls
// (multi_vae_model.py:44-53,72-78; neural_matrix_factorization_model.py:59-64): y = act(x W + b) forward and
// the two transposed products backward.  One tiled kernel serves all of them:
//   C[M,N] = act( op(A)[M,K] * op(B)[K,N] + bias[N] )
// op() selects the storage order of each operand (K-contiguous or X-contiguous), so x W, dY W^T and X^T dY
// run without materialising a transpose.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles of
// 32x32), BK = 32, both operands staged through LDS with a register prefetch of the next K tile, one
// __syncthreads per K tile (two LDS buffers).  LDS images are chosen per storage order so that every
// ds_read_b32 of an MFMA operand is bank-conflict free:
//   K-contiguous operand  -> tile[X][BK+1]   (read (x, k) at x*33 + k)
//   X-contiguous operand  -> tile[BK][BX+4]  (read (x, k) at k*132 + x, float4 stores stay 16-B aligned)
// Small-MN / huge-K products (dH = dLogits * W4^T: 512 x 600 x 26744) are split along K over gridDim.z; the
// partial tiles go to a workspace and k_gemm_reduce sums them in a fixed order (deterministic) and applies the
// epilogue.  Numerics: fp32 fma chains in k order (MFMA f32 is exact fp32, MI355X_MICROARCH.md).
#include "el_common.h"

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
    float* ws;       // split-K partials [gridDim.z][M][N] or NULL
    int vecA, vecB;
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

static int gemm_splits(el_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
    const int64_t target = (int64_t)ctx->cus * 2;
    if (tiles >= target / 2 || K < 4 * GBK) return 1;
    int64_t s = (target + tiles - 1) / tiles;
    const int64_t maxs = K / (2 * GBK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

extern "C" size_t el_gemm_ws_bytes(el_ctx* ctx, int64_t M, int64_t N, int64_t K) {
    if (!ctx) return 0;
    const int s = gemm_splits(ctx, M, N, K);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * 4 : 0;
}

// C[M,N] = act(op(A) op(B) + bias).  transA = 0: A is [M,K] (lda >= K); 1: A is stored [K,M] (lda >= M).
// transB = 0: B is [K,N] (ldb >= N); 1: B is stored [N,K] (ldb >= K).
extern "C" int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           const float* bias, int act, void* ws, size_t ws_bytes) {
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
    int splits = gemm_splits(ctx, M, N, K);
    if (splits > 1 && (ws == nullptr || ws_bytes < (size_t)splits * M * N * 4)) splits = 1;
    p.kchunk = ((K + splits - 1) / splits + GBK - 1) / GBK * GBK;
    if (p.kchunk < GBK) p.kchunk = GBK;
    splits = (int)((K + p.kchunk - 1) / p.kchunk);
    if (splits < 1) splits = 1;
    p.ws = splits > 1 ? (float*)ws : nullptr;
    hipStream_t s = (hipStream_t)stream;
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
    EL_CHECK_LAUNCH();
    if (splits > 1) {
        const int64_t n = M * N;
        EL_LAUNCH("k_gemm_reduce", k_gemm_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.ws, splits, M, N, C,
                  ldc, bias, act);
        EL_CHECK_LAUNCH();
    }
    return 0;
}
