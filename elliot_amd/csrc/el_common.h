// Shared host/device helpers for libelliot_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/elliot_hip.h"

#include <string>
#include <vector>

struct el_timing_rec {
    const char* name;
    hipEvent_t a, b;
};

// Switches of the library: ONE table (el_ctx.hip), read from the environment (EL_<NAME>) once, in el_ctx_create, and settable per
// context with el_ctx_set_option -- nothing in the library calls getenv after that.  Every entry selects between forms that are
// both product code (a test compares them, or bench.py reports one beside the other); measured-slower experiments do not live here.
struct el_options {
    double ichunk = 0;               // positions per lane group of the BPR item segments (0: by batch and catalogue size)
    double uchunk = 0;               // ... of the user segments
    double loop_graph = 1;           // el_bprmf_train_loop: small-batch epochs as one captured graph
    double gemm_split = 1;           // dense products >= 2 GFLOP on the three-plane bf16 kernel (0: the fp32 matrix instruction)
    double gemm_xcd = 1;             // XCD-aware tile order of that kernel
    double nmf_side = 1;             // NeuMF: weight-gradient products on the library's second stream
    double nmf_head4 = 1;            // NeuMF head: four samples per wave where the rows are 16-byte aligned (0: one sample per wave)
    double vae_side = 1;             // Mult-VAE: the same
    double nmf_screen_maxfrac = 0.5; // NeuMF scoring: the screen is kept when it leaves the exact kernel at most this share of the pairs
    double screen_stride = 0;        // screened top-k: catalogue stride of the first pass (0: by catalogue size)
    double screen_ka = 0;            // ... and the list length it keeps
    double screen_prof = 0;          // ... per-pass clock counters (profiling builds of the passes)
};

struct el_ctx {
    el_options opt;
    int device;
    int cus;
    int64_t hbm_bytes;
    char arch[64];
    // optional per-kernel timing (el_timing_enable): hipEvents recorded on the launch stream
    bool timing;
    std::string timing_only;   // el_timing_filter: when non-empty only launches of this name are bracketed
    bool tuning = false;   // el_tuning_mode: optimiser launches use the *_tune kernel instantiations
    float* zeros = nullptr; // 256 bytes of zeros in device memory (source of out-of-range LDS-DMA lanes, el_gemm.hip)
    // el_bprmf_train_loop: pinned staging copy of the caller's step-size table (the caller's array may be freed on return)
    float* lr_pinned = nullptr;
    size_t lr_pinned_cap = 0;
    hipEvent_t lr_copied = nullptr;
    std::vector<el_timing_rec> pending;
    std::vector<hipEvent_t> pool;
    // screened top-k: what the item-side image in the last workspace was derived from (EL_TOPK_ITEMS_UNCHANGED)
    const void* prep_ws = nullptr;
    const float* prep_Gi = nullptr;
    const float* prep_Bi = nullptr;
    int64_t prep_I = 0;
    int prep_F = 0;
    // last screened el_score_topk call: where its per-user record counters and its flagged-user counter live (el_topk_screen_stats)
    const int32_t* scr_cnt = nullptr;
    const int32_t* scr_flagged = nullptr;
    int64_t scr_users = 0;
    // el_nmf_score_topk: what the PI image (item-side layer-1 projection) in the last workspace was derived from
    const void* nmf_ws = nullptr;
    const float* nmf_Imlp = nullptr;
    const float* nmf_W1 = nullptr;
    int64_t nmf_I = 0;
    int nmf_E = 0, nmf_H1 = 0;
    size_t nmf_PI_off = 0, nmf_PIB_off = 0, nmf_Rn_off = 0;   // where the last call on nmf_ws put PI and its half-precision image (0: none)
    // second stream of the library (weight-gradient products of the Mult-VAE backward pass run beside the chain that produces the
    // input gradients) and the events that order the two; created on first use
    hipStream_t side = nullptr;
    hipEvent_t side_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool side_join_pending = false;      // el_nmf_train_step: the side stream's weight gradients are joined inside the apply half
    int64_t nmf_screen_cands = 0;        // last el_nmf_score_topk: pairs the exact kernel scored (/ users / I_local = the survival rate)
    bool nmf_screen_fallback = false;    // ... and whether a call that asked for the screen went without it
    // el_bprmf_train_loop: the captured small-batch step sequence (hipGraphExec_t) and the launch parameters it was built for
    void* loop_graph_exec = nullptr;
    std::vector<unsigned char> loop_graph_key;
};

extern thread_local el_ctx* g_el_cur_ctx;

// RAII: brackets one kernel launch with events when timing is enabled on the bound ctx.
struct ElKernelTimer {
    el_ctx* c;
    const char* name;
    hipStream_t s;
    hipEvent_t a, b;
    bool on;
    ElKernelTimer(const char* name_, hipStream_t s_) : c(g_el_cur_ctx), name(name_), s(s_), on(false) {
        if (c && c->timing && (c->timing_only.empty() || c->timing_only == name_)) {
            a = grab();
            b = grab();
            on = (a != nullptr && b != nullptr);
            if (on) (void)hipEventRecord(a, s);
        }
    }
    ~ElKernelTimer() {
        if (on) {
            (void)hipEventRecord(b, s);
            c->pending.push_back({name, a, b});
        }
    }
    hipEvent_t grab() {
        if (!c->pool.empty()) {
            hipEvent_t e = c->pool.back();
            c->pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

void el_set_error(const char* fmt, ...);
struct el_ctx;
__attribute__((visibility("hidden"))) bool el_side_stream_ready(el_ctx* ctx);      // (el_ctx.hip) creates ctx->side / side_ev on first use
// (el_gemm.hip; library-internal, not exported: the C ABI's el_gemm_f32 is its plain form)
__attribute__((visibility("hidden"))) int el_gemm_f32_x(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                                                        const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                                                        const float* bias, int act, const float* rmask, int64_t ldy, float* colsum,
                                                        float* colsum_part, void* ws, size_t ws_bytes, int* fused);
// out[c] = sum of part[0 .. P, c] in a fixed order (el_gemm.hip)
__attribute__((visibility("hidden"))) int el_colsum_finish(void* stream, const float* part, int P, int64_t C, float* out);

#define EL_CHECK_HIP(expr)                                                                \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            el_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                       \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

#define EL_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            el_set_error(__VA_ARGS__); \
            return 2;                  \
        }                              \
    } while (0)

#define EL_CHECK_LAUNCH() EL_CHECK_HIP(hipGetLastError())

// every kernel launch goes through here so that el_timing_enable() can bracket it with events
#define EL_LAUNCH(name, kern, grid, block, lds, stream, ...)               \
    do {                                                                   \
        ElKernelTimer _el_t(name, stream);                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);   \
    } while (0)

static inline int el_bind(el_ctx* ctx) {
    if (!ctx) {
        el_set_error("null el_ctx");
        return 2;
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) {
        el_set_error("hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
        return 1;
    }
    g_el_cur_ctx = ctx;
    return 0;
}

typedef unsigned long long u64;
typedef unsigned int u32;

#define EL_WAVE 64

// ---- order-preserving float <-> uint maps (top-k keys) ---------------------------
// key = ord(score) << 32 | ~item  : larger key == better (score desc, then index asc).
// -0.0f is canonicalised to +0.0f so that the key order equals IEEE compare order.
// key 0 (= ord of a negative NaN, item 0xffffffff) is the "empty slot" and sorts last.
__host__ __device__ inline u32 el_f2ord(float f) {
    union {
        float f;
        u32 u;
    } c;
    c.f = f;
    u32 b = c.u;
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float el_ord2f(u32 o) {
    union {
        float f;
        u32 u;
    } c;
    c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return c.f;
}
__host__ __device__ inline u64 el_make_key(float s, int32_t item) {
    return ((u64)el_f2ord(s) << 32) | (u64)(0xffffffffu - (u32)item);
}
__host__ __device__ inline int32_t el_key_item(u64 k) {
    return (int32_t)(0xffffffffu - (u32)(k & 0xffffffffull));
}
__host__ __device__ inline float el_key_score(u64 k) { return el_ord2f((u32)(k >> 32)); }

__host__ __device__ inline u64 el_d2ord(double d) {
    union {
        double d;
        u64 u;
    } c;
    c.d = d;
    u64 b = c.u;
    if (b == 0x8000000000000000ull) b = 0ull;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double el_ord2d(u64 o) {
    union {
        double d;
        u64 u;
    } c;
    c.u = (o & 0x8000000000000000ull) ? (o & 0x7fffffffffffffffull) : ~o;
    return c.d;
}

// ---- wave helpers -----------------------------------------------------------------
// LDS traffic of ONE wave is serviced in program order; this point keeps the compiler
// from moving LDS accesses across it and drains the wave's outstanding LDS operations.
// It is NOT a cross-wave barrier.
__device__ __forceinline__ void el_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt/expcnt untouched
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T>
__device__ __forceinline__ T el_group_sum(T v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// true in every lane of an aligned group of `width` lanes (a power of two <= 64) when any of them holds true
__device__ __forceinline__ bool el_group_any(bool v, int width) {
    const unsigned long long b = __ballot(v);
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long m = width >= 64 ? ~0ull : (((1ull << width) - 1ull) << (lane & ~(width - 1)));
    return (b & m) != 0ull;
}

// first position p in [lo, hi) with idx[p] >= x (idx ascending), hi if none
__device__ __forceinline__ int64_t el_lower_bound(const int32_t* __restrict__ idx, int64_t lo, int64_t hi,
                                                  int32_t x) {
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (idx[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
__device__ __forceinline__ bool el_row_contains(const int32_t* __restrict__ idx, int64_t lo, int64_t hi,
                                                int32_t x) {
    int64_t p = el_lower_bound(idx, lo, hi, x);
    return p < hi && idx[p] == x;
}

// ---- Philox4x32-10 (counter-based RNG; Salmon et al. 2011) ------------------------
struct el_philox4 {
    u32 x, y, z, w;
};
__host__ __device__ inline el_philox4 el_philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
    const u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        u64 p0 = (u64)M0 * c0, p1 = (u64)M1 * c2;
        u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
        u32 n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    el_philox4 o = {c0, c1, c2, c3};
    return o;
}

// Keras-2.3 Adam, sparse-apply arithmetic, one element:
//   m <- m*b1 ; m += g*(1-b1) ; v <- v*b2 ; v += (g*g)*(1-b2) ; theta -= (lr_t*m)/(sqrt(v)+eps)
__device__ __forceinline__ void el_adam_elem(float& th, float& m, float& v, float g, float lr_t, float b1, float b2,
                                             float omb1, float omb2, float eps) {
    m = m * b1 + g * omb1;
    v = v * b2 + (g * g) * omb2;
    th = th - (lr_t * m) / (sqrtf(v) + eps);
}

// ---- the gradient-free Adam step of the deferred decay on packed fp32 instructions -------------------------------------------
// A row that waits for its postponed updates is replayed with g = 0: per element and step one IEEE square root and one IEEE
// division, which the compiler expands into ~37 VALU issue slots (v_sqrt + neighbour test + scaling; div_scale x 2, v_rcp, five
// fma, div_fmas, div_fixup).  el_adam_replay2 takes the SAME step on two elements at once with v_pk_mul / v_pk_add / v_pk_fma:
//   sqrt   v_rsq + Goldschmidt / Markstein in fma only: y = rsq(v); g = v y; h = y / 2; r = 1/2 - h g; g += g r; h += h r;
//          g += (v - g g) h -- equal to sqrtf() for EVERY float in [2^-96, 2^96] (checked exhaustively on the device, 1.6e9 inputs:
//          scripts/exp/replay_math.hip, tests/test_gpu_bpr.py::test_packed_replay_arithmetic_is_exact)
//   div    the compiler's own expansion without v_div_scale / the scale of v_div_fmas / v_div_fixup, which are the identity while the
//          numerator is above 2^-103, the quotient normal and the exponents less than 96 apart
// -- 18.5 issue slots per element and step, the same bits.  el_replay_ok is the guard: inside it the next EL_REPLAY_CHUNK steps
// stay in the ranges above (m shrinks by 0.9 per step, v by 0.999; lr in [2^-30, 1]); a wave with any element outside (zeros,
// denormals, huge values) takes those steps with el_adam_elem.
typedef float el_f2 __attribute__((ext_vector_type(2)));
#define EL_REPLAY_CHUNK 8

__device__ __forceinline__ bool el_replay_ok(float m, float v) {
    const u32 mb = __float_as_uint(m) & 0x7fffffffu, vb = __float_as_uint(v);
    return (mb - ((127u - 58u) << 23)) < (88u << 23) &&          // |m| in [2^-58, 2^30)
           (vb - ((127u - 89u) << 23)) < (149u << 23);           // v in [2^-89, 2^60)
}
__device__ __forceinline__ bool el_replay_lr_ok(float lr) { return lr >= 9.3132257e-10f && lr <= 1.0f; }     // [2^-30, 1]

__device__ __forceinline__ el_f2 el_pk_sqrt(el_f2 v) {           // == sqrtf per element for v in [2^-96, 2^96]
    const el_f2 half = {0.5f, 0.5f};
    el_f2 y;
    y.x = __builtin_amdgcn_rsqf(v.x), y.y = __builtin_amdgcn_rsqf(v.y);
    el_f2 g = v * y, h = y * 0.5f;
    const el_f2 r = __builtin_elementwise_fma(-h, g, half);
    g = __builtin_elementwise_fma(g, r, g);
    h = __builtin_elementwise_fma(h, r, h);
    const el_f2 d = __builtin_elementwise_fma(-g, g, v);
    return __builtin_elementwise_fma(d, h, g);
}
__device__ __forceinline__ el_f2 el_pk_div(el_f2 num, el_f2 den) {  // == num / den per element inside the no-scaling region
    const el_f2 one = {1.f, 1.f};
    el_f2 r0;
    r0.x = __builtin_amdgcn_rcpf(den.x), r0.y = __builtin_amdgcn_rcpf(den.y);
    const el_f2 e = __builtin_elementwise_fma(-den, r0, one);
    const el_f2 r1 = __builtin_elementwise_fma(e, r0, r0);
    const el_f2 q0 = num * r1;
    const el_f2 m0 = __builtin_elementwise_fma(-den, q0, num);
    const el_f2 q1 = __builtin_elementwise_fma(m0, r1, q0);
    const el_f2 m1 = __builtin_elementwise_fma(-den, q1, num);
    return __builtin_elementwise_fma(m1, r1, q1);
}
__device__ __forceinline__ void el_adam_replay2(el_f2& th, el_f2& m, el_f2& v, float lr) {
    const el_f2 zero = {0.f, 0.f};
    m = m * 0.9f + zero;                                          // m*b1 + g*(1-b1), v*b2 + (g*g)*(1-b2) with g = 0
    v = v * 0.999f + zero;
    th = th - el_pk_div(m * lr, el_pk_sqrt(v) + 1e-7f);           // theta - (lr_t m) / (sqrt(v) + eps)
}

// The same step inside a run of consecutive steps (round 5): sqrt(v) of step k is sqrt(0.999 v) of step k - 1 -- the correctly rounded
// root s and the refined half-reciprocal h of the step before, scaled by sqrt(b2) and 1 / sqrt(b2), are a seed of the accuracy
// v_rsq gives (2^-22; v' = RN(b2 v), s = RN(sqrt v), two constant roundings, one product rounding), so the refinement and the final
// residual step RN(g + h (v' - g g)) that follow return the correctly rounded root again (the residual is exact in the fma, a square
// root is never a rounding tie) -- without the transcendental instruction, a quarter-rate one.  Inside the guard of el_replay_ok m
// and v stay normal and non-zero through a chunk, so the `+ 0` of m b1 + g (1 - b1) with g = 0 (which only turns a -0 product into
// +0) is the identity there and is left out.  Checked against el_adam_elem on 2^31 random 8-step runs on every box
// (el_selftest_replay_math, out[2]) and by the bit-for-bit tests of the deferred decay.
#define EL_SQRT_B2 0.9994998574256897f       // RN(sqrt(0.999f))
#define EL_RSQRT_B2 1.0005003213882446f      // RN(1 / sqrt(0.999f))
template <bool FIRST>
__device__ __forceinline__ void el_adam_replay2s(el_f2& th, el_f2& m, el_f2& v, float lr, el_f2& G, el_f2& H) {
    const el_f2 half = {0.5f, 0.5f};
    m = m * 0.9f;
    v = v * 0.999f;
    el_f2 g, h;
    if (FIRST) {
        el_f2 y;
        y.x = __builtin_amdgcn_rsqf(v.x), y.y = __builtin_amdgcn_rsqf(v.y);
        g = v * y, h = y * 0.5f;
    } else {
        g = G * EL_SQRT_B2, h = H * EL_RSQRT_B2;
    }
    const el_f2 r = __builtin_elementwise_fma(-h, g, half);
    g = __builtin_elementwise_fma(g, r, g);
    h = __builtin_elementwise_fma(h, r, h);
    const el_f2 d = __builtin_elementwise_fma(-g, g, v);
    G = __builtin_elementwise_fma(d, h, g);                        // == sqrtf(v) per element
    H = h;
    th = th - el_pk_div(m * lr, G + 1e-7f);                        // theta - (lr_t m) / (sqrt(v) + eps)
}

// ns gradient-free steps on the VW elements a lane holds (wave-uniform ns; lr of step k at lrs(k)).  Elements at the m = v = 0
// fixed point of the step (rows or slots that never had a gradient, lanes past the end of a row) do not keep a wave off the packed
// path: they ride along and get their values back.
template <int VW, typename LR>
__device__ __forceinline__ void el_adam_replay(float (&th)[VW], float (&mm)[VW], float (&vv)[VW], int ns, LR lrs) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#ifdef EL_REPLAY_SIMPLE                                            // (A/B builds: the plain loop of rounds 3)
    for (int s = 0; s < ns; ++s) {
        const float lr = lrs(s);
#pragma unroll
        for (int x = 0; x < VW; ++x) el_adam_elem(th[x], mm[x], vv[x], 0.0f, lr, b1, b2, omb1, omb2, eps);
    }
    return;
#endif
    for (int s0 = 0; s0 < ns; s0 += EL_REPLAY_CHUNK) {
        const int c = ns - s0 < EL_REPLAY_CHUNK ? ns - s0 : EL_REPLAY_CHUNK;
        // the chunk's step sizes first, all loads in flight together (a load per step in front of its arithmetic serialises the
        // ring's latency with it: measured +25 % on the whole kernel)
        float lrv[EL_REPLAY_CHUNK];
#pragma unroll
        for (int k = 0; k < EL_REPLAY_CHUNK; ++k) lrv[k] = lrs(s0 + (k < c ? k : 0));
#ifdef EL_NO_PACKED_REPLAY                                         // (A/B builds: scripts/exp/build_variants.sh)
        bool fast = false, dead = true;
        if (false) {
#else
        bool fast = VW >= 2, dead = true;
        if (VW >= 2) {
#endif
            bool ok = true;
#pragma unroll
            for (int x = 0; x < VW; ++x) {
                ok = ok & el_replay_ok(mm[x], vv[x]);
                dead = dead & (mm[x] == 0.f) & (vv[x] == 0.f);
            }
            ok = ok | dead;
#pragma unroll
            for (int k = 0; k < EL_REPLAY_CHUNK; ++k) ok = ok & el_replay_lr_ok(lrv[k]);
            fast = __ballot(!ok) == 0ull;
        }
        if (fast) {
            el_f2 T[VW >= 2 ? VW / 2 : 1], M[VW >= 2 ? VW / 2 : 1], V[VW >= 2 ? VW / 2 : 1];
#ifndef EL_REPLAY_UNSEEDED
            el_f2 SG[VW >= 2 ? VW / 2 : 1], SH[VW >= 2 ? VW / 2 : 1];    // sqrt(v) and its half-reciprocal of the step before
#pragma unroll
            for (int x = 0; x + 1 < VW; x += 2) {
                T[x / 2].x = th[x], T[x / 2].y = th[x + 1];
                M[x / 2].x = mm[x], M[x / 2].y = mm[x + 1];
                V[x / 2].x = vv[x], V[x / 2].y = vv[x + 1];
            }
#pragma unroll
            for (int x = 0; x + 1 < VW; x += 2) el_adam_replay2s<true>(T[x / 2], M[x / 2], V[x / 2], lrv[0], SG[x / 2], SH[x / 2]);   // (c >= 1)
#pragma unroll
            for (int k = 1; k < EL_REPLAY_CHUNK; ++k) {
                if (k < c) {
#pragma unroll
                    for (int x = 0; x + 1 < VW; x += 2) el_adam_replay2s<false>(T[x / 2], M[x / 2], V[x / 2], lrv[k], SG[x / 2], SH[x / 2]);
                }
            }
#else                                                                  // (A/B builds: every step seeds its square root with v_rsq)
#pragma unroll
            for (int x = 0; x + 1 < VW; x += 2) {
                T[x / 2].x = th[x], T[x / 2].y = th[x + 1];
                M[x / 2].x = mm[x], M[x / 2].y = mm[x + 1];
                V[x / 2].x = vv[x], V[x / 2].y = vv[x + 1];
            }
#pragma unroll
            for (int k = 0; k < EL_REPLAY_CHUNK; ++k) {
                if (k < c) {
#pragma unroll
                    for (int x = 0; x + 1 < VW; x += 2) el_adam_replay2(T[x / 2], M[x / 2], V[x / 2], lrv[k]);
                }
            }
#endif
#pragma unroll
            for (int x = 0; x + 1 < VW; x += 2) {
                th[x] = dead ? th[x] : T[x / 2].x, th[x + 1] = dead ? th[x + 1] : T[x / 2].y;
                mm[x] = dead ? 0.f : M[x / 2].x, mm[x + 1] = dead ? 0.f : M[x / 2].y;
                vv[x] = dead ? 0.f : V[x / 2].x, vv[x + 1] = dead ? 0.f : V[x / 2].y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < EL_REPLAY_CHUNK; ++k) {
                if (k < c) {
#pragma unroll
                    for (int x = 0; x < VW; ++x) el_adam_elem(th[x], mm[x], vv[x], 0.0f, lrv[k], b1, b2, omb1, omb2, eps);
                }
            }
        }
    }
}

// ---- the gradient-free Adam steps of a waiting row in closed form (el_bprmf_state.replay_series; round 6) ------------------------
// n gradient-free steps from (theta_0, m_0, v_0) are m_n = b1^n m_0, v_n = b2^n v_0 and
//     theta_n = theta_0 - m_0 sum_{k=1..n} lr_k b1^k / (r^k a_0 + eps),        a_0 = sqrt(v_0), r = sqrt(b2).
// With den = a_0 + eps, u = a_0 / den in [0, 1) and z_k = 1 - r^k (small: ~k / 2000), r^k a_0 + eps = den (1 - u z_k), so
//     1 / (r^k a_0 + eps) = (1 / den) (1 + u z_k + u^2 z_k^2 + u^3 z_k^3 + O(z_k^4))
// and the sum splits into FOUR ROW-LEVEL scalars that do not depend on the element,
//     T0 = sum lr_k b1^k,  Z1 = sum lr_k b1^k z_k,  Z2 = sum lr_k b1^k z_k^2,  Z3 = sum lr_k b1^k z_k^3        (all terms positive),
// and an O(1) update per element: theta_n = theta_0 - (m_0 / den) (T0 + u (Z1 + u (Z2 + u Z3))).  The weights b1^k cut the sums off
// (b1^k < 2^-149 past k = 980): a gap of any length costs at most 1024 scalar terms per ROW, and one square root + one reciprocal per
// ELEMENT, against one IEEE square root and one IEEE division per element AND STEP of the step-by-step replay.  Valid for every
// v_0 >= 0 (u = 0 gives Keras' m / eps exactly); m_0 = v_0 = 0 stays put.  This is NOT the same fp32 rounding sequence as Keras'
// per-step update: against the exact-arithmetic recurrence it is as close as the fp32 step-by-step form is (relative 1.5e-6 of the
// move, scripts/exp/series_check.py; tests/test_gpu_bpr.py::test_series_replay_*), which is what north_star's 1e-4 on the loss and
// the oracle tolerances of the parity tests ask for -- the bit-for-bit mode stays available (replay_series = 0).
struct el_series {
    float T0, Z1, Z2, Z3, p1, p2;      // the four sums, b1^n, b2^n
};
#define EL_OM_SQRT_B2 5.0011865e-4f    // 1 - sqrt(0.999f), rounded from the double value (NOT 1 - EL_SQRT_B2: the slope of z_k counts)
#define EL_SERIES_MAX 1024

template <typename LR>
__device__ __forceinline__ el_series el_adam_series_sums(int n, LR lrs) {
    const float b1 = 0.9f, b2 = 0.999f;
    el_series s;
    s.T0 = s.Z1 = s.Z2 = s.Z3 = 0.f;
    float p1 = 1.f, p2 = 1.f, z = 0.f;
    const int c = n < EL_SERIES_MAX ? n : EL_SERIES_MAX;
    for (int k0 = 0; k0 < c; k0 += 8) {
        float lrv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) lrv[k] = lrs(k0 + k < c ? k0 + k : k0);    // the chunk's step sizes in flight together
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k0 + k < c) {
                p1 *= b1;
                p2 *= b2;
                z = __builtin_fmaf(z, EL_SQRT_B2, EL_OM_SQRT_B2);              // 1 - r^k = (1 - r^(k-1)) r + (1 - r)
                const float w = lrv[k] * p1;
                float wz = w * z;
                s.T0 += w;
                s.Z1 += wz;
                wz *= z;
                s.Z2 += wz;
                wz *= z;
                s.Z3 += wz;
            }
        }
    }
    if (n > EL_SERIES_MAX) {
        p1 = 0.f;                                                              // 0.9^1024 < 2^-149
        p2 = powf(b2, (float)n);
    }
    s.p1 = p1, s.p2 = p2;
    return s;
}

template <int VW>
__device__ __forceinline__ void el_adam_series_apply(float (&th)[VW], float (&mm)[VW], float (&vv)[VW], const el_series& s) {
#pragma unroll
    for (int x = 0; x < VW; ++x) {
        const float a0 = __builtin_amdgcn_sqrtf(vv[x]);
        const float inv = __builtin_amdgcn_rcpf(a0 + 1e-7f);
        const float u = a0 * inv;
        const float S = __builtin_fmaf(u, __builtin_fmaf(u, __builtin_fmaf(u, s.Z3, s.Z2), s.Z1), s.T0);
        th[x] = th[x] - (mm[x] * inv) * S;
        mm[x] = mm[x] * s.p1;
        vv[x] = vv[x] * s.p2;
    }
}

// a waiting row brought forward ns steps: step by step (the same bits as Keras' every-row pass) or in closed form
template <int VW, bool SER, typename LR>
__device__ __forceinline__ void el_adam_catchup(float (&th)[VW], float (&mm)[VW], float (&vv)[VW], int ns, LR lrs) {
    if (SER) {
        const el_series s = el_adam_series_sums(ns, lrs);
        el_adam_series_apply<VW>(th, mm, vv, s);
    } else {
        el_adam_replay<VW>(th, mm, vv, ns, lrs);
    }
}

// ---- per-user record of the samplers (el_bpr_sampler_meta_build) -------------------------------------------------------
struct __attribute__((aligned(64))) SamplerRec {
    int64_t r0;      // row start in the positives CSR
    int32_t len;     // row length
    u32 pad;
    u32 sig[12];     // 384-bit membership signature of the row
};
#define EL_SIG_BITS 384u
__device__ __forceinline__ u32 el_sig_bit(int32_t item) { return ((u32)item * 0x9E3779B1u) % EL_SIG_BITS; }

// the user's record in one 64-byte read; -> row bounds; sig_maybe(cand): false = certainly not in the row
struct SamplerUser {
    int64_t r0, r1;
    uint4 sg[3];
    __device__ __forceinline__ void load(const SamplerRec* __restrict__ meta, u32 u) {
        const uint4* rec = reinterpret_cast<const uint4*>(meta + u);
        const uint4 h = rec[0];
        sg[0] = rec[1], sg[1] = rec[2], sg[2] = rec[3];
        r0 = (int64_t)(((u64)h.y << 32) | (u64)h.x);
        r1 = r0 + (int64_t)(int32_t)h.z;
    }
    __device__ __forceinline__ bool maybe(int32_t cand) const {
        const u32 b = el_sig_bit(cand), w = b >> 5;
        const uint4 q = w < 4 ? sg[0] : (w < 8 ? sg[1] : sg[2]);
        const u32 word = (w & 3u) == 0 ? q.x : ((w & 3u) == 1 ? q.y : ((w & 3u) == 2 ? q.z : q.w));
        return ((word >> (b & 31u)) & 1u) != 0u;
    }
};
