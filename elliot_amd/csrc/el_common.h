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

struct el_ctx {
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
    // el_nmf_score_topk: what the PI image (item-side layer-1 projection) in the last workspace was derived from
    const void* nmf_ws = nullptr;
    const float* nmf_Imlp = nullptr;
    const float* nmf_W1 = nullptr;
    int64_t nmf_I = 0;
    int nmf_E = 0, nmf_H1 = 0;
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
