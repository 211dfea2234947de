// Experiment (not part of the library): dense TF-Adam pass over a [U][F] table with
//   SoA  theta / g / m / v in four arrays (distance tunable), vs
//   AoS  one [U][4][F] array (a row's theta, g, m, v adjacent: one address stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 ldnt(const float4* p) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void stnt(float4* p, float4 v) { f4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p)); }

__device__ __forceinline__ void upd(float& th, float g, float& m, float& v, float lr, float b1, float b2, float eps) {
    m = m * b1 + g * (1.f - b1);
    v = v * b2 + (g * g) * (1.f - b2);
    th = th - (lr * m) / (sqrtf(v) + eps);
}

__global__ __launch_bounds__(256) void k_soa(float4* th, float4* g, float4* m, float4* v, long n4, float lr) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
        float4 t = ldnt(th + e), gg = ldnt(g + e), mm = ldnt(m + e), vv = ldnt(v + e);
        upd(t.x, gg.x, mm.x, vv.x, lr, .9f, .999f, 1e-7f); upd(t.y, gg.y, mm.y, vv.y, lr, .9f, .999f, 1e-7f);
        upd(t.z, gg.z, mm.z, vv.z, lr, .9f, .999f, 1e-7f); upd(t.w, gg.w, mm.w, vv.w, lr, .9f, .999f, 1e-7f);
        stnt(th + e, t); stnt(m + e, mm); stnt(v + e, vv);
        if (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f) stnt(g + e, make_float4(0, 0, 0, 0));
    }
}

// AoS: row r occupies floats [r*4F, (r+1)*4F): theta F | g F | m F | v F.  F4 = F/4 float4 per part.
__global__ __launch_bounds__(256) void k_aos(float4* a, long rows, int F4, float lr) {
    long stride = (long)gridDim.x * blockDim.x;
    long n4 = rows * F4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
        long r = e / F4; int c = (int)(e - r * F4);
        float4* base = a + r * 4 * F4 + c;
        float4 t = ldnt(base), gg = ldnt(base + F4), mm = ldnt(base + 2 * F4), vv = ldnt(base + 3 * F4);
        upd(t.x, gg.x, mm.x, vv.x, lr, .9f, .999f, 1e-7f); upd(t.y, gg.y, mm.y, vv.y, lr, .9f, .999f, 1e-7f);
        upd(t.z, gg.z, mm.z, vv.z, lr, .9f, .999f, 1e-7f); upd(t.w, gg.w, mm.w, vv.w, lr, .9f, .999f, 1e-7f);
        stnt(base, t); stnt(base + 2 * F4, mm); stnt(base + 3 * F4, vv);
        if (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f) stnt(base + F4, make_float4(0, 0, 0, 0));
    }
}

int main() {
    const long U = 1000000; const int F = 128; const long n = U * F;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const long gaps[] = {0, 1 << 20, 3 << 20, 5 << 20};
    for (long gap : gaps) {
        long stride = n + gap / 4;
        float* buf; CK(hipMalloc(&buf, stride * 4 * 4)); CK(hipMemset(buf, 0, stride * 4 * 4));
        for (int it = 0; it < 3; ++it) k_soa<<<256 * 16, 256>>>((float4*)buf, (float4*)(buf + stride), (float4*)(buf + 2 * stride), (float4*)(buf + 3 * stride), n / 4, 1e-3f);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) k_soa<<<256 * 16, 256>>>((float4*)buf, (float4*)(buf + stride), (float4*)(buf + 2 * stride), (float4*)(buf + 3 * stride), n / 4, 1e-3f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("SoA gap %8ld B: %.3f ms  (%.0f GB/s algorithmic)\n", gap, ms / 10, 24.0 * n / (ms / 10) / 1e6);
        CK(hipFree(buf));
    }
    {
        float* buf; CK(hipMalloc(&buf, n * 4 * 4)); CK(hipMemset(buf, 0, n * 4 * 4));
        for (int grid : {256 * 8, 256 * 16, 256 * 32}) {
            for (int it = 0; it < 3; ++it) k_aos<<<grid, 256>>>((float4*)buf, U, F / 4, 1e-3f);
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it) k_aos<<<grid, 256>>>((float4*)buf, U, F / 4, 1e-3f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("AoS grid %6d: %.3f ms  (%.0f GB/s algorithmic)\n", grid, ms / 10, 24.0 * n / (ms / 10) / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
