// Developer experiment: how fast does a bare stream of v_mfma_f32_32x32x16_bf16 run in the issue patterns of k_screen_pass?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// CH accumulator chains per wave, KS k-steps per chain round; operands live in registers (loaded once)
template <int CH, int KS, int NT>
__global__ __launch_bounds__(NT) void k_rate(const uint4* __restrict__ src, float* out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a[KS], b[CH][KS];
    for (int k = 0; k < KS; ++k) {
        uint4 v = src[(tid * 37 + k * 11) & 4095];
        a[k] = *reinterpret_cast<bf16x8*>(&v);
        for (int c = 0; c < CH; ++c) {
            uint4 w = src[(tid * 13 + k * 7 + c * 101) & 4095];
            b[c][k] = *reinterpret_cast<bf16x8*>(&w);
        }
    }
    floatx16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[c][k], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 123.456f) out[tid] = s;
}

// the fp32 matrix instruction of el_gemm.hip / el_topk.hip: v_mfma_f32_32x32x2_f32, operands in registers
template <int CH, int KS, int NT>
__global__ __launch_bounds__(NT) void k_rate_f32(const float* __restrict__ src, float* out, int iters) {
    const int tid = threadIdx.x;
    float a[KS], b[CH][KS];
    for (int k = 0; k < KS; ++k) {
        a[k] = src[(tid * 37 + k * 11) & 16383];
        for (int c = 0; c < CH; ++c) b[c][k] = src[(tid * 13 + k * 7 + c * 101) & 16383];
    }
    floatx16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[c][k], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 123.456f) out[tid] = s;
}

template <int CH, int KS, int NT>
static void run_f32(const char* name, const float* src, float* out, int wgs, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate_f32<CH, KS, NT><<<wgs, NT>>>(src, out, iters);
    CK(hipEventRecord(e0));
    k_rate_f32<CH, KS, NT><<<wgs, NT>>>(src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)wgs * (NT / 64) * iters * KS * CH * 32.0 * 32 * 2 * 2;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.0f %% of 157.3)\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 1.573);
}

template <int CH, int KS, int NT>
static void run(const char* name, const uint4* src, float* out, int wgs, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<CH, KS, NT><<<wgs, NT>>>(src, out, iters);
    CK(hipEventRecord(e0));
    k_rate<CH, KS, NT><<<wgs, NT>>>(src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)wgs * (NT / 64) * iters * KS * CH * 32.0 * 32 * 16 * 2;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.0f %% of 2500)\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 25.0);
}

int main() {
    uint4* src; float* out;
    CK(hipMalloc(&src, 4096 * 16)); CK(hipMalloc(&out, 4096));
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(3);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random bf16 around +-1
    CK(hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice));
    const int it = 20000;
    run<2, 8, 512>("2 chains, 8 waves/WG, 1 WG/CU (pass 2)", src, out, 256, it);
    run<2, 8, 512>("2 chains, 8 waves/WG, 2 WG/CU", src, out, 512, it);
    run<4, 8, 512>("4 chains, 8 waves/WG, 1 WG/CU", src, out, 256, it / 2);
    run<1, 8, 512>("1 chain, 8 waves/WG, 1 WG/CU", src, out, 256, it * 2);
    run<2, 8, 256>("2 chains, 4 waves/WG, 1 WG/CU (1 wave/SIMD)", src, out, 256, it);
    run<4, 8, 256>("4 chains, 4 waves/WG, 1 WG/CU (1 wave/SIMD)", src, out, 256, it / 2);
    run<2, 8, 1024>("2 chains, 16 waves/WG, 1 WG/CU (4 waves/SIMD)", src, out, 256, it);
    // zero operands: the DVFS reference point
    CK(hipMemset(src, 0, 4096 * 16));
    run<2, 8, 512>("2 chains, 8 waves/WG, 1 WG/CU, ZERO operands", src, out, 256, it);
    // fp32 matrix instruction
    float* hf = (float*)malloc(16384 * 4);
    for (int i = 0; i < 16384; ++i) hf[i] = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(src, hf, 16384 * 4, hipMemcpyHostToDevice));
    run_f32<4, 8, 256>("f32 32x32x2: 4 chains, 4 waves/WG, 2 WG/CU", (const float*)src, out, 512, it / 2);
    run_f32<4, 8, 512>("f32 32x32x2: 4 chains, 8 waves/WG, 1 WG/CU", (const float*)src, out, 256, it / 2);
    CK(hipMemset(src, 0, 16384 * 4));
    run_f32<4, 8, 256>("f32 32x32x2: 4 chains, ZERO operands", (const float*)src, out, 512, it / 2);
    return 0;
}
