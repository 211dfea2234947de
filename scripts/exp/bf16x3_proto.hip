// Prototype (developer experiment): C[M,N] = A[M,K] * B[K,N] in fp32-equivalent accuracy on the bf16 matrix cores.
// Each fp32 operand value is split into three bf16 pieces (hi + mid + lo = the value to 2^-24 relative); six of the nine piece
// products (hh, hm, mh, mm, hl, lh) are accumulated in fp32.  Operands are pre-packed into stage-contiguous chunks so the kernel's
// global -> LDS traffic is pure lane-linear LDS-DMA.   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x3 bf16x3_proto.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned int u32;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 f2bf(float x) {
    u32 b = __float_as_uint(x);
    b += 0x7fffu + ((b >> 16) & 1u);
    return b >> 16;
}
__device__ __forceinline__ float bf2f(u32 h) { return __uint_as_float(h << 16); }

// packed operand: [row block of 128][k block of 16][plane 3][k half 2][row 128][8 bf16]  (12 KiB per (row block, k block))
constexpr int RB = 128, KBK = 16, CHUNK = 3 * 2 * RB * 8;   // bf16 elements per chunk

// src is [rows][K] (trans = 0, ld >= K) or [K][rows] (trans = 1, ld >= rows)
__global__ void k_pack(const float* __restrict__ src, int64_t rows, int64_t K, int64_t ld, int trans, u16* __restrict__ dst, int64_t nkb) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (row, 8-k group)
    const int64_t rows_p = (rows + RB - 1) / RB * RB;
    const int64_t groups = nkb * 2;
    if (t >= rows_p * groups) return;
    int64_t row, g;
    if (trans) { row = t % rows_p; g = t / rows_p; } else { g = t % groups; row = t / groups; }
    const int64_t kb = g >> 1; const int half = (int)(g & 1);
    u16 out[3][8];
    for (int j = 0; j < 8; ++j) {
        const int64_t k = kb * KBK + half * 8 + j;
        float v = 0.f;
        if (row < rows && k < K) v = trans ? src[k * ld + row] : src[row * ld + k];
        const u32 h = f2bf(v); const float r1 = v - bf2f(h);
        const u32 m = f2bf(r1); const float r2 = r1 - bf2f(m);
        const u32 l = f2bf(r2);
        out[0][j] = (u16)h; out[1][j] = (u16)m; out[2][j] = (u16)l;
    }
    const int64_t rb = row / RB; const int r = (int)(row % RB);
    u16* base = dst + (rb * nkb + kb) * CHUNK;
    for (int p = 0; p < 3; ++p) {
        uint4 v;
        v.x = out[p][0] | ((u32)out[p][1] << 16); v.y = out[p][2] | ((u32)out[p][3] << 16);
        v.z = out[p][4] | ((u32)out[p][5] << 16); v.w = out[p][6] | ((u32)out[p][7] << 16);
        *reinterpret_cast<uint4*>(base + ((p * 2 + half) * RB + r) * 8) = v;
    }
}

// 256 threads = 4 waves, tile 128 (M) x 128 (N); wave w: M rows (w >> 1) * 64, N cols (w & 1) * 64 -> 2 x 2 MFMA tiles.
// MFMA "a" operand = B' rows (N), "b" operand = A' rows (M): a lane ends up with 4 consecutive N of one M row -> float4 stores.
template <int NPROD>
__global__ __launch_bounds__(256, 3) void k_gemm_bf16x3(const u16* __restrict__ Ap, const u16* __restrict__ Bp, float* __restrict__ C,
                                                         int64_t M, int64_t N, int64_t ldc, int nkb, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 stages][A chunk 12 KiB | B chunk 12 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int col = lane & 31, hi = lane >> 5;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const u16* ga = Ap + (int64_t)tm * nkb * CHUNK;
    const u16* gb = Bp + (int64_t)tn * nkb * CHUNK;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int CHB = CHUNK * 2;            // bytes per chunk (12288)
    auto gload = [&](int kb) {
        char* base = smem + (kb & 1) * (2 * CHB);
        const char* sa = reinterpret_cast<const char*>(ga + (int64_t)kb * CHUNK);
        const char* sb = reinterpret_cast<const char*>(gb + (int64_t)kb * CHUNK);
#pragma unroll
        for (int q = 0; q < 3; ++q) {          // 12 KiB = 12 pieces of 1 KiB, 4 waves x 3
            const int piece = q * 4 + wave_u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sa + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(base + CHB + piece * 1024), 16, 0, 0);
        }
    };
    gload(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0)
        __syncthreads();
        if (kb + 1 < nkb) gload(kb + 1);
        const char* sA = smem + (kb & 1) * (2 * CHB);
        const char* sB = sA + CHB;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[t][p] = *reinterpret_cast<const bf16x8*>(sA + (((p * 2 + hi) * RB) + (wave >> 1) * 64 + t * 32 + col) * 16);
                fb[t][p] = *reinterpret_cast<const bf16x8*>(sB + (((p * 2 + hi) * RB) + (wave & 1) * 64 + t * 32 + col) * 16);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                floatx16 c = acc[a][b];
                // smallest terms first
                if (NPROD >= 6) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][2], fa[a][0], c, 0, 0, 0);    // B lo * A hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][2], c, 0, 0, 0);    // B hi * A lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][1], fa[a][1], c, 0, 0, 0);    // mid * mid
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][1], fa[a][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][0], fa[a][0], c, 0, 0, 0);
                acc[a][b] = c;
            }
    }
    // D[i = N index within tile][j = M index]: lane (col = M row, hi) holds N = 8q + 4hi + {0..3}
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int64_t m = (int64_t)tm * 128 + (wave >> 1) * 64 + a * 32 + col;
        if (m >= M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t n0 = (int64_t)tn * 128 + (wave & 1) * 64 + b * 32 + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = n0 + 8 * q;
                floatx4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                if (n + 3 < N) *reinterpret_cast<floatx4*>(C + m * ldc + n) = v;
                else for (int e = 0; e < 4; ++e) if (n + e < N) C[m * ldc + n + e] = v[e];
            }
        }
    }
}

int main(int argc, char** argv) {
    int64_t M = argc > 1 ? atoll(argv[1]) : 512, N = argc > 2 ? atoll(argv[2]) : 26744, K = argc > 3 ? atoll(argv[3]) : 600;
    const int64_t nkb = (K + KBK - 1) / KBK, Mp = (M + RB - 1) / RB * RB, Np = (N + RB - 1) / RB * RB;
    std::vector<float> hA(M * K), hB(K * N);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto& x : hB) x = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC; u16 *pA, *pB;
    CK(hipMalloc(&dA, M * K * 4)); CK(hipMalloc(&dB, K * N * 4)); CK(hipMalloc(&dC, M * N * 4));
    CK(hipMalloc(&pA, Mp / RB * nkb * CHUNK * 2)); CK(hipMalloc(&pB, Np / RB * nkb * CHUNK * 2));
    CK(hipMemcpy(dA, hA.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), K * N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int tiles_n = (int)(Np / RB), tiles = (int)(Mp / RB) * tiles_n;
    for (int nprod : {6, 3}) {
        float t_pack = 0, t_gemm = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            k_pack<<<(unsigned)((Mp * nkb * 2 + 255) / 256), 256>>>(dA, M, K, K, 0, pA, nkb);           // A [M][K]
            k_pack<<<(unsigned)((Np * nkb * 2 + 255) / 256), 256>>>(dB, N, K, N, 1, pB, nkb);           // B [K][N] -> rows = N
            CK(hipEventRecord(e1));
            if (nprod == 6) k_gemm_bf16x3<6><<<tiles, 256, 2 * 2 * CHUNK * 2>>>(pA, pB, dC, M, N, N, (int)nkb, tiles_n);
            else k_gemm_bf16x3<3><<<tiles, 256, 2 * 2 * CHUNK * 2>>>(pA, pB, dC, M, N, N, (int)nkb, tiles_n);
            CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
            float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
            if (it >= 2) { t_pack += a / 4; t_gemm += b / 4; }
        }
        std::vector<float> hC(M * N);
        CK(hipMemcpy(hC.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
        double maxrel = 0, maxrel32 = 0;
        for (int s = 0; s < 400; ++s) {
            const int64_t m = rand() % M, n = rand() % N;
            double ref = 0, mag = 0; float f32 = 0.f;
            for (int64_t k = 0; k < K; ++k) { ref += (double)hA[m * K + k] * hB[k * N + n]; mag += fabs((double)hA[m * K + k] * hB[k * N + n]); f32 = fmaf(hA[m * K + k], hB[k * N + n], f32); }
            maxrel = fmax(maxrel, fabs(hC[m * N + n] - ref) / mag);
            maxrel32 = fmax(maxrel32, fabs((double)f32 - ref) / mag);
        }
        printf("M=%lld N=%lld K=%lld products=%d: pack %.3f ms, gemm %.3f ms = %.1f TFLOP/s-equivalent (with pack %.1f); max |err| / sum|a b| = %.3g (fp32 fma chain: %.3g)\n",
               (long long)M, (long long)N, (long long)K, nprod, t_pack, t_gemm, 2.0 * M * N * K / t_gemm * 1e-9, 2.0 * M * N * K / (t_gemm + t_pack) * 1e-9, maxrel, maxrel32);
    }
    return 0;
}
