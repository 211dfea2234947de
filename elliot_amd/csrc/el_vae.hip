// Mult-VAE train step / predict.
//
// Replaces VariationalAutoEncoder.call / train_step / predict (multi_vae_model.py:115-155; Encoder :32-64,
// Decoder :67-83, Sampling :20-29) and the dense-row sampler feeding it (sparse_sampler.py:19-25).
//
//   x~ = l2_normalize(x) -> dropout -> h = tanh(x~ W1 + b1) -> [mu | logvar] = h [Wm | Wv] + [bm | bv]
//   z = mu + exp(logvar/2) eps -> h2 = tanh(z W3 + b3) -> logits = h2 W4 + b4
//   loss = -mean_b sum_i log_softmax(logits)_i x_i + anneal * (-1/2) mean_{b,j}(logvar - mu^2 - exp(logvar) + 1)
//
// The reference multiplies a 99.7 %-sparse dense [B, I] block by W1; here the batch stays CSR and the first
// layer is a gather-sum of W1 rows (k_vae_enc1), its weight gradient one dense GEMM against the densified batch (k_vae_densify).  Every
// other product is a dense fp32 MFMA GEMM (el_gemm.hip) with bias/tanh fused into the epilogue; softmax +
// multinomial NLL + its gradient are one row-wise kernel over the logits.  Adam uses the arithmetic of TF's
// dense ApplyAdam kernel (m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= (m*alpha)/(sqrt(v)+eps)).
// The L2 kernel_regularizers declared by the reference are never added to its loss: no-op here too.
#include "el_common.h"

extern "C" int el_gemm_f32(el_ctx* ctx, void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           const float* bias, int act, void* ws, size_t ws_bytes);

// ---- first layer: CSR rows x W1 ------------------------------------------------------------------------
// keep-probability mask of nonzero (user, item): Philox keyed by (seed, step); 1/(1-rate) scaling as Keras Dropout
__device__ __forceinline__ float vae_drop_scale(float rate, u64 seed, u32 step, u32 user, u32 item) {
    if (rate <= 0.f) return 1.f;
    el_philox4 r = el_philox4x32_10(user, item, step, 0u, (u32)seed, (u32)(seed >> 32));
    const float uni = (float)(r.x >> 8) * (1.0f / 16777216.0f);
    return uni < rate ? 0.f : 1.0f / (1.0f - rate);
}

// one workgroup (NW = 16 waves) per batch row; wave w takes nonzeros w, w+NW, ...; lane owns float4 chunks
// c = lane, lane+64, ... of the H hidden units; the partial rows are combined through LDS in a fixed order.
// (Four waves per row left the kernel's duration to its heaviest row: a user with 3000 interactions is a chain of 190
//  dependent index -> W1-row round trips per wave; sixteen waves cut that chain fourfold and the 512 rows still fill the chip.)
constexpr int ENC_NW = 16;
template <int CPL>
__global__ __launch_bounds__(ENC_NW * 64) void k_vae_enc1(const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices, const float* __restrict__ W1,
                                                  const float* __restrict__ b1, int64_t B, int H, float rate, u64 seed,
                                                  u32 step, float* __restrict__ h, float* __restrict__ rnorm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* part = reinterpret_cast<float4*>(smem);   // [ENC_NW][H/4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const int32_t user = rows[b];
    const int64_t r0 = indptr[user], r1 = indptr[user + 1];
    // K.l2_normalize(x, axis=1): x / sqrt(max(sum x^2, 1e-12)); x is the binary train row (sp_i_train)
    const float nrm = 1.0f / sqrtf(fmaxf((float)(r1 - r0), 1e-12f));
    float acc[CPL][4];
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
    const int H4 = H >> 2;
    // ENC_U nonzeros of a wave are in flight together: a heavy row (thousands of items) is a chain of dependent
    // index -> W1-row loads, and with one workgroup per batch row that chain, not bandwidth, was the kernel's duration
    constexpr int ENC_U = 4;
    for (int64_t e0 = r0 + wave * ENC_U; e0 < r1; e0 += ENC_NW * ENC_U) {
        float wv[ENC_U];
        const float4* wr[ENC_U];
#pragma unroll
        for (int t = 0; t < ENC_U; ++t) {
            const int64_t e = e0 + t;
            const int32_t item = e < r1 ? indices[e] : 0;
            wv[t] = e < r1 ? nrm * vae_drop_scale(rate, seed, step, (u32)user, (u32)item) : 0.f;
            wr[t] = reinterpret_cast<const float4*>(W1 + (int64_t)item * H);
        }
        float4 tv[ENC_U][CPL];
#pragma unroll
        for (int t = 0; t < ENC_U; ++t)
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int c = lane + q * 64;
                tv[t][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wv[t] != 0.f && c < H4) tv[t][q] = wr[t][c];
            }
#pragma unroll
        for (int t = 0; t < ENC_U; ++t)                       // same summation order as one nonzero at a time
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                acc[q][0] += wv[t] * tv[t][q].x;
                acc[q][1] += wv[t] * tv[t][q].y;
                acc[q][2] += wv[t] * tv[t][q].z;
                acc[q][3] += wv[t] * tv[t][q].w;
            }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane + q * 64;
        if (c < H4) part[wave * H4 + c] = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    }
    __syncthreads();
    float* hb = h + b * (int64_t)H;
    for (int c = threadIdx.x; c < H4; c += ENC_NW * 64) {
        float4 t = part[c];
#pragma unroll
        for (int w = 1; w < ENC_NW; ++w) {                  // fixed order: run-to-run identical sums
            const float4 q = part[w * H4 + c];
            t.x += q.x;
            t.y += q.y;
            t.z += q.z;
            t.w += q.w;
        }
        const float4 bb = reinterpret_cast<const float4*>(b1)[c];
        float4 o;
        o.x = tanhf(t.x + bb.x);
        o.y = tanhf(t.y + bb.y);
        o.z = tanhf(t.z + bb.z);
        o.w = tanhf(t.w + bb.w);
        reinterpret_cast<float4*>(hb)[c] = o;
    }
    if (threadIdx.x == 0) rnorm[b] = nrm;
}

// ---- dW1 = x~^T dh on the batch's nonzeros ---------------------------------------------------------------------------
// x~ (the l2-normalised, dropped-out batch rows) has ~130 nonzeros per row of 26 744: as a dense GEMM (round 1: densify +
// 26744 x 600 x 512 on the MFMA pipe, 0.21 ms) 99.5 % of the products are zeros.  Here the batch is transposed on the fly --
// count per item (atomics on int counters), exclusive scan, fill (atomic cursors), and one wave per item that SORTS its list of
// batch rows before it adds coef(b, item) * dh[b, :] in ascending b -- so the sums do not depend on the order the atomics
// landed in (deterministic, like the GEMM it replaces) and every row of gW1 is written exactly once (zeros for items the
// batch does not touch: gW1 is a dense variable's gradient, Keras' dense Adam reads all of it).
__global__ __launch_bounds__(256) void k_vae_w1_count(const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices, int32_t* __restrict__ cnt) {
    const int32_t user = rows[blockIdx.x];
    const int64_t r0 = indptr[user], r1 = indptr[user + 1];
    for (int64_t e = r0 + threadIdx.x; e < r1; e += 256) atomicAdd(cnt + indices[e], 1);
}

// off[0..I] = exclusive scan of cnt[0..I), cursor = off; one workgroup of 16 waves, coalesced: wave w owns a contiguous chunk,
// pass 1 adds it up, pass 2 (after the 16 chunk sums are prefixed) writes 64 offsets per iteration from a wave-level scan
// ... and lists the HEAVY items (more than W1_LIGHT batch rows) in heavy[1..], their number in heavy[0]
constexpr int W1_LIGHT = 16;
__global__ __launch_bounds__(1024) void k_vae_w1_scan(const int32_t* __restrict__ cnt, int64_t I, int32_t* __restrict__ off,
                                                      int32_t* __restrict__ cursor, int32_t* __restrict__ heavy) {
    __shared__ int32_t tot[16];
    __shared__ int32_t nheavy;
    if (threadIdx.x == 0) nheavy = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = ((I + 15) / 16 + 63) / 64 * 64, i0 = wave * chunk, i1 = (i0 + chunk < I) ? i0 + chunk : I;
    int32_t s = 0;
    for (int64_t i = i0 + lane; i < i1; i += 64) s += cnt[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) tot[wave] = s;
    __syncthreads();
    int32_t carry = 0;
    for (int w = 0; w < wave; ++w) carry += tot[w];
    for (int64_t base = i0; base < i1; base += 64) {
        const int64_t i = base + lane;
        const int32_t v = i < i1 ? cnt[i] : 0;
        int32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (i < i1) {
            off[i] = carry + inc - v;
            cursor[i] = carry + inc - v;
            if (v > W1_LIGHT) heavy[1 + atomicAdd(&nheavy, 1)] = (int32_t)i;     // (few items: LDS atomics; any order)
        }
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 1023) off[I] = carry;                 // the last wave's final carry = the grand total (its chunk may be empty)
    __syncthreads();
    if (threadIdx.x == 0) heavy[0] = nheavy;
}

__global__ __launch_bounds__(256) void k_vae_w1_fill(const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, int32_t* __restrict__ cursor,
                                                     int32_t* __restrict__ pairs) {
    const int32_t b = blockIdx.x, user = rows[b];
    const int64_t r0 = indptr[user], r1 = indptr[user + 1];
    for (int64_t e = r0 + threadIdx.x; e < r1; e += 256) pairs[atomicAdd(cursor + indices[e], 1)] = b;
}

// one workgroup (W1_NW = 16 waves) per item.  The item's list of batch rows is turned into presence flags in LDS (batch rows
// are distinct), wave w then walks ITS contiguous range of b ascending -- so the order of the sum is fixed whatever order the
// atomics of k_vae_w1_fill landed in, without sorting -- four rows in flight at a time; the 16 partial rows are combined in wave
// order.  A popular item sits in most rows of the batch: with one wave per item its 500 dependent row reads were the
// kernel's whole duration.
// LIGHT items (at most W1_LIGHT = 16 batch rows; most of the catalogue): one wave per item, four items per workgroup, no LDS --
// lane t < n holds list entry t, its rank among the entries gives the ascending order, four rows in flight at a time.
template <int CPL>
__global__ __launch_bounds__(256) void k_vae_w1_light(const int32_t* __restrict__ rows, const float* __restrict__ rnorm,
                                                      const int32_t* __restrict__ off, const int32_t* __restrict__ pairs,
                                                      const float* __restrict__ dh, int64_t I, int H, float rate, u64 seed, u32 step,
                                                      float* __restrict__ gW1) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= I) return;
    const int o0 = off[item], n = off[item + 1] - o0;
    if (n > W1_LIGHT) return;                                // k_vae_w1_rows writes this row
    const int H4 = H >> 2;
    float4 acc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) {
        const int myb = lane < n ? pairs[o0 + lane] : 0x7fffffff;
        float mycf = 0.f;
        if (lane < n) mycf = rnorm[myb] * vae_drop_scale(rate, seed, step, (u32)rows[myb], (u32)item);
        int rank = 0;
        for (int t = 0; t < n; ++t) rank += (__shfl(myb, t, 64) < myb) ? 1 : 0;      // batch rows are distinct
        for (int r0 = 0; r0 < n; r0 += 4) {
            int bb[4];
            float cf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u64 who = __ballot(lane < n && rank == r0 + t);
                const int src = who ? __ffsll((long long)who) - 1 : 0;
                bb[t] = who ? __shfl(myb, src, 64) : -1;
                cf[t] = who ? __shfl(mycf, src, 64) : 0.f;
            }
            float4 v[4][CPL];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int c = lane + q * 64;
                    v[t][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bb[t] >= 0 && c < H4) v[t][q] = reinterpret_cast<const float4*>(dh + (int64_t)bb[t] * H)[c];
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    acc[q].x += cf[t] * v[t][q].x;
                    acc[q].y += cf[t] * v[t][q].y;
                    acc[q].z += cf[t] * v[t][q].z;
                    acc[q].w += cf[t] * v[t][q].w;
                }
        }
    }
    float4* dst = reinterpret_cast<float4*>(gW1 + item * (int64_t)H);
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane + q * 64;
        if (c < H4) dst[c] = acc[q];
    }
}

// HEAVY items (listed by k_vae_w1_scan): a fixed grid walks the list.
constexpr int W1_NW = 16;
template <int CPL>
__global__ __launch_bounds__(W1_NW * 64) void k_vae_w1_rows(const int32_t* __restrict__ rows, const float* __restrict__ rnorm,
                                                          const int32_t* __restrict__ off, const int32_t* __restrict__ pairs,
                                                          const float* __restrict__ dh, int64_t I, int H, int cap, float rate, u64 seed,
                                                          u32 step, float* __restrict__ gW1, const int32_t* __restrict__ heavy) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H4 = H >> 2;
    float4* part = reinterpret_cast<float4*>(smem);                          // [W1_NW][H4]
    unsigned char* present = reinterpret_cast<unsigned char*>(part + (size_t)W1_NW * H4);   // [cap]
    const int nh = heavy[0];
  for (int hi = blockIdx.x; hi < nh; hi += gridDim.x) {
    const int64_t item = heavy[1 + hi];
    float4* dst = reinterpret_cast<float4*>(gW1 + item * (int64_t)H);
    const int o0 = off[item], n = off[item + 1] - o0;
    __syncthreads();                                         // (previous item's partials have been read)
    for (int t = tid; t < cap; t += W1_NW * 64) present[t] = 0;
    __syncthreads();
    for (int t = tid; t < n; t += W1_NW * 64) present[pairs[o0 + t]] = 1;
    __syncthreads();
    float4 acc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int R = cap / W1_NW;                               // rows of the batch per wave (cap >= 1024: R >= 64)
    for (int b0 = wave * R; b0 < (wave + 1) * R; b0 += 64) {
        u64 m = __ballot(present[b0 + lane] != 0);
        while (m) {
            int bb[4];
            float cf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {                    // up to four present rows, their coefficients and row reads in flight
                bb[t] = -1;
                cf[t] = 0.f;
                if (m) {
                    bb[t] = b0 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    cf[t] = rnorm[bb[t]] * vae_drop_scale(rate, seed, step, (u32)rows[bb[t]], (u32)item);
                }
            }
            float4 v[4][CPL];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int c = lane + q * 64;
                    v[t][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bb[t] >= 0 && c < H4) v[t][q] = reinterpret_cast<const float4*>(dh + (int64_t)bb[t] * H)[c];
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    acc[q].x += cf[t] * v[t][q].x;
                    acc[q].y += cf[t] * v[t][q].y;
                    acc[q].z += cf[t] * v[t][q].z;
                    acc[q].w += cf[t] * v[t][q].w;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = lane + q * 64;
        if (c < H4) part[wave * H4 + c] = acc[q];
    }
    __syncthreads();
    for (int c = tid; c < H4; c += W1_NW * 64) {
        float4 t = part[c];
#pragma unroll
        for (int w = 1; w < W1_NW; ++w) {
            const float4 q = part[w * H4 + c];
            t.x += q.x;
            t.y += q.y;
            t.z += q.z;
            t.w += q.w;
        }
        dst[c] = t;
    }
  }
}

// dense image of the normalised (and dropped-out) batch rows: xd[b, item] = x~(b, item), zero elsewhere.
// dW1 = xd^T dh is then one dense MFMA GEMM (the reference multiplies the dense block too) -- deterministic,
// and 4x faster here than scattering 600-float rows with atomics onto popular items.
__global__ __launch_bounds__(256) void k_vae_densify(const int32_t* __restrict__ rows, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, const float* __restrict__ rnorm,
                                                     int64_t B, int64_t I, float rate, u64 seed, u32 step,
                                                     float* __restrict__ xd) {
    const int64_t b = blockIdx.x;
    const int32_t user = rows[b];
    const int64_t r0 = indptr[user], r1 = indptr[user + 1];
    const float nrm = rnorm[b];
    for (int64_t e = r0 + threadIdx.x; e < r1; e += 256) {
        const int32_t item = indices[e];
        xd[b * I + item] = nrm * vae_drop_scale(rate, seed, step, (u32)user, (u32)item);
    }
}

// ---- sampling + KL ------------------------------------------------------------------------------------------
// mv = [mu | logvar] [B, 2L]; z = mu + exp(logvar/2) eps (multi_vae_model.py:25-29);
// loss += anneal * (-1/2) * mean_{b,j}(logvar - mu^2 - exp(logvar) + 1)   (:119-121, :137)
__global__ __launch_bounds__(256) void k_vae_sample(const float* __restrict__ mv, const float* __restrict__ eps,
                                                    int64_t B, int L, float anneal, float* __restrict__ z,
                                                    double* loss_out, int64_t Bd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (t < B * L) {
        const int64_t b = t / L;
        const int j = (int)(t - b * L);
        const float mu = mv[b * 2 * L + j], lv = mv[b * 2 * L + L + j];
        const float e = eps ? eps[t] : 0.f;
        z[t] = mu + expf(0.5f * lv) * e;
        term = lv - mu * mu - expf(lv) + 1.0f;
    }
    __shared__ float wsum[4];
    float wl = el_group_sum(term, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = wl;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out) {
        const double tot = (double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3];
        if (tot != 0.0 && anneal != 0.f) atomicAdd(loss_out, (double)anneal * (-0.5) * tot / ((double)Bd * (double)L));
    }
}

// d[mu | logvar] from dz and the KL term
__global__ __launch_bounds__(256) void k_vae_dmv(const float* __restrict__ dz, const float* __restrict__ mv,
                                                 const float* __restrict__ eps, int64_t B, int L, float anneal,
                                                 float* __restrict__ dmv, int64_t Bd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * L) return;
    const int64_t b = t / L;
    const int j = (int)(t - b * L);
    const float mu = mv[b * 2 * L + j], lv = mv[b * 2 * L + L + j];
    const float e = eps ? eps[t] : 0.f;
    const float g = dz[t];
    const float kscale = anneal / ((float)Bd * (float)L);
    dmv[b * 2 * L + j] = g + kscale * mu;
    dmv[b * 2 * L + L + j] = g * e * 0.5f * expf(0.5f * lv) + kscale * 0.5f * (expf(lv) - 1.0f);
}

// in place: d <- d * (1 - y^2)   (tanh backward)
__global__ __launch_bounds__(256) void k_tanh_bwd(float* __restrict__ d, const float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const float yy = y[t];
        d[t] = d[t] * (1.0f - yy * yy);
    }
}


// ---- row-wise log-softmax / multinomial NLL / gradient ---------------------------------------------------
// One workgroup per batch row over logits[b, 0..I).
//   mode 0 (predict): logits <- log_softmax(logits)                         (multi_vae_model.py:153-155)
//   mode 1 (train)  : loss += -(1/B) sum_i log_softmax_i x_i ;  logits <- d loss / d logits
//                     = (softmax * sum_i x_i - x) / B                        (:131-137)
constexpr int SMX_NT = 1024;
// one workgroup of 1024 threads per batch row (26 744 logits = 107 KB, L2-resident after the first sweep); 16-byte accesses
// when the rows are 16-byte aligned (I % 4 == 0)
template <bool V4>
__global__ __launch_bounds__(SMX_NT) void k_vae_softmax(float* __restrict__ logits, const int32_t* __restrict__ rows,
                                                        const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, int64_t B, int64_t I,
                                                        int mode, double* loss_out, int64_t Bd) {
    constexpr int NWV = SMX_NT / 64;
    __shared__ float red[NWV];
    __shared__ float bc;
    const int64_t b = blockIdx.x;
    float* row = logits + b * I;
    float4* row4 = reinterpret_cast<float4*>(row);
    const int64_t I4 = V4 ? (I >> 2) : 0;
    const int tid = threadIdx.x;
    float m = -INFINITY;
    if (V4) {
        for (int64_t i = tid; i < I4; i += SMX_NT) {
            const float4 v = row4[i];
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    } else {
        for (int64_t i = tid; i < I; i += SMX_NT) m = fmaxf(m, row[i]);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float t = red[0];
        for (int w = 1; w < NWV; ++w) t = fmaxf(t, red[w]);
        bc = t;
    }
    __syncthreads();
    m = bc;
    float s = 0.f;
    if (V4) {
        for (int64_t i = tid; i < I4; i += SMX_NT) {
            const float4 v = row4[i];
            s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
        }
    } else {
        for (int64_t i = tid; i < I; i += SMX_NT) s += expf(row[i] - m);
    }
    s = el_group_sum(s, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < NWV; ++w) t += red[w];
        bc = t;
    }
    __syncthreads();
    const float lse = m + logf(bc);
    if (mode == 0) {
        if (V4) {
            for (int64_t i = tid; i < I4; i += SMX_NT) {
                float4 v = row4[i];
                v.x -= lse, v.y -= lse, v.z -= lse, v.w -= lse;
                row4[i] = v;
            }
        } else {
            for (int64_t i = tid; i < I; i += SMX_NT) row[i] = row[i] - lse;
        }
        return;
    }
    const int32_t user = rows[b];
    const int64_t r0 = indptr[user], r1 = indptr[user + 1];
    const float cnt = (float)(r1 - r0);          // sum_i x_i for the binary train row
    // NLL over the row's positives (reads the ORIGINAL logits: do this before overwriting)
    float nll = 0.f;
    for (int64_t e = r0 + tid; e < r1; e += SMX_NT) nll += row[indices[e]] - lse;
    nll = el_group_sum(nll, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = nll;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < NWV; ++w) tot += (double)red[w];
        atomicAdd(loss_out, -tot / (double)Bd);
    }
    const float invB = 1.0f / (float)Bd;
    const float sc = cnt * invB;
    if (V4) {
        for (int64_t i = tid; i < I4; i += SMX_NT) {
            float4 v = row4[i];
            v.x = expf(v.x - lse) * sc, v.y = expf(v.y - lse) * sc, v.z = expf(v.z - lse) * sc, v.w = expf(v.w - lse) * sc;
            row4[i] = v;
        }
    } else {
        for (int64_t i = tid; i < I; i += SMX_NT) row[i] = expf(row[i] - lse) * sc;
    }
    __syncthreads();
    for (int64_t e = r0 + tid; e < r1; e += SMX_NT) row[indices[e]] -= invB;
}

// TF dense ApplyAdam arithmetic; resets the gradient buffer
__global__ __launch_bounds__(256) void k_adam_apply_dense(float* __restrict__ th, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                          float alpha, float b1, float b2, float eps, int zero_g) {
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const float gg = g[e];
        float mm = m[e], vv = v[e];
        mm = mm + (gg - mm) * omb1;
        vv = vv + (gg * gg - vv) * omb2;
        th[e] = th[e] - (mm * alpha) / (sqrtf(vv) + eps);
        m[e] = mm;
        v[e] = vv;
        if (zero_g && gg != 0.f) g[e] = 0.f;
    }
}

// ---- host orchestration -----------------------------------------------------------------------------------------
static unsigned grid1d(int64_t n, el_ctx* ctx) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->cus * 16;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

static int vae_check(const el_vae_state* st, int64_t B) {
    EL_REQUIRE(st != nullptr, "el_vae: null state");
    EL_REQUIRE(st->I >= 1 && st->H >= 4 && st->L >= 1 && st->H % 4 == 0, "el_vae: bad dims (H must be a multiple of 4)");
    EL_REQUIRE(st->H <= 1024, "el_vae: intermediate_dim > 1024 unsupported in this build");
    EL_REQUIRE(B >= 1 && B <= st->Bmax, "el_vae: batch %lld exceeds Bmax %lld", (long long)B, (long long)st->Bmax);
    for (int t = 0; t < 8; ++t) EL_REQUIRE(st->w[t] != nullptr, "el_vae: null weight %d", t);
    EL_REQUIRE(st->h && (st->dae || st->mv) && st->z && st->h2 && st->logits && st->rnorm, "el_vae: null activation buffer");
    return 0;
}

#define EL_VAE_ENC1(KERN, ...)                                                                          \
    do {                                                                                                \
        const int cpl = (st->H / 4 + 63) / 64;                                                          \
        const unsigned g = (unsigned)B;                                                                 \
        const size_t lds = (size_t)st->H * 4 * ENC_NW;                                                  \
        if (cpl <= 1) EL_LAUNCH(#KERN, (KERN<1>), dim3(g), dim3(ENC_NW * 64), lds, s, __VA_ARGS__);     \
        else if (cpl <= 2) EL_LAUNCH(#KERN, (KERN<2>), dim3(g), dim3(ENC_NW * 64), lds, s, __VA_ARGS__); \
        else if (cpl <= 3) EL_LAUNCH(#KERN, (KERN<3>), dim3(g), dim3(ENC_NW * 64), lds, s, __VA_ARGS__); \
        else EL_LAUNCH(#KERN, (KERN<4>), dim3(g), dim3(ENC_NW * 64), lds, s, __VA_ARGS__);              \
    } while (0)

// out[n] = sum_b X[b, n] in a FIXED order (el_gemm.hip: el_colsum_finish -- 16 columns x 64 contiguous row shares per workgroup, the
// shares added in ascending order): the bias gradients carried the last float atomics of the step's variables (k_colsum added 32-row
// slabs in whatever order they finished)
static int colsum(hipStream_t s, const float* X, int64_t B, int64_t N, float* out) {
    return el_colsum_finish(s, X, (int)B, N, out);
}

static int vae_forward(el_ctx* ctx, hipStream_t s, const el_vae_state* st, const int64_t* indptr, const int32_t* indices,
                       const int32_t* rows, int64_t B, const float* eps, float anneal, float rate, u64 seed, u32 step,
                       double* loss_out, int64_t Bd) {
    const int H = st->H, L = st->L;
    const int64_t I = st->I;
    EL_VAE_ENC1(k_vae_enc1, rows, indptr, indices, st->w[0], st->w[1], B, H, rate, seed, step, st->h, st->rnorm);
    if (st->dae) {
        // MultiDAE encoder (multi_dae_model.py:32-51): z = tanh(h Wm + bm), no log-variance head, no sampling, no KL
        if (int rc = el_gemm_f32(ctx, s, 0, 0, B, L, H, st->h, H, st->w[2], L, st->z, L, st->w[3], 1 /*tanh*/, st->ws, st->ws_bytes)) return rc;
    } else {
        if (int rc = el_gemm_f32(ctx, s, 0, 0, B, 2 * L, H, st->h, H, st->w[2], 2 * L, st->mv, 2 * L, st->w[3], 0, st->ws, st->ws_bytes)) return rc;
        EL_LAUNCH("k_vae_sample", k_vae_sample, dim3((unsigned)((B * L + 255) / 256)), dim3(256), 0, s, st->mv, eps, B, L, anneal, st->z, loss_out, Bd);
    }
    if (int rc = el_gemm_f32(ctx, s, 0, 0, B, H, L, st->z, L, st->w[4], H, st->h2, H, st->w[5], 1 /*tanh*/, st->ws, st->ws_bytes)) return rc;
    if (int rc = el_gemm_f32(ctx, s, 0, 0, B, I, H, st->h2, H, st->w[6], I, st->logits, I, st->w[7], 0, st->ws, st->ws_bytes)) return rc;
    EL_CHECK_LAUNCH();
    return 0;
}

// forward, loss (batch means over Bd rows: Bd = B, or the global batch when several ranks share a step) and backward: the eight
// gradient buffers st->g[] are complete on exit
// Adam on the variables whose bit is set in `mask` (bit k = variable k of W1 b1 [Wm|Wv] [bm|bv] W3 b3 W4 b4); defined below
static int vae_apply_vars(el_ctx* ctx, hipStream_t s, const el_vae_state* st, float lr_t, unsigned mask);

static int vae_grads(el_ctx* ctx, hipStream_t s, const el_vae_state* st, const int64_t* indptr, const int32_t* indices,
                     const int32_t* rows, int64_t B, int64_t Bd, const float* eps, float anneal, float dropout_rate,
                     uint64_t dropout_seed, int32_t step, double* loss_out) {
    if (int rc = vae_check(st, B)) return rc;
    EL_REQUIRE(indptr && indices && rows && loss_out && step >= 1 && Bd >= B, "el_vae: bad arguments");
    for (int t = 0; t < 8; ++t) EL_REQUIRE(st->g[t], "el_vae: gradient buffers missing");
    EL_REQUIRE(st->dh2 && (st->dae || st->dmv) && st->dh && st->dz, "el_vae_train_step: backward buffers missing");
    const int H = st->H, L = st->L;
    const int64_t I = st->I;
    if (int rc = vae_forward(ctx, s, st, indptr, indices, rows, B, eps, anneal, dropout_rate, dropout_seed, (u32)step, loss_out, Bd)) return rc;
    // loss + dlogits (in place)
    if (I % 4 == 0 && ((uintptr_t)st->logits & 15) == 0)
        EL_LAUNCH("k_vae_softmax", k_vae_softmax<true>, dim3((unsigned)B), dim3(SMX_NT), 0, s, st->logits, rows, indptr, indices, B, I, 1, loss_out, Bd);
    else
        EL_LAUNCH("k_vae_softmax", k_vae_softmax<false>, dim3((unsigned)B), dim3(SMX_NT), 0, s, st->logits, rows, indptr, indices, B, I, 1, loss_out, Bd);
    float* dl = st->logits;
    // Two streams (round 5).  The chain dl -> dh2 -> dz -> d[mu|logvar] -> dh -> dW1 is serial and, below the 26 744-wide product,
    // made of 512 x {200..600}^2 products that a few dozen workgroups finish in ~20 us each: latency, not throughput.  The weight
    // gradients dW4, dW3, dW[m|v] and the five bias gradients (column sums) depend on the chain but nothing depends on them until
    // the optimiser, so they run on the library's second stream beside it (fork after each link, one join at the end).  The side
    // products take the upper half of the workspace: enabled when each half holds what its products need (a host that sizes the
    // workspace 2 x el_gemm_ws_bytes gets it; the option vae_side = 0 turns it off: bench.py's per-kernel breakdown pass runs one
    // stream), and when dz does not alias z (dW3 reads z while the chain writes dz).
    const bool side_env = ctx->opt.vae_side != 0;
    const int64_t LL = st->dae ? L : 2 * L;
    const size_t w1_bytes = ((size_t)(4 * (I + 1) + 4) * 4 + 255) & ~(size_t)255;     // dW1's index arrays (below)
    const size_t half = (st->ws_bytes / 2) & ~(size_t)255;
    size_t need_side = el_gemm_ws_bytes(ctx, H, I, B), need_main = el_gemm_ws_bytes(ctx, B, H, I);
    {
        const size_t a = el_gemm_ws_bytes(ctx, L, H, B), b = el_gemm_ws_bytes(ctx, H, LL, B);
        const size_t c = el_gemm_ws_bytes(ctx, B, L, H), d = el_gemm_ws_bytes(ctx, B, H, LL), w1 = (size_t)(4 * (I + 1) + 4) * 4;
        need_side = need_side > a ? need_side : a, need_side = need_side > b ? need_side : b;
        need_main = need_main > c ? need_main : c, need_main = need_main > d ? need_main : d, need_main = need_main > w1 ? need_main : w1;
    }
    bool side_on = side_env && st->ws != nullptr && half >= need_side + w1_bytes && half >= need_main && st->dz != st->z;
    if (side_on) side_on = el_side_stream_ready(ctx);
    hipStream_t ss = side_on ? ctx->side : s;
    void* ws1 = st->ws;
    const size_t wsb1 = side_on ? half : st->ws_bytes;
    void* ws2 = side_on ? (void*)((char*)st->ws + half) : st->ws;
    const size_t wsb2 = side_on ? half - w1_bytes : st->ws_bytes;
    // From here on every way out of this function joins the side stream (whatever was forked onto it so far): a failing product or
    // launch in the middle of the backward pass must not leave work in flight on ctx->side that the caller's next call -- or its free
    // of dl / dh2 / g[] / the workspace -- would race with.
    struct Join {
        el_ctx* c; hipStream_t s, ss; bool on;
        ~Join() {
            if (!on) return;
            (void)hipEventRecord(c->side_ev[7], ss);
            (void)hipStreamWaitEvent(s, c->side_ev[7], 0);
        }
    } join{ctx, s, ss, side_on};
    // dW1 = x~^T dhpre by sparse transposition of the batch (k_vae_w1_*) when its scratch fits.  Its index arrays (per-item counts,
    // offsets, cursors, the heavy-item list) depend on the batch alone, the (item-ordered) list of batch rows goes into the logits
    // buffer once dl is consumed: with two streams the side stream builds all of it beside the chain (arrays at the end of its half
    // of the workspace), and the main stream only runs the two kernels that need dh.
    const int cpl1 = (H / 4 + 63) / 64;
    const bool sparse_w1 = B <= 2048 && I < (1LL << 24) && H % 4 == 0 && cpl1 <= 4 && st->ws != nullptr &&
                           wsb1 >= (size_t)(4 * (I + 1) + 4) * 4 && (((uintptr_t)st->dh | (uintptr_t)st->g[0]) & 15) == 0;
    int32_t* cnt = side_on ? (int32_t*)((char*)st->ws + half + wsb2) : (int32_t*)st->ws;
    int32_t* off = cnt + (I + 1);
    int32_t* cursor = off + (I + 1);
    int32_t* heavy = cursor + (I + 1);                              // [1 + I]
    int32_t* pairs = (int32_t*)st->logits;                          // <= B * I entries: every batch row has <= I nonzeros
    auto w1_index = [&](hipStream_t q, bool wait_dl_readers) -> int {
        EL_CHECK_HIP(hipMemsetAsync(cnt, 0, (size_t)(I + 1) * 4, q));
        EL_LAUNCH("k_vae_w1_count", k_vae_w1_count, dim3((unsigned)B), dim3(256), 0, q, rows, indptr, indices, cnt);
        EL_LAUNCH("k_vae_w1_scan", k_vae_w1_scan, dim3(1), dim3(1024), 0, q, cnt, I, off, cursor, heavy);
        if (wait_dl_readers) EL_CHECK_HIP(hipStreamWaitEvent(q, ctx->side_ev[4], 0));      // dh2 = dl W4^T (main stream) has read dl
        EL_LAUNCH("k_vae_w1_fill", k_vae_w1_fill, dim3((unsigned)B), dim3(256), 0, q, rows, indptr, indices, cursor, pairs);
        return 0;
    };
    auto fork = [&](int k) -> int {                       // what the main stream has produced so far is visible to the side stream
        if (!side_on) return 0;
        EL_CHECK_HIP(hipEventRecord(ctx->side_ev[k], s));
        EL_CHECK_HIP(hipStreamWaitEvent(ss, ctx->side_ev[k], 0));
        return 0;
    };
    // decoder output layer
    if (int rc = fork(0)) return rc;
    if (int rc = el_gemm_f32(ctx, ss, 1, 0, H, I, B, st->h2, H, dl, I, st->g[6], I, nullptr, 0, ws2, wsb2)) return rc;   // dW4 = h2^T dl
    if (int rc = colsum(ss, dl, B, I, st->g[7])) return rc;
    if (side_on) EL_CHECK_HIP(hipEventRecord(ctx->side_ev[6], ss));                                                       // dl consumed on the side
    if (int rc = el_gemm_f32(ctx, s, 0, 1, B, H, I, dl, I, st->w[6], I, st->dh2, H, nullptr, 0, ws1, wsb1)) return rc;   // dh2 = dl W4^T
    // (Measured in round 5 and not kept: the output layer's Adam step on the side stream as soon as its gradients are complete --
    //  0.770-0.781 against 0.768 ms per step; the dW1 index on the side stream -- 0.831-0.837 against 0.782: the side stream then waits for
    //  the chain's dl W4^T before it may overwrite dl, and everything queued behind that wait starts late.)
    EL_LAUNCH("k_tanh_bwd", k_tanh_bwd, dim3(grid1d(B * H, ctx)), dim3(256), 0, s, st->dh2, st->h2, B * H);
    if (int rc = fork(1)) return rc;
    if (int rc = el_gemm_f32(ctx, ss, 1, 0, L, H, B, st->z, L, st->dh2, H, st->g[4], H, nullptr, 0, ws2, wsb2)) return rc;  // dW3 = z^T dh2pre
    if (int rc = colsum(ss, st->dh2, B, H, st->g[5])) return rc;
    // (one stream: dz may reuse the z buffer after dW3 consumed z)
    if (int rc = el_gemm_f32(ctx, s, 0, 1, B, L, H, st->dh2, H, st->w[4], H, st->dz, L, nullptr, 0, ws1, wsb1)) return rc;   // dz = dh2pre W3^T
    if (st->dae) {
        EL_LAUNCH("k_tanh_bwd", k_tanh_bwd, dim3(grid1d(B * L, ctx)), dim3(256), 0, s, st->dz, st->z, B * L);                       // through tanh
        if (int rc = fork(2)) return rc;
        if (int rc = el_gemm_f32(ctx, ss, 1, 0, H, L, B, st->h, H, st->dz, L, st->g[2], L, nullptr, 0, ws2, wsb2)) return rc;    // dWm
        if (int rc = colsum(ss, st->dz, B, L, st->g[3])) return rc;
        if (int rc = el_gemm_f32(ctx, s, 0, 1, B, H, L, st->dz, L, st->w[2], L, st->dh, H, nullptr, 0, ws1, wsb1)) return rc;   // dh
    } else {
        EL_LAUNCH("k_vae_dmv", k_vae_dmv, dim3((unsigned)((B * L + 255) / 256)), dim3(256), 0, s, st->dz, st->mv, eps, B, L, anneal, st->dmv, Bd);
        if (int rc = fork(2)) return rc;
        if (int rc = el_gemm_f32(ctx, ss, 1, 0, H, 2 * L, B, st->h, H, st->dmv, 2 * L, st->g[2], 2 * L, nullptr, 0, ws2, wsb2)) return rc;  // dWmv
        if (int rc = colsum(ss, st->dmv, B, 2 * L, st->g[3])) return rc;
        if (int rc = el_gemm_f32(ctx, s, 0, 1, B, H, 2 * L, st->dmv, 2 * L, st->w[2], 2 * L, st->dh, H, nullptr, 0, ws1, wsb1)) return rc;  // dh = dmv Wmv^T
    }
    EL_LAUNCH("k_tanh_bwd", k_tanh_bwd, dim3(grid1d(B * H, ctx)), dim3(256), 0, s, st->dh, st->h, B * H);
    if (int rc = fork(3)) return rc;
    if (int rc = colsum(ss, st->dh, B, H, st->g[1])) return rc;
    if (side_on) EL_CHECK_HIP(hipStreamWaitEvent(s, ctx->side_ev[6], 0));       // the logits buffer is free again (dW1's scratch lives there)
    // (nothing is forked after this point: the Join guard above records the side stream's end and waits for it on every return below)
    if (sparse_w1) {
        if (int rc = w1_index(s, false)) return rc;                                       // on this stream, here (dl is consumed: ev 6 above)
        int cap = 1024;                                             // presence flags: >= 64 batch rows per wave
        while (cap < B) cap <<= 1;
        const size_t lds = (size_t)W1_NW * (H / 4) * 16 + (size_t)cap;
        const unsigned gl = (unsigned)((I + 3) / 4), gh = (unsigned)(ctx->cus * 2);
#define EL_W1R(CPL_)                                                                                                                   \
    do {                                                                                                                               \
        EL_LAUNCH("k_vae_w1_light", (k_vae_w1_light<CPL_>), dim3(gl), dim3(256), 0, s, rows, st->rnorm, off, pairs, st->dh, I, H,      \
                  dropout_rate, (u64)dropout_seed, (u32)step, st->g[0]);                                                               \
        EL_LAUNCH("k_vae_w1_rows", (k_vae_w1_rows<CPL_>), dim3(gh), dim3(W1_NW * 64), lds, s, rows, st->rnorm, off, pairs, st->dh, I, H, \
                  cap, dropout_rate, (u64)dropout_seed, (u32)step, st->g[0], heavy);                                                   \
    } while (0)
        if (cpl1 <= 1) EL_W1R(1);
        else if (cpl1 <= 2) EL_W1R(2);
        else if (cpl1 <= 3) EL_W1R(3);
        else EL_W1R(4);
#undef EL_W1R
        EL_CHECK_LAUNCH();
        return 0;
    }
    EL_CHECK_HIP(hipMemsetAsync(st->logits, 0, (size_t)B * I * 4, s));
    EL_LAUNCH("k_vae_densify", k_vae_densify, dim3((unsigned)B), dim3(256), 0, s, rows, indptr, indices, st->rnorm, B, I,
              dropout_rate, (u64)dropout_seed, (u32)step, st->logits);
    if (int rc = el_gemm_f32(ctx, s, 1, 0, I, H, B, st->logits, I, st->dh, H, st->g[0], H, nullptr, 0, ws1, wsb1)) return rc;
    EL_CHECK_LAUNCH();
    return 0;
}

// Adam on the ten variables (W1 b1 [Wm|Wv] [bm|bv] W3 b3 W4 b4)
// Keras dense Adam on all eight variables of the model from ONE launch (eight launches of 3-100 us kernels were mostly
// launch gaps), 16 bytes per lane: m += (g - m)(1 - b1); v += (g g - v)(1 - b2); theta -= (m alpha) / (sqrt(v) + eps)
struct AdamOct {
    float* th[8];
    float* g[8];
    float* m[8];
    float* v[8];
    int64_t n[8];
};
__global__ __launch_bounds__(256) void k_adam_apply_oct(AdamOct t, float alpha, float b1, float b2, float eps) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        float *th = t.th[k], *g = t.g[k], *m = t.m[k], *v = t.v[k];
        const int64_t n = t.n[k];
        int64_t done = 0;
        if ((((uintptr_t)th | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
            f4 *th4 = reinterpret_cast<f4*>(th), *m4 = reinterpret_cast<f4*>(m), *v4 = reinterpret_cast<f4*>(v);
            const f4* g4 = reinterpret_cast<const f4*>(g);
            const int64_t n4 = n >> 2;
            for (int64_t e = tid; e < n4; e += stride) {
                f4 a = __builtin_nontemporal_load(th4 + e), mm = __builtin_nontemporal_load(m4 + e), vv = __builtin_nontemporal_load(v4 + e);
                const f4 gg = __builtin_nontemporal_load(g4 + e);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    mm[x] = mm[x] + (gg[x] - mm[x]) * omb1;
                    vv[x] = vv[x] + (gg[x] * gg[x] - vv[x]) * omb2;
                    a[x] = a[x] - (mm[x] * alpha) / (sqrtf(vv[x]) + eps);
                }
                __builtin_nontemporal_store(a, th4 + e);
                __builtin_nontemporal_store(mm, m4 + e);
                __builtin_nontemporal_store(vv, v4 + e);
            }
            done = n4 << 2;
        }
        for (int64_t e = done + tid; e < n; e += stride) {
            const float gg = g[e];
            float mm = m[e], vv = v[e];
            mm = mm + (gg - mm) * omb1;
            vv = vv + (gg * gg - vv) * omb2;
            th[e] = th[e] - (mm * alpha) / (sqrtf(vv) + eps);
            m[e] = mm;
            v[e] = vv;
        }
    }
}

static int vae_apply_vars(el_ctx* ctx, hipStream_t s, const el_vae_state* st, float lr_t, unsigned mask) {
    for (int t = 0; t < 8; ++t) EL_REQUIRE(st->g[t] && st->m[t] && st->v[t], "el_vae: optimiser buffers missing");
    const int H = st->H, L = st->L;
    const int64_t I = st->I;
    const int64_t LL = st->dae ? L : 2 * L;
    const int64_t sizes[8] = {I * H, H, (int64_t)H * LL, LL, (int64_t)L * H, H, (int64_t)H * I, I};
    AdamOct t;
    int64_t big = 0;
    for (int k = 0; k < 8; ++k) {
        t.th[k] = (float*)st->w[k], t.g[k] = (float*)st->g[k], t.m[k] = (float*)st->m[k], t.v[k] = (float*)st->v[k];
        t.n[k] = ((mask >> k) & 1u) ? sizes[k] : 0;
        if (t.n[k] > big) big = t.n[k];
    }
    if (big == 0) return 0;
    EL_LAUNCH("k_adam_apply_dense", k_adam_apply_oct, dim3(grid1d(big / 4 + 1, ctx)), dim3(256), 0, s, t, lr_t, 0.9f, 0.999f, 1e-7f);
    EL_CHECK_LAUNCH();
    return 0;
}

static int vae_apply(el_ctx* ctx, hipStream_t s, const el_vae_state* st, float lr_t, unsigned done_mask = 0u) {
    return vae_apply_vars(ctx, s, st, lr_t, 0xFFu & ~done_mask);
}

extern "C" int el_vae_train_step(el_ctx* ctx, void* stream, const el_vae_state* st, const int64_t* indptr,
                                 const int32_t* indices, const int32_t* rows, int64_t B, const float* eps, float anneal,
                                 float dropout_rate, uint64_t dropout_seed, int32_t step, float lr_t, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = vae_grads(ctx, (hipStream_t)stream, st, indptr, indices, rows, B, B, eps, anneal, dropout_rate, dropout_seed, step, loss_out)) return rc;
    return vae_apply(ctx, (hipStream_t)stream, st, lr_t);
}

extern "C" int el_vae_grads(el_ctx* ctx, void* stream, const el_vae_state* st, const int64_t* indptr, const int32_t* indices,
                            const int32_t* rows, int64_t B, int64_t B_global, const float* eps, float anneal, float dropout_rate,
                            uint64_t dropout_seed, int32_t step, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    return vae_grads(ctx, (hipStream_t)stream, st, indptr, indices, rows, B, B_global, eps, anneal, dropout_rate, dropout_seed, step, loss_out);
}

extern "C" int el_vae_apply(el_ctx* ctx, void* stream, const el_vae_state* st, float lr_t) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(st != nullptr, "el_vae_apply: null state");
    return vae_apply(ctx, (hipStream_t)stream, st, lr_t);
}

// log_softmax(logits) of the batch rows into st->logits [B, I] (multi_vae_model.py:145-155; dropout off)
extern "C" int el_vae_predict(el_ctx* ctx, void* stream, const el_vae_state* st, const int64_t* indptr,
                              const int32_t* indices, const int32_t* rows, int64_t B, const float* eps) {
    if (int rc = el_bind(ctx)) return rc;
    if (int rc = vae_check(st, B)) return rc;
    EL_REQUIRE(indptr && indices && rows, "el_vae_predict: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = vae_forward(ctx, s, st, indptr, indices, rows, B, eps, 0.f, 0.f, 0, 0, nullptr, B)) return rc;
    if (st->I % 4 == 0 && ((uintptr_t)st->logits & 15) == 0)
        EL_LAUNCH("k_vae_softmax", k_vae_softmax<true>, dim3((unsigned)B), dim3(SMX_NT), 0, s, st->logits, rows, indptr, indices, B, st->I, 0, nullptr, B);
    else
        EL_LAUNCH("k_vae_softmax", k_vae_softmax<false>, dim3((unsigned)B), dim3(SMX_NT), 0, s, st->logits, rows, indptr, indices, B, st->I, 0, nullptr, B);
    EL_CHECK_LAUNCH();
    return 0;
}
