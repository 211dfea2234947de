// MF2020 ("Neural Collaborative Filtering vs. Matrix Factorization Revisited"): point-wise logistic SGD in fp64, one sample after the other.
//
// Replaces MFModel.train_step (latent_factor_models/MF2020/MF_model.py:80-113): for every (user, item, rating) of the batch, IN ORDER,
//     prediction = gb + ub[u] + ib[i] + <P[u], Q[i]>;  sigmoid, loss (the two numerically stable branches of :92-99);  grad = rating - sigmoid
//     P[u] += lr (grad Q[i] - reg P[u]);  Q[i] += lr (grad P[u]' - reg Q[i])   -- P[u]' is the UPDATED row: `uf_` is a NumPy view of the
//     row that :103 has just modified in place (the same aliasing as BPRMF_model.py:108-113) --
//     ub[u] += lr (grad - reg ub);  ib[i] += lr (grad - reg ib);  gb += lr (grad - reg gb)                 (scalars: the OLD values)
// The global bias is read and written by every sample: the updates form one strict chain, there is nothing to run side by side.  What a
// GPU can still do is take the chain off the memory system: ONE workgroup walks the batch in chunks of C samples;
//   phase A (256 threads)  the chunk's samples into LDS; its distinct users and items get LDS slots (two small open-addressing hash
//                          tables, atomicCAS on LDS); their factor rows and biases are fetched from HBM into the slots -- every load of
//                          the chunk in flight together
//   phase B (one wave)     the C updates in order, entirely on LDS and registers: lane f holds element f of the two rows, the dot product
//                          is one wave reduction, exp / log / the division in fp64; the global bias stays in a register
//   phase C (256 threads)  the slots back to HBM
// What sits on the chain per sample: two LDS row reads, the 64-lane fp64 sum (data-parallel-primitive row operations, no LDS crossbar),
// one fp64 exp and one division, the row updates.  The loss is NOT on it: phase B leaves the sample's prediction in LDS and phase C
// computes the chunk's loss terms on all threads and adds them in sample order (`loss += this_loss`, :108).  Against ~20 us per sample
// for the reference's NumPy calls; a chain cannot go faster than its links.  The dot product's summation order is the reduction
// tree's, NumPy's is its BLAS': results agree to the last bits of fp64 (tests: 1e-12), not bit for bit -- as for BPRMF's
// el_bprsgd_apply.
#include "el_common.h"

#define MF20_THREADS 256
#define MF20_HASH 512          // entries per hash table (>= 2 x the largest chunk)

struct Mf20Lds {
    int32_t* s_u;      // [C] sample fields
    int32_t* s_i;
    float* s_y;
    int32_t* su;       // [C] slot of the sample's user / item
    int32_t* si;
    int32_t* hk_u;     // [MF20_HASH] hash keys (-1 empty) and slots, users / items
    int32_t* hs_u;
    int32_t* hk_i;
    int32_t* hs_i;
    int32_t* key_u;    // [C] row id held by a slot
    int32_t* key_i;
    int32_t* cnt;      // [2] slots in use
    double* pred;      // [C] the samples' predictions (loss terms: phase C)
    double* lterm;     // [C]
    double* bu;        // [C] slot biases
    double* bi;
    double* P;         // [C, F] slot rows
    double* Q;
};

__device__ __forceinline__ unsigned mf20_hash(int32_t k) { return ((unsigned)k * 0x9E3779B1u) >> 23; }   // 9 bits

__device__ __forceinline__ void mf20_insert(int32_t* hk, int32_t* hs, int32_t* key_of, int32_t* cnt, int32_t k) {
    unsigned h = mf20_hash(k) & (MF20_HASH - 1);
    for (;;) {
        const int32_t old = atomicCAS(&hk[h], -1, k);
        if (old == -1) {                       // this thread owns the key: a slot for it
            const int32_t slot = atomicAdd(cnt, 1);
            hs[h] = slot;
            key_of[slot] = k;
            return;
        }
        if (old == k) return;
        h = (h + 1) & (MF20_HASH - 1);
    }
}
__device__ __forceinline__ int32_t mf20_lookup(const int32_t* hk, const int32_t* hs, int32_t k) {
    unsigned h = mf20_hash(k) & (MF20_HASH - 1);
    while (hk[h] != k) h = (h + 1) & (MF20_HASH - 1);
    return hs[h];
}

// sum of v over the 64 lanes, returned to every lane: row operations of the data-parallel primitives on the two halves of the double
// (quad swaps, half-row and row mirrors, then the row broadcasts 15 / 31), the total read from lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double mf20_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double mf20_wave_sum(double v) {
    v = mf20_dpp_add<0xB1, 0xf>(v);       // quad_perm [1, 0, 3, 2]
    v = mf20_dpp_add<0x4E, 0xf>(v);       // quad_perm [2, 3, 0, 1]
    v = mf20_dpp_add<0x141, 0xf>(v);      // row_half_mirror
    v = mf20_dpp_add<0x140, 0xf>(v);      // row_mirror: every lane holds its row's 16-lane sum
    v = mf20_dpp_add<0x142, 0xa>(v);      // row_bcast15 into rows 1, 3
    v = mf20_dpp_add<0x143, 0xc>(v);      // row_bcast31 into rows 2, 3: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(MF20_THREADS) void k_mf2020_seq(el_mf2020_state st, const int32_t* __restrict__ samples, int64_t n, int C,
                                                             double* loss_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf20_lds[];
    const int F = st.F, tid = threadIdx.x, lane = tid & 63;
    Mf20Lds L;
    {
        unsigned char* p = mf20_lds;
        L.P = (double*)p, p += (size_t)C * F * 8;
        L.Q = (double*)p, p += (size_t)C * F * 8;
        L.bu = (double*)p, p += (size_t)C * 8;
        L.bi = (double*)p, p += (size_t)C * 8;
        L.pred = (double*)p, p += (size_t)C * 8;
        L.lterm = (double*)p, p += (size_t)C * 8;
        L.s_u = (int32_t*)p, p += (size_t)C * 4;
        L.s_i = (int32_t*)p, p += (size_t)C * 4;
        L.s_y = (float*)p, p += (size_t)C * 4;
        L.su = (int32_t*)p, p += (size_t)C * 4;
        L.si = (int32_t*)p, p += (size_t)C * 4;
        L.key_u = (int32_t*)p, p += (size_t)C * 4;
        L.key_i = (int32_t*)p, p += (size_t)C * 4;
        L.hk_u = (int32_t*)p, p += MF20_HASH * 4;
        L.hs_u = (int32_t*)p, p += MF20_HASH * 4;
        L.hk_i = (int32_t*)p, p += MF20_HASH * 4;
        L.hs_i = (int32_t*)p, p += MF20_HASH * 4;
        L.cnt = (int32_t*)p;
    }
    double gb = *st.gb, loss = 0.0;             // (wave 0 carries them through the chunks)
    const double lr = st.lr, reg = st.reg;
    for (int64_t c0 = 0; c0 < n; c0 += C) {
        const int cs = (int)((n - c0 < C) ? n - c0 : C);
        // ---- phase A: samples, slots, rows
        for (int h = tid; h < MF20_HASH; h += MF20_THREADS) L.hk_u[h] = L.hk_i[h] = -1;
        if (tid < 2) L.cnt[tid] = 0;
        if (tid < cs) {
            const int32_t* sp = samples + (c0 + tid) * 3;
            L.s_u[tid] = sp[0], L.s_i[tid] = sp[1], L.s_y[tid] = (float)sp[2];
        }
        __syncthreads();
        if (tid < cs) {
            mf20_insert(L.hk_u, L.hs_u, L.key_u, L.cnt, L.s_u[tid]);
            mf20_insert(L.hk_i, L.hs_i, L.key_i, L.cnt + 1, L.s_i[tid]);
        }
        __syncthreads();
        if (tid < cs) {
            L.su[tid] = mf20_lookup(L.hk_u, L.hs_u, L.s_u[tid]);
            L.si[tid] = mf20_lookup(L.hk_i, L.hs_i, L.s_i[tid]);
        }
        const int nu = L.cnt[0], ni = L.cnt[1];
        for (int x = tid; x < nu * F; x += MF20_THREADS) L.P[x] = st.P[(int64_t)L.key_u[x / F] * F + x % F];
        for (int x = tid; x < ni * F; x += MF20_THREADS) L.Q[x] = st.Q[(int64_t)L.key_i[x / F] * F + x % F];
        for (int x = tid; x < nu; x += MF20_THREADS) L.bu[x] = st.bu[L.key_u[x]];
        for (int x = tid; x < ni; x += MF20_THREADS) L.bi[x] = st.bi[L.key_i[x]];
        __syncthreads();
        // ---- phase B: the chunk's updates, in order, one wave
        if (tid < 64) {
            int su_n = L.su[0], si_n = L.si[0];
            float y_n = L.s_y[0];
            for (int s = 0; s < cs; ++s) {
                const int su = su_n, si = si_n;
                const double y = (double)y_n;
                if (s + 1 < cs) su_n = L.su[s + 1], si_n = L.si[s + 1], y_n = L.s_y[s + 1];      // (not on the chain: fetched a sample ahead)
                double dot = 0.0;
                for (int f = lane; f < F; f += 64) dot += L.P[su * F + f] * L.Q[si * F + f];
                dot = mf20_wave_sum(dot);
                const double ub = L.bu[su], ib = L.bi[si];
                const double pred = gb + ub + ib + dot;                       // :89
                if (lane == 0) L.pred[s] = pred;
                double sig;
                if (pred > 0) sig = 1.0 / (1.0 + exp(-pred));                 // :92-94
                else {                                                        // :97-98
                    const double ep = exp(pred);
                    sig = ep / (1.0 + ep);
                }
                const double grad = y - sig;                                  // :101
                for (int f = lane; f < F; f += 64) {
                    const double uf = L.P[su * F + f], itf = L.Q[si * F + f];
                    const double ufn = uf + lr * (grad * itf - reg * uf);     // :103
                    L.P[su * F + f] = ufn;
                    L.Q[si * F + f] = itf + lr * (grad * ufn - reg * itf);    // :104 (the updated user row: a view)
                }
                if (lane == 0) {
                    L.bu[su] = ub + lr * (grad - reg * ub);                   // :105
                    L.bi[si] = ib + lr * (grad - reg * ib);                   // :106
                }
                gb = gb + lr * (grad - reg * gb);                             // :107
                el_wave_lds_sync();                                           // the next sample reads what this one wrote
            }
        }
        __syncthreads();
        // ---- phase C: the chunk's loss terms (:94-99), every thread one sample; added in sample order below (:108)
        if (tid < cs) {
            const double pred = L.pred[tid], y = (double)L.s_y[tid];
            double this_loss;
            if (pred > 0) this_loss = log(1.0 + exp(-pred)) + (1.0 - y) * pred;
            else this_loss = -y * pred + log(1.0 + exp(pred));
            L.lterm[tid] = this_loss;
        }
        // ---- slots back to the tables
        for (int x = tid; x < nu * F; x += MF20_THREADS) st.P[(int64_t)L.key_u[x / F] * F + x % F] = L.P[x];
        for (int x = tid; x < ni * F; x += MF20_THREADS) st.Q[(int64_t)L.key_i[x / F] * F + x % F] = L.Q[x];
        for (int x = tid; x < nu; x += MF20_THREADS) st.bu[L.key_u[x]] = L.bu[x];
        for (int x = tid; x < ni; x += MF20_THREADS) st.bi[L.key_i[x]] = L.bi[x];
        __syncthreads();
        if (tid == 0)
            for (int s2 = 0; s2 < cs; ++s2) loss += L.lterm[s2];
        __syncthreads();
    }
    if (tid == 0) {
        *st.gb = gb;
        if (loss_out) *loss_out += loss;
    }
}

static size_t mf20_lds_bytes(int C, int F) {
    return (size_t)2 * C * F * 8 + (size_t)4 * C * 8 + (size_t)7 * C * 4 + (size_t)4 * MF20_HASH * 4 + 16;
}

// Replaces: MFModel.train_step (MF2020/MF_model.py:80-113) on one batch of (user, item, rating) rows, in order.
//   samples: int32 [n, 3] device (the rows custom_sampler_rendle.Sampler.step yields, :84-85); *loss_out (device double) += sum of losses
extern "C" int el_mf2020_train(el_ctx* ctx, void* stream, const el_mf2020_state* stp, const int32_t* samples, int64_t n, double* loss_out) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(stp && stp->P && stp->Q && stp->bu && stp->bi && stp->gb, "el_mf2020_train: null state");
    EL_REQUIRE(stp->F >= 1 && stp->F <= 512 && stp->U >= 1 && stp->I >= 1, "el_mf2020_train: bad shape (F=%d)", stp->F);
    EL_REQUIRE(samples != nullptr || n == 0, "el_mf2020_train: null samples");
    if (n <= 0) return 0;
    int C = 128;                                              // samples per chunk: as many as 128 KB of LDS rows allow
    while (C > 8 && mf20_lds_bytes(C, stp->F) > (size_t)(128 << 10)) C >>= 1;
    const size_t lds = mf20_lds_bytes(C, stp->F);
    EL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mf2020_seq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    EL_LAUNCH("k_mf2020_seq", k_mf2020_seq, dim3(1), dim3(MF20_THREADS), lds, (hipStream_t)stream, *stp, samples, n, C, loss_out);
    EL_CHECK_LAUNCH();
    return 0;
}
