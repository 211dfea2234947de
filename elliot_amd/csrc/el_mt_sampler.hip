// Exact replay of the reference's BPR triplet stream on the device.
//
// Replaces dataset/samplers/custom_sampler.py:14-46 bit for bit: the global legacy NumPy RandomState (MT19937,
// np.random.seed(42) at :15), np.random.randint's masked rejection over successive 32-bit outputs (0 draws when the
// range has a single value), draw order u -> position of i in list(set(pos(u))) -> j repeated while j in pos(u).
//
// MT19937 is sequential and every sample consumes a data-dependent number of words, so the work is split:
//   1. k_mt_generate   one workgroup regenerates W tempered words from the 624-word state (the twist of one block is
//                      three dependent parallel phases: k<227 uses old words only, then two phases on fresh words)
//   2. k_mt_candidates for EVERY word position p in parallel: "if a sample started here, which triplet comes out and
//                      where does the next sample start?"  (speculative; only ~1/4 of the positions are real starts)
//   3. k_mt_chase      one lane follows start -> next(start) n times and copies the n real triplets out
//   4. k_mt_generate   (state-only) replays the generator up to the last consumed word and stores the new state
// Pointer chasing is n dependent L2 reads (~0.1 us each); everything else is parallel.
#include "el_common.h"

#define MT_N 624
#define MT_M 397

__device__ __forceinline__ u32 mt_temper(u32 y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ u32 mt_mix(u32 cur, u32 nxt, u32 far) {
    const u32 y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// state[0..623] words, state[624] = pos (624 means "block exhausted").  Emits `nwords` outputs starting at the
// current position into out (if out != NULL) and stores the state reached after `advance` outputs.
__global__ __launch_bounds__(256) void k_mt_generate(u32* __restrict__ state, u32* __restrict__ out, int64_t nwords,
                                                     int64_t advance, const u32* __restrict__ init_state) {
    __shared__ u32 mt[MT_N];
    __shared__ u32 snap[MT_N];
    const int tid = threadIdx.x;
    const u32* src = init_state ? init_state : state;
    for (int k = tid; k < MT_N; k += 256) mt[k] = src[k];
    __syncthreads();
    const int64_t pos0 = (int64_t)src[MT_N];
    const int64_t total = nwords > advance ? nwords : advance;
    int64_t emitted = 0;
    // the state to save is the block containing output number `advance` (0-based index of the NEXT word to draw)
    // next-word position inside its block: (pos0 + advance) relative to block boundaries
    int64_t cur_pos = pos0;       // position in the current block
    bool saved = false;
    int64_t save_pos = 0;
    while (true) {
        // words available in this block: [cur_pos, 624)
        const int64_t avail = MT_N - cur_pos;
        // does the save point fall in this block?  (advance == emitted + d with 0 <= d <= avail; d == avail only when
        // no further block is needed -> pos 624 is a legal stored position, numpy stores it that way too)
        if (!saved && advance - emitted <= avail) {
            for (int k = tid; k < MT_N; k += 256) snap[k] = mt[k];
            save_pos = cur_pos + (advance - emitted);
            saved = true;
        }
        if (out) {
            for (int64_t k = tid; k < avail; k += 256)
                if (emitted + k < nwords) out[emitted + k] = mt_temper(mt[cur_pos + k]);
        }
        emitted += avail;
        if (emitted >= total) break;
        __syncthreads();
        // twist: three dependent phases + the last word.  Inside a phase every new word needs the OLD value of its
        // right neighbour, so all reads complete before any write (each thread owns at most one word per phase).
        {
            const int k = tid;
            u32 v = 0;
            if (k < MT_N - MT_M) v = mt_mix(mt[k], mt[k + 1], mt[k + MT_M]);
            __syncthreads();
            if (k < MT_N - MT_M) mt[k] = v;
            __syncthreads();
        }
        {
            const int k = MT_N - MT_M + tid;
            u32 v = 0;
            if (k < 2 * (MT_N - MT_M)) v = mt_mix(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            __syncthreads();
            if (k < 2 * (MT_N - MT_M)) mt[k] = v;
            __syncthreads();
        }
        {
            const int k = 2 * (MT_N - MT_M) + tid;
            u32 v = 0;
            if (k < MT_N - 1) v = mt_mix(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            __syncthreads();
            if (k < MT_N - 1) mt[k] = v;
            __syncthreads();
        }
        if (tid == 0) mt[MT_N - 1] = mt_mix(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        __syncthreads();
        cur_pos = 0;
    }
    __syncthreads();
    if (saved) {
        for (int k = tid; k < MT_N; k += 256) state[k] = snap[k];
        if (tid == 0) state[MT_N] = (u32)save_pos;
    }
}

struct MtCursor {
    const u32* w;
    int64_t p, end;
    bool ok;
    __device__ __forceinline__ u32 next() {
        if (p >= end) {
            ok = false;
            return 0u;
        }
        return w[p++];
    }
    // np.random.randint(n): no draw when n == 1, else masked rejection
    __device__ __forceinline__ u32 randint(u32 n) {
        const u32 rng = n - 1u;
        if (rng == 0u) return 0u;
        u32 m = rng;
        m |= m >> 1;
        m |= m >> 2;
        m |= m >> 4;
        m |= m >> 8;
        m |= m >> 16;
        u32 v;
        do {
            v = next() & m;
        } while (ok && v > rng);
        return v;
    }
};

// cand[p] = {u, i, j, next start} of the sample that would start at word p (next = -1: ran out of words)
__global__ __launch_bounds__(256) void k_mt_candidates(const u32* __restrict__ words, int64_t W,
                                                       const int64_t* __restrict__ lists_indptr,
                                                       const int32_t* __restrict__ lists_items,
                                                       const int64_t* __restrict__ pos_indptr,
                                                       const int32_t* __restrict__ pos_indices, u32 U, u32 I,
                                                       int4* __restrict__ cand) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W) return;
    MtCursor c;
    c.w = words;
    c.p = p;
    c.end = W;
    c.ok = true;
    const u32 u = c.randint(U);                                 // custom_sampler.py:32
    int4 r = make_int4(0, 0, 0, -1);
    if (c.ok) {
        const int64_t l0 = lists_indptr[u], l1 = lists_indptr[u + 1];
        const u32 lui = (u32)(l1 - l0);
        const u32 ip = lui ? c.randint(lui) : 0u;               // :37
        if (c.ok && lui) {
            const int32_t it = lists_items[l0 + ip];
            const int64_t r0 = pos_indptr[u], r1 = pos_indptr[u + 1];
            u32 j = c.randint(I);                               // :39
            while (c.ok && el_row_contains(pos_indices, r0, r1, (int32_t)j)) j = c.randint(I);   // :40-41
            if (c.ok) r = make_int4((int)u, it, (int)j, (int)(c.p - p));
        }
    }
    cand[p] = r;
}

// follows the chain of real sample starts; result[0] = words consumed, result[1] = samples produced
__global__ void k_mt_chase(const int4* __restrict__ cand, int64_t W, int64_t n, int32_t* out_u, int32_t* out_i,
                           int32_t* out_j, int64_t* result) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t p = 0, t = 0;
    for (; t < n && p < W; ++t) {
        const int4 c = cand[p];
        if (c.w < 0) break;
        out_u[t] = c.x;
        out_i[t] = c.y;
        out_j[t] = c.z;
        p += c.w;
    }
    result[0] = p;
    result[1] = t;
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static int64_t mt_words_for(int64_t n) { return 16 * n + 8 * MT_N; }

extern "C" size_t el_bpr_sample_mt19937_ws_bytes(int64_t n) {
    if (n <= 0) return 0;
    const int64_t W = mt_words_for(n);
    return a256((size_t)W * 4) + a256((size_t)W * 16) + a256((MT_N + 1) * 4) + a256(16);
}

extern "C" int el_bpr_sample_mt19937(el_ctx* ctx, void* stream, uint32_t* mt_state, const int64_t* lists_indptr,
                                     const int32_t* lists_items, const int64_t* pos_indptr, const int32_t* pos_indices,
                                     int64_t U, int64_t I, int64_t n, int32_t* out_u, int32_t* out_i, int32_t* out_j,
                                     void* ws, size_t ws_bytes) {
    if (int rc = el_bind(ctx)) return rc;
    EL_REQUIRE(mt_state && lists_indptr && lists_items && pos_indptr && pos_indices && out_u && out_i && out_j,
               "el_bpr_sample_mt19937: null pointer");
    EL_REQUIRE(U >= 1 && U < 0x7fffffffLL && I >= 2 && I < 0x7fffffffLL, "el_bpr_sample_mt19937: U/I out of range");
    if (n <= 0) return 0;
    EL_REQUIRE(n <= (1LL << 26), "el_bpr_sample_mt19937: at most 2^26 samples per call");
    const int64_t W = mt_words_for(n);
    EL_REQUIRE(ws != nullptr && ws_bytes >= el_bpr_sample_mt19937_ws_bytes(n), "el_bpr_sample_mt19937: workspace too small");
    char* base = (char*)ws;
    u32* words = (u32*)base;
    int4* cand = (int4*)(base + a256((size_t)W * 4));
    u32* backup = (u32*)(base + a256((size_t)W * 4) + a256((size_t)W * 16));
    int64_t* result = (int64_t*)((char*)backup + a256((MT_N + 1) * 4));
    hipStream_t s = (hipStream_t)stream;
    EL_CHECK_HIP(hipMemcpyAsync(backup, mt_state, (MT_N + 1) * 4, hipMemcpyDeviceToDevice, s));
    EL_LAUNCH("k_mt_generate", k_mt_generate, dim3(1), dim3(256), 0, s, mt_state, words, W, (int64_t)0, (const u32*)backup);
    EL_LAUNCH("k_mt_candidates", k_mt_candidates, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, s, (const u32*)words, W,
              lists_indptr, lists_items, pos_indptr, pos_indices, (u32)U, (u32)I, cand);
    EL_LAUNCH("k_mt_chase", k_mt_chase, dim3(1), dim3(64), 0, s, (const int4*)cand, W, n, out_u, out_i, out_j, result);
    EL_CHECK_LAUNCH();
    // the number of consumed words decides the new generator state: one small synchronising read-back
    int64_t host_result[2] = {0, 0};
    EL_CHECK_HIP(hipMemcpyAsync(host_result, result, 16, hipMemcpyDeviceToHost, s));
    EL_CHECK_HIP(hipStreamSynchronize(s));
    EL_REQUIRE(host_result[1] == n, "el_bpr_sample_mt19937: ran out of generated words after %lld of %lld samples "
               "(rejection rate far above the provisioned 16 words/sample)", (long long)host_result[1], (long long)n);
    EL_LAUNCH("k_mt_generate", k_mt_generate, dim3(1), dim3(256), 0, s, mt_state, (u32*)nullptr, (int64_t)0, host_result[0],
              (const u32*)backup);
    EL_CHECK_LAUNCH();
    return 0;
}
