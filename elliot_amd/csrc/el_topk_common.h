// Shared pieces of the top-k kernels (el_topk.hip, el_topk_screen.hip).
#pragma once
#include "el_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct TopkParams {
    const float* Gu;
    const float* Gi;
    const float* Bi;
    int64_t u_start, u_stop, item_offset, I_local;
    int F;
    const int64_t* excl_indptr;
    const int32_t* excl_indices;
    const int64_t* cand_indptr;
    const int32_t* cand_indices;
    int k;
    int32_t* out_idx;
    float* out_val;
    // dense-preds variant
    const float* preds;
    int64_t ld;
    int dbg;  // experiment switches (EL_TOPK_DEBUG): 1 = skip the fused selection (GEMM-only timing)
    const int32_t* ulist;         // MFMA kernel: if set, process users u_start + ulist[skip .. min(*ulist_n, max)) instead of the range
    const int32_t* ulist_n;
    int ulist_skip, ulist_max;    // (ulist_max == 0: no upper limit)
    // MFMA kernel, item-split mode (gridDim.y = nsplit > 1): split s scores its slice of the item tiles and writes the
    // partial list of list entry e to row s * part_stride + e of out_idx / out_val (merged by k_topk_merge afterwards)
    int nsplit;
    int64_t part_stride;
};

// defined in el_topk.hip: fp32 MFMA kernel (must be eligible: F <= 256, k <= 40, no candidate list); honours p.ulist
int el_topk_launch_mfma(const TopkParams& p, hipStream_t st);
// defined in el_topk.hip: exact top-k of the users in p.ulist (device list, *p.ulist_n entries), parallel over users AND
// item slices so that a handful of users does not serialise on one workgroup.  scratch: el_topk_list_scratch_bytes().
size_t el_topk_list_scratch_bytes(int64_t n_users, int64_t I_local, int k);
int el_topk_run_list(const TopkParams& p, void* scratch, size_t scratch_bytes, hipStream_t st);

// ---- one-wave bitonic sort (descending) of n = 2^m u64 keys held in LDS --------------
__device__ __forceinline__ void el_wave_bitonic_desc(u64* a, int n, int lane) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                int i = 2 * t - (t & (stride - 1));
                int j = i + stride;
                bool desc = ((i & size) == 0);
                u64 x = a[i], y = a[j];
                bool sw = desc ? (x < y) : (x > y);
                if (sw) {
                    a[i] = y;
                    a[j] = x;
                }
            }
            el_wave_lds_sync();
        }
    }
}

// Sort one candidate list (n valid keys in a cap-slot LDS buffer), keep the best k.
// Whole wave participates, all arguments wave-uniform. Returns the new threshold.
__device__ __forceinline__ float el_wave_compact(u64* kb, int* cp, int cap, int k, int lane) {
    el_wave_lds_sync();
    int n = *cp;
    for (int t = n + lane; t < cap; t += 64) kb[t] = 0ull;
    el_wave_lds_sync();
    el_wave_bitonic_desc(kb, cap, lane);
    int nn = n < k ? n : k;
    if (lane == 0) *cp = nn;
    float nt = (nn >= k) ? el_key_score(kb[k - 1]) : -INFINITY;
    el_wave_lds_sync();
    return nt;
}

// Same, but first drops keys whose item is in the (sorted) exclusion row idx[e0,e1): the MFMA kernel
// inserts candidates unchecked and pays the membership test (a chain of dependent global loads) once
// per compaction for the whole buffer instead of once per insertion.  cap <= 64 here.
__device__ __forceinline__ float el_wave_compact_excl(u64* kb, int n, int& n_out, int cap, int k, int lane,
                                                      const int32_t* __restrict__ idx, int64_t e0, int64_t e1) {
    el_wave_lds_sync();
    bool drop = false;
    if (lane < cap) {
        if (lane < n) {
            if (e1 > e0) drop = el_row_contains(idx, e0, e1, el_key_item(kb[lane]));
            if (drop) kb[lane] = 0ull;
        } else {
            kb[lane] = 0ull;
        }
    }
    const int removed = __popcll(__ballot(drop));
    el_wave_lds_sync();
    el_wave_bitonic_desc(kb, cap, lane);
    n -= removed;
    int nn = n < k ? n : k;
    n_out = nn;
    float nt = (nn >= k) ? el_key_score(kb[k - 1]) : -INFINITY;
    el_wave_lds_sync();
    return nt;
}

// value held by the partner lane (l <-> l+32) -- v_permlane32_swap_b32, no LDS round trip
__device__ __forceinline__ u32 el_partner32(u32 x, int hi) {
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return hi ? r[0] : r[1];
}

// The r-th (0-based) masked item of a row, ascending, inside the local shard
// [off, off+I_local): what tf.where(mask, preds, -inf) + top_k pads with.
__device__ __forceinline__ int32_t el_fill_masked_range(const TopkParams& p, int64_t off, int64_t end, int64_t e0, int64_t e1,
                                                        int64_t c0, int64_t c1, int64_t r) {
    if (p.cand_indptr) {
        // masked = NOT candidate.  q* = #candidates (inside the range) with
        // cand[q] - off - q <= r ; answer = off + r + q*.
        int64_t lo = el_lower_bound(p.cand_indices, c0, c1, (int32_t)off);
        int64_t hi = el_lower_bound(p.cand_indices, c0, c1, (int32_t)(end > 0x7fffffffLL ? 0x7fffffffLL : end));
        int64_t a = lo, b = hi;
        while (a < b) {
            int64_t mid = (a + b) >> 1;
            int64_t f = (int64_t)p.cand_indices[mid] - off - (mid - lo);
            if (f <= r)
                a = mid + 1;
            else
                b = mid;
        }
        int64_t g = off + r + (a - lo);
        return g < end ? (int32_t)g : -1;
    }
    if (p.excl_indptr) {
        int64_t lo = el_lower_bound(p.excl_indices, e0, e1, (int32_t)off);
        int64_t hi = el_lower_bound(p.excl_indices, e0, e1, (int32_t)(end > 0x7fffffffLL ? 0x7fffffffLL : end));
        return (lo + r < hi) ? p.excl_indices[lo + r] : -1;
    }
    return -1;
}

__device__ __forceinline__ int32_t el_fill_masked(const TopkParams& p, int64_t e0, int64_t e1, int64_t c0,
                                                  int64_t c1, int64_t r) {
    return el_fill_masked_range(p, p.item_offset, p.item_offset + p.I_local, e0, e1, c0, c1, r);
}
