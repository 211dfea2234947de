// Rows of a sorted-segment gradient pass that were cut by chunk boundaries, finished WITHOUT floating-point atomics (round 6; the
// scheme of el_bpr_sorted.hip's k_bpr_item_combine, generic over what "finish" means: el_pwmf.hip stores the row to its gradient table).
//
// The segment pass gives every lane group a fixed chunk of sorted positions.  A segment that lies inside one chunk is finished there.
// A piece of a segment that is cut goes to a partial slot instead: slot 2 g + 1 for the piece that STARTS inside group g's chunk (the
// head piece: it also appends (row, g) to the split list), slot 2 g for a piece that started before the chunk (at most one of each per
// group).  The pieces of a listed row are then the head's and one per following lane group whose chunk still begins inside the segment
// (the sorted key at that chunk's first position is still this row), and they are added here in a FIXED order:
//   k_seg_combine       one lane group per listed row: lane k looks at group g0 + 1 + k; up to lpt - 1 continuations are summed in
//                       ascending order, longer segments go on the long list
//   k_seg_combine_long  one workgroup per row of the long list: each lane group adds a contiguous share in ascending order, eight loads
//                       in flight, the shares are added in lane-group order
// The same bits on every run.  FIN: a trivially copyable functor, fin(row, sub, lpt, g[CPL][VW], gb) called by the lanes of ONE lane
// group (lane `sub` holds the elements (sub + q lpt) VW + x of the row; gb = the bias part, used by sub == 0).
#pragma once
#include "el_common.h"

struct SegParts {
    const u32* keys;        // sorted keys of the pass
    u32 key_off;            // key - key_off = row
    int64_t n;              // sorted positions
    int chunk, lpt, F;
    int32_t* split;         // [1 + 2 (groups + 1)]: count, then (row, head lane group) pairs; NULL: k_seg_combine finds the heads itself
    int32_t* split_long;    // the same, filled by k_seg_combine
    float* part;            // [2 groups, F]
    float* part_b;          // [2 groups]
};

template <int VW>
__device__ __forceinline__ void sc_ld(const float* __restrict__ p, float (&v)[VW]) {
    if (VW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x, v[1 % VW] = t.y, v[2 % VW] = t.z, v[3 % VW] = t.w;
    } else {
#pragma unroll
        for (int x = 0; x < VW; ++x) v[x] = p[x];
    }
}
template <int VW>
__device__ __forceinline__ void sc_st(float* __restrict__ p, const float (&v)[VW]) {
    if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % VW], v[2 % VW], v[3 % VW]);
    else {
#pragma unroll
        for (int x = 0; x < VW; ++x) p[x] = v[x];
    }
}

template <int VW, int CPL, typename FIN>
__global__ __launch_bounds__(256) void k_seg_combine(SegParts sp, FIN fin) {
    const int F = sp.F, lpt = sp.lpt;
    const int64_t ent = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpt;
    const int sub = (int)(threadIdx.x & (lpt - 1));
    int64_t row, g0;
    if (sp.split) {                                             // listed heads
        if (ent >= (int64_t)sp.split[0]) return;
        row = (int64_t)sp.split[1 + 2 * ent];
        g0 = (int64_t)sp.split[2 + 2 * ent];
    } else {
        // no list (a contended counter costs more than it saves where chunks are a few positions long): lane group g looks at ITS chunk --
        // the last segment of chunk g is a cut head when it goes on into chunk g + 1 and started inside chunk g
        g0 = ent;
        const int64_t p0 = g0 * (int64_t)sp.chunk, p1 = p0 + sp.chunk;
        if (p1 >= sp.n) return;                                 // (the last chunk's last segment ends with the data)
        const u32 kl = sp.keys[p1 - 1];
        if (sp.keys[p1] != kl) return;                          // ends inside
        if (sp.keys[p0] == kl && p0 > 0 && sp.keys[p0 - 1] == kl) return;      // a middle piece: spans the whole chunk, started before it
        row = (int64_t)(kl - sp.key_off);
    }
    const int64_t pos = (g0 + 1 + sub) * (int64_t)sp.chunk;
    const bool cont = pos < sp.n && (int64_t)(sp.keys[pos] - sp.key_off) == row;
    const unsigned long long bal = __ballot(cont);
    const int lane0 = (int)(threadIdx.x & 63) & ~(lpt - 1);
    const unsigned long long mine = lpt >= 64 ? bal : ((bal >> lane0) & ((1ull << lpt) - 1ull));
    const unsigned long long full = lpt >= 64 ? ~0ull : ((1ull << lpt) - 1ull);
    if (mine == full) {                                         // lpt or more continuations: the workgroup kernel's
        if (sub == 0) {
            const int e2 = atomicAdd(sp.split_long, 1);
            sp.split_long[1 + 2 * e2] = (int32_t)row;
            sp.split_long[2 + 2 * e2] = (int32_t)g0;
        }
        return;
    }
    const int ncont = __builtin_ctzll(~mine);                   // consecutive continuations
    float acc[CPL][VW];
    float accb = sp.part_b[2 * g0 + 1];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int e = (sub + q * lpt) * VW;
#pragma unroll
        for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
        if (e < F) sc_ld<VW>(sp.part + (2 * g0 + 1) * F + e, acc[q]);
    }
    for (int k = 1; k <= ncont; k += 4) {
        float v[4][CPL][VW], vb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t sl = 2 * (g0 + (k + t <= ncont ? k + t : k));
            vb[t] = sp.part_b[sl];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
#pragma unroll
                for (int x = 0; x < VW; ++x) v[t][q][x] = 0.f;
                if (e < F) sc_ld<VW>(sp.part + sl * F + e, v[t][q]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (k + t > ncont) continue;
            accb += vb[t];
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int x = 0; x < VW; ++x) acc[q][x] += v[t][q][x];
        }
    }
    fin(row, sub, lpt, acc, accb);
}

template <int VW, int CPL, typename FIN>
__global__ __launch_bounds__(256) void k_seg_combine_long(SegParts sp, FIN fin) {
    const int F = sp.F, lpt = sp.lpt;
    const int nlist = sp.split_long[0];
    __shared__ int s_np;
    __shared__ __attribute__((aligned(16))) float s_red[4096];   // (256 / lpt) lane groups x F floats: lpt * VW * CPL >= F, CPL <= 4 -> <= 4096
    __shared__ float s_rb[32];
    const int ngl = 256 / lpt, gl = threadIdx.x / lpt, sub = threadIdx.x & (lpt - 1);
    for (int ent = blockIdx.x; ent < nlist; ent += gridDim.x) {
        const int64_t row = (int64_t)sp.split_long[1 + 2 * ent];
        const int64_t g0 = (int64_t)sp.split_long[2 + 2 * ent];
        if (threadIdx.x == 0) s_np = 0x7fffffff;
        __syncthreads();
        for (int base = 0; s_np == 0x7fffffff; base += 256) {
            const int64_t gq = g0 + 1 + base + threadIdx.x, pos = gq * (int64_t)sp.chunk;
            const bool cont = pos < sp.n && (int64_t)(sp.keys[pos] - sp.key_off) == row;
            if (!cont) atomicMin(&s_np, base + (int)threadIdx.x);
            __syncthreads();
        }
        const int ncont = s_np;                                // continuation partials: slots 2 (g0 + k), k = 1 .. ncont
        const int np = 1 + ncont;                              // + the head's
        const int per = (np + ngl - 1) / ngl;
        const int k0 = gl * per, k1 = (k0 + per < np) ? k0 + per : np;
        float acc[CPL][VW];
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int x = 0; x < VW; ++x) acc[q][x] = 0.f;
        float accb = 0.f;
        auto slot_of = [&](int k) { return k == 0 ? 2 * g0 + 1 : 2 * (g0 + k); };
        for (int k = k0; k < k1; k += 8) {
            float v[8][CPL][VW];
            float vb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int kk = k + t < k1 ? k + t : k;
                const int64_t sl = slot_of(kk);
                vb[t] = sp.part_b[sl];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int e = (sub + q * lpt) * VW;
#pragma unroll
                    for (int x = 0; x < VW; ++x) v[t][q][x] = 0.f;
                    if (e < F) sc_ld<VW>(sp.part + sl * F + e, v[t][q]);
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (k + t >= k1) continue;
                accb += vb[t];
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int x = 0; x < VW; ++x) acc[q][x] += v[t][q][x];
            }
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int e = (sub + q * lpt) * VW;
            if (e < F) sc_st<VW>(s_red + gl * F + e, acc[q]);
        }
        if (sub == 0) s_rb[gl] = accb;
        __syncthreads();
        if (gl == 0) {
            const int used = (np + per - 1) / per;              // lane groups that held partials
            float gb = 0.f;
            for (int h = 0; h < used; ++h) gb += s_rb[h];
            float gg[CPL][VW];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int e = (sub + q * lpt) * VW;
#pragma unroll
                for (int x = 0; x < VW; ++x) gg[q][x] = 0.f;
                if (e >= F) continue;
                for (int h = 0; h < used; ++h) {
                    float t4[VW];
                    sc_ld<VW>(s_red + h * F + e, t4);
#pragma unroll
                    for (int x = 0; x < VW; ++x) gg[q][x] += t4[x];
                }
            }
            fin(row, sub, lpt, gg, gb);
        }
        __syncthreads();
    }
}
