// Merge of the attention's key-split partials for one (query row, 4-channel chunk): the arithmetic of
// attention_combine_kernel (attention.hip), used by the fused layer kernels so that the merge costs no launch and
// msg makes no round trip through HBM.  All loads are issued before any is used (the split count is a run-time value:
// a plain loop over it serialises 2 * nsplit dependent HBM round trips per chunk).
#pragma once
#include "pdsc_common.h"

namespace pdsc {

constexpr int MERGE_MAX_SPLIT = 4;        // larger key splits go through attention_combine_kernel ...
constexpr int MERGE_MAX_SPLIT_BLOCK = 8;  // ... except in the workgroup-per-tile layer kernel (layer.hip: small problems)
constexpr int MERGE_MAX_SPLIT_H3 = 8;     // ... and in layer_h3.hip (r03: the per-GPU shares of the 8-GPU configurations -- 1-3 pairs of
                                          //     N = 5000 / 10000 -- are planned with 5-8 key splits: no combine launch, no msg round trip)

__device__ __forceinline__ f32x4 merge_partials_chunk(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                      size_t slot0, size_t sp_stride, int ns, int c4) {
    float mx[MERGE_MAX_SPLIT], ls[MERGE_MAX_SPLIT];
    f32x4 pv[MERGE_MAX_SPLIT];
#pragma unroll
    for (int sp = 0; sp < MERGE_MAX_SPLIT; ++sp) {
        const size_t slot = slot0 + (size_t)min(sp, ns - 1) * sp_stride;     // surplus slots repeat the last split
        const float2 ml = *reinterpret_cast<const float2*>(part_ml + slot * 2);
        mx[sp] = ml.x; ls[sp] = ml.y;
        pv[sp] = *reinterpret_cast<const f32x4*>(part_o + slot * PDSC_CHANNELS + c4);
    }
    float mmax = mx[0];
#pragma unroll
    for (int sp = 1; sp < MERGE_MAX_SPLIT; ++sp) mmax = fmaxf(mmax, mx[sp]);   // repeats do not change the maximum
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < MERGE_MAX_SPLIT; ++sp) {
        if (sp < ns) {                                                       // wave-uniform
            const float w = __builtin_amdgcn_exp2f(mx[sp] - mmax);
            L = fmaf(ls[sp], w, L);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(pv[sp][e], w, acc[e]);
        }
    }
    const float r = 1.0f / L;              // one correctly-rounded reciprocal per row, then multiplies (all merge sites agree)
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[e] * r;
    return v;
}


// Two-phase form for callers that merge several chunks per thread: issue the loads of ALL chunks first (one round trip
// of latency instead of one per chunk), then do the arithmetic.  NS = compile-time split count (registers for exactly
// NS splits, no duplicate loads).
template <int NS>
struct MergeLoads {
    float mx[NS], ls[NS];
    f32x4 pv[NS];
};

template <int NS>
__device__ __forceinline__ void merge_partials_load(MergeLoads<NS>& L, const float* __restrict__ part_o,
                                                    const float* __restrict__ part_ml, size_t slot0, size_t sp_stride, int c4) {
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
        const size_t slot = slot0 + (size_t)sp * sp_stride;
        const float2 ml = *reinterpret_cast<const float2*>(part_ml + slot * 2);
        L.mx[sp] = ml.x; L.ls[sp] = ml.y;
        L.pv[sp] = *reinterpret_cast<const f32x4*>(part_o + slot * PDSC_CHANNELS + c4);
    }
}

template <int NS>
__device__ __forceinline__ f32x4 merge_partials_finish(const MergeLoads<NS>& L) {
    float mmax = L.mx[0];
#pragma unroll
    for (int sp = 1; sp < NS; ++sp) mmax = fmaxf(mmax, L.mx[sp]);
    float den = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
        const float w = __builtin_amdgcn_exp2f(L.mx[sp] - mmax);
        den = fmaf(L.ls[sp], w, den);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(L.pv[sp][e], w, acc[e]);
    }
    const float r = 1.0f / den;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[e] * r;
    return v;
}

}  // namespace pdsc
