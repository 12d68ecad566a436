// Device code of the hypothesis-scoring kernel (score.hip; score_slp.hip compiles the same source with SLP vectorisation on).
#pragma once
#include "pdsc_common.h"

namespace pdsc {

constexpr int SC_SEEDS = 4;       // hypotheses per workgroup (all N points of the pair: no partial counts, no atomics)

// residual of one correspondence under one transform, in the reference's rounding order:
// pred = R p + t as a length-3 dot (fma chain) plus t, then torch.norm's fma chain.
__device__ __forceinline__ float residual(const float* __restrict__ T, float px, float py, float pz, float qx, float qy,
                                          float qz) {
    const float x = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
    const float y = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
    const float z = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
    return norm3(x - qx, y - qy, z - qz);
}

// the same residual without the final square root: torch.norm's radicand fma(dz,dz,fma(dy,dy,dx*dx))
__device__ __forceinline__ float residual_sq(const float* __restrict__ T, float px, float py, float pz, float qx, float qy,
                                             float qz) {
    const float x = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
    const float y = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
    const float z = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
    const float dx = x - qx, dy = y - qy, dz = z - qz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// `thr2` = sqrt_threshold_radicand(thr): `residual < thr` and `residual_sq < thr2` are the same predicate bit for bit, so the
// counts are those of the reference's `L2 < thr` without S*N correctly rounded square roots (half of this kernel's VALU work).
// One workgroup counts its SC_SEEDS hypotheses over ALL points of the pair and stores the totals.  r03 rebuilt this kernel while
// chasing forwards that depended on what else ran on the chip (tools/inflight_diverge_probe.py, profiles/r03_*_probe.txt; the
// account is in pointdsc_amd/pipeline.py):
//   * the counters used to be zeroed by hipMemsetAsync and filled by atomicAdds of per-workgroup partial counts.  With several
//     forwards in flight the memset was not reliably ordered before the atomics that followed it on the same stream (counters
//     far off on 16-35 % of replayed hipGraph forwards, and now and then on eager multi-stream forwards): no memset, no atomics,
//     no partial counts any more (launch_fill_u32 replaces the library's other hipMemsetAsync calls);
//   * score.hip IS COMPILED WITH -fno-slp-vectorize (pointdsc_amd/build.py).  The SLP vectoriser pairs the residual tests of
//     two seeds into packed fp32 instructions (v_pk_mul/fma/add_f32 with op_sel broadcasts of the just-loaded point); in that
//     form one half of the seed pairs came out a few votes short on 0.2-0.7 % of the forwards whenever kernels of other forwards
//     were co-resident -- transforms and points verified equal, a recount later in the same launch right, the affected half
//     moving with the instruction schedule (LDS staging, atomics, carry-chain vs ballot counting made no difference).  Scalar
//     fp32 code: 0 differing forwards in 8000-12000 per mode.  Cause below the ISA level not established;
//   * votes are counted per wavefront with ballot + popcount, the transforms sit in registers (no LDS staging, no barrier
//     before the loop).
// DBG (experiments builds, diagnostics): dbg[(b*S+s)*16 + 0..11] = the transform as this kernel read it, [12] = its count,
// [13] = the same count over points re-read with system-scope (cache-bypassing) loads, [14] = a third count, ordinary loads again
// VARIANT only names the instantiation: 0 = score.hip (built without SLP vectorisation), 1 = score_slp.hip (experiments builds: the
// same source with the compiler's default packed-fp32 pairing, kept as the reproducer of the miscount)
template <int DBG, int VARIANT = 0>
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ seed_trans, const float* __restrict__ src,
                                                    const float* __restrict__ tgt, float thr2, int* __restrict__ counts,
                                                    int NS, int S, const int* __restrict__ nvalid, float* __restrict__ dbg) {
    __shared__ int wsum[4][SC_SEEDS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s0 = blockIdx.x * SC_SEEDS, b = blockIdx.y;
    const int N = nvalid ? nvalid[b] : NS;        // ragged batches: only the pair's own correspondences vote (ragged.h)
    // the workgroup's transforms, read by every lane with ordinary vector loads (the offset is laundered through a VGPR so that the
    // compiler does not turn the uniform address into scalar loads: the scalar cache is one more cache to keep coherent with the
    // kernel that wrote seed_trans).  No LDS staging, no barrier before the loop.
    int vz = 0;
    asm volatile("" : "+v"(vz));
    float Ts[SC_SEEDS][12];
#pragma unroll
    for (int sl = 0; sl < SC_SEEDS; ++sl) {
        const float* tp = seed_trans + ((size_t)b * S + min(s0 + sl, S - 1)) * 16 + vz;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            Ts[sl][e] = tp[e];
            if constexpr (DBG != 0) if (dbg && t == 0 && s0 + sl < S) dbg[((size_t)b * S + s0 + sl) * 16 + e] = Ts[sl][e];
        }
    }
    const float* srcb = src + (size_t)b * NS * 3;
    const float* tgtb = tgt + (size_t)b * NS * 3;
    int cnt[SC_SEEDS];
#pragma unroll
    for (int s = 0; s < SC_SEEDS; ++s) cnt[s] = 0;
    for (int i = t; i < N; i += 256) {
        const float px = srcb[i * 3], py = srcb[i * 3 + 1], pz = srcb[i * 3 + 2];
        const float qx = tgtb[i * 3], qy = tgtb[i * 3 + 1], qz = tgtb[i * 3 + 2];
        // votes per wavefront: ballot + popcount (scalar adds)
#pragma unroll
        for (int s = 0; s < SC_SEEDS; ++s) cnt[s] += __popcll(__ballot(residual_sq(Ts[s], px, py, pz, qx, qy, qz) < thr2));
    }
#pragma unroll
    for (int s = 0; s < SC_SEEDS; ++s)
        if (lane == 0) wsum[wave][s] = cnt[s];
    __syncthreads();
    if (t < SC_SEEDS && s0 + t < S) counts[(size_t)b * S + s0 + t] = wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t];
    if constexpr (DBG != 0) {
        if (!dbg) return;
        if (t < SC_SEEDS && s0 + t < S) dbg[((size_t)b * S + s0 + t) * 16 + 12] = (float)(wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t]);
        __syncthreads();
        // the same count with every point re-read past the caches (system-scope loads)
#pragma unroll
        for (int s = 0; s < SC_SEEDS; ++s) cnt[s] = 0;
        for (int i = t; i < N; i += 256) {
            float p[3], q[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                p[e] = __hip_atomic_load(srcb + i * 3 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                q[e] = __hip_atomic_load(tgtb + i * 3 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int s = 0; s < SC_SEEDS; ++s) cnt[s] += residual_sq(Ts[s], p[0], p[1], p[2], q[0], q[1], q[2]) < thr2;
        }
#pragma unroll
        for (int s = 0; s < SC_SEEDS; ++s) {
            const int c = wave_sum(cnt[s]);
            if (lane == 0) wsum[wave][s] = c;
        }
        __syncthreads();
        if (t < SC_SEEDS && s0 + t < S) dbg[((size_t)b * S + s0 + t) * 16 + 13] = (float)(wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t]);
        __syncthreads();
        // ... and a third time with ordinary loads again
#pragma unroll
        for (int s = 0; s < SC_SEEDS; ++s) cnt[s] = 0;
        for (int i = t; i < N; i += 256) {
            const float px = srcb[i * 3], py = srcb[i * 3 + 1], pz = srcb[i * 3 + 2];
            const float qx = tgtb[i * 3], qy = tgtb[i * 3 + 1], qz = tgtb[i * 3 + 2];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < SC_SEEDS; ++s) cnt[s] += residual_sq(Ts[s], px, py, pz, qx, qy, qz) < thr2;
        }
#pragma unroll
        for (int s = 0; s < SC_SEEDS; ++s) {
            const int c = wave_sum(cnt[s]);
            if (lane == 0) wsum[wave][s] = c;
        }
        __syncthreads();
        if (t < SC_SEEDS && s0 + t < S) dbg[((size_t)b * S + s0 + t) * 16 + 14] = (float)(wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t]);
    }
}

}  // namespace pdsc
