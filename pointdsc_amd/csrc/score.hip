// a-10, a-11: hypothesis scoring / selection and the on-device post-refinement loop.
//   reference: models/PointDSC.py:325-335 (score S hypotheses on N correspondences, argmax, labels)
//              models/PointDSC.py:403-438 (post_refinement), utils/SE3.py:43-57 (transform)
// The reference materialises [S,N,3] predictions and [S,N] residuals; here every (seed, point) residual is
// formed in registers and only S integer counts leave the chip.  Bound: VALU/latency (30*S*N flop, 24*N+64*S
// bytes); reported as time only.
#include "pdsc_common.h"
#include "ragged.h"

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
//   * THIS FILE IS COMPILED WITH -fno-slp-vectorize (pointdsc_amd/build.py).  The SLP vectoriser pairs the residual tests of
//     two seeds into packed fp32 instructions (v_pk_mul/fma/add_f32 with op_sel broadcasts of the just-loaded point); in that
//     form one half of the seed pairs came out a few votes short on 0.2-0.7 % of the forwards whenever kernels of other forwards
//     were co-resident -- transforms and points verified equal, a recount later in the same launch right, the affected half
//     moving with the instruction schedule (LDS staging, atomics, carry-chain vs ballot counting made no difference).  Scalar
//     fp32 code: 0 differing forwards in 8000-12000 per mode.  Cause below the ISA level not established;
//   * votes are counted per wavefront with ballot + popcount, the transforms sit in registers (no LDS staging, no barrier
//     before the loop).
// DBG (experiments builds, diagnostics): dbg[(b*S+s)*16 + 0..11] = the transform as this kernel read it, [12] = its count,
// [13] = the same count over points re-read with system-scope (cache-bypassing) loads, [14] = a third count, ordinary loads again
template <int DBG>
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

float*& score_debug_slot() {       // (experiments builds) where the next launch_score_hypotheses leaves its diagnostics; NULL = nowhere
    static thread_local float* p = nullptr;
    return p;
}

// first argmax over the S counts, emit the winning transform and its inlier labels
__global__ __launch_bounds__(1024) void select_best_kernel(const int* __restrict__ counts, const float* __restrict__ seed_trans,
                                                           const float* __restrict__ src, const float* __restrict__ tgt,
                                                           float thr, int* __restrict__ best_out,
                                                           float* __restrict__ initial_trans, float* __restrict__ labels,
                                                           int NS, int S, const int* __restrict__ nvalid) {
    const int N = nvalid ? nvalid[blockIdx.x] : NS;
    __shared__ unsigned long long wbest[16];
    __shared__ float Tb[16];
    __shared__ int best_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    // key = count << 32 | (0xFFFFFFFF - index): max key == highest count, lowest index among equals
    unsigned long long key = 0;
    for (int s = t; s < S; s += 1024) {
        const unsigned long long kk = ((unsigned long long)(unsigned)counts[(size_t)b * S + s] << 32) | (0xFFFFFFFFu - (unsigned)s);
        key = kk > key ? kk : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
    }
    if (lane == 0) wbest[wave] = key;
    __syncthreads();
    if (t == 0) {
        unsigned long long k = wbest[0];
        for (int w = 1; w < 16; ++w) k = wbest[w] > k ? wbest[w] : k;
        best_s = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFu));
        if (best_out) best_out[b] = best_s;
    }
    __syncthreads();
    if (t < 16) {
        const float v = seed_trans[((size_t)b * S + best_s) * 16 + t];
        Tb[t] = v;
        initial_trans[(size_t)b * 16 + t] = v;
    }
    __syncthreads();
    const float* srcb = src + (size_t)b * NS * 3;
    const float* tgtb = tgt + (size_t)b * NS * 3;
    for (int i = t; i < N; i += 1024) {
        const float r = residual(Tb, srcb[i * 3], srcb[i * 3 + 1], srcb[i * 3 + 2], tgtb[i * 3], tgtb[i * 3 + 1], tgtb[i * 3 + 2]);
        labels[(size_t)b * NS + i] = r < thr ? 1.0f : 0.0f;
    }
    for (int i = N + t; i < NS; i += 1024) labels[(size_t)b * NS + i] = 0.0f;       // padding rows of a ragged batch
}

// post_refinement: one persistent 512-thread workgroup per pair runs the whole <=max_iters loop.
__global__ __launch_bounds__(512) void refine_kernel(const float* __restrict__ initial_trans, const float* __restrict__ src,
                                                      const float* __restrict__ tgt, float thr, int max_iters,
                                                      float* __restrict__ final_trans, int* __restrict__ solves, int NS,
                                                      const int* __restrict__ nvalid) {
    __shared__ float red[8 * 9];
    __shared__ float Tc[16];
    const int t = threadIdx.x, b = blockIdx.x;
    const int N = nvalid ? nvalid[b] : NS;
    const float* srcb = src + (size_t)b * NS * 3;
    const float* tgtb = tgt + (size_t)b * NS * 3;
    if (t < 16) Tc[t] = initial_trans[(size_t)b * 16 + t];
    __syncthreads();
    int prev = 0, solved = 0;
    for (int it = 0; it < max_iters; ++it) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // count, sum w, sum w*a (3), sum w*b (3)
        for (int i = t; i < N; i += 512) {
            const float px = srcb[i * 3], py = srcb[i * 3 + 1], pz = srcb[i * 3 + 2];
            const float qx = tgtb[i * 3], qy = tgtb[i * 3 + 1], qz = tgtb[i * 3 + 2];
            const float r = residual(Tc, px, py, pz, qx, qy, qz);
            if (r < thr) {
                const float q = r / thr;
                const float w = 1.0f / (1.0f + q * q);             // models/PointDSC.py:435
                acc[0] += 1.0f;
                acc[1] += w;
                acc[2] = fmaf(px, w, acc[2]); acc[3] = fmaf(py, w, acc[3]); acc[4] = fmaf(pz, w, acc[4]);
                acc[5] = fmaf(qx, w, acc[5]); acc[6] = fmaf(qy, w, acc[6]); acc[7] = fmaf(qz, w, acc[7]);
            }
        }
        block_sum<8, 8>(acc, red);
        const int n_inl = (int)acc[0];                              // exact: counts < 2^24
        if (n_inl == prev) break;                                   // abs(int(inlier_num - previous)) < 1
        prev = n_inl;
        const float den = acc[1] + 1e-6f;
        const float cA[3] = {acc[2] / den, acc[3] / den, acc[4] / den};
        const float cB[3] = {acc[5] / den, acc[6] / den, acc[7] / den};
        float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = t; i < N; i += 512) {
            const float px = srcb[i * 3], py = srcb[i * 3 + 1], pz = srcb[i * 3 + 2];
            const float qx = tgtb[i * 3], qy = tgtb[i * 3 + 1], qz = tgtb[i * 3 + 2];
            const float r = residual(Tc, px, py, pz, qx, qy, qz);
            if (r < thr) {
                const float q = r / thr;
                const float w = 1.0f / (1.0f + q * q);
                const float am[3] = {px - cA[0], py - cA[1], pz - cA[2]};
                const float bm[3] = {qx - cB[0], qy - cB[1], qz - cB[2]};
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int c = 0; c < 3; ++c) H[rr * 3 + c] = fmaf(am[rr] * w, bm[c], H[rr * 3 + c]);
            }
        }
        block_sum<9, 8>(H, red);
        __syncthreads();                     // everyone has finished reading Tc for this iteration
        if (t == 0) kabsch_from_covariance(H, cA, cB, Tc);
        ++solved;
        __syncthreads();
    }
    if (t < 16) final_trans[(size_t)b * 16 + t] = Tc[t];
    if (t == 0 && solves) solves[b] = solved;
}

}  // namespace pdsc

namespace pdsc {

int launch_score_hypotheses(const float* seed_trans, const float* src, const float* tgt, float thr, int* counts, int bs, int N, int S,
                            const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(seed_trans && src && tgt && counts, "pdsc_score_hypotheses: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_score_hypotheses: bs=%d N=%d S=%d", bs, N, S);
    dim3 grid(ceil_div(S, SC_SEEDS), bs);
#ifdef PDSC_EXPERIMENTS
    if (float* dbg = score_debug_slot()) {
        hipLaunchKernelGGL(score_kernel<1>, grid, dim3(256), 0, st, seed_trans, src, tgt, sqrt_threshold_radicand(thr), counts, N, S, nvalid, dbg);
        return check_launch("pdsc_score_hypotheses");
    }
#endif
    hipLaunchKernelGGL(score_kernel<0>, grid, dim3(256), 0, st, seed_trans, src, tgt, sqrt_threshold_radicand(thr), counts, N, S, nvalid, (float*)nullptr);
    return check_launch("pdsc_score_hypotheses");
}

int launch_select_best(const int* counts, const float* seed_trans, const float* src, const float* tgt, float thr, int* best,
                       float* initial_trans, float* labels, int bs, int N, int S, const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(counts && seed_trans && src && tgt && initial_trans && labels, "pdsc_select_best: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_select_best: bs=%d N=%d S=%d", bs, N, S);
    hipLaunchKernelGGL(select_best_kernel, dim3(bs), dim3(1024), 0, st, counts, seed_trans, src, tgt, thr, best, initial_trans, labels, N, S,
                       nvalid);
    return check_launch("pdsc_select_best");
}

int launch_post_refinement(const float* initial_trans, const float* src, const float* tgt, float threshold, int max_iters,
                           float* final_trans, int* solves, int bs, int N, const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(initial_trans && src && tgt && final_trans, "pdsc_post_refinement: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && max_iters >= 0, "pdsc_post_refinement: bs=%d N=%d iters=%d", bs, N, max_iters);
    hipLaunchKernelGGL(refine_kernel, dim3(bs), dim3(512), 0, st, initial_trans, src, tgt, threshold, max_iters, final_trans, solves, N, nvalid);
    return check_launch("pdsc_post_refinement");
}

}  // namespace pdsc

extern "C" int pdsc_score_hypotheses(const float* seed_trans, const float* src, const float* tgt, float thr, int* counts,
                                     int bs, int N, int S, void* stream) {
    return pdsc::launch_score_hypotheses(seed_trans, src, tgt, thr, counts, bs, N, S, nullptr, (hipStream_t)stream);
}

extern "C" int pdsc_select_best(const int* counts, const float* seed_trans, const float* src, const float* tgt, float thr,
                                int* best, float* initial_trans, float* labels, int bs, int N, int S, void* stream) {
    return pdsc::launch_select_best(counts, seed_trans, src, tgt, thr, best, initial_trans, labels, bs, N, S, nullptr, (hipStream_t)stream);
}

extern "C" int pdsc_post_refinement(const float* initial_trans, const float* src, const float* tgt, float threshold,
                                    int max_iters, float* final_trans, int* solves, int bs, int N, void* stream) {
    return pdsc::launch_post_refinement(initial_trans, src, tgt, threshold, max_iters, final_trans, solves, bs, N, nullptr, (hipStream_t)stream);
}
