// a-10, a-11: hypothesis scoring / selection and the on-device post-refinement loop.
//   reference: models/PointDSC.py:325-335 (score S hypotheses on N correspondences, argmax, labels)
//              models/PointDSC.py:403-438 (post_refinement), utils/SE3.py:43-57 (transform)
// The reference materialises [S,N,3] predictions and [S,N] residuals; here every (seed, point) residual is
// formed in registers and only S integer counts leave the chip.  Bound: VALU/latency (30*S*N flop, 24*N+64*S
// bytes); reported as time only.
#include "pdsc_common.h"
#include "ragged.h"
#include "score_kernel.h"

namespace pdsc {

float*& score_debug_slot() {       // (experiments builds) where the next launch_score_hypotheses leaves its diagnostics; NULL = nowhere
    static thread_local float* p = nullptr;
    return p;
}

// first argmax over the S counts, emit the winning transform and its inlier labels
// (NT = threads of the workgroup that runs it: 1024 in select_best_kernel, 512 in select_refine_kernel -- the argmax is order
//  independent and the labels elementwise, so the results do not depend on NT; Tb_out: the winning 4x4 stays in shared memory)
template <int NT>
__device__ __forceinline__ void select_best_body(const int* __restrict__ counts, const float* __restrict__ seed_trans,
                                                           const float* __restrict__ src, const float* __restrict__ tgt,
                                                           float thr, int* __restrict__ best_out,
                                                           float* __restrict__ initial_trans, float* __restrict__ labels,
                                                           int NS, int S, const int* __restrict__ nvalid, int b, float* __restrict__ Tb) {
    const int N = nvalid ? nvalid[b] : NS;
    __shared__ unsigned long long wbest[NT / 64];
    __shared__ int best_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // key = count << 32 | (0xFFFFFFFF - index): max key == highest count, lowest index among equals
    unsigned long long key = 0;
    for (int s = t; s < S; s += NT) {
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
        for (int w = 1; w < NT / 64; ++w) k = wbest[w] > k ? wbest[w] : k;
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
    for (int i = t; i < N; i += NT) {
        const float r = residual(Tb, srcb[i * 3], srcb[i * 3 + 1], srcb[i * 3 + 2], tgtb[i * 3], tgtb[i * 3 + 1], tgtb[i * 3 + 2]);
        labels[(size_t)b * NS + i] = r < thr ? 1.0f : 0.0f;
    }
    for (int i = N + t; i < NS; i += NT) labels[(size_t)b * NS + i] = 0.0f;       // padding rows of a ragged batch
}

__global__ __launch_bounds__(1024) void select_best_kernel(const int* __restrict__ counts, const float* __restrict__ seed_trans,
                                                           const float* __restrict__ src, const float* __restrict__ tgt,
                                                           float thr, int* __restrict__ best_out,
                                                           float* __restrict__ initial_trans, float* __restrict__ labels,
                                                           int NS, int S, const int* __restrict__ nvalid) {
    __shared__ float Tb[16];
    select_best_body<1024>(counts, seed_trans, src, tgt, thr, best_out, initial_trans, labels, NS, S, nvalid, blockIdx.x, Tb);
}

// post_refinement: one persistent 512-thread workgroup per pair runs the whole <=max_iters loop.
// the loop of post_refinement for pair b, run by one 512-thread workgroup (shared by refine_kernel and select_refine_kernel)
__device__ __forceinline__ void refine_body(const float* __restrict__ initial_trans, const float* __restrict__ src,
                                                      const float* __restrict__ tgt, float thr, int max_iters,
                                                      float* __restrict__ final_trans, int* __restrict__ solves, int NS,
                                                      const int* __restrict__ nvalid, int* __restrict__ trace,
                                                      const unsigned int* __restrict__ range_flag, int b, const float* __restrict__ T0_shared) {
    __shared__ float red[8 * 9];
    __shared__ float Tc[16];
    const int t = threadIdx.x;
    const int N = nvalid ? nvalid[b] : NS;
    const float* srcb = src + (size_t)b * NS * 3;
    const float* tgtb = tgt + (size_t)b * NS * 3;
    if (t < 16) Tc[t] = T0_shared ? T0_shared[t] : initial_trans[(size_t)b * 16 + t];
    // trace (optional) [bs][PDSC_REFINE_TRACE]: the inlier count of every iteration evaluated (models/PointDSC.py:424-425), -1 after
    // the last one -- the discrete part of the loop, compared by the parity census with the reference's own sequence
    if (trace && t < PDSC_REFINE_TRACE) trace[(size_t)b * PDSC_REFINE_TRACE + t] = -1;
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
        if (trace && t == 0 && it < PDSC_REFINE_TRACE) trace[(size_t)b * PDSC_REFINE_TRACE + it] = n_inl;
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
    // fp16 range sentinel (pdsc_common.h): a pair whose activations left the range of the split-precision arithmetic was computed from
    // inf / NaN operands somewhere in the encoder -- its pose is returned as NaN, never as a plausible-looking wrong motion
    const bool poisoned = range_flag && range_flag[b] != 0u;
    if (t < 16) final_trans[(size_t)b * 16 + t] = poisoned ? __builtin_nanf("") : Tc[t];
    if (t == 0 && solves) solves[b] = solved;
}

__global__ __launch_bounds__(512) void refine_kernel(const float* __restrict__ initial_trans, const float* __restrict__ src,
                                                      const float* __restrict__ tgt, float thr, int max_iters,
                                                      float* __restrict__ final_trans, int* __restrict__ solves, int NS,
                                                      const int* __restrict__ nvalid, int* __restrict__ trace,
                                                      const unsigned int* __restrict__ range_flag) {
    refine_body(initial_trans, src, tgt, thr, max_iters, final_trans, solves, NS, nvalid, trace, range_flag, blockIdx.x, nullptr);
}

// r06: the two single-workgroup-per-pair launches at the end of the forward as one (one launch less on its chain): select the best
// hypothesis and emit its labels (select_best_body), then refine it (refine_body) -- the same two bodies, the pose handed over in
// shared memory instead of through initial_trans (which is still written: the parity census reads it).
__global__ __launch_bounds__(512) void select_refine_kernel(const int* __restrict__ counts, const float* __restrict__ seed_trans,
                                                            const float* __restrict__ src, const float* __restrict__ tgt, float thr,
                                                            float refine_thr, int max_iters, int* __restrict__ best_out,
                                                            float* __restrict__ initial_trans, float* __restrict__ labels,
                                                            float* __restrict__ final_trans, int* __restrict__ solves, int NS, int S,
                                                            const int* __restrict__ nvalid, int* __restrict__ trace,
                                                            const unsigned int* __restrict__ range_flag, unsigned int* __restrict__ range_report) {
    __shared__ float Tb[16];
    const int b = blockIdx.x;
    // (pdsc_set_range_report) the pair's range word also goes to pinned host memory: the caller that waits for the stream anyway reads
    // it there without a copy of its own
    if (range_report && threadIdx.x == 0) range_report[b] = range_flag ? range_flag[b] : 0u;
    select_best_body<512>(counts, seed_trans, src, tgt, thr, best_out, initial_trans, labels, NS, S, nvalid, b, Tb);
    __syncthreads();
    refine_body(initial_trans, src, tgt, refine_thr, max_iters, final_trans, solves, NS, nvalid, trace, range_flag, b, Tb);
}

}  // namespace pdsc

namespace pdsc {

int launch_score_hypotheses(const float* seed_trans, const float* src, const float* tgt, float thr, int* counts, int bs, int N, int S,
                            const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(seed_trans && src && tgt && counts, "pdsc_score_hypotheses: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_score_hypotheses: bs=%d N=%d S=%d", bs, N, S);
    dim3 grid(ceil_div(S, SC_SEEDS), bs);
#ifdef PDSC_EXPERIMENTS
    if (env_int("PDSC_SCORE_SLP", 0))        // A/B knob: the same kernel compiled WITH SLP vectorisation (score_slp.hip), the r03 reproducer
        return launch_score_hypotheses_slp(seed_trans, src, tgt, sqrt_threshold_radicand(thr), counts, bs, N, S, nvalid, st);
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

int launch_select_and_refine(const int* counts, const float* seed_trans, const float* src, const float* tgt, float thr, float refine_thr,
                             int max_iters, int* best, float* initial_trans, float* labels, float* final_trans, int* solves, int bs, int N,
                             int S, const int* nvalid, hipStream_t st, int* trace, const unsigned int* range_flag, unsigned int* range_report) {
    PDSC_REQUIRE(counts && seed_trans && src && tgt && initial_trans && labels && final_trans, "pdsc_select_best / pdsc_post_refinement: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0 && max_iters >= 0, "pdsc_select_best / pdsc_post_refinement: bs=%d N=%d S=%d iters=%d", bs, N, S, max_iters);
    hipLaunchKernelGGL(select_refine_kernel, dim3(bs), dim3(512), 0, st, counts, seed_trans, src, tgt, thr, refine_thr, max_iters, best, initial_trans,
                       labels, final_trans, solves, N, S, nvalid, trace, range_flag, range_report);
    return check_launch("pdsc_select_best + pdsc_post_refinement");
}

int launch_post_refinement(const float* initial_trans, const float* src, const float* tgt, float threshold, int max_iters,
                           float* final_trans, int* solves, int bs, int N, const int* nvalid, hipStream_t st, int* trace,
                           const unsigned int* range_flag) {
    PDSC_REQUIRE(initial_trans && src && tgt && final_trans, "pdsc_post_refinement: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && max_iters >= 0, "pdsc_post_refinement: bs=%d N=%d iters=%d", bs, N, max_iters);
    hipLaunchKernelGGL(refine_kernel, dim3(bs), dim3(512), 0, st, initial_trans, src, tgt, threshold, max_iters, final_trans, solves, N, nvalid, trace, range_flag);
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
    return pdsc::launch_post_refinement(initial_trans, src, tgt, threshold, max_iters, final_trans, solves, bs, N, nullptr, (hipStream_t)stream, nullptr);
}
