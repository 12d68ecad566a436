// a-7, a-8, a-9: the per-seed solver -- k x k feature*spatial compatibility, power iteration, weighted
// Procrustes with a register-resident Jacobi 3x3 decomposition.
//   reference: models/PointDSC.py:257-282 (matrices + weight normalisation), :347-358 (power iteration),
//              models/common.py:7-45 (rigid_transform_3d; torch.svd is done on the HOST there),
//              utils/SE3.py:73-96 (integrate_trans).
// One WAVEFRONT per seed (seed_solve_kernel below), lane i owns neighbour i (k <= 64).  Zero host round trips (the reference
// pays one D2H+H2D per SVD batch and one sync per power iteration); k x k never leaves the CU.
#include "pdsc_common.h"

namespace pdsc {

// chosen iterate = first iteration at which EVERY seed of the pair passed allclose (the reference breaks
// there), else the last one.
__device__ __forceinline__ int chosen_iterate(unsigned int mask, int num_iter) {
    const unsigned int m = num_iter >= 32 ? mask : (mask & ((1u << num_iter) - 1u));
    return m ? (__ffs((int)m) - 1) : (num_iter - 1);
}

// weighted Procrustes of one seed, part 1: lane = neighbour, v = its eigenvector entry (0 for lanes >= k).
// models/PointDSC.py:282 (weight normalisation), models/common.py:17-33.  Lane 0 leaves the 3x3 covariance H and the two
// centroids in the seed's 16-float slot (H[9] | cA[3] | cB[3] | -); part 2, kabsch_batch_kernel, turns every slot into the
// 4x4 in place with ONE THREAD PER SEED: the fp64 Jacobi is ~20 k serial cycles, which as "lane 0 of a wavefront per seed"
// left 63 of 64 lanes idle and dominated the fused solver (150 of 255 us at 16 000 seeds).
__device__ __forceinline__ void seed_procrustes(int lane, bool valid, float v, float ax, float ay, float az, float bx, float by,
                                                float bz, float* __restrict__ trans_out, float* __restrict__ w_out) {
    float w = v / (wave_sum_dpp(v) + 1e-6f);                       // models/PointDSC.py:282
    if (w < 0.f) w = 0.f;                                      // models/common.py:20 (weight_threshold = 0)
    w = valid ? w : 0.f;
    if (w_out && valid) w_out[lane] = w;
    const float den = wave_sum_dpp(w) + 1e-6f;
    float cA[3] = {wave_sum_dpp(ax * w) / den, wave_sum_dpp(ay * w) / den, wave_sum_dpp(az * w) / den};
    float cB[3] = {wave_sum_dpp(bx * w) / den, wave_sum_dpp(by * w) / den, wave_sum_dpp(bz * w) / den};
    const float am[3] = {ax - cA[0], ay - cA[1], az - cA[2]};
    const float bm[3] = {bx - cB[0], by - cB[1], bz - cB[2]};
    float H[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[r * 3 + c] = wave_sum_dpp(am[r] * w * bm[c]);
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) trans_out[e] = H[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) { trans_out[9 + e] = cA[e]; trans_out[12 + e] = cB[e]; }
    }
}

// part 2 (models/common.py:35-45 + utils/SE3.py:73-96): slot (H | cA | cB) -> row-major 4x4, one thread per seed
__global__ __launch_bounds__(64) void kabsch_batch_kernel(float* __restrict__ seed_trans, int count) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    float* slot = seed_trans + (size_t)i * 16;
    float H[9], cA[3], cB[3], T[16];
#pragma unroll
    for (int e = 0; e < 9; ++e) H[e] = slot[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) { cA[e] = slot[9 + e]; cB[e] = slot[12 + e]; }
    kabsch_from_covariance(H, cA, cB, T);
#pragma unroll
    for (int e = 0; e < 16; ++e) slot[e] = T[e];
}

// only_if_early_exit: the fused solver below already wrote the hypotheses of the LAST iterate; this launch then re-solves
// a pair's seeds only when the reference's global early exit picked an earlier iterate (rare), and returns at once otherwise.
__global__ __launch_bounds__(64) void seed_transform_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                            const int* __restrict__ knn_idx,
                                                            const float* __restrict__ eig_iters,
                                                            const unsigned int* __restrict__ conv_mask,
                                                            float* __restrict__ seed_trans, float* __restrict__ seed_w,
                                                            int N, int S, int k, int num_iter, int only_if_early_exit) {
    const int lane = threadIdx.x;
    const int s = blockIdx.x, b = blockIdx.y;
    const bool valid = lane < k;
    int it = num_iter - 1;
    if (num_iter > 0) it = chosen_iterate(conv_mask[b], num_iter);
    if (only_if_early_exit && it == num_iter - 1) return;
    const int idx = knn_idx[((size_t)b * S + s) * k + (valid ? lane : 0)];
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    float v = valid ? 1.0f : 0.0f;
    if (num_iter > 0) {
        v = eig_iters[(((size_t)b * S + s) * num_iter + it) * PDSC_MAX_K + lane];
        v = valid ? v : 0.f;
    }
    seed_procrustes(lane, valid, v, srcb[idx * 3], srcb[idx * 3 + 1], srcb[idx * 3 + 2], tgtb[idx * 3], tgtb[idx * 3 + 1],
                    tgtb[idx * 3 + 2], seed_trans + ((size_t)b * S + s) * 16, seed_w ? seed_w + ((size_t)b * S + s) * k : nullptr);
}

// ---- fused per-seed solver (a-7 + a-8 + a-9 in one launch): ONE WAVEFRONT PER SEED -----------------------------------------
// NB = ceil(k / 16) blocks of 16 neighbours.  The feature rows go straight from global memory (L2-resident: 2.5 MB per pair)
// into MFMA operand registers -- lane (row r = lane & 15 of a block, k-slot kq = lane >> 4) holds channels 16 q + 4 kq + e,
// q = 0..7, of neighbour 16 R + r: 8 float4 per block -- and the k x k Gram is accumulated on v_mfma_f32_16x16x4_f32 for
// the NB (NB + 1) / 2 tiles on / above the diagonal only (M is symmetric bit for bit: same products, same order).  The
// spatial term is evaluated in the accumulator layout, M goes to a wave-private LDS patch (mirrored), every lane < k then
// owns one row in registers for the power iterations (v broadcast through LDS), and the same lanes (= neighbours) feed the
// weighted Procrustes of the last iterate.  No workgroup barrier anywhere: the 4 waves of a workgroup are independent.
constexpr int SV_WAVES = 4;
template <int NB>
struct SolveLds {
    static constexpr int KP = 16 * NB, MLD = KP + 1;
    float M[KP * MLD];
    float pts[KP][8];
    float v[KP];
};
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NB>
__global__ __launch_bounds__(64 * SV_WAVES) void seed_solve_kernel(const float* __restrict__ normed, const float* __restrict__ src,
                                                                   const float* __restrict__ tgt, const int* __restrict__ knn_idx,
                                                                   const float* __restrict__ sigma, const float* __restrict__ sigma_spat,
                                                                   float* __restrict__ eig_iters, unsigned int* __restrict__ conv_mask,
                                                                   float* __restrict__ seed_M, float* __restrict__ seed_trans,
                                                                   float* __restrict__ seed_w, int N, int S, int k, int num_iter) {
    using L = SolveLds<NB>;
    constexpr int KP = L::KP, MLD = L::MLD;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x * SV_WAVES + wave, b = blockIdx.y;
    if (s >= S) return;                                   // wave-uniform; no workgroup barrier below
    L& sh = *reinterpret_cast<L*>(lds_raw + (size_t)wave * sizeof(L));
    const bool valid = lane < k;
    const int* idxp = knn_idx + ((size_t)b * S + s) * k;
    const int idx = idxp[valid ? lane : 0];
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    const float* nb = normed + (size_t)b * N * PDSC_CHANNELS;
    const float ax = srcb[idx * 3], ay = srcb[idx * 3 + 1], az = srcb[idx * 3 + 2];
    const float bx = tgtb[idx * 3], by = tgtb[idx * 3 + 1], bz = tgtb[idx * 3 + 2];
    if (lane < KP) {
        *reinterpret_cast<f32x4*>(&sh.pts[lane][0]) = f32x4{ax, ay, az, 0.f};
        *reinterpret_cast<f32x4*>(&sh.pts[lane][4]) = f32x4{bx, by, bz, 0.f};
    }
    // operand fragments: block R, chunk q -> float4 of neighbour min(16 R + (lane & 15), k - 1)
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 frag[NB][8];
#pragma unroll
    for (int R = 0; R < NB; ++R) {
        const int nbr = __shfl(idx, min(16 * R + r16, k - 1), 64);
        const float* row = nb + (size_t)nbr * PDSC_CHANNELS + 4 * kq;
#pragma unroll
        for (int q = 0; q < 8; ++q) frag[R][q] = *reinterpret_cast<const f32x4*>(row + 16 * q);
    }
    const float sg = sigma[0], sig2 = sg * sg, rsig2 = 1.0f / sig2;
    const float sd = sigma_spat[0], sd2 = sd * sd, rsd2 = 1.0f / sd2;
    wave_lds_sync();                                      // pts visible to the wave
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = I; J < NB; ++J) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[I][q][e], frag[J][q][e], acc, 0, 0, 0);
            // accumulator: column j = 16 J + (lane & 15), rows i = 16 I + 4 (lane >> 4) + r
            const int j = 16 * J + r16;
            const f32x4 pj = *reinterpret_cast<const f32x4*>(&sh.pts[j][0]);
            const f32x4 qj = *reinterpret_cast<const f32x4*>(&sh.pts[j][4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + 4 * kq + r;
                const f32x4 pi = *reinterpret_cast<const f32x4*>(&sh.pts[i][0]);
                const f32x4 qi = *reinterpret_cast<const f32x4*>(&sh.pts[i][4]);
                // 1-ulp hardware sqrt and multiplication by the (refined) reciprocals instead of the IEEE sequences: the
                // entries of M feed a power iteration, nothing is thresholded on them (24 elements per lane x ~50 VALU ops
                // saved; this kernel is VALU-bound)
                const float fm = fmaxf(1.0f - (1.0f - acc[r]) * rsig2, 0.0f);
                const float dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
                const float ex = qi[0] - qj[0], ey = qi[1] - qj[1], ez = qi[2] - qj[2];
                const float ds = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);          // ((a-b)**2).sum(-1) ** 0.5
                const float dt = __builtin_amdgcn_sqrtf((ex * ex + ey * ey) + ez * ez);
                const float df = ds - dt;
                const float sm = fmaxf(1.0f - (df * df) * rsd2, 0.0f);
                const float m = (i == j || i >= k || j >= k) ? 0.0f : fm * sm;
                sh.M[i * MLD + j] = m;
                if (I != J) sh.M[j * MLD + i] = m;        // mirror (block-uniform condition)
            }
        }
    }
    wave_lds_sync();
    if (seed_M) {
        for (int e = lane; e < k * k; e += 64) seed_M[((size_t)b * S + s) * k * k + e] = sh.M[(e / k) * MLD + (e % k)];
    }
    // power iteration: v <- M v / (||M v|| + 1e-6), every iterate kept, allclose flag per iteration
    float mrow[KP];
    {
        const float* mr = sh.M + min(lane, KP - 1) * MLD;
#pragma unroll
        for (int j = 0; j < KP; ++j) mrow[j] = mr[j];
    }
    float v = valid ? 1.0f : 0.0f, last = v;
    unsigned int bits = 0;
    float* out = eig_iters + ((size_t)b * S + s) * num_iter * PDSC_MAX_K;
    for (int it = 0; it < num_iter; ++it) {
        if (lane < KP) sh.v[lane] = v;
        wave_lds_sync();
        float nv = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < KP / 4; ++j4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(&sh.v[4 * j4]);     // broadcast
#pragma unroll
            for (int e = 0; e < 4; ++e) nv = fmaf(mrow[4 * j4 + e], vv[e], nv);  // the j = 0..k-1 chain (rows / columns >= k are 0)
        }
        nv = valid ? nv : 0.f;
        const float nrm = sqrtf(wave_sum_dpp(nv * nv));
        v = nv / (nrm + 1e-6f);
        out[it * PDSC_MAX_K + lane] = v;
        const bool close = fabsf(v - last) <= (1e-8f + 1e-5f * fabsf(last));   // torch.allclose(v, last)
        if (__all(close || !valid)) bits |= (1u << it);
        last = v;
        wave_lds_sync();                                  // everyone has read sh.v before it is overwritten
    }
    if (lane == 0) atomicAnd(conv_mask + b, bits);
    if (seed_trans)                                       // hypothesis of the LAST iterate (the common case, see seed_transform_kernel)
        seed_procrustes(lane, valid, valid ? v : 0.f, ax, ay, az, bx, by, bz, seed_trans + ((size_t)b * S + s) * 16,
                        seed_w ? seed_w + ((size_t)b * S + s) * k : nullptr);
}

template <int NB>
static int launch_seed_solve(const float* normed, const float* src, const float* tgt, const int* knn_idx, const float* sigma,
                             const float* sigma_spat, float* eig_iters, unsigned int* conv_mask, float* seed_M, float* seed_trans,
                             float* seed_w, int bs, int N, int S, int k, int num_iter, hipStream_t st) {
    const size_t lds_bytes = SV_WAVES * sizeof(SolveLds<NB>);
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&seed_solve_kernel<NB>), lds_bytes, "pdsc_seed_solve(dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(seed_solve_kernel<NB>, dim3(ceil_div(S, SV_WAVES), bs), dim3(64 * SV_WAVES), lds_bytes, st, normed, src, tgt,
                       knn_idx, sigma, sigma_spat, eig_iters, conv_mask, seed_M, seed_trans, seed_w, N, S, k, num_iter);
    return check_launch("pdsc_seed_solve");
}

// general rigid_transform_3d(A, B, weights, weight_threshold): one 256-thread workgroup per problem
__global__ __launch_bounds__(256) void rigid_transform_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              const float* __restrict__ W, float wthr,
                                                              float* __restrict__ T, int n) {
    __shared__ float red[4 * 9];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* a = A + (size_t)b * n * 3;
    const float* bb = B + (size_t)b * n * 3;
    const float* w = W ? W + (size_t)b * n : nullptr;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = t; i < n; i += 256) {
        float wi = w ? w[i] : 1.0f;
        if (wi < wthr) wi = 0.f;
        acc[0] += wi;
        acc[1] = fmaf(a[i * 3], wi, acc[1]); acc[2] = fmaf(a[i * 3 + 1], wi, acc[2]); acc[3] = fmaf(a[i * 3 + 2], wi, acc[3]);
        acc[4] = fmaf(bb[i * 3], wi, acc[4]); acc[5] = fmaf(bb[i * 3 + 1], wi, acc[5]); acc[6] = fmaf(bb[i * 3 + 2], wi, acc[6]);
    }
    block_sum<7, 4>(acc, red);
    const float den = acc[0] + 1e-6f;
    const float cA[3] = {acc[1] / den, acc[2] / den, acc[3] / den};
    const float cB[3] = {acc[4] / den, acc[5] / den, acc[6] / den};
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = t; i < n; i += 256) {
        float wi = w ? w[i] : 1.0f;
        if (wi < wthr) wi = 0.f;
        const float am[3] = {a[i * 3] - cA[0], a[i * 3 + 1] - cA[1], a[i * 3 + 2] - cA[2]};
        const float bm[3] = {bb[i * 3] - cB[0], bb[i * 3 + 1] - cB[1], bb[i * 3 + 2] - cB[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[r * 3 + c] = fmaf(am[r] * wi, bm[c], H[r * 3 + c]);
    }
    block_sum<9, 4>(H, red);
    if (t == 0) kabsch_from_covariance(H, cA, cB, T + (size_t)b * 16);
}

// r06: the early-exit fix-up and the 3x3 decompositions in ONE launch (one launch less on the forward's chain).  A block = one wavefront =
// 64 consecutive seeds.  Phase 1 (rare: only pairs whose global early exit picked an earlier iterate): the wavefront re-solves each of
// its seeds of such a pair exactly as seed_transform_kernel does (lane = neighbour); phase 2: kabsch_batch_kernel's body, one thread
// per seed.  Same functions on the same inputs: bit-identical to the two launches it replaces.
__global__ __launch_bounds__(64) void fixup_kabsch_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                          const int* __restrict__ knn_idx, const float* __restrict__ eig_iters,
                                                          const unsigned int* __restrict__ conv_mask, float* __restrict__ seed_trans,
                                                          float* __restrict__ seed_w, int N, int S, int k, int num_iter, int count) {
    const int lane = threadIdx.x, base = blockIdx.x * 64;
    // the common case first: no pair this block's seeds belong to took the early exit (a handful of mask words, not one per seed)
    bool any = false;
    {
        const int last = min(base + 63, count - 1);
        for (int b = base / S; b <= last / S; ++b) any |= chosen_iterate(conv_mask[b], num_iter) != num_iter - 1;
    }
    for (int j = 0; any && j < 64 && base + j < count; ++j) {
        const int i = base + j, b = i / S, s = i - b * S;
        const int it = chosen_iterate(conv_mask[b], num_iter);                      // (uniform)
        if (it == num_iter - 1) continue;
        const bool valid = lane < k;
        const int idx = knn_idx[((size_t)b * S + s) * k + (valid ? lane : 0)];
        const float* srcb = src + (size_t)b * N * 3;
        const float* tgtb = tgt + (size_t)b * N * 3;
        float v = eig_iters[(((size_t)b * S + s) * num_iter + it) * PDSC_MAX_K + lane];
        v = valid ? v : 0.f;
        seed_procrustes(lane, valid, v, srcb[idx * 3], srcb[idx * 3 + 1], srcb[idx * 3 + 2], tgtb[idx * 3], tgtb[idx * 3 + 1],
                        tgtb[idx * 3 + 2], seed_trans + ((size_t)b * S + s) * 16, seed_w ? seed_w + ((size_t)b * S + s) * k : nullptr);
    }
    if (any) __threadfence();        // lane 0's slot stores above are read back by the slots' own threads below
    const int i = base + lane;
    if (i >= count) return;
    float* slot = seed_trans + (size_t)i * 16;
    float H[9], cA[3], cB[3], T[16];
#pragma unroll
    for (int e = 0; e < 9; ++e) H[e] = slot[e];          // (first read of these lines in this kernel: nothing stale can be cached)
#pragma unroll
    for (int e = 0; e < 3; ++e) { cA[e] = slot[9 + e]; cB[e] = slot[12 + e]; }
    kabsch_from_covariance(H, cA, cB, T);
#pragma unroll
    for (int e = 0; e < 16; ++e) slot[e] = T[e];
}

// The reference's power iteration stops ALL matrices of a call at the first iteration where every one of them passes
// allclose (models/PointDSC.py:347-358).  In testing mode a call is one pair; the validation forward is batched, so
// there the masks of all pairs are AND-ed.
__global__ void conv_mask_all_pairs_kernel(unsigned int* conv_mask, int bs) {
    unsigned int m = 0xffffffffu;
    for (int b = 0; b < bs; ++b) m &= conv_mask[b];
    __syncthreads();
    for (int b = threadIdx.x; b < bs; b += blockDim.x) conv_mask[b] = m;
}

}  // namespace pdsc

extern "C" int pdsc_conv_mask_all_pairs(unsigned int* conv_mask, int bs, void* stream) {
    PDSC_REQUIRE(conv_mask && bs > 0, "pdsc_conv_mask_all_pairs: bad argument");
    hipLaunchKernelGGL(pdsc::conv_mask_all_pairs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, conv_mask, bs);
    return pdsc::check_launch("pdsc_conv_mask_all_pairs");
}

extern "C" int pdsc_seed_power_iteration(const float* normed, const float* src, const float* tgt, const int* knn_idx,
                                         const float* sigma, const float* sigma_spat, float* eig_iters,
                                         unsigned int* conv_mask, float* seed_M, int bs, int N, int S, int k,
                                         int num_iterations, void* stream) {
    PDSC_REQUIRE(normed && src && tgt && knn_idx && sigma && sigma_spat && eig_iters && conv_mask,
                 "pdsc_seed_power_iteration: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_seed_power_iteration: bs=%d N=%d S=%d", bs, N, S);
    PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K, "pdsc_seed_power_iteration: k=%d (max %d)", k, PDSC_MAX_K);
    PDSC_REQUIRE(num_iterations >= 0 && num_iterations <= PDSC_MAX_POWER_ITERS,
                 "pdsc_seed_power_iteration: num_iterations=%d (max %d)", num_iterations, PDSC_MAX_POWER_ITERS);
    return pdsc_seed_solve(normed, src, tgt, knn_idx, sigma, sigma_spat, eig_iters, conv_mask, seed_M, nullptr, nullptr, bs, N, S, k,
                           num_iterations, stream);
}

namespace pdsc {
int launch_seed_solve_forward(const float* normed, const float* src, const float* tgt, const int* knn_idx, const float* sigma,
                              const float* sigma_spat, float* eig_iters, unsigned int* conv_mask, float* seed_trans, float* seed_weights,
                              int bs, int N, int S, int k, int num_iterations, bool mask_ready, hipStream_t st) {
    PDSC_REQUIRE(normed && src && tgt && knn_idx && sigma && sigma_spat && eig_iters && conv_mask && seed_trans, "pdsc_seed_solve: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_seed_solve: bs=%d N=%d S=%d", bs, N, S);
    PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K, "pdsc_seed_solve: k=%d (max %d)", k, PDSC_MAX_K);
    PDSC_REQUIRE(num_iterations >= 0 && num_iterations <= PDSC_MAX_POWER_ITERS, "pdsc_seed_solve: num_iterations=%d (max %d)",
                 num_iterations, PDSC_MAX_POWER_ITERS);
    if (!mask_ready)
        if (const int rc = launch_fill_u32(conv_mask, 0xFFFFFFFFu, (size_t)bs, st); rc != PDSC_OK) return rc;
    const int nb = ceil_div(k, 16);
#define PDSC_SOLVE(NBV) launch_seed_solve<NBV>(normed, src, tgt, knn_idx, sigma, sigma_spat, eig_iters, conv_mask, nullptr, seed_trans, \
                                               seed_weights, bs, N, S, k, num_iterations, st)
    const int rc = nb == 1 ? PDSC_SOLVE(1) : nb == 2 ? PDSC_SOLVE(2) : nb == 3 ? PDSC_SOLVE(3) : PDSC_SOLVE(4);
#undef PDSC_SOLVE
    if (rc != PDSC_OK) return rc;
    // the reference's global early exit (models/PointDSC.py:354-356) may have picked an earlier iterate for some pair (its seeds are
    // re-solved) -- and every slot becomes its 4x4: one launch (fixup_kabsch_kernel)
    if (num_iterations <= 0)        // (no iterate to choose: plain decompositions)
        hipLaunchKernelGGL(kabsch_batch_kernel, dim3(ceil_div(bs * S, 64)), dim3(64), 0, st, seed_trans, bs * S);
    else
        hipLaunchKernelGGL(fixup_kabsch_kernel, dim3(ceil_div(bs * S, 64)), dim3(64), 0, st, src, tgt, knn_idx, eig_iters, conv_mask, seed_trans,
                           seed_weights, N, S, k, num_iterations, bs * S);
    return check_launch("pdsc_seed_solve(fix-up + kabsch)");
}
}  // namespace pdsc

extern "C" int pdsc_seed_solve(const float* normed, const float* src, const float* tgt, const int* knn_idx, const float* sigma,
                               const float* sigma_spat, float* eig_iters, unsigned int* conv_mask, float* seed_M, float* seed_trans,
                               float* seed_weights, int bs, int N, int S, int k, int num_iterations, void* stream) {
    PDSC_REQUIRE(normed && src && tgt && knn_idx && sigma && sigma_spat && eig_iters && conv_mask, "pdsc_seed_solve: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0, "pdsc_seed_solve: bs=%d N=%d S=%d", bs, N, S);
    PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K, "pdsc_seed_solve: k=%d (max %d)", k, PDSC_MAX_K);
    PDSC_REQUIRE(num_iterations >= 0 && num_iterations <= PDSC_MAX_POWER_ITERS, "pdsc_seed_solve: num_iterations=%d (max %d)",
                 num_iterations, PDSC_MAX_POWER_ITERS);
    hipStream_t st = (hipStream_t)stream;
    if (const int rc = pdsc::launch_fill_u32(conv_mask, 0xFFFFFFFFu, (size_t)bs, st); rc != PDSC_OK) return rc;
    const int nb = pdsc::ceil_div(k, 16);
#define PDSC_SOLVE(NBV) pdsc::launch_seed_solve<NBV>(normed, src, tgt, knn_idx, sigma, sigma_spat, eig_iters, conv_mask, seed_M, seed_trans, \
                                                     seed_weights, bs, N, S, k, num_iterations, st)
    int rc = nb == 1 ? PDSC_SOLVE(1) : nb == 2 ? PDSC_SOLVE(2) : nb == 3 ? PDSC_SOLVE(3) : PDSC_SOLVE(4);
#undef PDSC_SOLVE
    if (rc != PDSC_OK || !seed_trans) return rc;
    if (num_iterations <= 0) {
        hipLaunchKernelGGL(pdsc::kabsch_batch_kernel, dim3(pdsc::ceil_div(bs * S, 64)), dim3(64), 0, st, seed_trans, bs * S);
        return pdsc::check_launch("pdsc_seed_solve(kabsch)");
    }
    // the reference's global early exit (models/PointDSC.py:354-356) picked an earlier iterate for some pair: re-solve its seeds
    hipLaunchKernelGGL(pdsc::seed_transform_kernel, dim3(S, bs), dim3(64), 0, st, src, tgt, knn_idx, eig_iters, conv_mask, seed_trans,
                       seed_weights, N, S, k, num_iterations, 1);
    rc = pdsc::check_launch("pdsc_seed_solve(early-exit fix-up)");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(pdsc::kabsch_batch_kernel, dim3(pdsc::ceil_div(bs * S, 64)), dim3(64), 0, st, seed_trans, bs * S);
    return pdsc::check_launch("pdsc_seed_solve(kabsch)");
}

extern "C" int pdsc_seed_transforms(const float* src, const float* tgt, const int* knn_idx, const float* eig_iters,
                                    const unsigned int* conv_mask, float* seed_trans, float* seed_weights, int bs, int N,
                                    int S, int k, int num_iterations, void* stream) {
    PDSC_REQUIRE(src && tgt && knn_idx && eig_iters && conv_mask && seed_trans, "pdsc_seed_transforms: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0 && k >= 1 && k <= PDSC_MAX_K, "pdsc_seed_transforms: bs=%d N=%d S=%d k=%d", bs, N, S, k);
    PDSC_REQUIRE(num_iterations >= 0 && num_iterations <= PDSC_MAX_POWER_ITERS, "pdsc_seed_transforms: num_iterations=%d", num_iterations);
    hipLaunchKernelGGL(pdsc::seed_transform_kernel, dim3(S, bs), dim3(64), 0, (hipStream_t)stream, src, tgt, knn_idx, eig_iters,
                       conv_mask, seed_trans, seed_weights, N, S, k, num_iterations, 0);
    const int rc = pdsc::check_launch("pdsc_seed_transforms");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(pdsc::kabsch_batch_kernel, dim3(pdsc::ceil_div(bs * S, 64)), dim3(64), 0, (hipStream_t)stream, seed_trans, bs * S);
    return pdsc::check_launch("pdsc_seed_transforms(kabsch)");
}

extern "C" int pdsc_rigid_transform_3d(const float* A, const float* B, const float* weights, float weight_threshold,
                                       float* T, int bs, int n, void* stream) {
    PDSC_REQUIRE(T && (n == 0 || (A && B)), "pdsc_rigid_transform_3d: null pointer");
    PDSC_REQUIRE(bs > 0 && n >= 0, "pdsc_rigid_transform_3d: bs=%d n=%d", bs, n);
    hipLaunchKernelGGL(pdsc::rigid_transform_kernel, dim3(bs), dim3(256), 0, (hipStream_t)stream, A, B, weights,
                       weight_threshold, T, n);
    return pdsc::check_launch("pdsc_rigid_transform_3d");
}
