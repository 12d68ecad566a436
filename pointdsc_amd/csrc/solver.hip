// a-7, a-8, a-9: the per-seed solver -- k x k feature*spatial compatibility, power iteration, weighted
// Procrustes with a register-resident Jacobi 3x3 decomposition.
//   reference: models/PointDSC.py:257-282 (matrices + weight normalisation), :347-358 (power iteration),
//              models/common.py:7-45 (rigid_transform_3d; torch.svd is done on the HOST there),
//              utils/SE3.py:73-96 (integrate_trans).
// One workgroup per seed, lane i of every wave owns neighbour i (k <= 64); the k x k matrix build (k^2 x 128 MACs, the
// only sizeable part) is split column-wise over the 4 waves.  Nothing here is bandwidth- or MFMA-bound
// (S * ~0.5 MFLOP); the point is zero host round trips (the reference pays one D2H+H2D per SVD batch and
// one sync per power iteration) and k x k never leaving LDS.
#include "pdsc_common.h"

namespace pdsc {

constexpr int FS_LD = PDSC_CHANNELS + 4;     // padded feature row (16 distinct bank slots for b128 reads)
constexpr int MS_LD = PDSC_MAX_K + 1;

constexpr int SP_WAVES = 4;

__global__ __launch_bounds__(64 * SP_WAVES) void seed_power_kernel(const float* __restrict__ normed, const float* __restrict__ src,
                                                        const float* __restrict__ tgt, const int* __restrict__ knn_idx,
                                                        const float* __restrict__ sigma, const float* __restrict__ sigma_spat,
                                                        float* __restrict__ eig_iters, unsigned int* __restrict__ conv_mask,
                                                        float* __restrict__ seed_M, int N, int S, int k, int num_iter) {
    extern __shared__ __attribute__((aligned(16))) float Fs[];        // [k][FS_LD]
    __shared__ __attribute__((aligned(16))) float Ms[PDSC_MAX_K * MS_LD];
    __shared__ __attribute__((aligned(16))) float pts[PDSC_MAX_K][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x, b = blockIdx.y;
    const bool valid = lane < k;
    const int* idxp = knn_idx + ((size_t)b * S + s) * k;
    const int idx = idxp[valid ? lane : 0];
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    const float* nb = normed + (size_t)b * N * PDSC_CHANNELS;
    if (wave == 0) {
        pts[lane][0] = srcb[idx * 3]; pts[lane][1] = srcb[idx * 3 + 1]; pts[lane][2] = srcb[idx * 3 + 2];
        pts[lane][4] = tgtb[idx * 3]; pts[lane][5] = tgtb[idx * 3 + 1]; pts[lane][6] = tgtb[idx * 3 + 2];
    }
    // gather the k feature rows (32 float4 chunks each), two rows per wave instruction
    for (int e = threadIdx.x; e < k * 32; e += 64 * SP_WAVES) {
        const int j = e >> 5, c = e & 31;
        const int rj = __shfl(idx, j, 64);           // every wave holds the same idx vector
        *reinterpret_cast<f32x4*>(Fs + j * FS_LD + c * 4) = *reinterpret_cast<const f32x4*>(nb + (size_t)rj * PDSC_CHANNELS + c * 4);
    }
    __syncthreads();

    const float sg = sigma[0], sig2 = sg * sg;
    const float sd = sigma_spat[0], sd2 = sd * sd;
    // k x k feature Gram on the exact fp32 MFMA: wave w owns the 32 x 32 tile (rows 32(w>>1).., columns 32(w&1)..) of the
    // k <= 64 neighbours; rows >= k of Fs are never written, their products land in rows / columns >= k that are skipped.
    // Accumulator lane = column j, register r = row i: the spatial term is evaluated in the same layout (the points of
    // row i are LDS broadcasts) and the finished M goes to LDS row-major for the power iteration.
    {
        const int rb = wave >> 1, cb = wave & 1;
        const int jcol = 32 * cb + (lane & 31), hh = lane >> 5;
        if (32 * rb < k && 32 * cb < k) {              // wave-uniform: tiles entirely beyond k have nothing to do
            const float* arow = Fs + min(32 * rb + (lane & 31), k - 1) * FS_LD + 4 * hh;
            const float* brow = Fs + min(jcol, k - 1) * FS_LD + 4 * hh;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(arow + 8 * q);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(brow + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
            }
            const int jc = min(jcol, k - 1);
            const float jx = pts[jc][0], jy = pts[jc][1], jz = pts[jc][2];
            const float kx = pts[jc][4], ky = pts[jc][5], kz = pts[jc][6];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (i < k && jcol < k) {
                    const float fm = fmaxf(1.0f - (1.0f - acc[r]) / sig2, 0.0f);
                    const float dx = pts[i][0] - jx, dy = pts[i][1] - jy, dz = pts[i][2] - jz;
                    const float ex = pts[i][4] - kx, ey = pts[i][5] - ky, ez = pts[i][6] - kz;
                    const float ds = sqrtf((dx * dx + dy * dy) + dz * dz);          // ((a-b)**2).sum(-1) ** 0.5
                    const float dt = sqrtf((ex * ex + ey * ey) + ez * ez);
                    const float df = ds - dt;
                    const float sm = fmaxf(1.0f - (df * df) / sd2, 0.0f);
                    const float m = (i == jcol) ? 0.0f : fm * sm;
                    Ms[i * MS_LD + jcol] = m;
                    if (seed_M) seed_M[(((size_t)b * S + s) * k + i) * k + jcol] = m;
                }
            }
        }
    }
    __syncthreads();                                 // Ms complete
    if (wave != 0) return;                           // the power iteration is one wave's work
    // power iteration: v <- M v / (||M v|| + 1e-6), every iterate kept, allclose flag per iteration
    float v = valid ? 1.0f : 0.0f;
    float last = v;
    unsigned int bits = 0;
    float* out = eig_iters + ((size_t)b * S + s) * num_iter * PDSC_MAX_K;
    float mrow[PDSC_MAX_K];                          // this lane's row of M, in registers for all iterations
    {
        const float* mr = Ms + lane * MS_LD;
#pragma unroll
        for (int j = 0; j < PDSC_MAX_K; ++j) mrow[j] = j < k ? mr[j] : 0.f;
    }
    for (int it = 0; it < num_iter; ++it) {
        float nv = 0.f;
#pragma unroll
        for (int j = 0; j < PDSC_MAX_K; ++j)
            // j >= k: mrow[j] = 0 and v[j] = 0 add an exact zero, so this is the j = 0..k-1 chain without 64 branches
            nv = fmaf(mrow[j], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), j)), nv);
        nv = valid ? nv : 0.f;
        const float nrm = sqrtf(wave_sum(nv * nv));
        v = nv / (nrm + 1e-6f);
        out[it * PDSC_MAX_K + lane] = v;
        const bool close = fabsf(v - last) <= (1e-8f + 1e-5f * fabsf(last));   // torch.allclose(v, last)
        if (__all(close || !valid)) bits |= (1u << it);
        last = v;
    }
    if (lane == 0) atomicAnd(conv_mask + b, bits);
}

// chosen iterate = first iteration at which EVERY seed of the pair passed allclose (the reference breaks
// there), else the last one.
__device__ __forceinline__ int chosen_iterate(unsigned int mask, int num_iter) {
    const unsigned int m = num_iter >= 32 ? mask : (mask & ((1u << num_iter) - 1u));
    return m ? (__ffs((int)m) - 1) : (num_iter - 1);
}

__global__ __launch_bounds__(64) void seed_transform_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                            const int* __restrict__ knn_idx,
                                                            const float* __restrict__ eig_iters,
                                                            const unsigned int* __restrict__ conv_mask,
                                                            float* __restrict__ seed_trans, float* __restrict__ seed_w,
                                                            int N, int S, int k, int num_iter) {
    const int lane = threadIdx.x;
    const int s = blockIdx.x, b = blockIdx.y;
    const bool valid = lane < k;
    const int idx = knn_idx[((size_t)b * S + s) * k + (valid ? lane : 0)];
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    float v = valid ? 1.0f : 0.0f;
    if (num_iter > 0) {
        const int it = chosen_iterate(conv_mask[b], num_iter);
        v = eig_iters[(((size_t)b * S + s) * num_iter + it) * PDSC_MAX_K + lane];
        v = valid ? v : 0.f;
    }
    float w = v / (wave_sum(v) + 1e-6f);                       // models/PointDSC.py:282
    if (w < 0.f) w = 0.f;                                      // models/common.py:20 (weight_threshold = 0)
    w = valid ? w : 0.f;
    if (seed_w && valid) seed_w[((size_t)b * S + s) * k + lane] = w;
    const float ax = srcb[idx * 3], ay = srcb[idx * 3 + 1], az = srcb[idx * 3 + 2];
    const float bx = tgtb[idx * 3], by = tgtb[idx * 3 + 1], bz = tgtb[idx * 3 + 2];
    const float den = wave_sum(w) + 1e-6f;
    float cA[3] = {wave_sum(ax * w) / den, wave_sum(ay * w) / den, wave_sum(az * w) / den};
    float cB[3] = {wave_sum(bx * w) / den, wave_sum(by * w) / den, wave_sum(bz * w) / den};
    const float am[3] = {ax - cA[0], ay - cA[1], az - cA[2]};
    const float bm[3] = {bx - cB[0], by - cB[1], bz - cB[2]};
    float H[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[r * 3 + c] = wave_sum(am[r] * w * bm[c]);
    if (lane == 0) kabsch_from_covariance(H, cA, cB, seed_trans + ((size_t)b * S + s) * 16);
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
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(conv_mask, 0xFF, sizeof(unsigned int) * bs, st) != hipSuccess) return pdsc::check_launch("memset");
    const size_t lds_bytes = (size_t)k * pdsc::FS_LD * sizeof(float);
    hipLaunchKernelGGL(pdsc::seed_power_kernel, dim3(S, bs), dim3(64 * pdsc::SP_WAVES), lds_bytes, st, normed, src, tgt, knn_idx, sigma,
                       sigma_spat, eig_iters, conv_mask, seed_M, N, S, k, num_iterations);
    return pdsc::check_launch("pdsc_seed_power_iteration");
}

extern "C" int pdsc_seed_transforms(const float* src, const float* tgt, const int* knn_idx, const float* eig_iters,
                                    const unsigned int* conv_mask, float* seed_trans, float* seed_weights, int bs, int N,
                                    int S, int k, int num_iterations, void* stream) {
    PDSC_REQUIRE(src && tgt && knn_idx && eig_iters && conv_mask && seed_trans, "pdsc_seed_transforms: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && S > 0 && k >= 1 && k <= PDSC_MAX_K, "pdsc_seed_transforms: bs=%d N=%d S=%d k=%d", bs, N, S, k);
    PDSC_REQUIRE(num_iterations >= 0 && num_iterations <= PDSC_MAX_POWER_ITERS, "pdsc_seed_transforms: num_iterations=%d", num_iterations);
    hipLaunchKernelGGL(pdsc::seed_transform_kernel, dim3(S, bs), dim3(64), 0, (hipStream_t)stream, src, tgt, knn_idx, eig_iters,
                       conv_mask, seed_trans, seed_weights, N, S, k, num_iterations);
    return pdsc::check_launch("pdsc_seed_transforms");
}

extern "C" int pdsc_rigid_transform_3d(const float* A, const float* B, const float* weights, float weight_threshold,
                                       float* T, int bs, int n, void* stream) {
    PDSC_REQUIRE(T && (n == 0 || (A && B)), "pdsc_rigid_transform_3d: null pointer");
    PDSC_REQUIRE(bs > 0 && n >= 0, "pdsc_rigid_transform_3d: bs=%d n=%d", bs, n);
    hipLaunchKernelGGL(pdsc::rigid_transform_kernel, dim3(bs), dim3(256), 0, (hipStream_t)stream, A, B, weights,
                       weight_threshold, T, n);
    return pdsc::check_launch("pdsc_rigid_transform_3d");
}
