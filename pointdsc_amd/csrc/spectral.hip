// f-3: the spectral-matching baseline, i.e. the N x N power iteration of
//   reference baseline_scripts/baseline_3DMatch.py:19-53 (SM):
//     diff = corr[i] - corr[j];  d = |diff[0:3]| - |diff[3:6]|             (corr = centred corr_pos [N,6])
//     M    = max(0, 4.5 - d^2 / 2 / sigma^2), sigma = inlier_threshold / 3, zero diagonal
//     v    = 1;  10 x { v = M v;  v = v / (|v| + 1e-6) }
//     labels = 1 for the int(N * top_ratio) largest entries of v;  pred_trans = rigid_transform_3d(src, tgt, v * labels)
// (the same matrix-vector power iteration is what models/PointDSC.py:170 (commented) / cal_confidence run on N x N.)
//
// Two kernels, both HBM-bound:
//   sm_matrix_kernel : writes M once, 4 N^2 bytes (tiles of 64 x 256, keypoints from L1/L2; the matrix is symmetric
//                      but is written plainly -- it is read 10 times, written once)
//   sm_matvec_kernel : y = M v_raw * scale, one wave per row block, lanes stride the columns with float4 loads
//                      (1 KiB per wave instruction, fully coalesced), v in LDS, wave-shuffle row reductions; the
//                      normalisation of iteration t is folded into iteration t+1's read of v (per-block partial sums of
//                      y^2 are summed in a fixed order, so the result is deterministic), one launch per iteration.
// 4 N^2 bytes per iteration: 100 MB at N = 5000, i.e. the whole matrix fits the 256 MiB Infinity Cache.
#include <math.h>
#include "pdsc_common.h"

namespace pdsc {

constexpr int SMV_ROWS = 16;             // rows per workgroup in the mat-vec (4 waves x 4 rows)
constexpr int SMV_MAX_BLOCKS = 4096;     // partial-sum slots

__global__ __launch_bounds__(256) void sm_matrix_kernel(const float* __restrict__ corr, float sigma2, float* __restrict__ M,
                                                        long long ld, int N) {
    // thread -> 4 consecutive columns of one row; block = 64 rows x 256 columns (4 rows x 64 float4 per pass)
    const int b = blockIdx.z;
    const float* c = corr + (size_t)b * N * 6;
    float* Mb = M + (size_t)b * N * ld;
    const int j4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
    if (j4 >= ld) return;
    float cj[4][6];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = min(j4 + e, N - 1);
#pragma unroll
        for (int d = 0; d < 6; ++d) cj[e][d] = c[j * 6 + d];
    }
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + 4 * r;
        if (i >= N) break;
        float ci[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) ci[d] = c[i * 6 + d];
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ax = ci[0] - cj[e][0], ay = ci[1] - cj[e][1], az = ci[2] - cj[e][2];
            const float bx = ci[3] - cj[e][3], by = ci[4] - cj[e][4], bz = ci[5] - cj[e][5];
            const float ds = sqrtf((ax * ax + ay * ay) + az * az);              // torch.sum(diff ** 2, -1) ** 0.5
            const float dt = sqrtf((bx * bx + by * by) + bz * bz);
            const float d = ds - dt;
            float m = fmaxf(0.0f, 4.5f - ((d * d) / 2.0f) / sigma2);  // 4.5 - M**2 / 2 / sigma**2
            if (j4 + e == i || j4 + e >= N) m = 0.0f;
            out[e] = m;
        }
        *reinterpret_cast<f32x4*>(Mb + (size_t)i * ld + j4) = out;
    }
}

// y[i] = sum_j M[i][j] * (v[j] * scale),  scale = 1 / (sqrt(sum of the previous iteration's partials) + 1e-6) (or 1);
// partial_out[block] = sum of y[i]^2 over the block's rows.
__global__ __launch_bounds__(256) void sm_matvec_kernel(const float* __restrict__ M, long long ld, const float* __restrict__ v,
                                                        const float* __restrict__ partial_in, int n_partial_in,
                                                        float* __restrict__ y, float* __restrict__ partial_out, int N) {
    extern __shared__ __attribute__((aligned(16))) float vs[];      // [ld] scaled v, zero padded
    __shared__ float wsum[4];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float scale = 1.0f;
    if (partial_in) {
        float s = 0.f;
        for (int p = 0; p < n_partial_in; ++p) s += partial_in[(size_t)b * SMV_MAX_BLOCKS + p];   // fixed order: deterministic
        scale = 1.0f / (sqrtf(s) + 1e-6f);
    }
    const float* vb = v + (size_t)b * N;
    for (int j = t; j < ld; j += 256) vs[j] = j < N ? vb[j] * scale : 0.f;
    __syncthreads();
    const float* Mb = M + (size_t)b * N * ld;
    float sq = 0.f;
    {
        // the wave's 4 rows advance together: 4 independent 1-KiB loads in flight per step
        constexpr int R = SMV_ROWS / 4;
        const int i0 = blockIdx.x * SMV_ROWS + wave * R;
        const float* row[R];
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            row[r] = Mb + (size_t)min(i0 + r, N - 1) * ld;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
        }
        // columns >= N never enter the sums: a caller's padding columns [N, ld) may hold anything (0 * NaN would poison a row)
        const int n4 = (N + 3) & ~3;                                // <= ld (ld is a multiple of 4)
        for (int j = lane * 4; j < n4; j += 256) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(vs + j);
            f32x4 m[R];
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] = *reinterpret_cast<const f32x4*>(row[r] + j);
            if (j + 4 > N) {                                        // the one ragged group of the row
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[r][e] = j + e < N ? m[r][e] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(m[r][e], x[e], acc[r][e]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float yi = wave_sum((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]));
            if (i0 + r < N) {                                       // wave-uniform
                if (lane == 0) y[(size_t)b * N + i0 + r] = yi;
                sq = fmaf(yi, yi, sq);
            }
        }
    }
    if (lane == 0) wsum[wave] = sq;
    __syncthreads();
    if (t == 0) partial_out[(size_t)b * SMV_MAX_BLOCKS + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// final normalisation + selection mask: eig = y * scale; weights = eig * [rank(eig) < num_top]  (stable descending
// rank by counting, ties by ascending index -- torch.argsort is unspecified there)
__global__ __launch_bounds__(256) void sm_finish_kernel(const float* __restrict__ y, const float* __restrict__ partial_in,
                                                        int n_partial_in, int num_top, float* __restrict__ eig,
                                                        float* __restrict__ labels, float* __restrict__ weights, int N) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    float s = 0.f;
    for (int p = 0; p < n_partial_in; ++p) s += partial_in[(size_t)b * SMV_MAX_BLOCKS + p];
    const float scale = 1.0f / (sqrtf(s) + 1e-6f);
    const float* yb = y + (size_t)b * N;
    const float ei = yb[i] * scale;
    int cnt = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        if (j < N) {
            const float ej = yb[j] * scale;
            cnt += (ej > ei) || (ej == ei && j < i);
        }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) {
        const float lab = cnt < num_top ? 1.0f : 0.0f;
        eig[(size_t)b * N + i] = ei;
        labels[(size_t)b * N + i] = lab;
        weights[(size_t)b * N + i] = ei * lab;
    }
}


// ---- cal_confidence (reference models/PointDSC.py:366-401): confidence of a spectral-matching solution -------------
// One 1024-thread workgroup per pair for the O(N) vector steps (fixed-order reductions: deterministic), sm_matvec_kernel
// for every M x.  B = M - lambda1 v v^T is never formed: B x = M x - lambda1 v (v . x).
constexpr int CC_THREADS = 1024;
template <int NV>
__device__ __forceinline__ void cc_block_sum(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < CC_THREADS / 64; ++w) t += red[w * NV + i];
        v[i] = t;
    }
}
// step 0: lambda1 = (v . Mv) / (v . v); x <- 1.   method 0 / 2 finish here (conf = lambda1, or v . Mv / N)
__global__ __launch_bounds__(CC_THREADS) void cc_rayleigh_kernel(const float* __restrict__ v, const float* __restrict__ Mv, int method,
                                                                 float* __restrict__ lambda1, float* __restrict__ x,
                                                                 float* __restrict__ conf, int N) {
    __shared__ double red[CC_THREADS / 64 * 2];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mv + (size_t)b * N;
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        acc[0] += (double)vb[i] * (double)yb[i];
        acc[1] += (double)vb[i] * (double)vb[i];
        if (x) x[(size_t)b * N + i] = 1.0f;
    }
    cc_block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        const float l1 = (float)acc[0] / (float)acc[1];
        lambda1[b] = l1;
        if (method == 0) conf[b] = l1;
        if (method == 2) conf[b] = (float)acc[0] / (float)N;
    }
}
// one deflated power step: z = Mx - lambda1 v (v . x);  x <- z / (|z| + 1e-6)
__global__ __launch_bounds__(CC_THREADS) void cc_deflate_kernel(const float* __restrict__ v, const float* __restrict__ Mx,
                                                                const float* __restrict__ lambda1, float* __restrict__ x, int N) {
    __shared__ double red[CC_THREADS / 64];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mx + (size_t)b * N;
    float* xb = x + (size_t)b * N;
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) s[0] += (double)vb[i] * (double)xb[i];
    cc_block_sum<1>(s, red);
    const float coef = lambda1[b] * (float)s[0];
    double n2[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        const float z = yb[i] - coef * vb[i];
        n2[0] += (double)z * (double)z;
    }
    cc_block_sum<1>(n2, red);
    const float inv = 1.0f / ((float)sqrt(n2[0]) + 1e-6f);
    for (int i = threadIdx.x; i < N; i += CC_THREADS) xb[i] = (yb[i] - coef * vb[i]) * inv;
}
// lambda2 = (x . Bx) / (x . x);  conf = lambda1 / lambda2
__global__ __launch_bounds__(CC_THREADS) void cc_ratio_kernel(const float* __restrict__ v, const float* __restrict__ Mx,
                                                              const float* __restrict__ lambda1, const float* __restrict__ x,
                                                              float* __restrict__ conf, int N) {
    __shared__ double red[CC_THREADS / 64 * 3];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mx + (size_t)b * N;
    const float* xb = x + (size_t)b * N;
    double acc[3] = {0.0, 0.0, 0.0};                 // v.x, x.Mx, x.x
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        acc[0] += (double)vb[i] * (double)xb[i];
        acc[1] += (double)xb[i] * (double)yb[i];
        acc[2] += (double)xb[i] * (double)xb[i];
    }
    cc_block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        const double l1 = (double)lambda1[b];
        const float l2 = (float)((acc[1] - l1 * acc[0] * acc[0]) / acc[2]);     // x.Bx = x.Mx - lambda1 (v.x)^2
        conf[b] = lambda1[b] / l2;
    }
}

}  // namespace pdsc

using namespace pdsc;

extern "C" size_t pdsc_sm_workspace_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    const size_t ld = (size_t)pdsc_compat_ld(N);
    return (size_t)bs * N * ld * 4 + (size_t)bs * N * 4 * 3 + (size_t)bs * SMV_MAX_BLOCKS * 4 * 2 + 1024;
}

extern "C" int pdsc_sm_baseline(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                                int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                                void* workspace, size_t workspace_bytes, int bs, int N, void* stream) {
    PDSC_REQUIRE(corr_pos && src_keypts && tgt_keypts && pred_trans && pred_labels && workspace, "pdsc_sm_baseline: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 1 && num_top >= 0 && num_top <= N && num_iterations >= 1, "pdsc_sm_baseline: bs=%d N=%d top=%d iters=%d",
                 bs, N, num_top, num_iterations);
    PDSC_REQUIRE(ceil_div(N, SMV_ROWS) <= SMV_MAX_BLOCKS, "pdsc_sm_baseline: N=%d too large (max %d)", N, SMV_ROWS * SMV_MAX_BLOCKS);
    if (workspace_bytes < pdsc_sm_workspace_bytes(bs, N)) {
        set_error("pdsc_sm_baseline: workspace %zu < %zu bytes", workspace_bytes, pdsc_sm_workspace_bytes(bs, N));
        return PDSC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long ld = pdsc_compat_ld(N);
    float* M = (float*)workspace;
    float* va = M + (size_t)bs * N * ld;
    float* vb = va + (size_t)bs * N;
    float* wts = vb + (size_t)bs * N;
    float* pa = wts + (size_t)bs * N;
    float* pb = pa + (size_t)bs * SMV_MAX_BLOCKS;
    // sigma = inlier_threshold / 3 in double (python floats), sigma ** 2 in double, then the fp32 divisor of a tensor op
    const double sigma = (double)inlier_threshold / 3.0;
    const float sigma2 = (float)(sigma * sigma);
    hipLaunchKernelGGL(sm_matrix_kernel, dim3(ceil_div((int)ld, 256), ceil_div(N, 64), bs), dim3(256), 0, st, corr_pos, sigma2, M, ld, N);
    int rc = check_launch("pdsc_sm_baseline(matrix)");
    if (rc != PDSC_OK) return rc;
    // v0 = 1 (bit pattern of 1.0f)
    rc = launch_fill_u32((unsigned int*)va, 0x3f800000u, (size_t)bs * N, st);      // a kernel, like every other fill of the library (no hipMemsetAsync)
    if (rc != PDSC_OK) return rc;
    rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sm_matvec_kernel), 160 * 1024 - 64, "pdsc_sm_baseline(dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    const int nblocks = ceil_div(N, SMV_ROWS);
    float *vin = va, *vout = vb, *pin = nullptr, *pout = pa;
    for (int it = 0; it < num_iterations; ++it) {
        hipLaunchKernelGGL(sm_matvec_kernel, dim3(nblocks, bs), dim3(256), (size_t)ld * sizeof(float), st, M, ld, vin, pin, nblocks, vout,
                           pout, N);
        rc = check_launch("pdsc_sm_baseline(matvec)");
        if (rc != PDSC_OK) return rc;
        float* tv = vin; vin = vout; vout = tv;
        pin = pout; pout = (pout == pa) ? pb : pa;
    }
    // vin = last y (unnormalised), pin = its partial sums
    float* eig = leading_eig ? leading_eig : vout;
    hipLaunchKernelGGL(sm_finish_kernel, dim3(ceil_div(N, 4), bs), dim3(256), 0, st, vin, pin, nblocks, num_top, eig, pred_labels, wts, N);
    rc = check_launch("pdsc_sm_baseline(finish)");
    if (rc != PDSC_OK) return rc;
    return pdsc_rigid_transform_3d(src_keypts, tgt_keypts, wts, 0.0f, pred_trans, bs, N, stream);
}

extern "C" size_t pdsc_cal_confidence_workspace_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * N * 4 * 2 + (size_t)bs * SMV_MAX_BLOCKS * 4 + (size_t)bs * 16 + 1024;
}

extern "C" int pdsc_cal_confidence(const float* M, long long ld, const float* leading_eig, int method, int num_iterations,
                                   float* confidence, void* workspace, size_t workspace_bytes, int bs, int N, void* stream) {
    PDSC_REQUIRE(M && leading_eig && confidence && workspace, "pdsc_cal_confidence: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && ld >= N && ld % 4 == 0, "pdsc_cal_confidence: bs=%d N=%d ld=%lld (ld >= N, multiple of 4)", bs, N, ld);
    PDSC_REQUIRE(method >= 0 && method <= 2 && num_iterations >= 0, "pdsc_cal_confidence: method=%d iterations=%d", method, num_iterations);
    PDSC_REQUIRE(ceil_div(N, SMV_ROWS) <= SMV_MAX_BLOCKS && (size_t)ld * sizeof(float) <= 160 * 1024 - 64,
                 "pdsc_cal_confidence: N=%d too large", N);
    if (workspace_bytes < pdsc_cal_confidence_workspace_bytes(bs, N)) {
        set_error("pdsc_cal_confidence: workspace %zu < %zu bytes", workspace_bytes, pdsc_cal_confidence_workspace_bytes(bs, N));
        return PDSC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* y = (float*)workspace;
    float* x = y + (size_t)bs * N;
    float* part = x + (size_t)bs * N;
    float* lam = part + (size_t)bs * SMV_MAX_BLOCKS;
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&sm_matvec_kernel), 160 * 1024 - 64, "pdsc_cal_confidence(dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
    }
    const int nblocks = ceil_div(N, SMV_ROWS);
    auto matvec = [&](const float* in) {
        hipLaunchKernelGGL(sm_matvec_kernel, dim3(nblocks, bs), dim3(256), (size_t)ld * sizeof(float), st, M, ld, in, (const float*)nullptr,
                           0, y, part, N);
        return check_launch("pdsc_cal_confidence(matvec)");
    };
    int rc = matvec(leading_eig);
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(cc_rayleigh_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, method, lam, method == 1 ? x : nullptr,
                       confidence, N);
    rc = check_launch("pdsc_cal_confidence(rayleigh)");
    if (rc != PDSC_OK || method != 1) return rc;
    for (int it = 0; it < num_iterations; ++it) {
        rc = matvec(x);
        if (rc != PDSC_OK) return rc;
        hipLaunchKernelGGL(cc_deflate_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, lam, x, N);
        rc = check_launch("pdsc_cal_confidence(deflate)");
        if (rc != PDSC_OK) return rc;
    }
    rc = matvec(x);
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(cc_ratio_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, lam, x, confidence, N);
    return check_launch("pdsc_cal_confidence(ratio)");
}
