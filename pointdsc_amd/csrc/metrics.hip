// f-4: the per-pair evaluation row on the device, so that an evaluation loop needs no
// device -> host copy of labels / poses per pair.
//   reference: libs/loss.py:44-51 (RE, TE, recall flag of TransformationLoss), :96-100 (precision / recall / F1 of
//              ClassificationLoss via sklearn), evaluation/test_3DMatch.py:90-98 (the stats row)
//   stats[b] = { success (re < re_thre && te < te_thre), RE [deg], TE [cm], #gt inliers, gt inlier ratio,
//                #predicted inliers that are gt inliers, precision, recall, F1 }
// One workgroup per pair; counts by block reduction, the 3x3 algebra on thread 0.
#include <math.h>
#include "pdsc_common.h"

namespace pdsc {

__global__ __launch_bounds__(256) void eval_stats_kernel(const float* __restrict__ trans, const float* __restrict__ gt_trans,
                                                         const float* __restrict__ pred_labels, const float* __restrict__ gt_labels,
                                                         float re_thre, float te_thre, float* __restrict__ stats, int N) {
    __shared__ float red[4 * 3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = pred_labels + (size_t)b * N;
    const float* g = gt_labels + (size_t)b * N;
    float cnt[3] = {0.f, 0.f, 0.f};                 // gt positives, predicted positives, true positives (exact in fp32 up to 2^24)
    for (int i = t; i < N; i += 256) {
        const bool gi = g[i] > 0.f, pi = p[i] > 0.f;      // `pred > 0` (:96), gt is 0/1
        cnt[0] += gi ? 1.f : 0.f;
        cnt[1] += pi ? 1.f : 0.f;
        cnt[2] += (gi && pi) ? 1.f : 0.f;
    }
    block_sum<3, 4>(cnt, red);
    if (t != 0) return;
    const float* T = trans + (size_t)b * 16;
    const float* G = gt_trans + (size_t)b * 16;
    // trace(R^T gR) = sum_ij R[i][j] * gR[i][j]
    float tr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) tr = fmaf(T[i * 4 + j], G[i * 4 + j], tr);
    const float c = fminf(fmaxf((tr - 1.0f) / 2.0f, -1.0f), 1.0f);
    const float re = acosf(c) * 180.0f / 3.14159265358979323846f;
    const float dx = T[3] - G[3], dy = T[7] - G[7], dz = T[11] - G[11];
    const float te = sqrtf((dx * dx + dy * dy) + dz * dz) * 100.0f;
    const float gt_pos = cnt[0], pr_pos = cnt[1], tp = cnt[2];
    const float precision = pr_pos > 0.f ? tp / pr_pos : 0.f;         // sklearn: 0 when nothing is predicted positive
    const float recall = gt_pos > 0.f ? tp / gt_pos : 0.f;
    const float denom = 2.f * tp + (pr_pos - tp) + (gt_pos - tp);
    const float f1 = denom > 0.f ? 2.f * tp / denom : 0.f;
    float* s = stats + (size_t)b * 9;
    s[0] = (te < te_thre && re < re_thre) ? 1.f : 0.f;
    s[1] = re; s[2] = te; s[3] = gt_pos; s[4] = gt_pos / (float)N; s[5] = tp; s[6] = precision; s[7] = recall; s[8] = f1;
}

}  // namespace pdsc

extern "C" int pdsc_eval_stats(const float* trans, const float* gt_trans, const float* pred_labels, const float* gt_labels,
                               float re_thre, float te_thre, float* stats, int bs, int N, void* stream) {
    PDSC_REQUIRE(trans && gt_trans && pred_labels && gt_labels && stats, "pdsc_eval_stats: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_eval_stats: bs=%d N=%d", bs, N);
    hipLaunchKernelGGL(pdsc::eval_stats_kernel, dim3(bs), dim3(256), 0, (hipStream_t)stream, trans, gt_trans, pred_labels, gt_labels,
                       re_thre, te_thre, stats, N);
    return pdsc::check_launch("pdsc_eval_stats");
}
