// tools/pk_f32_repro.hip, second translation unit: the scoring kernel of csrc/score_kernel.h built with -fno-slp-vectorize
// (scalar fp32 inner loop) -- the form the product library ships (pointdsc_amd/build.py).
#include "../pointdsc_amd/csrc/score_kernel.h"

void launch_scalar(const float* T, const float* src, const float* tgt, float thr2, int* counts, int N, int S, hipStream_t st) {
    hipLaunchKernelGGL((pdsc::score_kernel<0, 0>), dim3((S + 3) / 4, 1), dim3(256), 0, st, T, src, tgt, thr2, counts, N, S, (const int*)nullptr,
                       (float*)nullptr);
}
