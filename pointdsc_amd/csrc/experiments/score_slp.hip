// Experiments builds only: the scoring kernel of score_kernel.h compiled with the compiler's DEFAULT flags, i.e. with the SLP
// vectoriser pairing the inlier tests of two seeds into packed fp32 instructions (v_pk_mul/fma/add_f32 with op_sel broadcasts).
// That form lost a few votes on one half of the seed pairs while kernels of other forwards were co-resident (DESIGN.md §6
// "Forwards in flight: exactness"); score.hip is therefore built with -fno-slp-vectorize (pointdsc_amd/build.py) and this file
// keeps the failing form selectable (PDSC_SCORE_SLP=1) so that the finding stays reproducible:
//     PDSC_SCORE_SLP=1 PROBE_B=2 PROBE_MODE=plain python tools/inflight_diverge_probe.py
#include "pdsc_common.h"
#include "ragged.h"
#include "score_kernel.h"

namespace pdsc {

#ifdef PDSC_EXPERIMENTS
int launch_score_hypotheses_slp(const float* seed_trans, const float* src, const float* tgt, float thr2, int* counts, int bs, int N, int S,
                                const int* nvalid, hipStream_t st) {
    dim3 grid(ceil_div(S, SC_SEEDS), bs);
    hipLaunchKernelGGL((score_kernel<0, 1>), grid, dim3(256), 0, st, seed_trans, src, tgt, thr2, counts, N, S, nvalid, (float*)nullptr);
    return check_launch("pdsc_score_hypotheses(slp)");
}
#endif

}  // namespace pdsc
