// Shared host/device helpers for libpointdsc_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/pointdsc_hip.h"

#define PDSC_WAVE 64

// the 16-bit element of every split-precision ("x3") operand -- hi or lo part of an fp32 value, split_layout.h -- and the matrix
// instruction that multiplies them
typedef _Float16 sp16;
typedef sp16 sp16x2 __attribute__((ext_vector_type(2)));
typedef sp16 sp16x4 __attribute__((ext_vector_type(4)));
typedef sp16 sp16x8 __attribute__((ext_vector_type(8)));
#define PDSC_MFMA_X3 __builtin_amdgcn_mfma_f32_32x32x16_f16

namespace pdsc {

// ---- host-side error plumbing ---------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define PDSC_REQUIRE(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            pdsc::set_error(__VA_ARGS__);             \
            return PDSC_ERR_ARG;                      \
        }                                             \
    } while (0)

// Dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) of kernel `fn`, done once per (kernel, device) --
// the attribute is per device -- and checked: returns PDSC_OK or PDSC_ERR_LAUNCH with the error text set.  Thread-safe.
int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what);

// buffer := value, as a kernel launch on `st` (NOT hipMemsetAsync: profiles/r03_p_diverge_probe.txt -- with several forwards in
// flight, and above all inside replayed hipGraphs, the zeroing of the hypothesis counters was not reliably ordered before the
// scoring kernel's atomicAdds that follow it on the same stream; a kernel node is)
int launch_fill_u32(unsigned int* p, unsigned int value, size_t count, hipStream_t st);
int launch_copy_u32(unsigned int* dst, const unsigned int* src, size_t count, hipStream_t st);      // dst[i] = src[i], likewise a kernel
// linear.hip: the first two layers of the confidence head in one launch (bit-identical to two pdsc_linear launches)
int launch_classifier_hidden(const float* X, const float* W1, const float* b1, const float* W2, const float* b2, float* H2, int M,
                             hipStream_t st);

// fp16 range sentinel (r06): the split-precision arithmetic carries activations as fp16 hi + lo pairs, so |x| must stay below 65504.
// Every site that converts an activation notes its magnitude (range_note: one v_max3_f32 per two values); a wavefront whose maximum
// reaches the bound sets the word of its pair in `range_flag` ([bs] u32, workspace entry "range_flag": zeroed by the forward's first
// launch, read by its last one, which turns the pose of such a pair into NaN -- a result that is wrong is never returned silently).
// run_forward publishes the array through this thread-local slot for the duration of a call (like layer_nvalid_slot, ragged.h);
// stage-level calls outside a forward leave it NULL and the kernels skip the report.
unsigned int*& range_flag_slot();
constexpr float PDSC_F16_RANGE = 65504.0f;

// opt-in event timing of the roofline kernels (api.hip); no-ops unless pdsc_profile_enable() was called
void profile_mark_begin(int kind, hipStream_t st);
void profile_mark_end(int kind, hipStream_t st);

// Tuning / A-B knobs (PDSC_* environment variables, listed in DESIGN.md) exist in EXPERIMENTS builds only
// (-DPDSC_EXPERIMENTS: libpointdsc_hip_exp.so, built by `python -m pointdsc_amd.build --experiments` for tools/ab_*.py).  In
// the product library env_int / env_str return the default without looking at the environment: nothing outside
// pdsc_config can change what a drop-in module computes or which kernel it runs.
#ifdef PDSC_EXPERIMENTS
static inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline const char* env_str(const char* name) { return getenv(name); }
#else
static inline int env_int(const char*, int dflt) { return dflt; }
static inline const char* env_str(const char*) { return nullptr; }
#endif

// The smallest fp32 x with sqrtf(x) >= thr (host sqrtf is correctly rounded, like the device's): since the correctly
// rounded square root is monotone, `sqrt(x) >= thr` <=> `x >= x*` and `sqrt(x) < thr` <=> `x < x*`, bit for bit -- lets
// the N^2 / S*N predicate loops compare the radicand and skip the ~20-instruction IEEE sqrt.  thr <= 0 -> 0; NaN -> NaN.
static inline float sqrt_threshold_radicand(float thr) {
    float x = 0.f;
    if (thr > 0.f) {
        x = thr * thr;
        while (x > 0.f && sqrtf(nextafterf(x, 0.f)) >= thr) x = nextafterf(x, 0.f);
        while (sqrtf(x) < thr) x = nextafterf(x, INFINITY);
    } else if (thr != thr) {
        x = thr;
    }
    return x;
}

static inline long long round_up(long long x, long long m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers ----------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// running maximum of |x| over the values a lane converts to fp16: ONE v_max3_f32 per two values (|abs| source modifiers; v_max ignores
// NaN operands: a NaN input is not an overflow and propagates on its own, as it does in the reference).  The values are in/out operands
// of the statement although it does not change them: the conversion that follows then depends on it, so the scheduler cannot sink the
// note past the point where the fp32 values die (as plain fmaxf calls it sank them to the end of the pinned pipelines of layer_h3.hip
// and kept every fp32 source alive: 295 spilled registers).
__device__ __forceinline__ void range_note(float& r, float& x0, float& x1) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(r), "+v"(x0), "+v"(x1));
}
__device__ __forceinline__ void range_note(float& r, f32x4& v) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(r), "+v"(v[0]), "+v"(v[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(r), "+v"(v[2]), "+v"(v[3]));
}
// end of a wavefront's work: one store per wavefront and only when the bound was reached (all writers store the same value)
__device__ __forceinline__ void range_report(unsigned int* flag, int pair, float r) {
    if (flag && __builtin_amdgcn_ballot_w64(!(r < PDSC_F16_RANGE)) != 0ull && (threadIdx.x & 63) == 0) flag[pair] = 1u;
}

// Euclidean norm of a 3-vector with torch's CPU/GPU reduction order: sqrt(fma(z,z,fma(y,y,x*x))).
// (oracle/pointdsc_oracle.py:pairwise_dist documents the measurement.)  sqrtf is IEEE-correct
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ float norm3(float x, float y, float z) {
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Wave-wide float sum on the VALU's data-parallel primitives (DPP) instead of six ds_bpermute round trips through the LDS
// crossbar (~100 cycles each): pair swap, quad swap, half-row mirror, row mirror leave every lane of a 16-lane row with
// the row total (both partners of every step add the same two numbers, so the copies are bit-identical); row_bcast:15 /
// row_bcast:31 fold the four rows into lane 63; one v_readlane broadcasts.  Deterministic, but ANOTHER summation order than
// wave_sum above -- used where a wavefront owns a whole small problem (per-seed solver), not where bit patterns are pinned.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define PDSC_DPP_ADD(ctrl, row_mask)                                                                                    \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    PDSC_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
    PDSC_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
    PDSC_DPP_ADD(0x141, 0xf);     // row_half_mirror
    PDSC_DPP_ADD(0x140, 0xf);     // row_mirror
    PDSC_DPP_ADD(0x142, 0xa);     // row_bcast:15 -> rows 1, 3
    PDSC_DPP_ADD(0x143, 0xc);     // row_bcast:31 -> rows 2, 3
#undef PDSC_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum of NV floats per thread; result valid in every thread.  `red` needs NV*nwaves floats.
template <int NV, int NWAVES>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();  // protect `red` from the previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) s += red[w * NV + i];
        v[i] = s;
    }
}

// ---- Kabsch from a 3x3 weighted covariance -----------------------------------------------------
// reference models/common.py:35-42: U,S,V = svd(H); R = V diag(1,1,det(V U^T)) U^T; t = cB - R cA.
// Computed as: eigen-decomposition of H^T H (cyclic Jacobi, fp64) -> V, sigma; u_i = H v_i / sigma_i for the
// two leading directions, u_3 = u_1 x u_2 (so det U = +1), which yields the same R whenever rank(H) >= 2
// (DESIGN.md "3x3 SVD").  Input H row-major fp32, centroids fp32; output row-major 4x4 fp32.
__device__ inline void kabsch_from_covariance(const float* H32, const float* cA, const float* cB, float* T) {
    double H[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[i][j] = (double)H32[i * 3 + j];
    // A = H^T H (symmetric)
    double A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            const double apq = A[p][q];
            if (fabs(apq) < 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // A <- A J
                const double arp = A[r][p], arq = A[r][q];
                A[r][p] = c * arp - s * arq;
                A[r][q] = s * arp + c * arq;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // A <- J^T A
                const double apr = A[p][r], aqr = A[q][r];
                A[p][r] = c * apr - s * aqr;
                A[q][r] = s * apr + c * aqr;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {  // V <- V J
                const double vrp = V[r][p], vrq = V[r][q];
                V[r][p] = c * vrp - s * vrq;
                V[r][q] = s * vrp + c * vrq;
            }
        }
    }
    // order eigenvalues descending: i0 >= i1 >= i2
    int i0 = 0, i1 = 1, i2 = 2;
    double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
    if (e0 < e1) { double t = e0; e0 = e1; e1 = t; int ti = i0; i0 = i1; i1 = ti; }
    if (e0 < e2) { double t = e0; e0 = e2; e2 = t; int ti = i0; i0 = i2; i2 = ti; }
    if (e1 < e2) { double t = e1; e1 = e2; e2 = t; int ti = i1; i1 = i2; i2 = ti; }
    (void)i2;
    // select columns with compares (runtime-indexed local arrays would go to scratch memory)
#define PDSC_SEL3(r, i) ((i) == 0 ? V[r][0] : ((i) == 1 ? V[r][1] : V[r][2]))
    double v1[3] = {PDSC_SEL3(0, i0), PDSC_SEL3(1, i0), PDSC_SEL3(2, i0)};
    double v2[3] = {PDSC_SEL3(0, i1), PDSC_SEL3(1, i1), PDSC_SEL3(2, i1)};
#undef PDSC_SEL3
    // v3 = v1 x v2 makes (v1,v2,v3) right-handed; the sign of v3 is irrelevant below because it enters
    // R only through d * v3 u3^T with d = det(V) det(U) and u3 = u1 x u2 (det U = +1).
    double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    // u1 = H v1 / |H v1|
    double u1[3], u2[3], u3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u1[i] = H[i][0] * v1[0] + H[i][1] * v1[1] + H[i][2] * v1[2];
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (n1 > 1e-150) { u1[0] /= n1; u1[1] /= n1; u1[2] /= n1; }
    else { u1[0] = 1; u1[1] = 0; u1[2] = 0; }              // H == 0: any basis (R not defined by H)
#pragma unroll
    for (int i = 0; i < 3; ++i) u2[i] = H[i][0] * v2[0] + H[i][1] * v2[1] + H[i][2] * v2[2];
    double dp = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
    u2[0] -= dp * u1[0]; u2[1] -= dp * u1[1]; u2[2] -= dp * u1[2];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (n2 > 1e-12 * (n1 > 1e-150 ? n1 : 1.0) && n2 > 1e-150) { u2[0] /= n2; u2[1] /= n2; u2[2] /= n2; }
    else {  // rank <= 1: pick any unit vector orthogonal to u1
        int m = (fabs(u1[0]) <= fabs(u1[1]) && fabs(u1[0]) <= fabs(u1[2])) ? 0 : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
        const double e[3] = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
        double d2 = e[0] * u1[0] + e[1] * u1[1] + e[2] * u1[2];
        u2[0] = e[0] - d2 * u1[0]; u2[1] = e[1] - d2 * u1[1]; u2[2] = e[2] - d2 * u1[2];
        n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        u2[0] /= n2; u2[1] /= n2; u2[2] /= n2;
    }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    // with v3 = v1 x v2 and u3 = u1 x u2 both bases are right-handed: det(V) det(U) = +1, so
    // R = v1 u1^T + v2 u2^T + v3 u3^T is the proper rotation the reference's det-correction selects.
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ti = (double)cB[i] - (R[i][0] * (double)cA[0] + R[i][1] * (double)cA[1] + R[i][2] * (double)cA[2]);
        T[i * 4 + 0] = (float)R[i][0];
        T[i * 4 + 1] = (float)R[i][1];
        T[i * 4 + 2] = (float)R[i][2];
        T[i * 4 + 3] = (float)ti;
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// ---- exact fp32 sqrt / divide-by-invariant without the compiler's special-case scaffolding --------------
// sqrt: hipcc lowers a correctly rounded sqrtf to v_sqrt_f32 followed by exactly this one-ulp test (plus input
// scaling below 2^-96 and a zero/inf class check).  For x == 0 both residual tests fail and 0 is returned.
__device__ __forceinline__ float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x);
    const float ru = fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}

}  // namespace pdsc
