// Standalone reproducer for the packed-fp32 vote miscount of r03 (DESIGN.md "Forwards in flight: exactness"): NO PointDSC forward.
//
//   bash tools/pk_f32_repro_build.sh && tools/pk_f32_repro.bin [reps] [path to libpointdsc_hip.so] [comma list of hogs, default all]
//
// What r03 saw: the hypothesis-scoring kernel (csrc/score_kernel.h), in the form the SLP vectoriser gives it (two seeds per
// v_pk_mul/fma/add_f32, the just-loaded point broadcast with op_sel), came out a few votes short on the ODD seed of some pairs
// (the high half of the packed registers) in 0.2-0.7 % of the forwards -- only while kernels of other forwards were co-resident.
// This program launches the SAME kernel source (score_kernel.h is included, not copied) on a fixed synthetic problem whose
// exact counts are computed on the host with the same fma chain, next to synthetic co-resident kernels ("hogs") on other
// streams, and counts launches whose votes differ from the host's.  Forms of the inner loop:
//     scalar   score_kernel<0,0> built with -fno-slp-vectorize (pk_f32_repro_noslp.hip): what the product ships
//     slp      score_kernel<0,1> built with default flags: the compiler's packed form (the r03 failure)
//     pk_asm   hand-written v_pk_*_f32, operands broadcast into register pairs first: packed math, NO op_sel
//     pk_opsel hand-written v_pk_*_f32 reading the loaded point registers through op_sel / op_sel_hi: the compiler's operand form
//     pk_opsel_mov  the same op_sel forms on VALU-written copies of the point (v_mov first): op_sel without a VMEM-written source
// Hogs: none | mfma (bf16 32x32x16 chains, the matrix pipe + power) | valu (packed fma chains) | mem (HBM stream) | lds (ds_read_b128)
//       | att (the product's own split-precision attention launch -- LDS-DMA staging, s_setprio -- on random operands, through the C ABI of
//       libpointdsc_hip.so, if the path is given) | att32 (the exact-fp32 attention launch: fp32 MFMA, ordinary LDS staging)
//       | perm (v_permlane32_swap chains) | exp (v_exp_f32 chains) | ldsdma (buffer_load ... lds streams) | bar (s_barrier + LDS traffic) | cvt (v_cvt_pk_bf16_f32) | cvt_s (the same with an SGPR source) | mix (MFMA + v_exp + SGPR-source conversion + LDS reads in one loop) and mix-<x> (the mix without ingredient x; mix_vcvt: the conversion with VGPR sources only):
//       the ingredients of the split attention kernel one at a time.
// A mismatch table per (form, hog) goes to stdout; exit code 0 always (it is a probe, not a test).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../pointdsc_amd/csrc/score_kernel.h"

namespace pdsc {      // the two symbols pdsc_common.h expects from the library's host side
void set_error(const char*, ...) {}
int check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -1; }
}  // namespace pdsc

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e__ = (x);                                                                 \
        if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

void launch_scalar(const float* T, const float* src, const float* tgt, float thr2, int* counts, int N, int S, hipStream_t st);   // pk_f32_repro_noslp.hip

// ---- hand-written packed forms: one thread = one point, one workgroup = 4 seeds (two packed pairs), same grid as score_kernel ----
template <int OPSEL>
__global__ __launch_bounds__(256) void score_pk_asm_kernel(const float* __restrict__ seed_trans, const float* __restrict__ src,
                                                           const float* __restrict__ tgt, float thr2, int* __restrict__ counts, int N, int S) {
    __shared__ int wsum[4][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, s0 = blockIdx.x * 4;
    int vz = 0;
    asm volatile("" : "+v"(vz));
    // transforms of the two seed pairs, element e of both seeds of a pair side by side: (T_a[e], T_b[e])
    f2 Tp[2][12];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            Tp[p][e].x = seed_trans[(size_t)min(s0 + 2 * p, S - 1) * 16 + e + vz];
            Tp[p][e].y = seed_trans[(size_t)min(s0 + 2 * p + 1, S - 1) * 16 + e + vz];
        }
    int cnt[4] = {0, 0, 0, 0};
    for (int i = t; i < N; i += 256) {
        // the point as the compiler's loop holds it: two dwordx3 loads into consecutive registers
        f2 pxy, pz_, qxy, qz_;
        pxy.x = src[i * 3]; pxy.y = src[i * 3 + 1]; pz_.x = src[i * 3 + 2]; pz_.y = 0.f;
        qxy.x = tgt[i * 3]; qxy.y = tgt[i * 3 + 1]; qz_.x = tgt[i * 3 + 2]; qz_.y = 0.f;
        if constexpr (OPSEL == 2) {
            // copies written by the VALU: the packed instructions below then never read a register a VMEM load wrote
            f2 a, b, c, d;
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(a.x), "=&v"(a.y), "=&v"(b.x), "=&v"(b.y) : "v"(pxy.x), "v"(pxy.y), "v"(pz_.x), "v"(pz_.y));
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(c.x), "=&v"(c.y), "=&v"(d.x), "=&v"(d.y) : "v"(qxy.x), "v"(qxy.y), "v"(qz_.x), "v"(qz_.y));
            pxy = a; pz_ = b; qxy = c; qz_ = d;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f2 r[3];
#pragma unroll
            for (int row = 0; row < 3; ++row) {
                f2 acc;
                if constexpr (OPSEL) {
                    // x = fma(T2, pz, fma(T1, py, T0 * px)) + T3 with px / py / pz taken from (px,py) / (pz,_) by op_sel
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(pxy), "v"(Tp[p][row * 4 + 0]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(Tp[p][row * 4 + 1]), "v"(pxy));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(Tp[p][row * 4 + 2]), "v"(pz_));
                    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "v"(Tp[p][row * 4 + 3]));
                    if (row == 0) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(qxy));
                    if (row == 1) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(qxy));
                    if (row == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(qz_));
                } else {
                    const float pe[3] = {pxy.x, pxy.y, pz_.x};
                    const float qe[3] = {qxy.x, qxy.y, qz_.x};
                    f2 bx = {pe[0], pe[0]}, by = {pe[1], pe[1]}, bz = {pe[2], pe[2]}, bq = {qe[row], qe[row]};
                    asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz), "+v"(bq));        // materialise the broadcast pairs: no op_sel below
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(bx), "v"(Tp[p][row * 4 + 0]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(Tp[p][row * 4 + 1]), "v"(by));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(Tp[p][row * 4 + 2]), "v"(bz));
                    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "v"(Tp[p][row * 4 + 3]));
                    asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(bq));
                }
                r[row] = acc;
            }
            f2 d2;
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d2) : "v"(r[0]));
            asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d2) : "v"(r[1]));
            asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d2) : "v"(r[2]));
            cnt[2 * p] += __popcll(__ballot(d2.x < thr2));
            cnt[2 * p + 1] += __popcll(__ballot(d2.y < thr2));
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (lane == 0) wsum[wave][s] = cnt[s];
    __syncthreads();
    if (t < 4 && s0 + t < S) counts[s0 + t] = wsum[0][t] + wsum[1][t] + wsum[2][t] + wsum[3][t];
}

// ---- what do the wrong lanes compute?  Operands chosen so that every possible mis-selection gives a distinct product: a = (1, 2), b = (10, 100).
//   r0 = v_pk_mul_f32 a, b op_sel_hi:[0,1]   right: (1*10, 1*100) = (10, 100);  select dropped -> hi = 2*100 = 200
//   r1 = v_pk_mul_f32 a, b op_sel:[1,0]      right: (2*10, 2*100) = (20, 200);  select dropped -> lo = 1*10  = 10
//   r2 = v_pk_mul_f32 a, b                   (default selects)                  always (10, 200)
// Every thread repeats the three instructions `iters` times on fresh copies and counts the outcomes that differ from the right ones:
// tally[0..5] = wrong r0.lo, r0.hi, r1.lo, r1.hi, r2.lo, r2.hi;  tally[6 + k] = how often the wrong value was the "select dropped" one
__global__ __launch_bounds__(256) void select_probe_kernel(unsigned long long* __restrict__ tally, int iters) {
    unsigned bad[6] = {0, 0, 0, 0, 0, 0}, dropped[2] = {0, 0};
    for (int it = 0; it < iters; ++it) {
        f2 a = {1.0f, 2.0f}, b = {10.0f, 100.0f}, r0, r1, r2;
        asm volatile("" : "+v"(a), "+v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r0) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r1) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b));
        bad[0] += r0.x != 10.0f; bad[1] += r0.y != 100.0f; bad[2] += r1.x != 20.0f; bad[3] += r1.y != 200.0f;
        bad[4] += r2.x != 10.0f; bad[5] += r2.y != 200.0f;
        dropped[0] += r0.y == 200.0f; dropped[1] += r1.x == 10.0f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (bad[k]) atomicAdd(&tally[k], (unsigned long long)bad[k]);
    if (dropped[0]) atomicAdd(&tally[6], (unsigned long long)dropped[0]);
    if (dropped[1]) atomicAdd(&tally[7], (unsigned long long)dropped[1]);
}

// ... and the victim's actual dependent chain (mul -> fma -> fma -> add -> add with the same operand-select forms), on small integer
// operands so that every mis-selection gives a distinct exact value:  px,py = 1,2  pz = 3  qx = 0.5;  T0 = (10,20) T1 = (100,200)
// T2 = (1000,2000) T3 = (5,7):  right = (10 + 200 + 3000 + 5 - 0.5, 20 + 400 + 6000 + 7 - 0.5) = (3214.5, 6426.5).
// wrong[0] = number of wrong results; wrong[1 + 2k], wrong[2 + 2k] = the k-th wrong (lo, hi) pair as float bits (first 31 kept)
// the same chain keeping a copy of the accumulator pair after every instruction: on a wrong end result the five intermediate pairs of
// the first cases are stored -- which instruction of the chain went wrong, and how
__global__ __launch_bounds__(256) void chain_trace_kernel(unsigned* __restrict__ wrong, float* __restrict__ trace, int iters) {
    for (int it = 0; it < iters; ++it) {
        f2 pxy = {1.0f, 2.0f}, pz_ = {3.0f, 0.0f}, qxy = {0.5f, 0.25f}, T0 = {10.f, 20.f}, T1 = {100.f, 200.f}, T2 = {1000.f, 2000.f}, T3 = {5.f, 7.f}, a1, a2, a3, a4, a5;
        asm volatile("" : "+v"(pxy), "+v"(pz_), "+v"(qxy), "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(a1) : "v"(pxy), "v"(T0));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(a2) : "v"(T1), "v"(pxy), "v"(a1));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(a3) : "v"(T2), "v"(pz_), "v"(a2));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a4) : "v"(T3), "v"(a3));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a5) : "v"(a4), "v"(qxy));
        if (a5.x != 3214.5f || a5.y != 6426.5f) {
            const unsigned k = atomicAdd(&wrong[0], 1u);
            if (k < 4) {
                float* t = trace + k * 10;
                t[0] = a1.x; t[1] = a1.y; t[2] = a2.x; t[3] = a2.y; t[4] = a3.x; t[5] = a3.y; t[6] = a4.x; t[7] = a4.y; t[8] = a5.x; t[9] = a5.y;
            }
        }
    }
}

// NOPS: wait states (s_nop NOPS-1) placed between the dependent packed instructions: 0 = none (back to back, as the compiler emits them)
template <int NOPS>
__global__ __launch_bounds__(256) void chain_probe_kernel(unsigned* __restrict__ wrong, int iters) {
    for (int it = 0; it < iters; ++it) {
        f2 pxy = {1.0f, 2.0f}, pz_ = {3.0f, 0.0f}, qxy = {0.5f, 0.25f}, T0 = {10.f, 20.f}, T1 = {100.f, 200.f}, T2 = {1000.f, 2000.f}, T3 = {5.f, 7.f}, acc;
        asm volatile("" : "+v"(pxy), "+v"(pz_), "+v"(qxy), "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3));
#define PK_GAP() do { if constexpr (NOPS > 0) asm volatile("s_nop %0" :: "n"(NOPS - 1)); } while (0)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(pxy), "v"(T0));
        PK_GAP();
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(T1), "v"(pxy));
        PK_GAP();
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(T2), "v"(pz_));
        PK_GAP();
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "v"(T3));
        PK_GAP();
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(qxy));
#undef PK_GAP
        if (acc.x != 3214.5f || acc.y != 6426.5f) {
            const unsigned k = atomicAdd(&wrong[0], 1u);
            if (k < 31) { wrong[1 + 2 * k] = __builtin_bit_cast(unsigned, acc.x); wrong[2 + 2 * k] = __builtin_bit_cast(unsigned, acc.y); }
        }
    }
}

// ---- hogs ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hog_mfma(float* out, int iters) {
    f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x * 3 + e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void hog_valu(float* out, int iters) {
    f2 a = {1.0f + threadIdx.x * 1e-6f, 0.5f}, b = {0.999f, 1.001f}, c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f2{(float)j, (float)-j};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += c[j].x + c[j].y;
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void hog_mem(const pdsc::f32x4* __restrict__ buf, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const pdsc::f32x4 v = __builtin_nontemporal_load(buf + i);
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void hog_lds(float* out, int iters) {
    __shared__ float4 sm[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = float4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    float s = 0.f;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const float4 v = sm[idx & 2047];
        s += v.x + v.w;
        idx += 257;
    }
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void hog_perm(float* out, int iters) {       // v_permlane32_swap chains (the attention kernel's half_max / chunk_for_store)
    unsigned a = threadIdx.x * 2654435761u, b = threadIdx.x ^ 0x9e3779b9u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a = r[0] + 1u; b = r[1] ^ a;
        }
    }
    if (a == 0x12345u && b == 0x54321u) out[0] = 1.f;
}
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void hog_cvt(float* out, int iters) {        // v_cvt_pk_bf16_f32 chains (the attention kernel's P hi/lo split)
    f2 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = f2{0.001f * (threadIdx.x + j), 1.5f + j};
    unsigned acc = 0u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x2v h = __builtin_convertvector(x[j], bf16x2v);
            const unsigned u = __builtin_bit_cast(unsigned, h);
            acc ^= u;
            x[j].x = __builtin_bit_cast(float, u << 16) * 1.0001f;
            x[j].y = __builtin_bit_cast(float, u & 0xffff0000u) * 0.9999f;
        }
    if (acc == 0x12345u) out[0] = 1.f;
}
// v_cvt_pk_bf16_f32 with an SGPR as its second source: the form the element-wise P split of the attention loop compiled to before
// r04's pairwise split (`v_cvt_pk_bf16_f32 v149, v199, s0`, 16 per tile) -- the library whose attention launch triggers the miscount
// has it, the one that does not has not
__global__ __launch_bounds__(256) void hog_cvt_s(float* out, int iters) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.001f * (threadIdx.x + j) + 1.0f;
    unsigned acc = 0u;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned u;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, s0" : "=v"(u) : "v"(x[j]));
            acc ^= u;
            x[j] = __builtin_bit_cast(float, u << 16) * 1.0001f;
        }
    if (acc == 0x12345u) out[0] = 1.f;
}
// the attention loop's mix in one synthetic kernel: bf16 MFMA chains with v_exp, SGPR-operand conversions, shifts and LDS reads between them.
// Template switches take one ingredient out at a time (CVT: 0 none, 1 both sources VGPRs, 2 second source an SGPR).
template <bool MFMA, bool EXP, int CVT, bool LDS>
__global__ __launch_bounds__(256) void hog_mix(float* out, int iters) {
    __shared__ float4 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = float4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f16v acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x * 3 + e)); }
    float x = -0.01f * threadIdx.x, s = 0.f;
    unsigned w = 0u;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (MFMA) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            float p = x * 0.999f;
            if (EXP) p = __builtin_amdgcn_exp2f(x);
            unsigned u = __builtin_bit_cast(unsigned, p) >> 16;
            if (CVT == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, s0" : "=v"(u) : "v"(p));
            if (CVT == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u) : "v"(p));
            w ^= u;
            x = p - __builtin_bit_cast(float, u << 16) - 0.5f;
            if (LDS) {
                const float4 v = sm[idx & 1023];
                s += v.x;
                idx += 17;
            }
        }
    }
    if (s + acc[0][0] + acc[1][3] == 123.456f && w == 7u) out[0] = s;
}
__global__ __launch_bounds__(256) void hog_exp(float* out, int iters) {        // transcendental unit
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = -0.001f * (threadIdx.x + j);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_exp2f(x[j]) - 1.0f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void hog_ldsdma(const float* __restrict__ buf, size_t bytes, float* out, int iters) {     // buffer_load ... lds (LDS-DMA)
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
    int off = (blockIdx.x * 4 + wave) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(stage[wave] + j * 1024), 16, lane * 16, off + j * 1024, 0, 0);
        off = (off + 1048576) & 0x3fffffff;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (stage[wave][lane] == 77 && iters < 0) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void hog_bar(float* out, int iters) {        // barriers + ds_read/ds_write traffic
    __shared__ float sm[1024];
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        sm[(threadIdx.x + it) & 1023] = s + it;
        __syncthreads();
        s += sm[(threadIdx.x * 7 + it) & 1023];
        __syncthreads();
    }
    if (s == 123.456f) out[0] = s;
}

// ---- host ------------------------------------------------------------------------------------------------------------------
static float frand(unsigned& st) { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); }

static void rot(float ax, float ay, float az, float ang, float* R) {
    const float n = sqrtf(ax * ax + ay * ay + az * az), x = ax / n, y = ay / n, z = az / n, c = cosf(ang), s = sinf(ang), C = 1 - c;
    const float M[9] = {c + x * x * C, x * y * C - z * s, x * z * C + y * s, y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                        z * x * C - y * s, z * y * C + x * s, c + z * z * C};
    memcpy(R, M, sizeof(M));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    const char* libpath = argc > 2 ? argv[2] : nullptr;
    const char* only = argc > 3 ? argv[3] : nullptr;
    const int N = 5000, S = 500, LAUNCHES = 6;       // launches of the form per hog launch (each into its own counter slice)
    unsigned st = 12345u;
    std::vector<float> src(N * 3), tgt(N * 3), T((size_t)S * 16, 0.f);
    float Rg[9];
    rot(0.3f, -0.5f, 0.8f, 0.9f, Rg);
    const float tg[3] = {0.2f, -0.4f, 0.1f};
    for (int i = 0; i < N; ++i) {
        float p[3] = {frand(st) * 3.f, frand(st) * 3.f, frand(st) * 3.f};
        for (int e = 0; e < 3; ++e) src[i * 3 + e] = p[e];
        const bool inl = frand(st) < 0.22f;
        for (int r = 0; r < 3; ++r)
            tgt[i * 3 + r] = inl ? Rg[r * 3] * p[0] + Rg[r * 3 + 1] * p[1] + Rg[r * 3 + 2] * p[2] + tg[r] + (frand(st) - 0.5f) * 0.06f : frand(st) * 3.f;
    }
    for (int s = 0; s < S; ++s) {       // hypotheses around the truth: residuals of the inliers spread across the threshold
        float dR[9], R[9];
        rot(frand(st) - 0.5f, frand(st) - 0.5f, frand(st) - 0.5f, frand(st) * 0.03f, dR);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = dR[r * 3] * Rg[c] + dR[r * 3 + 1] * Rg[3 + c] + dR[r * 3 + 2] * Rg[6 + c];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[(size_t)s * 16 + r * 4 + c] = R[r * 3 + c];
            T[(size_t)s * 16 + r * 4 + 3] = tg[r] + (frand(st) - 0.5f) * 0.05f;
        }
        T[(size_t)s * 16 + 15] = 1.f;
    }
    const float thr2 = pdsc::sqrt_threshold_radicand(0.10f);
    std::vector<int> want(S, 0);
    for (int s = 0; s < S; ++s) {
        const float* t = &T[(size_t)s * 16];
        for (int i = 0; i < N; ++i) {
            const float px = src[i * 3], py = src[i * 3 + 1], pz = src[i * 3 + 2];
            const float x = fmaf(t[2], pz, fmaf(t[1], py, t[0] * px)) + t[3];
            const float y = fmaf(t[6], pz, fmaf(t[5], py, t[4] * px)) + t[7];
            const float z = fmaf(t[10], pz, fmaf(t[9], py, t[8] * px)) + t[11];
            const float dx = x - tgt[i * 3], dy = y - tgt[i * 3 + 1], dz = z - tgt[i * 3 + 2];
            want[s] += fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < thr2;
        }
    }
    int wmin = want[0], wmax = want[0];
    for (int v : want) { wmin = v < wmin ? v : wmin; wmax = v > wmax ? v : wmax; }
    printf("problem: N=%d S=%d, host counts in [%d, %d]; %d reps x %d launches per (form, hog)\n", N, S, wmin, wmax, reps, LAUNCHES);

    float *dsrc, *dtgt, *dT, *dout, *dbig = nullptr;
    int* dcounts;
    CK(hipMalloc(&dsrc, src.size() * 4)); CK(hipMalloc(&dtgt, tgt.size() * 4)); CK(hipMalloc(&dT, T.size() * 4));
    CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dcounts, (size_t)LAUNCHES * S * 4));
    CK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtgt, tgt.data(), tgt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    const size_t big = (size_t)1 << 30;
    CK(hipMalloc(&dbig, big)); CK(hipMemset(dbig, 0, big));
    hipStream_t s_test, s_hog[2];
    CK(hipStreamCreate(&s_test)); CK(hipStreamCreate(&s_hog[0])); CK(hipStreamCreate(&s_hog[1]));

    // the product's attention launch as a hog (C ABI, random operands)
    typedef size_t (*bytes2_t)(int, int);
    typedef size_t (*bytes3_t)(int, int, int);
    typedef long long (*ld_t)(int);
    typedef int (*att_t)(const void*, const void*, const unsigned short*, long long, float*, void*, size_t, int, int, int, void*);
    typedef int (*att32_t)(const float*, const float*, long long, float*, void*, size_t, int, int, int, void*);
    att32_t att32 = nullptr;
    float *a32qkv = nullptr, *a32compat = nullptr;
    void* a32scr = nullptr;
    size_t a32scr_b = 0;
    att_t att = nullptr;
    void *aq = nullptr, *akv = nullptr, *ascr = nullptr, *acompat = nullptr;
    float* amsg = nullptr;
    size_t ascr_b = 0;
    long long ald = 0;
    const int ABS = 2, AN = 5000;
    if (libpath) {
        void* h = dlopen(libpath, RTLD_NOW);
        if (!h) fprintf(stderr, "dlopen(%s): %s\n", libpath, dlerror());
        else {
            att = (att_t)dlsym(h, "pdsc_sc_attention_split_u16");
            const size_t qb = ((bytes2_t)dlsym(h, "pdsc_split_q_bytes"))(ABS, AN), kb = ((bytes2_t)dlsym(h, "pdsc_split_kv_bytes"))(ABS, AN);
            ascr_b = ((bytes3_t)dlsym(h, "pdsc_attention_split_scratch_bytes"))(ABS, AN, 0);
            ald = ((ld_t)dlsym(h, "pdsc_compat_ld"))(AN);
            CK(hipMalloc(&aq, qb)); CK(hipMalloc(&akv, kb)); CK(hipMalloc(&ascr, ascr_b)); CK(hipMalloc(&amsg, (size_t)ABS * AN * 128 * 4));
            CK(hipMalloc(&acompat, (size_t)ABS * AN * ald * 2));
            std::vector<unsigned short> hq(qb / 2), hk(kb / 2), hc((size_t)ABS * AN * ald);
            for (auto& v : hq) v = (unsigned short)(0x3c00 + (int)(frand(st) * 512));      // bf16 bit patterns of moderate magnitude
            for (auto& v : hk) v = (unsigned short)(0x3c00 + (int)(frand(st) * 512));
            for (auto& v : hc) v = frand(st) < 0.1f ? (unsigned short)(frand(st) * 65535) : 0;
            CK(hipMemcpy(aq, hq.data(), qb, hipMemcpyHostToDevice)); CK(hipMemcpy(akv, hk.data(), kb, hipMemcpyHostToDevice));
            CK(hipMemcpy(acompat, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
            att32 = (att32_t)dlsym(h, "pdsc_sc_attention");
            a32scr_b = ((bytes3_t)dlsym(h, "pdsc_attention_scratch_bytes"))(ABS, AN, 0);
            CK(hipMalloc(&a32qkv, (size_t)ABS * AN * 384 * 4)); CK(hipMalloc(&a32compat, (size_t)ABS * AN * ald * 4)); CK(hipMalloc(&a32scr, a32scr_b ? a32scr_b : 16));
            std::vector<float> hqkv((size_t)ABS * AN * 384), hc32((size_t)ABS * AN * ald);
            for (auto& v : hqkv) v = frand(st) - 0.5f;
            for (auto& v : hc32) v = frand(st) < 0.1f ? frand(st) : 0.f;
            CK(hipMemcpy(a32qkv, hqkv.data(), hqkv.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(a32compat, hc32.data(), hc32.size() * 4, hipMemcpyHostToDevice));
        }
    }

    {   // select probe: alone, beside the pure MFMA loop, beside the mix neighbour
        unsigned long long* dtally;
        CK(hipMalloc(&dtally, 8 * sizeof(unsigned long long)));
        const char* names[3] = {"alone", "beside the pure MFMA loop", "beside the mix neighbour (MFMA interleaved with vector work)"};
        for (int mode = 0; mode < 3 && (!only || strstr(only, "probe")); ++mode) {
            CK(hipMemset(dtally, 0, 8 * sizeof(unsigned long long)));
            const int P_REPS = 200, P_WG = 512, P_IT = 2000;
            for (int rep = 0; rep < P_REPS; ++rep) {
                if (mode == 1) hipLaunchKernelGGL(hog_mfma, dim3(512), dim3(256), 0, s_hog[0], dout, 6000);
                if (mode == 2) hipLaunchKernelGGL((hog_mix<true, true, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                hipLaunchKernelGGL(select_probe_kernel, dim3(P_WG), dim3(256), 0, s_test, dtally, P_IT);
                CK(hipDeviceSynchronize());
            }
            unsigned long long t[8];
            CK(hipMemcpy(t, dtally, sizeof(t), hipMemcpyDeviceToHost));
            const double tot = (double)P_REPS * P_WG * 256 * P_IT;
            printf("select probe %-62s: of %.3g evaluations per form -- op_sel_hi:[0,1]: lo wrong %llu, hi wrong %llu (of which = the value with the select DROPPED: %llu); "
                   "op_sel:[1,0]: lo wrong %llu (select dropped: %llu), hi wrong %llu; default selects: lo wrong %llu, hi wrong %llu\n",
                   names[mode], tot, t[0], t[1], t[6], t[2], t[7], t[3], t[4], t[5]);
            fflush(stdout);
        }
        CK(hipFree(dtally));
        unsigned* dwrong;
        CK(hipMalloc(&dwrong, 64 * sizeof(unsigned)));
        for (int nops = 0; nops <= 3; ++nops)
        for (int mode = 0; mode < 3 && (!only || strstr(only, "probe")); ++mode) {
            if (nops > 0 && mode != 2) continue;
            const int gap = nops == 0 ? 0 : nops == 1 ? 1 : nops == 2 ? 2 : 8;
            CK(hipMemset(dwrong, 0, 64 * sizeof(unsigned)));
            const int P_REPS = 200, P_WG = 512, P_IT = 2000;
            for (int rep = 0; rep < P_REPS; ++rep) {
                if (mode == 1) hipLaunchKernelGGL(hog_mfma, dim3(512), dim3(256), 0, s_hog[0], dout, 6000);
                if (mode == 2) hipLaunchKernelGGL((hog_mix<true, true, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (gap == 0) hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(P_WG), dim3(256), 0, s_test, dwrong, P_IT);
                if (gap == 1) hipLaunchKernelGGL(chain_probe_kernel<1>, dim3(P_WG), dim3(256), 0, s_test, dwrong, P_IT);
                if (gap == 2) hipLaunchKernelGGL(chain_probe_kernel<2>, dim3(P_WG), dim3(256), 0, s_test, dwrong, P_IT);
                if (gap == 8) hipLaunchKernelGGL(chain_probe_kernel<8>, dim3(P_WG), dim3(256), 0, s_test, dwrong, P_IT);
                CK(hipDeviceSynchronize());
            }
            unsigned w[64];
            CK(hipMemcpy(w, dwrong, sizeof(w), hipMemcpyDeviceToHost));
            printf("chain probe, %d wait state(s) between the dependent packed instructions, %-62s: %u wrong results of %.3g; first wrong (lo, hi) pairs [right: (3214.5, 6426.5)]:",
                   gap, names[mode], w[0], (double)P_REPS * P_WG * 256 * P_IT);
            for (unsigned k = 0; k < (w[0] < 8 ? w[0] : 8); ++k) printf(" (%g, %g)", __builtin_bit_cast(float, w[1 + 2 * k]), __builtin_bit_cast(float, w[2 + 2 * k]));
            printf("\n");
            fflush(stdout);
        }
        {   // which instruction of the chain goes wrong
            float* dtrace;
            CK(hipMalloc(&dtrace, 40 * sizeof(float)));
            CK(hipMemset(dtrace, 0, 40 * sizeof(float)));
            CK(hipMemset(dwrong, 0, 64 * sizeof(unsigned)));
            for (int rep = 0; rep < 200 && (!only || strstr(only, "probe")); ++rep) {
                hipLaunchKernelGGL((hog_mix<true, true, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                hipLaunchKernelGGL(chain_trace_kernel, dim3(512), dim3(256), 0, s_test, dwrong, dtrace, 2000);
                CK(hipDeviceSynchronize());
            }
            unsigned w0;
            float tr[40];
            CK(hipMemcpy(&w0, dwrong, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(tr, dtrace, sizeof(tr), hipMemcpyDeviceToHost));
            printf("chain trace (every intermediate in its own register pair) beside the mix neighbour: %u wrong end results; right intermediates: "
                   "mul (10, 20) -> fma (210, 420) -> fma (3210, 6420) -> add (3215, 6427) -> add (3214.5, 6426.5)\n", w0);
            for (unsigned k = 0; k < (w0 < 4 ? w0 : 4); ++k)
                printf("    wrong case %u: mul (%g, %g) -> fma (%g, %g) -> fma (%g, %g) -> add (%g, %g) -> add (%g, %g)\n", k, tr[k * 10], tr[k * 10 + 1], tr[k * 10 + 2],
                       tr[k * 10 + 3], tr[k * 10 + 4], tr[k * 10 + 5], tr[k * 10 + 6], tr[k * 10 + 7], tr[k * 10 + 8], tr[k * 10 + 9]);
            CK(hipFree(dtrace));
        }
        CK(hipFree(dwrong));
    }
    const char* forms[] = {"scalar", "slp", "pk_asm", "pk_opsel", "pk_opsel_mov"};
    const char* hogs[] = {"none", "mfma", "valu", "mem", "lds", "mfma+mem", "att", "att32", "perm", "exp", "ldsdma", "bar", "cvt", "cvt_s", "mix", "mix-mfma", "mix-exp", "mix-cvt", "mix_vcvt", "mix-lds"};
    std::vector<int> got((size_t)LAUNCHES * S);
    for (int hg = 0; hg < 20; ++hg) {
        if ((hg == 6 && !att) || (hg == 7 && !att32)) continue;
        if (only) {      // exact token match in the comma list
            const size_t L = strlen(hogs[hg]);
            bool hit = false;
            for (const char* q = only; (q = strstr(q, hogs[hg])) != nullptr; q += L)
                if ((q == only || q[-1] == ',') && (q[L] == 0 || q[L] == ',')) { hit = true; break; }
            if (!hit) continue;
        }
        for (int f = 0; f < 5; ++f) {
            long bad_launches = 0, bad_seeds = 0, bad_odd = 0, short_votes = 0, over_votes = 0, launches = 0;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s_test));
            for (int rep = 0; rep < reps; ++rep) {
                // hog(s) first (about 1 ms each, 512 workgroups of 256 threads: two per CU, room left for the scoring workgroups)
                if (hg == 1 || hg == 5) hipLaunchKernelGGL(hog_mfma, dim3(512), dim3(256), 0, s_hog[0], dout, 6000);
                if (hg == 2) hipLaunchKernelGGL(hog_valu, dim3(512), dim3(256), 0, s_hog[0], dout, 6000);
                if (hg == 3 || hg == 5) hipLaunchKernelGGL(hog_mem, dim3(1024), dim3(256), 0, s_hog[1], (const pdsc::f32x4*)dbig, big / 16, dout);
                if (hg == 4) hipLaunchKernelGGL(hog_lds, dim3(512), dim3(256), 0, s_hog[0], dout, 60000);
                if (hg == 8) hipLaunchKernelGGL(hog_perm, dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 9) hipLaunchKernelGGL(hog_exp, dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 10) hipLaunchKernelGGL(hog_ldsdma, dim3(512), dim3(256), 0, s_hog[0], (const float*)dbig, big, dout, 40);
                if (hg == 13) hipLaunchKernelGGL(hog_cvt_s, dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 14) hipLaunchKernelGGL((hog_mix<true, true, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 15) hipLaunchKernelGGL((hog_mix<false, true, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 16) hipLaunchKernelGGL((hog_mix<true, false, 2, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 17) hipLaunchKernelGGL((hog_mix<true, true, 0, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 18) hipLaunchKernelGGL((hog_mix<true, true, 1, true>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 19) hipLaunchKernelGGL((hog_mix<true, true, 2, false>), dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 12) hipLaunchKernelGGL(hog_cvt, dim3(512), dim3(256), 0, s_hog[0], dout, 2500);
                if (hg == 11) hipLaunchKernelGGL(hog_bar, dim3(512), dim3(256), 0, s_hog[0], dout, 1500);
                if (hg == 7 && att32(a32qkv, a32compat, ald, amsg, a32scr, a32scr_b, ABS, AN, 0, s_hog[0]) != 0) { fprintf(stderr, "fp32 attention hog failed\n"); att32 = nullptr; break; }
                if (hg == 6 && att(aq, akv, (const unsigned short*)acompat, ald, amsg, ascr, ascr_b, ABS, AN, 0, s_hog[0]) != 0) { fprintf(stderr, "attention hog failed\n"); att = nullptr; break; }
                for (int l = 0; l < LAUNCHES; ++l) {
                    int* c = dcounts + (size_t)l * S;
                    const dim3 grid((S + 3) / 4, 1);
                    if (f == 0) launch_scalar(dT, dsrc, dtgt, thr2, c, N, S, s_test);
                    if (f == 1) hipLaunchKernelGGL((pdsc::score_kernel<0, 1>), grid, dim3(256), 0, s_test, dT, dsrc, dtgt, thr2, c, N, S, (const int*)nullptr, (float*)nullptr);
                    if (f == 2) hipLaunchKernelGGL(score_pk_asm_kernel<0>, grid, dim3(256), 0, s_test, dT, dsrc, dtgt, thr2, c, N, S);
                    if (f == 3) hipLaunchKernelGGL(score_pk_asm_kernel<1>, grid, dim3(256), 0, s_test, dT, dsrc, dtgt, thr2, c, N, S);
                    if (f == 4) hipLaunchKernelGGL(score_pk_asm_kernel<2>, grid, dim3(256), 0, s_test, dT, dsrc, dtgt, thr2, c, N, S);
                }
                CK(hipMemcpyAsync(got.data(), dcounts, got.size() * 4, hipMemcpyDeviceToHost, s_test));
                CK(hipStreamSynchronize(s_test));
                for (int l = 0; l < LAUNCHES; ++l) {
                    int nb = 0;
                    for (int s = 0; s < S; ++s) {
                        const int d = got[(size_t)l * S + s] - want[s];
                        if (d) { ++nb; bad_odd += s & 1; if (d < 0) short_votes -= d; else over_votes += d; }
                    }
                    bad_launches += nb > 0;
                    bad_seeds += nb;
                    ++launches;
                }
                CK(hipDeviceSynchronize());
            }
            CK(hipEventRecord(e1, s_test));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("hog %-9s form %-9s: %ld of %ld launches differ from the host counts; seeds off %ld (odd index %ld), votes short %ld, votes over %ld   [%.0f ms]\n",
                   hogs[hg], forms[f], bad_launches, launches, bad_seeds, bad_odd, short_votes, over_votes, ms);
            fflush(stdout);
        }
    }
    return 0;
}
