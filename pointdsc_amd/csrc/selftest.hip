// Test infrastructure inside the library: the synthetic "neighbour" kernel of tools/pk_f32_repro.hip (hog mix).
// A wave that interleaves bf16 MFMAs with ordinary vector work (v_exp, shifts, subtracts, LDS reads) makes packed fp32 instructions
// with operand selects (v_pk_*_f32 ... op_sel / op_sel_hi) in CO-RESIDENT waves of other kernels return wrong lanes -- 100 % of the
// launches in the standalone reproducer (profiles/r04_g_pk_f32_repro_mix.txt), against 0 for scalar fp32 and for packed fp32 with
// default selects.  The library ships no such instruction (pointdsc_amd/build.py, tools/isa_audit.py); this entry lets the GPU tests
// keep that neighbour on the chip while they check the library's entry points bit for bit
// (tests/test_gpu_parity.py::test_other_entry_points_stay_exact_beside_attention_launches).  It computes nothing of use.
#include "pdsc_common.h"

namespace pdsc {

typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void selftest_neighbour_kernel(float* __restrict__ sink, int iters) {
    __shared__ float4 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = float4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    st_bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x * 3 + e)); }
    float x = -0.01f * threadIdx.x, s = 0.f;
    unsigned w = 0u;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            const float p = __builtin_amdgcn_exp2f(x);
            const unsigned u = __builtin_bit_cast(unsigned, p) >> 16;
            w ^= u;
            x = p - __builtin_bit_cast(float, u << 16) - 0.5f;
            const float4 v = sm[idx & 1023];
            s += v.x;
            idx += 17;
        }
    }
    if (s + acc[0][0] + acc[1][3] == 123.456f && w == 7u) sink[0] = s;      // (never true: keeps the loop alive)
}

}  // namespace pdsc

// workgroups x 256 threads of the neighbour kernel, `iters` loop iterations each (2500 ~ 1 ms), on `stream`
extern "C" int pdsc_selftest_mfma_valu_neighbour(float* sink, int workgroups, int iters, void* stream) {
    PDSC_REQUIRE(sink && workgroups > 0 && workgroups <= 65536 && iters > 0, "pdsc_selftest_mfma_valu_neighbour: bad argument");
    hipLaunchKernelGGL(pdsc::selftest_neighbour_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, sink, iters);
    return pdsc::check_launch("pdsc_selftest_mfma_valu_neighbour");
}
