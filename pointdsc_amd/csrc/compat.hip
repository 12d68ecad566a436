// a-1: N x N spatial-consistency matrix build (reference models/PointDSC.py:150-153).
//
// HBM-write bound: 4*N*ld bytes out, 24*N bytes in.  One workgroup produces a 64-row x 256-column tile;
// every lane owns 4 consecutive columns (their 8 keypoints live in registers for the whole tile) and walks
// 16 rows whose keypoints are broadcast from LDS, so each store instruction is one fully coalesced 1 KiB
// row segment (float4 per lane).  Arithmetic is the reference's, bit for bit: the fma-chained norm of
// torch.norm, an IEEE division by sigma^2 and the clamp.
#include <stdlib.h>
#include "pdsc_common.h"

namespace pdsc {

constexpr int CT_ROWS = 64;    // rows per workgroup tile
constexpr int CT_COLS = 256;   // columns per workgroup tile (64 lanes x float4)

// CHEAP = timing probe only (tools/kernel_microbench.py): skips the sqrt/divide so that the store path can be
// measured in isolation; never used by the product path.
template <bool WRITE_DIST, bool CHEAP = false>
__global__ __launch_bounds__(256) void compat_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                     const float* __restrict__ sigma_spat,
                                                     float* __restrict__ compat, float* __restrict__ src_dist,
                                                     long long ld, int N) {
    __shared__ float rows_s[CT_ROWS][8];   // sx sy sz - tx ty tz -
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * CT_ROWS;
    const int j0 = blockIdx.x * CT_COLS;
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    const int t = threadIdx.x;
    if (t < CT_ROWS) {
        const int i = min(i0 + t, N - 1);
        rows_s[t][0] = srcb[i * 3 + 0]; rows_s[t][1] = srcb[i * 3 + 1]; rows_s[t][2] = srcb[i * 3 + 2];
        rows_s[t][4] = tgtb[i * 3 + 0]; rows_s[t][5] = tgtb[i * 3 + 1]; rows_s[t][6] = tgtb[i * 3 + 2];
    }
    const int lane = t & 63, wave = t >> 6;
    const int jc = j0 + lane * 4;
    float sx[4], sy[4], sz[4], tx[4], ty[4], tz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = min(jc + c, N - 1);
        sx[c] = srcb[j * 3 + 0]; sy[c] = srcb[j * 3 + 1]; sz[c] = srcb[j * 3 + 2];
        tx[c] = tgtb[j * 3 + 0]; ty[c] = tgtb[j * 3 + 1]; tz[c] = tgtb[j * 3 + 2];
    }
    const float sg = sigma_spat[0];
    const float s2 = sg * sg;                        // `self.sigma_spat ** 2` in fp32
    __syncthreads();
    if (jc >= ld) return;
    float* outb = compat + (size_t)b * N * ld;
    float* distb = WRITE_DIST ? src_dist + (size_t)b * N * ld : nullptr;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int il = wave * 16 + r;
        const int i = i0 + il;
        if (i >= N) break;
        const f32x4 ps = *reinterpret_cast<const f32x4*>(&rows_s[il][0]);
        const f32x4 pt = *reinterpret_cast<const f32x4*>(&rows_s[il][4]);
        f32x4 o, dd;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ds, v;
            if (CHEAP) {
                ds = ps[0] - sx[c];
                v = fmaxf(ds + (pt[0] - tx[c]) * s2, 0.0f);
            } else {
                ds = norm3(ps[0] - sx[c], ps[1] - sy[c], ps[2] - sz[c]);
                const float dt = norm3(pt[0] - tx[c], pt[1] - ty[c], pt[2] - tz[c]);
                const float df = ds - dt;
                v = fmaxf(1.0f - (df * df) / s2, 0.0f);
            }
            const bool valid = (jc + c) < N;
            o[c] = valid ? v : 0.0f;
            dd[c] = valid ? ds : 0.0f;
        }
        *reinterpret_cast<f32x4*>(outb + (size_t)i * ld + jc) = o;
        if (WRITE_DIST) *reinterpret_cast<f32x4*>(distb + (size_t)i * ld + jc) = dd;
    }
}

// ---- symmetric variant ----------------------------------------------------------------------------
// compat[i][j] == compat[j][i] bit for bit (negating a difference does not change its square), so only the
// tiles on and above the diagonal are computed (half the VALU work: 2 IEEE sqrt + 1 IEEE divide per element
// is what bounds the plain kernel, not HBM); every off-diagonal 128x128 tile is written twice: directly
// (float4 rows) and transposed through LDS (ds_read_b128 along j, dword stores of 64 consecutive i = 256 B).
constexpr int CS_T = 128;
constexpr int CS_LD = CS_T + 4;

template <bool SKIP_TRANSPOSE = false>   // true = timing probe only (wrong lower triangle), never shipped
__global__ __launch_bounds__(256, 2) void compat_sym_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                            const float* __restrict__ sigma_spat, float* __restrict__ compat,
                                                            long long ld, int N) {
    const int I = blockIdx.y, J = blockIdx.x, b = blockIdx.z;
    if (I > J) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ts = lds;                                  // [128][CS_LD] this tile, row-major
    float* rows_s = lds + CS_T * CS_LD;               // [128][8]
    const int i0 = I * CS_T, j0 = J * CS_T;
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    const int t = threadIdx.x;
    if (t < CS_T) {
        const int i = min(i0 + t, N - 1);
        rows_s[t * 8 + 0] = srcb[i * 3 + 0]; rows_s[t * 8 + 1] = srcb[i * 3 + 1]; rows_s[t * 8 + 2] = srcb[i * 3 + 2];
        rows_s[t * 8 + 4] = tgtb[i * 3 + 0]; rows_s[t * 8 + 5] = tgtb[i * 3 + 1]; rows_s[t * 8 + 6] = tgtb[i * 3 + 2];
    }
    const int l32 = t & 31, rg = t >> 5;
    const int jc = j0 + l32 * 4;
    float sx[4], sy[4], sz[4], tx[4], ty[4], tz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = min(jc + c, N - 1);
        sx[c] = srcb[j * 3 + 0]; sy[c] = srcb[j * 3 + 1]; sz[c] = srcb[j * 3 + 2];
        tx[c] = tgtb[j * 3 + 0]; ty[c] = tgtb[j * 3 + 1]; tz[c] = tgtb[j * 3 + 2];
    }
    const float sg = sigma_spat[0];
    const float s2 = sg * sg;
    float* outb = compat + (size_t)b * N * ld;
    __syncthreads();
    const bool offdiag = I != J;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int il = rg + 8 * r;
        const int i = i0 + il;
        const f32x4 ps = *reinterpret_cast<const f32x4*>(rows_s + il * 8);
        const f32x4 pt = *reinterpret_cast<const f32x4*>(rows_s + il * 8 + 4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ds = norm3(ps[0] - sx[c], ps[1] - sy[c], ps[2] - sz[c]);
            const float dt = norm3(pt[0] - tx[c], pt[1] - ty[c], pt[2] - tz[c]);
            const float df = ds - dt;
            const float v = fmaxf(1.0f - (df * df) / s2, 0.0f);
            o[c] = (jc + c) < N ? v : 0.0f;
        }
        if (i < N && jc < ld) *reinterpret_cast<f32x4*>(outb + (size_t)i * ld + jc) = o;
        if (offdiag) *reinterpret_cast<f32x4*>(Ts + il * CS_LD + 4 * l32) = o;
    }
    if (!offdiag || SKIP_TRANSPOSE) return;
    __syncthreads();
    // transposed copy: element (i,j) of this tile -> compat[j][i]
    const int il = t & 127, jr = t >> 7;
    const int i = i0 + il;                              // always < N here? no: the last row tile can be ragged
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int jq = 2 * p + jr;                      // group of 4 output rows
        const f32x4 v = *reinterpret_cast<const f32x4*>(Ts + il * CS_LD + 4 * jq);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + 4 * jq + c;
            if (j < N && i < N) outb[(size_t)j * ld + i] = v[c];
        }
    }
}

}  // namespace pdsc

extern "C" long long pdsc_compat_ld(int N) { return N <= 0 ? -1 : pdsc::round_up(N, 64); }

extern "C" int pdsc_spatial_compat(const float* src, const float* tgt, const float* sigma_spat, float* compat,
                                   float* src_dist, long long ld, int bs, int N, void* stream) {
    PDSC_REQUIRE(src && tgt && sigma_spat && compat, "pdsc_spatial_compat: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_spatial_compat: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= N && ld % 4 == 0, "pdsc_spatial_compat: ld=%lld must be >= N and a multiple of 4", ld);
    dim3 grid(pdsc::ceil_div((int)ld, pdsc::CT_COLS), pdsc::ceil_div(N, pdsc::CT_ROWS), bs);
    hipStream_t st = (hipStream_t)stream;
    static int variant = -1;
    const size_t sym_lds = (size_t)(pdsc::CS_T * pdsc::CS_LD + pdsc::CS_T * 8) * sizeof(float);
    if (variant < 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pdsc::compat_sym_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sym_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pdsc::compat_sym_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sym_lds);
        // tuning/A-B knob: 0 = full tiles, 1 = symmetric (shipped); 2, 3 = TIMING PROBES that produce wrong values
        // (2: symmetric without the transposed stores, 3: full tiles without sqrt/divide)
        const char* env = getenv("PDSC_COMPAT_VARIANT");
        variant = env ? atoi(env) : 1;
    }
    pdsc::profile_mark_begin(PDSC_PROF_COMPAT, st);
    if ((variant == 1 || variant == 2) && !src_dist) {
        const int nt = pdsc::ceil_div(N, pdsc::CS_T);
        if (variant == 1)
            hipLaunchKernelGGL(pdsc::compat_sym_kernel<false>, dim3(nt, nt, bs), dim3(256), sym_lds, st, src, tgt, sigma_spat, compat, ld, N);
        else
            hipLaunchKernelGGL(pdsc::compat_sym_kernel<true>, dim3(nt, nt, bs), dim3(256), sym_lds, st, src, tgt, sigma_spat, compat, ld, N);
    } else if (variant == 3 && !src_dist) {
        hipLaunchKernelGGL((pdsc::compat_kernel<false, true>), grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, nullptr, ld, N);
    } else if (src_dist)
        hipLaunchKernelGGL(pdsc::compat_kernel<true>, grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, src_dist, ld, N);
    else
        hipLaunchKernelGGL(pdsc::compat_kernel<false>, grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, nullptr, ld, N);
    pdsc::profile_mark_end(PDSC_PROF_COMPAT, st);
    return pdsc::check_launch("pdsc_spatial_compat");
}
