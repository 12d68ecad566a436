// a-1: N x N spatial-consistency matrix build (reference models/PointDSC.py:150-153).
//
//   compat[i][j] = max(0, 1 - (||s_i - s_j|| - ||t_i - t_j||)^2 / sigma_spat^2)
//
// Algorithmic traffic: 4*N*ld bytes out, 24*N bytes in -> HBM-write roofline.  What actually bounds a naive
// version is the VALU: two correctly rounded square roots and one correctly rounded divide per element (hipcc's
// IEEE sequences: ~60 VALU ops incl. v_div_scale/fmas/fixup and their VCC wait states); measured on MI355X
// the store path alone sustains 5.3 TB/s while the IEEE-math kernel reached 2.3 TB/s.  Two levers, both exact:
//   * hand-rolled exact math: the same algorithms hipcc's IEEE lowering uses (v_sqrt_f32 + one-ulp up/down
//     residual test; reciprocal refinement + two residual corrections for the divide) without the denormal
//     pre-scaling / class fix-ups that cannot trigger here, and with the reciprocal of sigma^2 hoisted out;
//   * symmetry: compat[i][j] == compat[j][i] bit for bit (negating a difference does not change its square),
//     so only tiles on/above the diagonal are computed and each off-diagonal tile is also written transposed
//     (through a 32-row LDS strip, 128-byte row segments).
// The arithmetic is the reference's to the bit: torch.norm evaluates sqrt(fma(dz,dz,fma(dy,dy,dx*dx))) (measured,
// oracle/pointdsc_oracle.py), then an IEEE division by sigma^2 (fp32 square) and the clamp.
#include <stdlib.h>
#include "pdsc_common.h"

#define PDSC_COMPAT_DEFAULT_VARIANT 3

namespace pdsc {

// (sqrt_rn: pdsc_common.h)
struct InvariantDivisor {   // divide by the same b many times: hipcc's v_rcp + refinement, hoisted
    float b, y;
    __device__ __forceinline__ explicit InvariantDivisor(float b_) : b(b_) {
        const float y0 = __builtin_amdgcn_rcpf(b_);
        const float e = fmaf(-b_, y0, 1.0f);
        y = fmaf(e, y0, y0);
    }
    // a / b: the residual-corrected quotient of the IEEE sequence (v_div_scale/v_div_fixup only matter for
    // operands near the exponent limits; a tiny or zero quotient is absorbed by the following `1 - q`)
    __device__ __forceinline__ float div(float a) const {
        const float q0 = a * y;
        const float r0 = fmaf(-b, q0, a);
        const float q1 = fmaf(r0, y, q0);
        const float r1 = fmaf(-b, q1, a);
        return fmaf(r1, y, q1);
    }
};

template <bool FASTMATH>
__device__ __forceinline__ float dist3(float x, float y, float z) {
    const float ss = fmaf(z, z, fmaf(y, y, x * x));
    return FASTMATH ? sqrt_rn(ss) : sqrtf(ss);
}
template <bool FASTMATH>
__device__ __forceinline__ float compat_from_dists(float ds, float dt, float s2, const InvariantDivisor& inv) {
    const float df = ds - dt;
    const float q = FASTMATH ? inv.div(df * df) : (df * df) / s2;
    return fmaxf(1.0f - q, 0.0f);
}

// ---- full-tile kernel: 64 rows x 256 columns per workgroup, float4 row segments ---------------------------
constexpr int CT_ROWS = 64;
constexpr int CT_COLS = 256;

template <bool WRITE_DIST, bool FASTMATH>
__global__ __launch_bounds__(256) void compat_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                     const float* __restrict__ sigma_spat,
                                                     float* __restrict__ compat, float* __restrict__ src_dist,
                                                     long long ld, int N) {
    __shared__ __attribute__((aligned(16))) float rows_s[CT_ROWS][8];   // sx sy sz - tx ty tz -
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
    const InvariantDivisor inv(s2);
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
            const float ds = dist3<FASTMATH>(ps[0] - sx[c], ps[1] - sy[c], ps[2] - sz[c]);
            const float dt = dist3<FASTMATH>(pt[0] - tx[c], pt[1] - ty[c], pt[2] - tz[c]);
            const float v = compat_from_dists<FASTMATH>(ds, dt, s2, inv);
            const bool valid = (jc + c) < N;
            o[c] = valid ? v : 0.0f;
            dd[c] = valid ? ds : 0.0f;
        }
        *reinterpret_cast<f32x4*>(outb + (size_t)i * ld + jc) = o;
        if (WRITE_DIST) *reinterpret_cast<f32x4*>(distb + (size_t)i * ld + jc) = dd;
    }
}

// ---- symmetric kernel: 128x128 tiles on/above the diagonal, transposed copy through a 32-row LDS strip ------
constexpr int CS_T = 128;
constexpr int CS_STRIP = 32;
constexpr int CS_LD = CS_T + 4;

template <bool FASTMATH>
__global__ __launch_bounds__(256) void compat_sym_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                         const float* __restrict__ sigma_spat, float* __restrict__ compat,
                                                         long long ld, int N) {
    const int I = blockIdx.y, J = blockIdx.x, b = blockIdx.z;
    if (I > J) return;
    __shared__ __attribute__((aligned(16))) float Ts[CS_STRIP * CS_LD];   // one 32 x 128 strip, row-major
    __shared__ __attribute__((aligned(16))) float rows_s[CS_T * 8];
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
    const InvariantDivisor inv(s2);
    float* outb = compat + (size_t)b * N * ld;
    const bool offdiag = I != J;
    __syncthreads();
    for (int strip = 0; strip < CS_T / CS_STRIP; ++strip) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = rg + 8 * r;                      // row inside the strip
            const int il = strip * CS_STRIP + sl;
            const int i = i0 + il;
            const f32x4 ps = *reinterpret_cast<const f32x4*>(rows_s + il * 8);
            const f32x4 pt = *reinterpret_cast<const f32x4*>(rows_s + il * 8 + 4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ds = dist3<FASTMATH>(ps[0] - sx[c], ps[1] - sy[c], ps[2] - sz[c]);
                const float dt = dist3<FASTMATH>(pt[0] - tx[c], pt[1] - ty[c], pt[2] - tz[c]);
                const float v = compat_from_dists<FASTMATH>(ds, dt, s2, inv);
                o[c] = (jc + c) < N ? v : 0.0f;
            }
            if (i < N && jc < ld) *reinterpret_cast<f32x4*>(outb + (size_t)i * ld + jc) = o;
            if (offdiag) *reinterpret_cast<f32x4*>(Ts + sl * CS_LD + 4 * l32) = o;
        }
        if (offdiag) {                                       // block-uniform
            __syncthreads();
            // element (i, j) of the strip -> compat[j][i]: lane = i (32 consecutive floats = 128 B per output row)
            const int sl = t & 31, jq0 = t >> 5;
            const int i = i0 + strip * CS_STRIP + sl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int jq = jq0 + 8 * p;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Ts + sl * CS_LD + 4 * jq);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = j0 + 4 * jq + c;
                    if (j < N && i < N) outb[(size_t)j * ld + i] = v[c];
                }
            }
            __syncthreads();
        }
    }
}

// ---- unorm16 variant for the split-precision attention (the only consumer of the matrix in the forward) -----------
// value u = round(c * 65535) (v_cvt_pknorm_u16_f32), stored in the attention kernel's tile order: inside every group of 32
// keys, key 8g + 4h + e sits at position 16h + 4g + e -- the 16 keys a lane half of the S^T accumulator holds are contiguous
// (attention_split.hip).  A 16-byte chunk q of a group (positions 8q .. 8q+7) therefore holds the keys
// 16(q&1) + 4(q>>1) + {0,1,2,3,8,9,10,11}.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ int c16_pos(int r) { return 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3); }   // r in [0, 32)
__host__ __device__ __forceinline__ int c16_chunk_base(int q) { return 16 * (q & 1) + 4 * (q >> 1); }            // first key of chunk q

// r03: the shipped kernel.  The r02 kernel (below, experiments builds) paid the correctly rounded sqrt / divide of the fp32
// matrix (~50 VALU instructions per element: instruction-bound at 0.38 of the HBM rate, PMC r02_c) for a result that is then
// rounded to 16 bits.  Here c is evaluated with the hardware's 1-ulp v_sqrt_f32 and a multiplication by 1/sigma^2 on packed
// fp32 math (v_pk_*_f32: two elements per instruction): ~16 issue slots per element.  |c - c_fp32| <= 2 (ulp(d_src) +
// ulp(d_tgt)) |d_src - d_tgt| / sigma^2 < 2e-5 at the clamp (|d_src - d_tgt| -> sigma), -> 0 towards c = 1: the stored value
// is within 2 units of round(c_fp32 * 65535), equal to it for ~99 % of the non-zero entries (tests); the diagonal is
// exactly 65535, the matrix is symmetric bit for bit (each unordered pair is evaluated once).
// Work decomposition: one workgroup = one 128 x 128 tile on / above the diagonal, one wavefront = a 64 x 64 quarter, one
// lane = an 8 x 8 micro-block whose rows AND columns are the key sets of a 16-byte chunk -- so the lane holds, in
// registers, whole chunks of 8 rows (direct tile) and of 8 columns (the mirrored tile): no LDS transposition, and every
// store instruction of a wave writes 8 x 128 whole, aligned bytes in both orientations.
constexpr int CF_T = 128;
__device__ __forceinline__ int cf_point_slot(int r) { return (r + (r >> 4)) * 8; }      // floats; one pad point per 16: conflict-free b128 reads

template <bool EDGE, bool CLAMP>
__device__ __forceinline__ void compat_fast_tile(const float* __restrict__ pts, unsigned short* __restrict__ outb, long long ld,
                                                 int N, int i0, int j0, bool offdiag, float ninv) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ri = (w >> 1) * 8 + (lane >> 3), cj = (w & 1) * 8 + (lane & 7);       // chunk index of the rows / columns (0..15)
    const int rbase = 32 * (ri >> 2) + c16_chunk_base(ri & 3);                      // first of this lane's 8 tile rows
    const int cbase = 32 * (cj >> 2) + c16_chunk_base(cj & 3);
    // columns in registers, as pairs (m, m+1)
    f32x2 sx[4], sy[4], sz[4], tx[4], ty[4], tz[4];
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
        const int m0 = 2 * cp, c0 = cbase + (m0 & 3) + 8 * (m0 >> 2);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(CF_T + c0));
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(CF_T + c0) + 4);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(CF_T + c0 + 1));
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(CF_T + c0 + 1) + 4);
        sx[cp] = f32x2{a0[0], a1[0]}; sy[cp] = f32x2{a0[1], a1[1]}; sz[cp] = f32x2{a0[2], a1[2]};
        tx[cp] = f32x2{b0[0], b1[0]}; ty[cp] = f32x2{b0[1], b1[1]}; tz[cp] = f32x2{b0[2], b1[2]};
    }
    unsigned tr[8][4];                            // mirrored tile: chunk of column m = rows (2rp, 2rp+1) packed in word rp
    // r04: NO operand selects on the packed fp32 instructions.  v_pk_*_f32 with a non-default op_sel / op_sel_hi (a scalar or an
    // inline constant broadcast into both halves, which is what `f32x2{x, x} - v` compiles to) returned wrong lanes whenever the
    // wave shared a CU with the split attention kernel (tools/pk_f32_repro.hip, profiles/r04_pk_f32_repro*.txt: 16-24 % of the
    // launches; never with default selects).  Every broadcast therefore lives in a real register PAIR (materialised through an
    // empty asm, so the compiler cannot fold it back into an operand select): constants once per lane, the row point once per row.
    f32x2 one2 = {1.0f, 1.0f}, ninv2 = {ninv, ninv}, zero2 = {0.f, 0.f};
    asm volatile("" : "+v"(one2), "+v"(ninv2), "+v"(zero2));
    const int colpos = j0 + 32 * (cj >> 2) + 8 * (cj & 3);                           // tile-order position of this lane's column chunk
    const int rowpos = i0 + 32 * (ri >> 2) + 8 * (ri & 3);
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
        const int m0 = 2 * rp, r0 = rbase + (m0 & 3) + 8 * (m0 >> 2);
        const f32x4 ps0 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(r0));
        const f32x4 pt0 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(r0) + 4);
        const f32x4 ps1 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(r0 + 1));
        const f32x4 pt1 = *reinterpret_cast<const f32x4*>(pts + cf_point_slot(r0 + 1) + 4);
        unsigned d0[4], d1[4];
        f32x2 bs[2][3], bt[2][3];                 // the two row points, every coordinate in both halves of a register pair
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                bs[rr][e] = f32x2{(rr ? ps1 : ps0)[e], (rr ? ps1 : ps0)[e]};
                bt[rr][e] = f32x2{(rr ? pt1 : pt0)[e], (rr ? pt1 : pt0)[e]};
                asm volatile("" : "+v"(bs[rr][e]), "+v"(bt[rr][e]));
            }
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            f32x2 o[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const f32x2 dx = bs[rr][0] - sx[cp], dy = bs[rr][1] - sy[cp], dz = bs[rr][2] - sz[cp];
                const f32x2 ex = bt[rr][0] - tx[cp], ey = bt[rr][1] - ty[cp], ez = bt[rr][2] - tz[cp];
                const f32x2 a = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                const f32x2 b = __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ey, ey, ex * ex));
                const f32x2 ds = {__builtin_amdgcn_sqrtf(a[0]), __builtin_amdgcn_sqrtf(a[1])};
                const f32x2 dt = {__builtin_amdgcn_sqrtf(b[0]), __builtin_amdgcn_sqrtf(b[1])};
                const f32x2 df = ds - dt;
                f32x2 v = __builtin_elementwise_fma(df * df, ninv2, one2);           // 1 - df^2 / sigma^2
                if (CLAMP) v = __builtin_elementwise_max(v, zero2);
                o[rr] = v;
            }
            if (EDGE) {
                const int mc = 2 * cp, jc = j0 + cbase + (mc & 3) + 8 * (mc >> 2);
                const int ir = i0 + r0;
                const bool vj0 = jc < N, vj1 = jc + 1 < N, vi0 = ir < N, vi1 = ir + 1 < N;
                o[0][0] = (vi0 && vj0) ? o[0][0] : 0.f; o[0][1] = (vi0 && vj1) ? o[0][1] : 0.f;
                o[1][0] = (vi1 && vj0) ? o[1][0] : 0.f; o[1][1] = (vi1 && vj1) ? o[1][1] : 0.f;
            }
            d0[cp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[0][0], o[0][1]));
            d1[cp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[1][0], o[1][1]));
            tr[2 * cp][rp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[0][0], o[1][0]));
            tr[2 * cp + 1][rp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[0][1], o[1][1]));
        }
        const int i = i0 + r0;
        if (!EDGE || (i < N && colpos < ld)) *reinterpret_cast<u32x4*>(outb + (size_t)i * ld + colpos) = u32x4{d0[0], d0[1], d0[2], d0[3]};
        if (!EDGE || (i + 1 < N && colpos < ld)) *reinterpret_cast<u32x4*>(outb + (size_t)(i + 1) * ld + colpos) = u32x4{d1[0], d1[1], d1[2], d1[3]};
    }
    if (offdiag) {                                // block-uniform
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int j = j0 + cbase + (m & 3) + 8 * (m >> 2);
            if (!EDGE || (j < N && rowpos < ld)) *reinterpret_cast<u32x4*>(outb + (size_t)j * ld + rowpos) = u32x4{tr[m][0], tr[m][1], tr[m][2], tr[m][3]};
        }
    }
}

template <bool CLAMP>
__global__ __launch_bounds__(256) void compat_sym_u16_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                             const float* __restrict__ sigma_spat,
                                                             unsigned short* __restrict__ compat, long long ld, int N, int nt) {
    // workgroup -> tile (I <= J) of the upper triangle, rows of tiles back to back: start(I) = I nt - I (I - 1) / 2
    const int idx = blockIdx.x, b = blockIdx.y;
    int I = (int)(((float)(2 * nt + 1) - __builtin_sqrtf((float)((2 * nt + 1) * (2 * nt + 1) - 8 * idx))) * 0.5f);
    I = max(0, min(I, nt - 1));
    while (I > 0 && I * nt - I * (I - 1) / 2 > idx) --I;
    while ((I + 1) * nt - (I + 1) * I / 2 <= idx) ++I;
    const int J = I + idx - (I * nt - I * (I - 1) / 2);
    __shared__ __attribute__((aligned(16))) float pts[(2 * CF_T + 2 * CF_T / 16) * 8];      // tile rows, then tile columns: sx sy sz - tx ty tz -
    const int i0 = I * CF_T, j0 = J * CF_T;
    const float* srcb = src + (size_t)b * N * 3;
    const float* tgtb = tgt + (size_t)b * N * 3;
    const int t = threadIdx.x;
    {
        const int p = min((t < CF_T ? i0 : j0 - CF_T) + t, N - 1);
        float* d = pts + cf_point_slot(t);
        *reinterpret_cast<f32x4*>(d) = f32x4{srcb[p * 3 + 0], srcb[p * 3 + 1], srcb[p * 3 + 2], 0.f};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{tgtb[p * 3 + 0], tgtb[p * 3 + 1], tgtb[p * 3 + 2], 0.f};
    }
    const float sg = sigma_spat[0];
    const float s2 = sg * sg;                        // `self.sigma_spat ** 2` in fp32
    const float y0 = __builtin_amdgcn_rcpf(s2);
    const float ninv = -fmaf(fmaf(-s2, y0, 1.0f), y0, y0);      // -(1 / sigma^2), one Newton step on the 1-ulp reciprocal
    unsigned short* outb = compat + (size_t)b * N * ld;
    __syncthreads();
    if (i0 + CF_T > N || j0 + CF_T > N)              // block-uniform: only the last row / column of tiles masks and bounds-checks
        compat_fast_tile<true, CLAMP>(pts, outb, ld, N, i0, j0, I != J, ninv);
    else
        compat_fast_tile<false, CLAMP>(pts, outb, ld, N, i0, j0, I != J, ninv);
}

#ifdef PDSC_EXPERIMENTS
// r02 kernel (A/B record): u = round(c_fp32 * 65535) of the bit-exact fp32 matrix -- the exact sqrt / divide above, mirrored
// tile transposed through a 32-row LDS strip.  0.38 of the HBM rate (instruction-bound).
constexpr int C16_PITCH = 2 * CS_T + 8;      // bytes per LDS strip row (128 u16 + pad)

__global__ __launch_bounds__(256) void compat_sym_u16_exact_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                                   const float* __restrict__ sigma_spat,
                                                                   unsigned short* __restrict__ compat, long long ld, int N) {
    const int I = blockIdx.y, J = blockIdx.x, b = blockIdx.z;
    if (I > J) return;
    __shared__ __attribute__((aligned(16))) unsigned char Ts[CS_STRIP * C16_PITCH];   // one 32 x 128 strip of u16, row-major
    __shared__ __attribute__((aligned(16))) float rows_s[CS_T * 8];
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
    const InvariantDivisor inv(s2);
    unsigned short* outb = compat + (size_t)b * N * ld;
    const bool offdiag = I != J;
    const int jpos = (jc & ~31) + c16_pos(jc & 31);          // the 4 columns jc..jc+3 are contiguous in tile order
    __syncthreads();
    for (int strip = 0; strip < CS_T / CS_STRIP; ++strip) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = rg + 8 * r;                      // row inside the strip
            const int il = strip * CS_STRIP + sl;
            const int i = i0 + il;
            const f32x4 ps = *reinterpret_cast<const f32x4*>(rows_s + il * 8);
            const f32x4 pt = *reinterpret_cast<const f32x4*>(rows_s + il * 8 + 4);
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ds = dist3<true>(ps[0] - sx[c], ps[1] - sy[c], ps[2] - sz[c]);
                const float dt = dist3<true>(pt[0] - tx[c], pt[1] - ty[c], pt[2] - tz[c]);
                const float v = compat_from_dists<true>(ds, dt, s2, inv);
                o[c] = ((jc + c) < N && i < N) ? v : 0.0f;
            }
            uint2 q;
            q.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[0], o[1]));
            q.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(o[2], o[3]));
            if (i < N && jc < ld) *reinterpret_cast<uint2*>(outb + (size_t)i * ld + jpos) = q;
            if (offdiag) *reinterpret_cast<uint2*>(Ts + sl * C16_PITCH + 8 * l32) = q;
        }
        if (offdiag) {                                       // block-uniform
            __syncthreads();
            // output row j, 16-B chunk q = tile-order positions 8q..8q+7 of the strip's 32 rows (one whole key group)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int idx = t + 256 * p2, jl = idx >> 2, q = idx & 3;
                unsigned w[4];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int p = 8 * q + m;
                    const int sl = 8 * ((p >> 2) & 3) + 4 * (p >> 4) + (p & 3);          // inverse of c16_pos
                    const unsigned v = *reinterpret_cast<const unsigned short*>(Ts + sl * C16_PITCH + 2 * jl);
                    if (m & 1) w[m >> 1] |= v << 16; else w[m >> 1] = v;
                }
                const int j = j0 + jl;
                if (j < N) *reinterpret_cast<u32x4*>(outb + (size_t)j * ld + i0 + strip * CS_STRIP + 8 * q) = u32x4{w[0], w[1], w[2], w[3]};
            }
            __syncthreads();
        }
    }
}
#endif  // PDSC_EXPERIMENTS

// self-test hook: the two hand-rolled exact primitives on arbitrary inputs (tests compare with IEEE results)
__global__ void exact_math_selftest_kernel(const float* __restrict__ x, float b, float* __restrict__ sq, float* __restrict__ dv,
                                           long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const InvariantDivisor inv(b);
    sq[i] = sqrt_rn(x[i]);
    dv[i] = inv.div(x[i]);
}

}  // namespace pdsc

extern "C" int pdsc_selftest_exact_math(const float* x, float divisor, float* sqrt_out, float* div_out, long long n,
                                        void* stream) {
    PDSC_REQUIRE(x && sqrt_out && div_out && n > 0, "pdsc_selftest_exact_math: bad argument");
    hipLaunchKernelGGL(pdsc::exact_math_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, divisor, sqrt_out, div_out, n);
    return pdsc::check_launch("pdsc_selftest_exact_math");
}

extern "C" long long pdsc_compat_ld(int N) { return N <= 0 ? -1 : pdsc::round_up(N, 64); }

extern "C" int pdsc_spatial_compat(const float* src, const float* tgt, const float* sigma_spat, float* compat,
                                   float* src_dist, long long ld, int bs, int N, void* stream) {
    PDSC_REQUIRE(src && tgt && sigma_spat && compat, "pdsc_spatial_compat: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_spatial_compat: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= N && ld % 4 == 0, "pdsc_spatial_compat: ld=%lld must be >= N and a multiple of 4", ld);
    hipStream_t st = (hipStream_t)stream;
    const dim3 full_grid(pdsc::ceil_div((int)ld, pdsc::CT_COLS), pdsc::ceil_div(N, pdsc::CT_ROWS), bs);
    const int nt = pdsc::ceil_div(N, pdsc::CS_T);
    const dim3 sym_grid(nt, nt, bs);
    pdsc::profile_mark_begin(PDSC_PROF_COMPAT, st);
    if (src_dist)
        hipLaunchKernelGGL((pdsc::compat_kernel<true, true>), full_grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, src_dist, ld, N);
    else {
#ifdef PDSC_EXPERIMENTS
        // A/B knob (all variants produce identical bits): 0 = full tiles + hipcc IEEE math, 1 = full tiles + hand-rolled exact
        // math, 2 = symmetric + IEEE math, 3 = symmetric + hand-rolled exact math (what ships)
        const int variant = pdsc::env_int("PDSC_COMPAT_VARIANT", PDSC_COMPAT_DEFAULT_VARIANT);
        if (variant == 0)
            hipLaunchKernelGGL((pdsc::compat_kernel<false, false>), full_grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, nullptr, ld, N);
        else if (variant == 1)
            hipLaunchKernelGGL((pdsc::compat_kernel<false, true>), full_grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, nullptr, ld, N);
        else if (variant == 2)
            hipLaunchKernelGGL((pdsc::compat_sym_kernel<false>), sym_grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, ld, N);
        else
#endif
        hipLaunchKernelGGL((pdsc::compat_sym_kernel<true>), sym_grid, dim3(256), 0, st, src, tgt, sigma_spat, compat, ld, N);
    }
    pdsc::profile_mark_end(PDSC_PROF_COMPAT, st);
    return pdsc::check_launch("pdsc_spatial_compat");
}

extern "C" int pdsc_spatial_compat_u16(const float* src, const float* tgt, const float* sigma_spat, unsigned short* compat_u16,
                                       long long ld, int bs, int N, void* stream) {
    PDSC_REQUIRE(src && tgt && sigma_spat && compat_u16, "pdsc_spatial_compat_u16: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_spatial_compat_u16: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= pdsc::round_up(N, 32) && ld % 32 == 0, "pdsc_spatial_compat_u16: ld=%lld must be a multiple of 32 and >= N", ld);
    hipStream_t st = (hipStream_t)stream;
    const int nt = pdsc::ceil_div(N, pdsc::CF_T);
    PDSC_REQUIRE(nt <= 1024, "pdsc_spatial_compat_u16: N=%d too large", N);
    const dim3 grid((unsigned)(nt * (nt + 1) / 2), bs);
    pdsc::profile_mark_begin(PDSC_PROF_COMPAT, st);
#ifdef PDSC_EXPERIMENTS
    // A/B knob (experiments builds only): 0 = r02 kernel (rounded exact fp32 matrix), 1 = fast kernel with an explicit
    // max(c, 0) in front of the conversion
    const int variant = pdsc::env_int("PDSC_COMPAT16_VARIANT", 2);
    if (variant == 0)
        hipLaunchKernelGGL(pdsc::compat_sym_u16_exact_kernel, dim3(nt, nt, bs), dim3(256), 0, st, src, tgt, sigma_spat, compat_u16, ld, N);
    else if (variant == 1)
        hipLaunchKernelGGL(pdsc::compat_sym_u16_kernel<true>, grid, dim3(256), 0, st, src, tgt, sigma_spat, compat_u16, ld, N, nt);
    else
#endif
    // v_cvt_pknorm_u16_f32 clamps to [0, 1] itself: bit-identical output with and without the explicit max (measured on every
    // entry of five matrices, profiles/r03_a_compat_bench.txt), 3 % faster without
    hipLaunchKernelGGL(pdsc::compat_sym_u16_kernel<false>, grid, dim3(256), 0, st, src, tgt, sigma_spat, compat_u16, ld, N, nt);
    pdsc::profile_mark_end(PDSC_PROF_COMPAT, st);
    return pdsc::check_launch("pdsc_spatial_compat_u16");
}
