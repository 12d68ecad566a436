// a-2 fused: the whole point-wise chain between two attention calls in ONE launch
//   tail of layer i   : feat  = featB + fc3( relu(fc2'( relu(fc1'(msg)) )) )         (fc_message, BN folded; reference
//                                                                                      models/PointDSC.py:43-45)
//   head of layer i+1 : featB = relu(pcn'(feat)) ; (q|k|v) = Wqkv featB + b            (models/PointDSC.py:75, :36-38)
// The five GEMMs of a 32-point tile run back to back in one 4-wave workgroup: activations never leave the
// CU (two 32x132 LDS tiles, ping-pong), each wave owns whole 32-column output tiles and reads the weight rows
// of its tile straight from L2 into registers (a weight tile is used by exactly one wave of the workgroup, so
// staging it in LDS would buy nothing), the next stage's weights are prefetched under the current stage's MFMAs.
// MFMA orientation: D = W_tile (A, rows = output channels) x X^T (B, columns = points), so the accumulator
// lane is a point and register r = 4g+e holds output channel n0+8g+4h+e: one float4 per (g) goes to LDS / HBM.
// Bound: MFMA (172 kFLOP per point per layer on v_mfma_f32_32x32x2_f32); weights stream from L2 (336 KB per tile).
#include <stdlib.h>
#include <type_traits>
#include "pdsc_common.h"
#include "split_layout.h"
#include "merge_partials.h"
#include "layer_args.h"
#include "ragged.h"

namespace pdsc {

constexpr int LF_ROWS = 32;                  // points per workgroup
constexpr int LF_LD = PDSC_CHANNELS + 4;     // padded LDS row (floats)
constexpr int LF_TILE = LF_ROWS * LF_LD;


#define LF_STAMP(k)                                                                                   \
    if (a.trace && lane == 0) a.trace[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + (k)] = __builtin_readcyclecounter();

template <int K>
__device__ __forceinline__ void load_w(const float* __restrict__ W, int n0, int l31, int h, f32x4 (&w)[K / 8]) {
    const float* p = W + (size_t)(n0 + l31) * K + 4 * h;
#pragma unroll
    for (int q = 0; q < K / 8; ++q) w[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
}

template <int K>
__device__ __forceinline__ void load_x(const float* Xs, int l31, int h, f32x4 (&x)[K / 8]) {
    const float* p = Xs + l31 * LF_LD + 4 * h;
#pragma unroll
    for (int q = 0; q < K / 8; ++q) x[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
}

template <int K>
__device__ __forceinline__ f32x16 mma_tile(const f32x4 (&w)[K / 8], const f32x4 (&x)[K / 8]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < K / 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q][e], x[q][e], acc, 0, 0, 0);
    return acc;
}

// acc -> (+bias)(relu)(+residual row from HBM) -> LDS tile Xo[point][col0 + 8g+4h .. +3]
template <bool RELU, bool RESID>
__device__ __forceinline__ void store_tile(const f32x16& acc, const float* __restrict__ bias, int n0, float* Xo, int col0,
                                           int l31, int h, const float* __restrict__ res_row) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + 8 * g + 4 * h);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = acc[4 * g + e] + bv[e];
            if (RELU) t = fmaxf(t, 0.f);
            v[e] = t;
        }
        if (RESID) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(res_row + n0 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rv[e] + v[e];
        }
        *reinterpret_cast<f32x4*>(Xo + l31 * LF_LD + col0 + 8 * g + 4 * h) = v;
    }
}

// coalesced copies between a 32x128 LDS tile and row-major global memory (ld floats per row)
__device__ __forceinline__ void tile_to_global(const float* Xs, float* __restrict__ dst, long long ld, int m0, int M, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = f & 31;
        if (m0 + row < M)
            *reinterpret_cast<f32x4*>(dst + (size_t)(m0 + row) * ld + 4 * c4) = *reinterpret_cast<const f32x4*>(Xs + row * LF_LD + 4 * c4);
    }
}
__device__ __forceinline__ void global_to_tile(const float* __restrict__ src, float* Xs, int m0, int M, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = f & 31;
        const int m = min(m0 + row, M - 1);
        *reinterpret_cast<f32x4*>(Xs + row * LF_LD + 4 * c4) = *reinterpret_cast<const f32x4*>(src + (size_t)m * PDSC_CHANNELS + 4 * c4);
    }
}

// 32x128 fp32 LDS tile (one of q / k / v for 32 points = one key tile) -> fp16 hi/lo streams (split_layout.h).
// WHICH: 0 = q rows, 1 = K image, 2 = V^T image.  `valid` = number of real points in the tile (the rest is zero).
template <int WHICH>
__device__ __forceinline__ void tile_to_split(const float* Xs, sp16* __restrict__ qrows, unsigned char* __restrict__ img,
                                              int valid, int t) {
    if (WHICH == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
            if (row < valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Xs + row * LF_LD + c4);
                sp16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sp16 x, y; split_sp16(v[e], x, y); hi[e] = x; lo[e] = y; }
                sp16* dst = qrows + (size_t)row * SPL_Q_LD + c4;
                *reinterpret_cast<sp16x4*>(dst) = hi;
                *reinterpret_cast<sp16x4*>(dst + PDSC_CHANNELS) = lo;
            }
        }
    } else if (WHICH == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = t + 256 * i, key = f >> 4, chunk = f & 15;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(Xs + key * LF_LD + 8 * chunk);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(Xs + key * LF_LD + 8 * chunk + 4);
            sp16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = key < valid ? (e < 4 ? v0[e & 3] : v1[e & 3]) : 0.f;
                sp16 x, y; split_sp16(v, x, y); hi[e] = x; lo[e] = y;
            }
            *reinterpret_cast<sp16x8*>(img + SPL_KH + spl_k_offset(key, chunk)) = hi;
            *reinterpret_cast<sp16x8*>(img + SPL_KL + spl_k_offset(key, chunk)) = lo;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = t + 256 * i, ch = f & 127, jh = f >> 7;
            sp16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = spl_v_key(jh, e);
                const float v = key < valid ? Xs[key * LF_LD + ch] : 0.f;
                sp16 x, y; split_sp16(v, x, y); hi[e] = x; lo[e] = y;
            }
            *reinterpret_cast<sp16x8*>(img + SPL_VH + spl_v_offset(ch, jh)) = hi;
            *reinterpret_cast<sp16x8*>(img + SPL_VL + spl_v_offset(ch, jh)) = lo;
        }
    }
}

// tail input: merged msg rows, or the merge of the attention's key-split partials (merge_partials.h)
__device__ __forceinline__ void msg_to_tile(const LayerArgs& a, int b, float* Xs, int m0, int M, int t) {
    if (a.msg) {
        global_to_tile(a.msg, Xs, m0, M, t);
        return;
    }
    // split count as a compile-time constant (wave-uniform switch): registers for exactly that many partials
    auto run = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        MergeLoads<NS> L[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
            const int m = min(m0 + row, M - 1);
            const size_t slot0 = (size_t)b * NS * a.Npad + (size_t)(m - b * a.N);
            merge_partials_load<NS>(L[i], a.part_o, a.part_ml, slot0, (size_t)a.Npad, c4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
            *reinterpret_cast<f32x4*>(Xs + row * LF_LD + c4) = merge_partials_finish<NS>(L[i]);
        }
    };
    // 5..8 splits (small problems: the attention plan splits the keys further to fill the chip): one chunk at a time --
    // registers for a single set of NS partials; costs four dependent round trips instead of one, still cheaper than the
    // attention_combine launch + the msg round trip it replaces
    auto run_seq = [&](auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
            const int m = min(m0 + row, M - 1);
            const size_t slot0 = (size_t)b * NS * a.Npad + (size_t)(m - b * a.N);
            MergeLoads<NS> L;
            merge_partials_load<NS>(L, a.part_o, a.part_ml, slot0, (size_t)a.Npad, c4);
            *reinterpret_cast<f32x4*>(Xs + row * LF_LD + c4) = merge_partials_finish<NS>(L);
        }
    };
    switch (a.nsplit) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run_seq(std::integral_constant<int, 5>{}); break;
        case 6: run_seq(std::integral_constant<int, 6>{}); break;
        case 7: run_seq(std::integral_constant<int, 7>{}); break;
        default: run_seq(std::integral_constant<int, 8>{}); break;
    }
}

constexpr int LF_XLD16 = PDSC_CHANNELS + 8;          // fp16 elements per row of a hi / lo activation tile (272 B)
constexpr int LF_XB_FLOATS = LF_ROWS * LF_XLD16;     // Xb doubles as the fp16 hi|lo image of featB: 2 * 32 * 136 * 2 B

template <bool HAS_TAIL, bool HAS_HEAD, bool QKV_X3>
__global__ __launch_bounds__(256, 3) void layer_fused_kernel(LayerArgs a) {
    __shared__ __attribute__((aligned(16))) float Xa[LF_TILE];
    __shared__ __attribute__((aligned(16))) float Xb[LF_XB_FLOATS > LF_TILE ? LF_XB_FLOATS : LF_TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * a.N + blockIdx.x * LF_ROWS;     // first row of this tile
    const int M = blockIdx.y * a.N + (a.nvalid ? a.nvalid[blockIdx.y] : a.N);     // end of this pair's rows
    if (m0 >= M) return;                                         // (ragged batches: tile past the pair's own rows; workgroup-uniform)

    LF_STAMP(0)
    f32x4 wpre[16];                                                  // PointCN weight tile of this wave (prefetched early)
    if (HAS_TAIL) {
        f32x4 w128[16], w64[8];
        // ---- fc1: 128 -> 64 (+BN, ReLU): tiles {0,1} on waves {0,1} ----
        if (wave < 2) load_w<128>(a.w1, 32 * wave, l31, h, w128);
        msg_to_tile(a, blockIdx.y, Xa, m0, M, t);
        LF_STAMP(1)
        __syncthreads();
        LF_STAMP(2)
        if (wave < 2) {
            f32x4 x[16];
            load_x<128>(Xa, l31, h, x);
            load_w<64>(a.w2, 32 * wave, l31, h, w64);                 // prefetch fc2 weights
            const f32x16 acc = mma_tile<128>(w128, x);
            store_tile<true, false>(acc, a.b1, 32 * wave, Xb, 32 * wave, l31, h, nullptr);
        }
        LF_STAMP(3)
        __syncthreads();
        // ---- fc2: 64 -> 64 (+BN, ReLU) ----
        f32x4 w3r[8];
        load_w<64>(a.w3, 32 * wave, l31, h, w3r);                     // prefetch fc3 weights (all waves)
        if (wave < 2) {
            f32x4 x[8];
            load_x<64>(Xb, l31, h, x);
            const f32x16 acc = mma_tile<64>(w64, x);
            store_tile<true, false>(acc, a.b2, 32 * wave, Xa, 32 * wave, l31, h, nullptr);
        }
        LF_STAMP(4)
        __syncthreads();
        // ---- fc3: 64 -> 128, + residual featB: tile = wave ----
        if (HAS_HEAD) load_w<128>(a.wp, 32 * wave, l31, h, wpre);      // PointCN weights of the head: under fc3's MFMAs
        {
            f32x4 x[8];
            load_x<64>(Xa, l31, h, x);
            const f32x16 acc = mma_tile<64>(w3r, x);
            const float* res_row = a.res + (size_t)min(m0 + l31, M - 1) * PDSC_CHANNELS;
            store_tile<false, true>(acc, a.b3, 32 * wave, Xb, 32 * wave, l31, h, res_row);
        }
        LF_STAMP(5)
        __syncthreads();
        LF_STAMP(6)
        if (a.feat_out) tile_to_global(Xb, a.feat_out, PDSC_CHANNELS, m0, M, t);
    } else {
        if (HAS_HEAD) load_w<128>(a.wp, 32 * wave, l31, h, wpre);
        global_to_tile(a.feat_in, Xb, m0, M, t);
        __syncthreads();
    }

    if (HAS_HEAD && !QKV_X3) {
        // ---- PointCN: 128 -> 128 (+BN, ReLU): tile = wave; input Xb, output Xa ----
        f32x4 w[16], x[16];
        load_x<128>(Xb, l31, h, x);
        {
            const f32x16 acc = mma_tile<128>(wpre, x);
            load_w<128>(a.wq, 32 * wave, l31, h, w);                  // prefetch first qkv tile
            store_tile<true, false>(acc, a.bp, 32 * wave, Xa, 32 * wave, l31, h, nullptr);
        }
        __syncthreads();
        tile_to_global(Xa, a.featB_out, PDSC_CHANNELS, m0, M, t);
        load_x<128>(Xa, l31, h, x);
        // ---- q|k|v: 128 -> 384 in three 128-column chunks staged through Xb ----
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n0 = 128 * c + 32 * wave;
            const f32x16 acc = mma_tile<128>(w, x);
            if (c < 2) load_w<128>(a.wq, n0 + 128, l31, h, w);        // prefetch next chunk's tile
            store_tile<false, false>(acc, a.bq, n0, Xb, 32 * wave, l31, h, nullptr);
            __syncthreads();
            if (a.qkv_out) tile_to_global(Xb, a.qkv_out + 128 * c, 3 * PDSC_CHANNELS, m0, M, t);
            if (a.qs) {
                unsigned char* img = a.kv + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SPL_TILE_STRIDE;
                const int valid = min(LF_ROWS, M - m0);
                if (c == 0) tile_to_split<0>(Xb, a.qs + (size_t)m0 * SPL_Q_LD, img, valid, t);
                else if (c == 1) tile_to_split<1>(Xb, nullptr, img, valid, t);
                else tile_to_split<2>(Xb, nullptr, img, valid, t);
            }
            if (c < 2) __syncthreads();
        }
    }
    if (HAS_HEAD && QKV_X3) {
        // ---- PointCN exactly as above (exact fp32: featB is the next residual) ... ----
        sp16* Xh = reinterpret_cast<sp16*>(Xb);
        sp16* Xl = Xh + LF_ROWS * LF_XLD16;
        sp16x8 wh[8], wl[8];
        {
            f32x4 x[16];
            load_x<128>(Xb, l31, h, x);
            // prefetch the first split qkv tile: lane (row l31, half h), step kk holds k = 16kk+8h..+7
            {
                const sp16* p = a.wq_split + (size_t)(32 * wave + l31) * PDSC_CHANNELS + 8 * h;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    wh[kk] = *reinterpret_cast<const sp16x8*>(p + 16 * kk);
                    wl[kk] = *reinterpret_cast<const sp16x8*>(p + (size_t)3 * PDSC_CHANNELS * PDSC_CHANNELS + 16 * kk);
                }
            }
            const f32x16 acc = mma_tile<128>(wpre, x);
            // (the first split qkv tile was requested above, under these MFMAs)
            LF_STAMP(7)
            __syncthreads();                                          // every wave holds its copy of Xb: Xb may be rewritten
            // ... stored twice: fp32 -> Xa (featB_out), fp16 hi|lo -> Xb (operand of the split-precision q|k|v GEMM)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * wave + 8 * g + 4 * h;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bp + col);
                f32x4 v;
                sp16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[4 * g + e] + bv[e], 0.f);
                    sp16 xh, xl; split_sp16(v[e], xh, xl); hi[e] = xh; lo[e] = xl;
                }
                *reinterpret_cast<f32x4*>(Xa + l31 * LF_LD + col) = v;
                *reinterpret_cast<sp16x4*>(Xh + l31 * LF_XLD16 + col) = hi;
                *reinterpret_cast<sp16x4*>(Xl + l31 * LF_XLD16 + col) = lo;
            }
        }
        LF_STAMP(8)
        __syncthreads();
        tile_to_global(Xa, a.featB_out, PDSC_CHANNELS, m0, M, t);
        LF_STAMP(9)
        // ---- q|k|v: 128 -> 384, three 128-column chunks, hi*hi + hi*lo + lo*hi on the fp16 matrix cores, staged via Xa ----
        unsigned char* img = a.kv ? a.kv + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SPL_TILE_STRIDE : nullptr;
        const int valid = min(LF_ROWS, M - m0);
        const int xo = l31 * LF_XLD16 + 8 * h;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n0 = 128 * c + 32 * wave;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc = zero;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const sp16x8 xh = *reinterpret_cast<const sp16x8*>(Xh + xo + 16 * kk);
                const sp16x8 xl = *reinterpret_cast<const sp16x8*>(Xl + xo + 16 * kk);
                acc = PDSC_MFMA_X3(wl[kk], xh, acc, 0, 0, 0);
                acc = PDSC_MFMA_X3(wh[kk], xl, acc, 0, 0, 0);
                acc = PDSC_MFMA_X3(wh[kk], xh, acc, 0, 0, 0);
            }
            if (c < 2) {                                              // prefetch next chunk's tile
                const sp16* p = a.wq_split + (size_t)(n0 + 128 + l31) * PDSC_CHANNELS + 8 * h;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    wh[kk] = *reinterpret_cast<const sp16x8*>(p + 16 * kk);
                    wl[kk] = *reinterpret_cast<const sp16x8*>(p + (size_t)3 * PDSC_CHANNELS * PDSC_CHANNELS + 16 * kk);
                }
            }
            if (c == 0) { LF_STAMP(10) }
            __syncthreads();                                          // previous readers of Xa are done
            store_tile<false, false>(acc, a.bq, n0, Xa, 32 * wave, l31, h, nullptr);
            __syncthreads();
            if (c == 0) { LF_STAMP(11) }
            if (a.qkv_out) tile_to_global(Xa, a.qkv_out + 128 * c, 3 * PDSC_CHANNELS, m0, M, t);
            if (a.qs) {
                if (c == 0) tile_to_split<0>(Xa, a.qs + (size_t)m0 * SPL_Q_LD, img, valid, t);
                else if (c == 1) tile_to_split<1>(Xa, nullptr, img, valid, t);
                else tile_to_split<2>(Xa, nullptr, img, valid, t);
            }
            if (c == 0) { LF_STAMP(12) }
        }
        LF_STAMP(13)
    }
}

template <bool T, bool H>
static int launch_layer(const LayerArgs& a, hipStream_t st) {
    if (H && a.wq_split)
        hipLaunchKernelGGL((layer_fused_kernel<T, H, H>), dim3(ceil_div(a.N, LF_ROWS), a.bs), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((layer_fused_kernel<T, H, false>), dim3(ceil_div(a.N, LF_ROWS), a.bs), dim3(256), 0, st, a);
    return check_launch("pdsc_layer_fused");
}

}  // namespace pdsc

static long long* g_layer_trace = nullptr;
long long* pdsc_layer_trace_buffer(void) { return g_layer_trace; }      // shared with layer_wave.hip
extern "C" int pdsc_layer_trace(long long* device_buffer) {       // diagnostics: see include/pointdsc_hip.h
    g_layer_trace = device_buffer;
    return PDSC_OK;
}

// Size rule, from the hipEvent-timed fused layer launch at N = 5000 (us, block / wave; tiles = pairs x 157): 2 pairs 37 / 55,
// 3 pairs (471 tiles) 43 / 61, 4 pairs (628) 56 / 53, 6 pairs (942) 70 / 56, 8 pairs 93 / 94, 16 pairs 190 / 163,
// 32 pairs 347 / 250.  Block up to two workgroups per CU.
// r02, wavefront-resident kernel with the H3 GEMMs (layer_h3.hip; whole forward, ms per step, block / wave,
// profiles/r02_j_ab_block_vs_h3.txt): N = 5000 x 1 pair (157 tiles) 1.200 / 1.235, 2 pairs (314) 1.708 / 1.664, 3 pairs (471)
// 1.985 / 1.929; N = 10000 x 1 (313) 2.682 / 2.682; N = 1000 x 1 (32) 0.572 / 0.659, x 4 (128) 0.641 / 0.712: block while the
// tiles leave CUs empty (about one workgroup per CU).
extern "C" int pdsc_layer_prefers_block(int bs, int N) {
    return (long long)bs * pdsc::ceil_div(N, pdsc::LF_ROWS) <= 288;
}

extern "C" int pdsc_layer_fused_split(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                                      const float* res, const float* feat_in, float* feat_out,
                                      float* featB_out, float* qkv_out, void* q_split, void* kv_tiles,
                                      const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                      const float* b3, const float* wp, const float* bp, const float* wq, const float* bq,
                                      const void* wq_split, int bs, int N, void* stream) {
    const bool tail = msg != nullptr || part_o != nullptr, head = featB_out != nullptr;
    PDSC_REQUIRE(tail || head, "pdsc_layer_fused: neither tail (msg / partials) nor head (featB_out) requested");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_layer_fused: bs=%d N=%d", bs, N);
    if (tail) {
        PDSC_REQUIRE(res && w1 && b1 && w2 && b2 && w3 && b3, "pdsc_layer_fused: tail needs res, fc1..fc3");
        if (!msg) PDSC_REQUIRE(part_ml && nsplit >= 1 && nsplit <= pdsc::MERGE_MAX_SPLIT_BLOCK && Npad >= N,
                               "pdsc_layer_fused: partials need part_ml, 1 <= nsplit <= %d, Npad >= N", pdsc::MERGE_MAX_SPLIT_BLOCK);
    } else PDSC_REQUIRE(feat_in, "pdsc_layer_fused: head-only needs feat_in");
    if (head) PDSC_REQUIRE((qkv_out || q_split) && wp && bp && wq && bq, "pdsc_layer_fused: head needs qkv_out or the split streams, pcn, qkv weights");
    else PDSC_REQUIRE(feat_out, "pdsc_layer_fused: tail-only needs feat_out");
    PDSC_REQUIRE((q_split == nullptr) == (kv_tiles == nullptr), "pdsc_layer_fused: q_split and kv_tiles go together");
    pdsc::LayerArgs a{msg, part_o, part_ml, nsplit, Npad, res, feat_in, feat_out, featB_out, qkv_out, w1, b1, w2, b2, w3, b3,
                      wp, bp, wq, bq, (const sp16*)wq_split, (sp16*)q_split, (unsigned char*)kv_tiles, N, bs, nullptr, nullptr, PDSC_LAYER_GEMM_F32, 0, 0, 0, g_layer_trace};
    a.nvalid = pdsc::layer_nvalid_slot();
    hipStream_t st = (hipStream_t)stream;
    // Two implementations.  layer_wave.hip (one wavefront per 32-point tile) wins once the tiles fill the chip; with few
    // tiles its serial 46k matrix-pipe cycles per tile are the launch time, and this file's kernel, which spreads a tile
    // over the four SIMDs of a CU, is faster (N = 1000, one pair: 0.68 vs 1.10 ms per forward).
    // PDSC_LAYER_VARIANT = block | wave overrides the size rule.
    const char* ev = pdsc::env_str("PDSC_LAYER_VARIANT");          // experiments builds only
    const int variant = !ev ? 0 : ev[0] == 'b' ? 1 : ev[0] == 'w' ? 2 : 0;
    const bool block_variant = variant == 1 || (variant == 0 && pdsc_layer_prefers_block(bs, N));
    if (!block_variant) {
        PDSC_REQUIRE(msg || !tail || nsplit <= pdsc::MERGE_MAX_SPLIT, "pdsc_layer_fused: the wavefront-per-tile kernel merges at most %d splits",
                     pdsc::MERGE_MAX_SPLIT);
        return pdsc::launch_layer_wave(a, tail, head, st);
    }
    if (tail && head) {
        pdsc::profile_mark_begin(PDSC_PROF_LAYER, st);
        const int rc = pdsc::launch_layer<true, true>(a, st);
        pdsc::profile_mark_end(PDSC_PROF_LAYER, st);
        return rc;
    }
    if (tail) return pdsc::launch_layer<true, false>(a, st);
    return pdsc::launch_layer<false, true>(a, st);
}

extern "C" int pdsc_layer_fused(const float* msg, const float* res, const float* feat_in, float* feat_out,
                                float* featB_out, float* qkv_out, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, const float* wp, const float* bp,
                                const float* wq, const float* bq, int M, void* stream) {
    if (featB_out) PDSC_REQUIRE(qkv_out, "pdsc_layer_fused: head needs qkv_out");
    // this entry point sees the batch as ONE run of M independent rows (bs = 1, N = M): the per-pair counts of a ragged forward
    // (layer_nvalid_slot) do not describe it -- with them in place the kernel took counts[0] for the row count of the whole batch
    // (r06: the exact-fp32 path takes ragged batches now).  Padding rows are computed like any row; nothing valid reads them.
    struct NoCounts {
        const int* saved;
        NoCounts() : saved(pdsc::layer_nvalid_slot()) { pdsc::layer_nvalid_slot() = nullptr; }
        ~NoCounts() { pdsc::layer_nvalid_slot() = saved; }
    } no_counts;
    return pdsc_layer_fused_split(msg, nullptr, nullptr, 0, 0, res, feat_in, feat_out, featB_out, qkv_out, nullptr, nullptr,
                                  w1, b1, w2, b2, w3, b3, wp, bp, wq, bq, nullptr, 1, M, stream);
}
