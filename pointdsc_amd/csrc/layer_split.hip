// a-2 fused, split-precision variant of layer.hip: the point-wise chain between two attention calls in ONE launch,
// with every GEMM on the fp16 matrix cores in the hi/lo split arithmetic of split_layout.h (x = hi + lo, product =
// hi*hi + hi*lo + lo*hi, fp32 accumulate: 3/16 of the matrix-pipe time of the exact fp32 MFMA, ~2^-16 relative error
// per product) and with the merge of the attention's key-split partials folded into the load of its input:
//   tail of layer i   : msg   = merge(partials)                       (attention_combine_kernel's job, no extra launch)
//                       feat  = featB + fc3( relu(fc2'( relu(fc1'(msg)) )) )        (reference models/PointDSC.py:43-45)
//   head of layer i+1 : featB = relu(pcn'(feat)) ; (q|k|v) = Wqkv featB + b          (models/PointDSC.py:75, :36-38)
//                       q, k, v leave as the fp16 hi/lo operand streams of attention_split.hip
// Weights are split once on the device (pdsc_wsplit_build): per matrix [out][in] fp16 hi, then [out][in] fp16 lo.
// Workgroup = 4 waves = one 32-point tile (never straddling two pairs); activations stay in LDS as fp16 hi/lo tiles
// [32 points][K] with a 272-byte row stride (17 chunks of 16 B: consecutive points rotate by one bank slot, every
// ds_read_b128 of a lane is base + immediate); fp32 results that leave the CU are staged through one fp32 tile.
// MFMA orientation: D = W_tile (A, rows = 32 output channels) x X^T (B, columns = points): the accumulator lane is a
// point and register r = 4g+e holds output channel n0+8g+4h+e.
// Bound: L2 weight stream (344 KB per workgroup) and launch latency; MFMA time is ~5 us per launch at M = 20 000.
#include "pdsc_common.h"
#include "split_layout.h"
#include "merge_partials.h"

namespace pdsc {

#ifdef PDSC_EXPERIMENTS      // the all-split layer kernel is an A/B record (r01: +3 % pairs/s at 3e-5 feature error): experiments builds only
constexpr int LX_ROWS = 32;                         // points per workgroup
constexpr int LX_XLD = PDSC_CHANNELS + 8;           // fp16 elements per activation row (272 B)
constexpr int LX_FLD = PDSC_CHANNELS + 4;           // floats per staging row
constexpr int LX_XTILE = LX_ROWS * LX_XLD;          // fp16 elements per hi (or lo) tile

struct LayerX3Args {
    const float* msg;        // [M][128] tail input (already merged), or NULL when the partials below are given
    const float* part_o;     // [bs][nsplit][Npad][128] un-normalised partial outputs of the attention key splits
    const float* part_ml;    // [bs][nsplit][Npad][2]   (reference exponent (log2), partial sum)
    int nsplit, Npad;
    const float* res;        // [M][128] tail residual (featB of this layer)
    const float* feat_in;    // [M][128] head-only input
    float* feat_out;         // [M][128] tail result (optional)
    float* featB_out;        // [M][128] head
    float* qkv_out;          // [M][384] head, optional fp32 copy of q|k|v
    sp16* qs;              // head: Q split stream
    unsigned char* kv;       // head: K/V tile stream
    const sp16 *w1, *w2, *w3, *wp, *wq;          // split weights (hi block, then lo block)
    const float *b1, *b2, *b3, *bp, *bq;
    int N, bs;
};

// A operand: rows n0..n0+31 of a split weight matrix [Nout][K]; lane (row l31, half h), step kk holds k = 16kk+8h..+7
template <int K>
__device__ __forceinline__ void load_w(const sp16* __restrict__ W, int nout, int n0, int l31, int h, sp16x8 (&wh)[K / 16],
                                       sp16x8 (&wl)[K / 16]) {
    const sp16* p = W + (size_t)(n0 + l31) * K + 8 * h;
#pragma unroll
    for (int kk = 0; kk < K / 16; ++kk) {
        wh[kk] = *reinterpret_cast<const sp16x8*>(p + 16 * kk);
        wl[kk] = *reinterpret_cast<const sp16x8*>(p + (size_t)nout * K + 16 * kk);
    }
}

// D[32 channels][32 points] = W_tile . X^T with X = Xh + Xl in LDS; small terms first
template <int K>
__device__ __forceinline__ f32x16 mma_tile(const sp16x8 (&wh)[K / 16], const sp16x8 (&wl)[K / 16], const sp16* Xh,
                                           const sp16* Xl, int l31, int h) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = zero;
    const int xo = l31 * LX_XLD + 8 * h;
#pragma unroll
    for (int kk = 0; kk < K / 16; ++kk) {
        const sp16x8 xh = *reinterpret_cast<const sp16x8*>(Xh + xo + 16 * kk);
        const sp16x8 xl = *reinterpret_cast<const sp16x8*>(Xl + xo + 16 * kk);
        acc = PDSC_MFMA_X3(wl[kk], xh, acc, 0, 0, 0);
        acc = PDSC_MFMA_X3(wh[kk], xl, acc, 0, 0, 0);
        acc = PDSC_MFMA_X3(wh[kk], xh, acc, 0, 0, 0);
    }
    return acc;
}

// acc -> (+bias)(relu)(+residual row from HBM) -> fp32 staging tile F and/or fp16 hi/lo activation tile, at columns
// col0 + 8g+4h .. +3 of row = point l31
template <bool RELU, bool RESID, bool TO_F, bool TO_X>
__device__ __forceinline__ void store_tile(const f32x16& acc, const float* __restrict__ bias, int n0, int col0, int l31, int h,
                                           const float* __restrict__ res_row, float* F, sp16* Xh, sp16* Xl) {
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
        const int col = col0 + 8 * g + 4 * h;
        if (TO_F) *reinterpret_cast<f32x4*>(F + l31 * LX_FLD + col) = v;
        if (TO_X) {
            sp16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { sp16 x, y; split_sp16(v[e], x, y); hi[e] = x; lo[e] = y; }
            *reinterpret_cast<sp16x4*>(Xh + l31 * LX_XLD + col) = hi;
            *reinterpret_cast<sp16x4*>(Xl + l31 * LX_XLD + col) = lo;
        }
    }
}

// 32x128 fp32 staging tile -> row-major global memory (ld floats per row), full 512-B rows
__device__ __forceinline__ void tile_to_global(const float* F, float* __restrict__ dst, long long ld, int m0, int M, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = f & 31;
        if (m0 + row < M)
            *reinterpret_cast<f32x4*>(dst + (size_t)(m0 + row) * ld + 4 * c4) = *reinterpret_cast<const f32x4*>(F + row * LX_FLD + 4 * c4);
    }
}

// fp32 rows (already merged, or merged here from the attention's key-split partials) -> fp16 hi/lo activation tile
__device__ __forceinline__ void rows_to_x(const LayerX3Args& a, const float* __restrict__ src, bool merge, int b, int m0, int M,
                                          sp16* Xh, sp16* Xl, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
        const int m = min(m0 + row, M - 1);
        f32x4 v;
        if (!merge) {
            v = *reinterpret_cast<const f32x4*>(src + (size_t)m * PDSC_CHANNELS + c4);
        } else {
            const size_t slot0 = (size_t)b * a.nsplit * a.Npad + (size_t)(m - b * a.N);
            v = merge_partials_chunk(a.part_o, a.part_ml, slot0, (size_t)a.Npad, a.nsplit, c4);
        }
        sp16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sp16 x, y; split_sp16(v[e], x, y); hi[e] = x; lo[e] = y; }
        *reinterpret_cast<sp16x4*>(Xh + row * LX_XLD + c4) = hi;
        *reinterpret_cast<sp16x4*>(Xl + row * LX_XLD + c4) = lo;
    }
}

// 32x128 fp32 staging tile (one of q / k / v for 32 points = one key tile) -> fp16 hi/lo streams (split_layout.h)
template <int WHICH>
__device__ __forceinline__ void stage_to_split(const float* F, sp16* __restrict__ qrows, unsigned char* __restrict__ img,
                                               int valid, int t) {
    if (WHICH == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
            if (row < valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(F + row * LX_FLD + c4);
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
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(F + key * LX_FLD + 8 * chunk);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(F + key * LX_FLD + 8 * chunk + 4);
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
                const float v = key < valid ? F[key * LX_FLD + ch] : 0.f;
                sp16 x, y; split_sp16(v, x, y); hi[e] = x; lo[e] = y;
            }
            *reinterpret_cast<sp16x8*>(img + SPL_VH + spl_v_offset(ch, jh)) = hi;
            *reinterpret_cast<sp16x8*>(img + SPL_VL + spl_v_offset(ch, jh)) = lo;
        }
    }
}

template <bool HAS_TAIL, bool HAS_HEAD>
__global__ __launch_bounds__(256, 3) void layer_x3_kernel(LayerX3Args a) {
    __shared__ __attribute__((aligned(16))) sp16 Xa[2 * LX_XTILE];     // hi | lo
    __shared__ __attribute__((aligned(16))) sp16 Xb[2 * LX_XTILE];
    __shared__ __attribute__((aligned(16))) float F[LX_ROWS * LX_FLD];
    sp16 *Xah = Xa, *Xal = Xa + LX_XTILE, *Xbh = Xb, *Xbl = Xb + LX_XTILE;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = b * a.N + blockIdx.x * LX_ROWS;     // first row of this tile
    const int M = (b + 1) * a.N;                        // end of this pair's rows
    constexpr int C = PDSC_CHANNELS, H = PDSC_CHANNELS / 2;

    if (HAS_TAIL) {
        sp16x8 w1h[8], w1l[8], w2h[4], w2l[4], w3h[4], w3l[4];
        // ---- fc1: 128 -> 64 (+BN, ReLU): tiles {0,1} on waves {0,1} ----
        if (wave < 2) load_w<C>(a.w1, H, 32 * wave, l31, h, w1h, w1l);
        rows_to_x(a, a.msg, a.msg == nullptr, b, m0, M, Xah, Xal, t);
        __syncthreads();
        if (wave < 2) {
            load_w<H>(a.w2, H, 32 * wave, l31, h, w2h, w2l);                 // prefetch fc2 weights
            const f32x16 acc = mma_tile<C>(w1h, w1l, Xah, Xal, l31, h);
            store_tile<true, false, false, true>(acc, a.b1, 32 * wave, 32 * wave, l31, h, nullptr, nullptr, Xbh, Xbl);
        }
        __syncthreads();
        // ---- fc2: 64 -> 64 (+BN, ReLU) ----
        load_w<H>(a.w3, C, 32 * wave, l31, h, w3h, w3l);                     // prefetch fc3 weights (all waves)
        if (wave < 2) {
            const f32x16 acc = mma_tile<H>(w2h, w2l, Xbh, Xbl, l31, h);
            store_tile<true, false, false, true>(acc, a.b2, 32 * wave, 32 * wave, l31, h, nullptr, nullptr, Xah, Xal);
        }
        __syncthreads();
        // ---- fc3: 64 -> 128, + residual featB: tile = wave; result is the layer's output feature ----
        {
            const f32x16 acc = mma_tile<H>(w3h, w3l, Xah, Xal, l31, h);
            const float* res_row = a.res + (size_t)min(m0 + l31, M - 1) * C;
            if (a.feat_out)
                store_tile<false, true, true, HAS_HEAD>(acc, a.b3, 32 * wave, 32 * wave, l31, h, res_row, F, Xbh, Xbl);
            else
                store_tile<false, true, false, HAS_HEAD>(acc, a.b3, 32 * wave, 32 * wave, l31, h, res_row, F, Xbh, Xbl);
        }
        __syncthreads();
        if (a.feat_out) {
            tile_to_global(F, a.feat_out, C, m0, M, t);
            if (HAS_HEAD) __syncthreads();                                   // F is rewritten by the head
        }
    } else {
        rows_to_x(a, a.feat_in, false, b, m0, M, Xbh, Xbl, t);
        __syncthreads();
    }

    if (HAS_HEAD) {
        // ---- PointCN: 128 -> 128 (+BN, ReLU): tile = wave; input Xb, output Xa (+ fp32 copy for featB_out) ----
        sp16x8 wh[8], wl[8];
        load_w<C>(a.wp, C, 32 * wave, l31, h, wh, wl);
        {
            const f32x16 acc = mma_tile<C>(wh, wl, Xbh, Xbl, l31, h);
            load_w<C>(a.wq, 3 * C, 32 * wave, l31, h, wh, wl);               // prefetch first qkv tile
            store_tile<true, false, true, true>(acc, a.bp, 32 * wave, 32 * wave, l31, h, nullptr, F, Xah, Xal);
        }
        __syncthreads();
        tile_to_global(F, a.featB_out, C, m0, M, t);
        // ---- q|k|v: 128 -> 384 in three 128-column chunks staged through F ----
        unsigned char* img = a.kv ? a.kv + ((size_t)b * gridDim.x + blockIdx.x) * SPL_TILE_STRIDE : nullptr;
        const int valid = min(LX_ROWS, M - m0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n0 = C * c + 32 * wave;
            const f32x16 acc = mma_tile<C>(wh, wl, Xah, Xal, l31, h);
            if (c < 2) load_w<C>(a.wq, 3 * C, n0 + C, l31, h, wh, wl);       // prefetch next chunk's tile
            __syncthreads();                                                 // previous readers of F are done
            store_tile<false, false, true, false>(acc, a.bq, n0, 32 * wave, l31, h, nullptr, F, nullptr, nullptr);
            __syncthreads();
            if (a.qkv_out) tile_to_global(F, a.qkv_out + C * c, 3 * C, m0, M, t);
            if (a.qs) {
                if (c == 0) stage_to_split<0>(F, a.qs + (size_t)m0 * SPL_Q_LD, img, valid, t);
                else if (c == 1) stage_to_split<1>(F, nullptr, img, valid, t);
                else stage_to_split<2>(F, nullptr, img, valid, t);
            }
        }
    }
}

template <bool T, bool H>
static int launch_layer_x3(const LayerX3Args& a, hipStream_t st) {
    hipLaunchKernelGGL((layer_x3_kernel<T, H>), dim3(ceil_div(a.N, LX_ROWS), a.bs), dim3(256), 0, st, a);
    return check_launch("pdsc_layer_fused_x3");
}

#endif  // PDSC_EXPERIMENTS

// fp32 weight matrix [n] -> fp16 hi [n] | fp16 lo [n]
__global__ __launch_bounds__(256) void wsplit_kernel(const float* __restrict__ src, sp16* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        sp16 hi, lo;
        split_sp16(src[i], hi, lo);
        dst[i] = hi;
        dst[n + i] = lo;
    }
}

}  // namespace pdsc

using namespace pdsc;

// element (fp16) offsets inside one layer's block of the split-weight buffer: pcn | qkv | fc1 | fc2 | fc3, each hi then lo
static const long long kWsplitElems[5] = {2LL * PDSC_CHANNELS * PDSC_CHANNELS, 2LL * 3 * PDSC_CHANNELS * PDSC_CHANNELS,
                                          2LL * (PDSC_CHANNELS / 2) * PDSC_CHANNELS, 2LL * (PDSC_CHANNELS / 2) * (PDSC_CHANNELS / 2),
                                          2LL * PDSC_CHANNELS * (PDSC_CHANNELS / 2)};
static const int kWsplitSection[5] = {PDSC_W_PCN_W, PDSC_W_QKV_W, PDSC_W_FC1_W, PDSC_W_FC2_W, PDSC_W_FC3_W};

static long long wsplit_layer_elems() {
    long long n = 0;
    for (int i = 0; i < 5; ++i) n += kWsplitElems[i];
    return n;
}

// after the hi|lo matrices of all layers: the fragment-ordered streams of layer_wave.hip, per layer [tail][head]
// ... in both GEMM formats (enum pdsc_layer_gemm): [tail F32][head F32][tail H3][head H3]
static long long wsplit_frag_layer_elems() { return (long long)(pdsc_wfrag_tail_bytes() + pdsc_wfrag_head_bytes()); }

extern "C" long long pdsc_wsplit_offset(const pdsc_config* cfg, int section, int layer) {
    if (!cfg || layer < 0 || layer >= cfg->num_layers) return -1;
    if (section >= PDSC_WS_FRAG_TAIL && section <= PDSC_WS_FRAG_HEAD_H3) {
        const long long tail = (long long)pdsc_wfrag_tail_bytes() / 2, head = (long long)pdsc_wfrag_head_bytes() / 2;
        const long long inside = section == PDSC_WS_FRAG_TAIL ? 0 : section == PDSC_WS_FRAG_HEAD ? tail
                               : section == PDSC_WS_FRAG_TAIL_H3 ? tail + head : 2 * tail + head;
        return (long long)cfg->num_layers * wsplit_layer_elems() + (long long)layer * wsplit_frag_layer_elems() + inside;
    }
    long long off = (long long)layer * wsplit_layer_elems();
    for (int i = 0; i < 5; ++i) {
        if (kWsplitSection[i] == section) return off;
        off += kWsplitElems[i];
    }
    return -1;
}

extern "C" size_t pdsc_wsplit_bytes(const pdsc_config* cfg) {
    if (!cfg || cfg->num_layers < 0) return 0;
    return (size_t)cfg->num_layers * (wsplit_layer_elems() + wsplit_frag_layer_elems()) * sizeof(sp16);
}

extern "C" int pdsc_wsplit_build(const pdsc_config* cfg, const float* wpack, void* wsplit, void* stream) {
    PDSC_REQUIRE(cfg && wpack && wsplit, "pdsc_wsplit_build: null pointer");
    for (int layer = 0; layer < cfg->num_layers; ++layer) {
        for (int i = 0; i < 5; ++i) {
            const long long src = pdsc_wpack_offset(cfg, kWsplitSection[i], layer), dst = pdsc_wsplit_offset(cfg, kWsplitSection[i], layer);
            PDSC_REQUIRE(src >= 0 && dst >= 0, "pdsc_wsplit_build: bad section");
            const long long n = kWsplitElems[i] / 2;
            hipLaunchKernelGGL(wsplit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wpack + src,
                               (sp16*)wsplit + dst, n);
        }
        auto W = [&](int section) { return wpack + pdsc_wpack_offset(cfg, section, layer); };
        for (int fmt = PDSC_LAYER_GEMM_F32; fmt <= PDSC_LAYER_GEMM_H3; ++fmt) {
            const bool h3 = fmt == PDSC_LAYER_GEMM_H3;
            int rc = pdsc_wfrag_build_tail_fmt(W(PDSC_W_FC1_W), W(PDSC_W_FC1_B), W(PDSC_W_FC2_W), W(PDSC_W_FC2_B), W(PDSC_W_FC3_W), W(PDSC_W_FC3_B),
                                               (sp16*)wsplit + pdsc_wsplit_offset(cfg, h3 ? PDSC_WS_FRAG_TAIL_H3 : PDSC_WS_FRAG_TAIL, layer),
                                               fmt, stream);
            if (rc != PDSC_OK) return rc;
            rc = pdsc_wfrag_build_head_fmt(W(PDSC_W_PCN_W), W(PDSC_W_PCN_B), W(PDSC_W_QKV_W), W(PDSC_W_QKV_B),
                                           (sp16*)wsplit + pdsc_wsplit_offset(cfg, h3 ? PDSC_WS_FRAG_HEAD_H3 : PDSC_WS_FRAG_HEAD, layer),
                                           fmt, stream);
            if (rc != PDSC_OK) return rc;
        }
    }
    return check_launch("pdsc_wsplit_build");
}

extern "C" int pdsc_layer_fused_x3(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                                   const float* res, const float* feat_in, float* feat_out, float* featB_out, float* qkv_out,
                                   void* q_split, void* kv_tiles, const void* w1, const float* b1, const void* w2,
                                   const float* b2, const void* w3, const float* b3, const void* wp, const float* bp,
                                   const void* wq, const float* bq, int bs, int N, void* stream) {
#ifndef PDSC_EXPERIMENTS
    (void)msg; (void)part_o; (void)part_ml; (void)nsplit; (void)Npad; (void)res; (void)feat_in; (void)feat_out; (void)featB_out; (void)qkv_out;
    (void)q_split; (void)kv_tiles; (void)w1; (void)b1; (void)w2; (void)b2; (void)w3; (void)b3; (void)wp; (void)bp; (void)wq; (void)bq; (void)bs; (void)N; (void)stream;
    set_error("pdsc_layer_fused_x3: the all-split layer kernel exists in experiments builds only (python -m pointdsc_amd.build --experiments)");
    return PDSC_ERR_ARG;
#else
    const bool tail = msg != nullptr || part_o != nullptr, head = featB_out != nullptr;
    PDSC_REQUIRE(tail || head, "pdsc_layer_fused_x3: neither tail (msg / partials) nor head (featB_out) requested");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_layer_fused_x3: bs=%d N=%d", bs, N);
    if (tail) {
        PDSC_REQUIRE(res && w1 && b1 && w2 && b2 && w3 && b3, "pdsc_layer_fused_x3: tail needs res, fc1..fc3");
        if (!msg) PDSC_REQUIRE(part_ml && nsplit >= 1 && nsplit <= MERGE_MAX_SPLIT && Npad >= N,
                               "pdsc_layer_fused_x3: partials need part_ml, 1 <= nsplit <= %d, Npad >= N", MERGE_MAX_SPLIT);
        PDSC_REQUIRE(head || feat_out, "pdsc_layer_fused_x3: tail-only needs feat_out");
    } else {
        PDSC_REQUIRE(feat_in, "pdsc_layer_fused_x3: head-only needs feat_in");
    }
    if (head) PDSC_REQUIRE((qkv_out || q_split) && wp && bp && wq && bq, "pdsc_layer_fused_x3: head needs qkv_out or the split streams, pcn, qkv weights");
    PDSC_REQUIRE((q_split == nullptr) == (kv_tiles == nullptr), "pdsc_layer_fused_x3: q_split and kv_tiles go together");
    LayerX3Args a{};
    a.msg = msg; a.part_o = part_o; a.part_ml = part_ml; a.nsplit = nsplit; a.Npad = Npad;
    a.res = res; a.feat_in = feat_in; a.feat_out = feat_out; a.featB_out = featB_out; a.qkv_out = qkv_out;
    a.qs = (sp16*)q_split; a.kv = (unsigned char*)kv_tiles;
    a.w1 = (const sp16*)w1; a.w2 = (const sp16*)w2; a.w3 = (const sp16*)w3; a.wp = (const sp16*)wp; a.wq = (const sp16*)wq;
    a.b1 = b1; a.b2 = b2; a.b3 = b3; a.bp = bp; a.bq = bq;
    a.N = N; a.bs = bs;
    hipStream_t st = (hipStream_t)stream;
    if (tail && head) return launch_layer_x3<true, true>(a, st);
    if (tail) return launch_layer_x3<true, false>(a, st);
    return launch_layer_x3<false, true>(a, st);
#endif
}
