// Building blocks of the wavefront-resident fused layer kernel (layer_wave.hip): weight-chunk streams, MFMA steps,
// fp16 hi/lo splitting, the wave-private LDS patch.
#pragma once
#include <type_traits>
#include "pdsc_common.h"
#include "split_layout.h"
#include "merge_partials.h"
#include "layer_args.h"

namespace pdsc {

constexpr int LW_WAVES = 4;      // independent wavefronts per workgroup
constexpr int LW_VLD = 36;       // floats per key row of the V transpose patch (32 channels + 4 pad)
constexpr int LW_QKV_BUFS = 4;   // weight-chunk buffers (32 registers each) during the split q|k|v projection
#ifndef LW_H3_BUFS
#define LW_H3_BUFS 3             // ... during every stage of the H3 variant (4 spills: 256 registers with 3)
#endif

struct WChunk {
    f32x4 v[8];      // 8 fp32 k-steps (q) of a weight tile, or 4 fp16 k-steps as (hi, lo) pairs
    float bias;      // first chunk of a tile only: bias of output channel n0 + l31 in lane-half 0, zero in lane-half 1
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

enum { ST_FC1 = 0, ST_FC2, ST_FC3, ST_PCN, ST_QKV };
struct ChunkDesc { int stage, tile, chunk, nchunks; };

template <bool T, bool H>
constexpr int num_chunks() { return (T ? 10 : 0) + (H ? 32 : 0); }

template <bool T>
constexpr ChunkDesc chunk_desc(int i) {
    if (T) {
        if (i < 4) return {ST_FC1, i / 2, i % 2, 2};
        i -= 4;
        if (i < 2) return {ST_FC2, i, 0, 1};
        i -= 2;
        if (i < 4) return {ST_FC3, i, 0, 1};
        i -= 4;
    }
    if (i < 8) return {ST_PCN, i / 2, i % 2, 2};
    i -= 8;
    return {ST_QKV, i / 2, i % 2, 2};
}

// ordinal of the output tile inside its stream (tail: fc1 0..1, fc2 2..3, fc3 4..7; head: pcn 0..3, q|k|v 4..15)
constexpr int tile_ordinal(const ChunkDesc d) {
    return d.stage == ST_FC1 ? d.tile : d.stage == ST_FC2 ? 2 + d.tile : d.stage == ST_FC3 ? 4 + d.tile : d.stage == ST_PCN ? d.tile : 4 + d.tile;
}
constexpr int LW_TAIL_CHUNKS = 10, LW_HEAD_CHUNKS = 32, LW_TAIL_TILES = 8, LW_HEAD_TILES = 16;

__device__ __forceinline__ void load_rows_f32(WChunk& w, const float* __restrict__ p) {
#pragma unroll
    for (int q = 0; q < 8; ++q) w.v[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
}

constexpr int LW_CHUNK_BYTES = 8192;      // one WChunk for all 64 lanes

// chunk `idx` of a fragment-ordered stream (pdsc_wfrag_build_*): slot s of lane l is the 16 bytes at s*1024 + l*16, i.e.
// every load instruction of the wave reads 1 KiB of consecutive memory (8 cache lines).  The natural [out][in] layout
// below makes the same instruction touch 64 different lines (row stride 256..512 B), which serialises in the L1.
// The bias fragments (256 B per output tile) follow the chunks of the stream.
__device__ __forceinline__ void load_chunk_frag(WChunk& w, const unsigned char* __restrict__ stream, int idx, int nchunks,
                                                int bias_tile /* -1: not the first chunk of a tile */, int lane) {
    const unsigned char* p = stream + (size_t)idx * LW_CHUNK_BYTES + lane * 16;
#pragma unroll
    for (int s = 0; s < 8; ++s) w.v[s] = *reinterpret_cast<const f32x4*>(p + 1024 * s);
    if (bias_tile >= 0) w.bias = *reinterpret_cast<const float*>(stream + (size_t)nchunks * LW_CHUNK_BYTES + bias_tile * 256 + lane * 4);
}

template <bool T, bool X3, bool FRAG>
__device__ __forceinline__ void load_chunk(WChunk& w, const LayerArgs& a, const int i, int lane) {
    const ChunkDesc d = chunk_desc<T>(i);
    const int bias_tile = d.chunk == 0 ? tile_ordinal(d) : -1;
    if (FRAG) {
        if (T && i < LW_TAIL_CHUNKS) load_chunk_frag(w, a.wf_tail, i, LW_TAIL_CHUNKS, bias_tile, lane);
        else load_chunk_frag(w, a.wf_head, i - (T ? LW_TAIL_CHUNKS : 0), LW_HEAD_CHUNKS, bias_tile, lane);
        return;
    }
    const int l31 = lane & 31, h = lane >> 5;
    const int n = 32 * d.tile + l31;
    if (d.chunk == 0) {
        const float* b = d.stage == ST_FC1 ? a.b1 : d.stage == ST_FC2 ? a.b2 : d.stage == ST_FC3 ? a.b3 : d.stage == ST_PCN ? a.bp : a.bq;
        const float bv = b[n];
        w.bias = h ? 0.f : bv;
    }
    switch (d.stage) {
        case ST_FC1: load_rows_f32(w, a.w1 + (size_t)n * 128 + 64 * d.chunk + 4 * h); break;
        case ST_FC2: load_rows_f32(w, a.w2 + (size_t)n * 64 + 4 * h); break;
        case ST_FC3: load_rows_f32(w, a.w3 + (size_t)n * 64 + 4 * h); break;
        case ST_PCN: load_rows_f32(w, a.wp + (size_t)n * 128 + 64 * d.chunk + 4 * h); break;
        default:
            if (X3) {
                const sp16* p = a.wq_split + (size_t)n * PDSC_CHANNELS + 64 * d.chunk + 8 * h;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w.v[2 * k] = *reinterpret_cast<const f32x4*>(p + 16 * k);
                    w.v[2 * k + 1] = *reinterpret_cast<const f32x4*>(p + (size_t)3 * PDSC_CHANNELS * PDSC_CHANNELS + 16 * k);
                }
            } else {
                load_rows_f32(w, a.wq + (size_t)n * 128 + 64 * d.chunk + 4 * h);
            }
    }
}

__device__ __forceinline__ void mma_f32(f32x16& acc, const WChunk& w, const f32x4* x) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v[q][e], x[q][e], acc, 0, 0, 0);
}

__device__ __forceinline__ void mma_x3(f32x16& acc, const WChunk& w, const sp16x8* xh, const sp16x8* xl) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const sp16x8 wh = __builtin_bit_cast(sp16x8, w.v[2 * i]), wl = __builtin_bit_cast(sp16x8, w.v[2 * i + 1]);
        acc = PDSC_MFMA_X3(wl, xh[i], acc, 0, 0, 0);
        acc = PDSC_MFMA_X3(wh, xl[i], acc, 0, 0, 0);
        acc = PDSC_MFMA_X3(wh, xh[i], acc, 0, 0, 0);
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(sp16 a, sp16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> packed hi pair and lo pair of the x3 split (split_layout.h: split_sp16_pair)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) { split_sp16_pair(x0, x1, hi, lo); }

// fp32 x4 -> packed hi (2 registers) and lo (2 registers)
__device__ __forceinline__ void split4(const f32x4& v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    split2(v[0], v[1], hi[0], lo[0]);
    split2(v[2], v[3], hi[1], lo[1]);
}

// ---- fp16 hi / scaled-lo split (the H3 arithmetic of the fc1..fc3 / PointCN GEMMs) -----------------------------------
// x = hi + lo' / 2048 with hi = f16(x), lo' = f16((x - hi) * 2048), both round-to-nearest-even.  hi carries 11
// significant bits, lo' the next 11: a product a*b evaluated as  a_hi*b_hi + (a_hi*b_lo' + a_lo'*b_hi) / 2048  on
// v_mfma_f32_32x32x16_f16 (fp32 accumulate, the two cross terms in their own accumulator) is exact to ~2^-21 relative,
// which is what lets the GEMMs that land on the residual stream leave the fp32 MFMA (1/16 of the f16 rate).  The scale keeps
// lo' in the normal range of fp16 whenever hi is (the attention operands' split, split_layout.h, is the same pair WITHOUT the
// scale -- one accumulator for all three terms -- and therefore has an absolute floor of 2^-25 on lo).  Range: |x| < 65504 (fp16); activations and weights of this network are O(1).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float H3_SCALE = 2048.0f, H3_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void split2h(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f16x2 hv = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    hi = __builtin_bit_cast(unsigned, hv);
    const f32x2 hf = __builtin_convertvector(hv, f32x2);
    const f16x2 lv = __builtin_convertvector(f32x2{(x0 - hf[0]) * H3_SCALE, (x1 - hf[1]) * H3_SCALE}, f16x2);
    lo = __builtin_bit_cast(unsigned, lv);
}

__device__ __forceinline__ void split4h(const f32x4& v, unsigned (&hi)[2], unsigned (&lo)[2]) {
    split2h(v[0], v[1], hi[0], lo[0]);
    split2h(v[2], v[3], hi[1], lo[1]);
}

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.
// afterwards, per lane: lower half (a, b) = (own a, partner's a);  upper half (a, b) = (partner's b, own b).
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// Lane-half h holds 4 consecutive channels (c0+4h..+3) of a value as packed hi[2] / lo[2].  Returns the 16-byte chunk
// this lane stores: lower half -> the 8 hi values of channels c0..c0+7, upper half -> the 8 lo values.
__device__ __forceinline__ u32x4 chunk_for_store(const unsigned (&hi)[2], const unsigned (&lo)[2]) {
    unsigned a0 = hi[0], b0 = lo[0], a1 = hi[1], b1 = lo[1];
    half_swap(a0, b0);        // lower: (own hi0, partner hi0); upper: (partner lo0, own lo0)
    half_swap(a1, b1);
    // lower half: own = channels 0..3, partner = 4..7 -> (a0, a1, b0, b1); upper: partner = 0..3 (a), own = 4..7 (b)
    return u32x4{a0, a1, b0, b1};
}

// B operand of one 16-wide k-step kk from the fp32 accumulator layout: lane-half h holds `a` = channels 16kk + 4h + e and
// `b` = channels 16kk + 8 + 4h + e of its point; the k-step wants channels 16kk + 8h .. +7 in lane-half h.  F16 selects the
// fp16 hi / scaled-lo split (H3) instead of the unscaled fp16 hi / lo split.
template <bool F16>
__device__ __forceinline__ void make_kstep(const f32x4& a_in, const f32x4& b_in, u32x4& oh, u32x4& ol, float& rmax) {
    f32x4 a = a_in, b = b_in;
    range_note(rmax, a);
    range_note(rmax, b);
    unsigned ha[2], la[2], hb[2], lb[2];
    if constexpr (F16) { split4h(a, ha, la); split4h(b, hb, lb); }
    else { split4(a, ha, la); split4(b, hb, lb); }
    half_swap(ha[0], hb[0]); half_swap(ha[1], hb[1]);
    half_swap(la[0], lb[0]); half_swap(la[1], lb[1]);
    oh = u32x4{ha[0], ha[1], hb[0], hb[1]};
    ol = u32x4{la[0], la[1], lb[0], lb[1]};
}

template <bool F16>
__device__ __forceinline__ void make_kstep(const f32x4& a, const f32x4& b, u32x4& oh, u32x4& ol) {
    unsigned ha[2], la[2], hb[2], lb[2];
    if constexpr (F16) { split4h(a, ha, la); split4h(b, hb, lb); }
    else { split4(a, ha, la); split4(b, hb, lb); }
    half_swap(ha[0], hb[0]); half_swap(ha[1], hb[1]);
    half_swap(la[0], lb[0]); half_swap(la[1], lb[1]);
    oh = u32x4{ha[0], ha[1], hb[0], hb[1]};
    ol = u32x4{la[0], la[1], lb[0], lb[1]};
}

// one weight chunk (4 k-steps as (hi, lo') slot pairs) in the H3 arithmetic: main accumulator <- hi*hi, cross <- the two
// hi*lo' terms (scaled by 2048; folded in by the tile epilogue)
__device__ __forceinline__ void mma_h3(f32x16& acc, f32x16& cross, const WChunk& w, const u32x4* xh, const u32x4* xl, bool first) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x8 wh = __builtin_bit_cast(f16x8, w.v[2 * i]), wl = __builtin_bit_cast(f16x8, w.v[2 * i + 1]);
        const f16x8 bh = __builtin_bit_cast(f16x8, xh[i]), bl = __builtin_bit_cast(f16x8, xl[i]);
        cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, (first && i == 0) ? zero : cross, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc, 0, 0, 0);
        cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, cross, 0, 0, 0);
    }
}

// ---- output staging ---------------------------------------------------------------------------------------------------
// In the accumulator layout a lane owns 16 bytes of 32 different rows, so a direct global store touches 64 cache lines
// and costs the L1 64 cycles (measured: 73 us of a 310 us launch at 32 pairs went into such stores).  Outputs therefore
// pass through a wave-private LDS patch of 32 rows x 144 B (128 B payload + 16 B pad) and leave as contiguous runs:
// lane = (row 8*it + lane/8, 16-byte piece lane%8), four passes.  Wave-private: ordering needs no workgroup barrier.
constexpr int LW_PROW = LW_VLD * 4;        // bytes per patch row

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace pdsc
