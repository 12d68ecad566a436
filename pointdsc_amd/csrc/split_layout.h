// bf16x3 operand streams of the split-precision attention (attention_split.hip), produced by layer.hip's head
// epilogue or by pdsc_pack_qkv_split.
//
// An fp32 value x is carried as two bf16 numbers  hi = bf16(x), lo = bf16(x - hi)  (round to nearest even both
// times; hi + lo reproduces x to 2^-17 relative).  A product a*b is evaluated on the bf16 matrix cores as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (the dropped a_lo*b_lo term is 2^-16 relative), i.e.
// three v_mfma_f32_32x32x16_bf16 per operand pair = 3/16 of the cost of the exact fp32 MFMA.
//
//   Q stream : [bs*N][256] bf16   row = (hi[0..127] | lo[0..127]), natural channel order
//   KV stream: [bs][ntiles][SPL_TILE_BYTES]  one 37 KiB block per tile of 32 keys, laid out as the exact LDS image
//              the attention kernel wants, so that the LDS-DMA copy is linear and fully coalesced:
//       +SPL_KH / +SPL_KL : K hi / lo   [32 keys][SPL_K_STRIDE = 272 B]: 16 chunks of 8 channels (16 B) + one pad chunk.
//                           The odd number of chunks per row rotates consecutive keys by one 16-B bank slot, so the
//                           column-slice ds_read_b128 (16 different keys, same chunk) is conflict-free AND every read of
//                           a lane is `lane base + immediate` (an XOR swizzle would need one address register per chunk).
//       +SPL_VH / +SPL_VL : V^T hi / lo [128 channels][SPL_V_STRIDE = 80 B]: 4 chunks of 8 keys (16 B) + one pad chunk;
//                           chunk jh = 2j+h holds, in order e = 0..7, keys 16j + 8(e>>2) + 4h + (e&3) -- the keys
//                           lane-half h holds in accumulator registers 8j..8j+7 of S^T = K Q^T.
//   keys >= N of the last tile and all pad chunks are zero.
#pragma once
#include "pdsc_common.h"

namespace pdsc {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPL_BK = 32;                       // keys per tile
constexpr int SPL_K_STRIDE = 272, SPL_V_STRIDE = 80;      // bytes per K row (key) / V^T row (channel), pad chunk included
constexpr int SPL_KH = 0, SPL_KL = 32 * SPL_K_STRIDE, SPL_VH = 2 * SPL_KL, SPL_VL = SPL_VH + 128 * SPL_V_STRIDE;
constexpr int SPL_TILE_BYTES = SPL_VL + 128 * SPL_V_STRIDE;     // 37888 = 37 KiB
constexpr int SPL_Q_LD = 2 * PDSC_CHANNELS;     // bf16 elements per row of the Q stream

__host__ __device__ __forceinline__ int spl_k_offset(int key, int chunk) { return key * SPL_K_STRIDE + (chunk << 4); }   // chunk 16 = pad
__host__ __device__ __forceinline__ int spl_v_offset(int ch, int jh) { return ch * SPL_V_STRIDE + (jh << 4); }          // jh 4 = pad
__host__ __device__ __forceinline__ int spl_v_key(int jh, int e) { return 16 * (jh >> 1) + 8 * (e >> 2) + 4 * (jh & 1) + (e & 3); }

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// pad chunks of one tile image (never read by the attention kernel; zeroed so the stream is deterministic)
__device__ __forceinline__ void spl_zero_pads(unsigned char* __restrict__ img, int t) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (t < 32) {
        *reinterpret_cast<f32x4*>(img + SPL_KH + spl_k_offset(t, 16)) = z;
        *reinterpret_cast<f32x4*>(img + SPL_KL + spl_k_offset(t, 16)) = z;
    }
    if (t < 128) {
        *reinterpret_cast<f32x4*>(img + SPL_VH + spl_v_offset(t, 4)) = z;
        *reinterpret_cast<f32x4*>(img + SPL_VL + spl_v_offset(t, 4)) = z;
    }
}

static inline int spl_num_tiles(int N) { return ceil_div(N, SPL_BK); }

// ---- point-fragment order (PF) of a [rows][128] fp32 matrix -----------------------------------------------------------
// Hand-off format between kernels whose lanes ARE points (the attention's accumulators -> the fused layer kernel's MFMA
// operands -> the next layer kernel's residual).  Rows are taken in tiles of 32; a tile is 16 KiB (as in row order) laid
// out as [q = 0..15][lane = 0..63][4 floats]: lane (l31 = lane & 31, h = lane >> 5) holds channels 8q + 4h .. + 3 of row
// l31 of the tile.  One wave instruction (q fixed) moves 1 KiB of consecutive memory -- in row order the same instruction
// touches 32 different cache lines for 32 bytes each.  Buffers are padded to whole tiles per pair; padding rows hold
// copies of the pair's last row.
__host__ __device__ __forceinline__ int pf_offset_floats(int q) { return q * 256; }     // + lane * 4 from the tile base
constexpr int PF_TILE_FLOATS = 32 * PDSC_CHANNELS;

}  // namespace pdsc
