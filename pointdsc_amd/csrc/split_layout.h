// bf16x3 operand streams of the split-precision attention (attention_split.hip), produced by layer.hip's head
// epilogue or by pdsc_pack_qkv_split.
//
// An fp32 value x is carried as two bf16 numbers  hi = bf16(x), lo = bf16(x - hi)  (round to nearest even both
// times; hi + lo reproduces x to 2^-17 relative).  A product a*b is evaluated on the bf16 matrix cores as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (the dropped a_lo*b_lo term is 2^-16 relative), i.e.
// three v_mfma_f32_32x32x16_bf16 per operand pair = 3/16 of the cost of the exact fp32 MFMA.
//
//   Q stream : [bs*N][256] bf16   row = (hi[0..127] | lo[0..127]), natural channel order
//   KV stream: [bs][ntiles][SPL_TILE_BYTES]  one 32 KiB block per tile of 32 keys, laid out as the exact LDS image
//              the attention kernel wants, so that the LDS-DMA copy is linear and fully coalesced:
//       +SPL_KH / +SPL_KL : K hi / lo   [32 keys][16 chunks of 8 channels (16 B)], chunk stored at chunk ^ (key & 15)
//                           (XOR swizzle: the column-slice ds_read_b128 of 16 different keys hits 16 different bank slots)
//       +SPL_VH / +SPL_VL : V^T hi / lo [128 channels][4 chunks of 8 keys (16 B)]; chunk jh = 2j+h holds, in order
//                           e = 0..7, keys 16j + 8(e>>2) + 4h + (e&3) -- the keys lane-half h holds in accumulator
//                           registers 8j..8j+7 of S^T = K Q^T -- stored at jh ^ ((channel>>2) & 3)
//   keys >= N of the last tile are zero.
#pragma once
#include "pdsc_common.h"

namespace pdsc {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPL_BK = 32;                       // keys per tile
constexpr int SPL_TILE_BYTES = 32768;
constexpr int SPL_KH = 0, SPL_KL = 8192, SPL_VH = 16384, SPL_VL = 24576;
constexpr int SPL_Q_LD = 2 * PDSC_CHANNELS;     // bf16 elements per row of the Q stream

__host__ __device__ __forceinline__ int spl_k_offset(int key, int chunk) { return key * 256 + ((chunk ^ (key & 15)) << 4); }
__host__ __device__ __forceinline__ int spl_v_offset(int ch, int jh) { return ch * 64 + ((jh ^ ((ch >> 2) & 3)) << 4); }
__host__ __device__ __forceinline__ int spl_v_key(int jh, int e) { return 16 * (jh >> 1) + 8 * (e >> 2) + 4 * (jh & 1) + (e & 3); }

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

static inline int spl_num_tiles(int N) { return ceil_div(N, SPL_BK); }

}  // namespace pdsc
