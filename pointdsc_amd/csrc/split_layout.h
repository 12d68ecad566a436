// bf16x3 operand streams of the split-precision attention (attention_split.hip), produced by layer.hip's head
// epilogue or by pdsc_pack_qkv_split.
//
// An fp32 value x is carried as two bf16 numbers  hi = bf16(x), lo = bf16(x - hi)  (round to nearest even both
// times; hi + lo reproduces x to 2^-17 relative).  A product a*b is evaluated on the bf16 matrix cores as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (the dropped a_lo*b_lo term is 2^-16 relative), i.e.
// three v_mfma_f32_32x32x16_bf16 per operand pair = 3/16 of the cost of the exact fp32 MFMA.
//
//   Q stream : [bs*N][256] bf16   row = (hi[0..127] | lo[0..127]), natural channel order
//   KV stream: [bs][ntiles][SPL_TILE_STRIDE]  one 32 KiB block (SPL_TILE_BYTES) per tile of 32 keys, laid out as the exact LDS image
//              the attention kernel wants, so that the LDS-DMA copy is linear and fully coalesced.  Both operands are
//              CHUNK-MAJOR (r02; r01 had key / channel rows with a pad chunk, 37 KiB):
//       +SPL_KH / +SPL_KL : K hi / lo   [16 chunks of 8 channels][32 keys][16 B]  (8 KiB per plane).  The MFMA A fragment
//                           of lane (key l31, half h) in k-step j is chunk 2j+h of key l31: the 32 lanes of a half read 512
//                           consecutive bytes -- the access the ds_read_b128 lane groups are built for, conflict-free with no
//                           padding -- and every read of a lane is `lane base + immediate` (+1 KiB per j).
//       +SPL_VH / +SPL_VL : V^T hi / lo [4 chunks of 8 keys][128 channels][16 B]  (8 KiB per plane); chunk jh = 2j+h
//                           holds, in order e = 0..7, keys 16j + 8(e>>2) + 4h + (e&3) -- the keys lane-half h holds in
//                           accumulator registers 8j..8j+7 of S^T = K Q^T.  Lane (channel 32c + l31, half h), step j reads
//                           chunk 2j+h of its channel: again 512 consecutive bytes per lane half.
//              On the producer side a chunk of one key / channel range is a run of whole, aligned cache lines: the layer
//              kernel's lane (key l31, half h) stores its K chunk straight from registers (2 x 512 B per instruction).
//   keys >= N of the last tile are zero.  No pad bytes.
#pragma once
#include "pdsc_common.h"

namespace pdsc {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPL_BK = 32;                       // keys per tile
constexpr int SPL_K_PLANE = 16 * 32 * 16, SPL_V_PLANE = 4 * 128 * 16;      // bytes per K / V^T plane (hi or lo): 8 KiB each
constexpr int SPL_KH = 0, SPL_KL = SPL_K_PLANE, SPL_VH = 2 * SPL_K_PLANE, SPL_VL = SPL_VH + SPL_V_PLANE;
constexpr int SPL_TILE_BYTES = SPL_VL + SPL_V_PLANE;     // 32768 = 32 KiB
// Images sit SPL_TILE_STRIDE apart in HBM, not back to back: an odd number of KiB keeps the eight XCDs, which walk their tile
// ranges in step, off a power-of-two address stride (precaution: with the un-pinned attention loop 32 and 37 KiB measured the
// same; r01's image size kept).  The 5 KiB between images are never read or written.
constexpr int SPL_TILE_STRIDE = 37 * 1024;
constexpr int SPL_Q_LD = 2 * PDSC_CHANNELS;     // bf16 elements per row of the Q stream

__host__ __device__ __forceinline__ int spl_k_offset(int key, int chunk) { return (chunk << 9) + (key << 4); }    // chunk 0..15
__host__ __device__ __forceinline__ int spl_v_offset(int ch, int jh) { return (jh << 11) + (ch << 4); }            // jh 0..3
__host__ __device__ __forceinline__ int spl_v_key(int jh, int e) { return 16 * (jh >> 1) + 8 * (e >> 2) + 4 * (jh & 1) + (e & 3); }

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
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
