// x3 operand streams of the split-precision attention (attention_split.hip), produced by the layer kernels' head epilogues or by
// pdsc_pack_qkv_split.
//
// An fp32 value x is carried as two fp16 numbers  hi = f16(x), lo = f16(x - hi)  (round to nearest even both times; lo is NOT
// scaled: hi carries 11 significant bits, lo the next 11 down to fp16's denormal floor of 6e-8 -- gfx950's f16 MFMA takes denormal
// inputs exactly, tools/f16_mfma_denorm_probe.hip -- so hi + lo reproduces x to max(2^-22 |x|, 3e-8)).  A product a*b is evaluated
// on the f16 matrix cores as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (11 x 11 bits: every partial product is exact
// in fp32; the dropped a_lo*b_lo term is 2^-22 relative), i.e. three v_mfma_f32_32x32x16_f16 per operand pair = 3/16 of the cost of
// the exact fp32 MFMA, ONE accumulator.  (r01-r04 carried bf16 hi/lo -- 8 + 8 bits, 2^-17 per operand: on trained KITTI-scale
// weights that left the logits 6e-3 off, profiles/r05_f_kitti_stage_trained.txt; fp16 costs the same MFMAs and conversions and
// needs |x| < 65504, which pdsc_encoder_range_probe checks for every activation that becomes such an operand.)
//
//   Q stream : [bs*N][256] fp16   row = (hi[0..127] | lo[0..127]), natural channel order
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

// (sp16, sp16x2/4/8, PDSC_MFMA_X3: pdsc_common.h)

constexpr int SPL_BK = 32;                       // keys per tile
constexpr int SPL_K_PLANE = 16 * 32 * 16, SPL_V_PLANE = 4 * 128 * 16;      // bytes per K / V^T plane (hi or lo): 8 KiB each
constexpr int SPL_KH = 0, SPL_KL = SPL_K_PLANE, SPL_VH = 2 * SPL_K_PLANE, SPL_VL = SPL_VH + SPL_V_PLANE;
constexpr int SPL_TILE_BYTES = SPL_VL + SPL_V_PLANE;     // 32768 = 32 KiB
// Images sit SPL_TILE_STRIDE apart in HBM, not back to back: an odd number of KiB keeps the eight XCDs, which walk their tile
// ranges in step, off a power-of-two address stride (precaution: with the un-pinned attention loop 32 and 37 KiB measured the
// same; r01's image size kept).  The 5 KiB between images are never read or written.
constexpr int SPL_TILE_STRIDE = 37 * 1024;
constexpr int SPL_Q_LD = 2 * PDSC_CHANNELS;     // 16-bit elements per row of the Q stream

__host__ __device__ __forceinline__ int spl_k_offset(int key, int chunk) { return (chunk << 9) + (key << 4); }    // chunk 0..15
__host__ __device__ __forceinline__ int spl_v_offset(int ch, int jh) { return (jh << 11) + (ch << 4); }            // jh 0..3
__host__ __device__ __forceinline__ int spl_v_key(int jh, int e) { return 16 * (jh >> 1) + 8 * (e >> 2) + 4 * (jh & 1) + (e & 3); }

__device__ __forceinline__ void split_sp16(float x, sp16& hi, sp16& lo) {
    hi = (sp16)x;
    lo = (sp16)(x - (float)hi);
}
// the same for a pair, with packed conversions (v_cvt_pk_f16_f32, round to nearest even): 6 VALU operations per pair;
// hi / lo = the two packed halves as one 32-bit register each
typedef float spf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_sp16_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const sp16x2 hv = __builtin_convertvector(spf32x2{x0, x1}, sp16x2);
    hi = __builtin_bit_cast(unsigned, hv);
    // lo = f16(x - hi) straight from the packed halves: v_fma_mixlo_f16 / v_fma_mixhi_f16 evaluate fma(hi, -1, x) with the fp16
    // source picked by op_sel and round the (exact: x - hi fits 13 bits) result into the low / high half of the destination --
    // two instructions per pair where convert-back, subtract, convert-again took five
    unsigned lv;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv) : "v"(hi), "v"(x1));
    lo = lv;
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
