// Arguments shared by the two implementations of the fused point-wise layer (layer.hip: one 4-wave workgroup per
// 32-point tile, activations in LDS; layer_wave.hip: one wavefront per tile, activations in registers).
#pragma once
#include "pdsc_common.h"

namespace pdsc {

struct LayerArgs {
    const float* msg;        // [M][128]  attention output            (tail), or NULL when the partials below are given
    const float* part_o;     // [bs][nsplit][Npad][128] un-normalised partial outputs of the attention key splits
    const float* part_ml;    // [bs][nsplit][Npad][2]   (reference exponent (log2), partial sum)
    int nsplit, Npad;
    const float* res;        // [M][128]  featB of this layer         (tail residual)
    const float* feat_in;    // [M][128]  used when there is no tail  (first head)
    float* feat_out;         // [M][128]  tail result, written when non-null
    float* featB_out;        // [M][128]  head
    float* qkv_out;          // [M][384]  head
    const float *w1, *b1, *w2, *b2, *w3, *b3;      // fc1 [64][128], fc2 [64][64], fc3 [128][64]
    const float *wp, *bp, *wq, *bq;                // pcn [128][128], qkv [384][128]
    const sp16* wq_split;  // optional: qkv weights as fp16 hi [384][128] | lo [384][128] -> the q|k|v projection runs in
                             // split precision (three fp16 MFMAs per operand pair); its error is of the order the attention's
                             // operand split already has, and q, k, v never touch the residual stream
    sp16* qs;              // head, optional: Q split stream   [bs*N][256]          (split_layout.h)
    unsigned char* kv;       // head, optional: K/V tile stream  [bs][tiles][32 KiB]  (split_layout.h)
    int N, bs;               // rows are bs pairs of N points; a workgroup's 32-point tile never straddles two pairs
    // optional (layer_wave.hip): the weights in MFMA-fragment order, one 8 KiB chunk per WChunk in the order the kernel
    // consumes them (pdsc_wfrag_build_tail / _head); when given they replace w1..w3 / wp, wq, wq_split
    const unsigned char* wf_tail;
    const unsigned char* wf_head;
    int gemm_format;         // format the fragment streams were built in: PDSC_LAYER_GEMM_F32 (fc1..fc3, pcn as fp32 rows) or
                             // PDSC_LAYER_GEMM_H3 (fp16 hi / scaled-lo pairs: those GEMMs run on v_mfma_f32_32x32x16_f16)
    int io_flags;            // enum pdsc_layer_io: which of part_o / res / featB_out are in point-fragment order (layer_h3.hip only)
    int stagger_cycles, stagger_mode;   // layer_h3.hip: start delay of half the first round's wavefronts (A/B knobs PDSC_LAYER_STAGGER, _MODE)
    long long* trace;        // diagnostics (pdsc_layer_trace): [workgroup][wave][16] shader-clock stamps, else NULL
    unsigned int* range_flag; // fp16 range sentinel (pdsc_common.h): [bs] words, or NULL outside a forward
    const int* nvalid;       // ragged batches (ragged.h): [bs] correspondences per pair (<= N): tiles past a pair's own rows are
                             // skipped, its last tile is padded / zeroed from ITS count; NULL = every pair has N rows
};

int launch_layer_wave(const LayerArgs& a, bool tail, bool head, hipStream_t st);      // layer_wave.hip
int launch_layer_h3(const LayerArgs& a, bool tail, bool head, hipStream_t st);        // layer_h3.hip (H3 fragment streams only)
int launch_layer_h3_coop(const LayerArgs& a, bool tail, bool head, hipStream_t st);   // layer_coop.hip (same contract, few tiles)
bool launch_layer_h3_fits(const LayerArgs& a, bool tail, bool head);                  // ... and only this output set

}  // namespace pdsc
