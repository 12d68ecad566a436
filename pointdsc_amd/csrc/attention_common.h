// Shared between the two attention kernels (attention.hip: exact fp32 MFMA; attention_split.hip: fp16x3 split).
#pragma once
#include "pdsc_common.h"

namespace pdsc {

struct AttArgs {
    const float* qkv;        // [bs*N][384]
    const float* compat;     // [bs][N][ld]
    long long ld;
    float* msg;              // [bs*N][128]
    float* part_o;           // [bs][nsplit][Npad][128]   un-normalised partial outputs
    float* part_ml;          // [bs][nsplit][Npad][2]     (running max (log2 domain), partial sum)
    int N, Npad, nsplit, num_tiles;
    const int* nvalid;       // ragged batches (r06): [bs] correspondences per pair (<= N; strides stay those of N), or NULL
};

// arguments of the split-precision kernels (attention_split.hip)
struct AttSplitArgs {
    const sp16* qs;             // [bs*N][256]  (hi | lo), q pre-scaled by log2(e)/sqrt(C)
    const unsigned char* kv;     // [bs][num_tiles][32 KiB]
    const void* compat;          // [bs][N][ld] fp32, or (C16) unorm16 in the tile order of pdsc_spatial_compat_u16
    long long ld;
    float* msg;                  // [bs*N][128]
    float* part_o;               // [bs][nsplit][Npad][128]
    float* part_ml;              // [bs][nsplit][Npad][2]
    int N, Npad, nsplit, num_tiles, nq, bs;
    int prio_mode;               // A/B knob PDSC_ATT_PRIO: 1 = s_setprio 1 for the younger half of the waves, 2 = for the older half
    int compute_all_waves;       // A/B knob PDSC_ATT_ALL_WAVES (experiments builds): 1 = waves without a valid query compute anyway (r01-r03)
    int compat_nt;               // A/B knob PDSC_ATT_COMPAT_NT: stream the compat slices with the non-temporal policy
    int items;                   // persistent form: number of (pair, key split, query block) items (= the one-item form's grid)
    int part_frag;               // key-split partials in point-fragment order (split_layout.h: PF), straight from the accumulators
    const int* nvalid;           // ragged batches: [bs] correspondences per pair (<= N), or NULL: every pair has N
    long long* trace;            // diagnostics (pdsc_attention_trace): [workgroup][wave][8] cycle sums, else NULL
    // leaf form (sc_attention_split_kernel<..., MG = true>): part_o / part_ml are [bs][nleaf][Npad][..] (point-fragment order)
    int nleaf;                   // leaves per pair (a multiple of nsplit; <= the tile count of the shortest pair)
};


typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// merge of the per-split partials (defined in attention.hip)
int launch_attention_combine(const AttArgs& a, int bs, hipStream_t st);
// exact-fp32 attention with an optional per-pair count array (ragged batches; pdsc_sc_attention = the same with NULL)
int launch_attention_fp32(const float* qkv, const float* compat, long long ld, float* msg, void* scratch, size_t scratch_bytes, int bs, int N,
                          int nsplit, const int* nvalid, hipStream_t st);

}  // namespace pdsc
