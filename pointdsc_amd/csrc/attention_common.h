// Shared between the two attention kernels (attention.hip: exact fp32 MFMA; attention_split.hip: bf16x3 split).
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
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// merge of the per-split partials (defined in attention.hip)
int launch_attention_combine(const AttArgs& a, int bs, hipStream_t st);

}  // namespace pdsc
