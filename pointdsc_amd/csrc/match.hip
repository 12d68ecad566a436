// f-2: correspondence construction, the step in front of the hot path.
//   reference: datasets/ThreeDMatch.py:283-290 (+305-308), demo_registration.py:101-108, datasets/KITTI.py (same lines),
//              torch variant evaluation/test_3DLoMatch.py:45-48
//     distance   = sqrt(2 - 2 * (src_desc @ tgt_desc.T) + 1e-6)          [Ns, Nt] never materialised here
//     source_idx = argmin(distance, axis=1)                              first index among equal distances (numpy)
//     mutual     : target_idx = argmin(distance, axis=0); keep i iff target_idx[source_idx[i]] == i
//     corr_pos   = concat(src_keypts[corr[:,0]], tgt_keypts[corr[:,1]]) - column mean         (in_dim = 6)
// One fused kernel does the Ns x Nt x D GEMM on the exact fp32 MFMA and the row arg-min: orientation D^T = T . S^T
// (A = 32 target descriptors from LDS, B = 32 source descriptors held in registers) leaves the accumulator lane = one
// source point and its 16 registers = 16 targets, so the arg-min is a lane-local scan in ascending target order
// (strict < keeps the first minimum); the target range is split over workgroups to fill the chip and the partial
// results meet in one 64-bit atomicMin on (distance bits << 32 | target index) -- equal distances resolve to the
// smaller index, like numpy.  NaN distances (2 - 2x + 1e-6 < 0) order first, as np.argmin orders them.
#include "pdsc_common.h"

namespace pdsc {

constexpr int MT_SRC = 128;      // source points per workgroup (32 per wave)
constexpr int MT_TGT = 128;      // target rows staged per barrier
constexpr int MT_MAXD = 64;      // descriptor length limit (FCGF 32, FPFH 33)

__device__ __forceinline__ unsigned long long match_key(float dist, int idx) {
    const unsigned int bits = (dist != dist) ? 0u : __float_as_uint(dist);      // dist >= 0 or NaN; NaN sorts first
    return ((unsigned long long)bits << 32) | (unsigned int)idx;
}

// MODE 1 (evaluation/test_3DLoMatch.py:45-46): source_idx = argmax_j <src_i, tgt_j> -- torch.argmax: first index among equal
// maxima, NaN counts as the maximum.  The key orders LARGER inner products first: bits = ~monotone(dot), NaN -> 0.
__device__ __forceinline__ unsigned int ip_bits(float dot) {
    if (dot != dot) return 0u;
    const unsigned int u = __float_as_uint(dot + 0.0f);                 // -0 -> +0 (equal for argmax)
    const unsigned int mono = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~mono;                                                        // 0 only for mono = 0xFFFFFFFF, a NaN pattern: reserved for NaN above
}
__device__ __forceinline__ float ip_from_bits(unsigned int bits) {
    const unsigned int mono = ~bits;
    return __uint_as_float((mono & 0x80000000u) ? (mono & 0x7fffffffu) : ~mono);
}

// keys[i] = min over the workgroup's target range of match_key(distance(i, j), j)
template <int MODE>
__global__ __launch_bounds__(256) void match_nn_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int Ns, int Nt,
                                                       int D, int tgt_per_split, unsigned long long* __restrict__ keys) {
    constexpr int KP = MT_MAXD;                      // descriptor columns held (zero padded)
    constexpr int LD = KP + 4;                       // LDS row stride in floats (16-B aligned, rotates bank slots)
    __shared__ __attribute__((aligned(16))) float Ts[MT_TGT * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int s0 = blockIdx.x * MT_SRC + wave * 32;
    const int j_begin = blockIdx.y * tgt_per_split, j_end = min(Nt, j_begin + tgt_per_split);
    const int nq = (D + 7) / 8;                      // 8-column groups in use

    // this lane's source descriptor fragment: k-slot (4q+e, half h) <-> column 8q+4h+e
    f32x4 sf[KP / 8];
    {
        const int srow = min(s0 + l31, Ns - 1);
#pragma unroll
        for (int q = 0; q < KP / 8; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 8 * q + 4 * h + e;
                sf[q][e] = c < D ? src[(size_t)srow * D + c] : 0.f;
            }
        }
    }
    float best = MODE == 0 ? INFINITY : -INFINITY;
    int best_j = 0x7fffffff;
    bool best_nan = false;

    // staging: descriptor lengths that are multiples of 4 (FCGF: 32) move as 16-byte pieces, 4 per thread and tile (r05: the
    // scalar loop below -- 32 dependent 4-byte loads per thread and tile, half of them zero fill -- was the other half of this
    // kernel's time); the columns past D that the last 8-column group reads are zeroed once
    const bool vec4 = (D & 3) == 0 && (reinterpret_cast<size_t>(tgt) & 15) == 0;
    if (vec4) {
        for (int idx = t; idx < MT_TGT * LD; idx += 256) Ts[idx] = 0.f;
    }
    for (int j0 = j_begin; j0 < j_end; j0 += MT_TGT) {
        __syncthreads();                             // previous tile consumed (first pass: the zero fill done)
        if (vec4) {
            const int c4n = D >> 2;
            for (int idx = t; idx < MT_TGT * c4n; idx += 256) {
                const int r = idx / c4n, c4 = idx - r * c4n;
                const int j = j0 + r;
                const f32x4 v = j < j_end ? *reinterpret_cast<const f32x4*>(tgt + (size_t)j * D + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(Ts + r * LD + 4 * c4) = v;
            }
        } else {
            for (int idx = t; idx < MT_TGT * KP; idx += 256) {
                const int r = idx / KP, c = idx - r * KP;
                const int j = j0 + r;
                Ts[r * LD + c] = (j < j_end && c < D) ? tgt[(size_t)j * D + c] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < MT_TGT / 32; ++sub) {
            if (j0 + 32 * sub >= j_end) break;       // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* trow = Ts + (32 * sub + l31) * LD + 4 * h;
            for (int q = 0; q < nq; ++q) {
                const f32x4 tf = *reinterpret_cast<const f32x4*>(trow + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tf[e], sf[q][e], acc, 0, 0, 0);
            }
            // lane = source point l31; register r = target j0 + 32 sub + (r&3) + 8 (r>>2) + 4 h.  Ascending target order
            // within a lane is r = 0..15; the other half of the targets lives in lane + 32 (merged at the end).
            // r05: the launch is bound by this vector work, not by its 16 MFMAs per block.  MODE 0: the correctly rounded square root
            // without the library call's special-case scaffolding (sqrt_rn, pdsc_common.h: v_sqrt_f32 + the one-ulp residual test hipcc
            // itself emits; radicands here are 1e-6 .. 4, or negative / NaN -> NaN like sqrtf), and a branch-free update: a NaN
            // distance becomes -inf, which nothing beats afterwards (np.argmin: the first NaN wins) and is mapped back at the end.
            // (Measured and dropped, profiles/r05_c_match_bench_radicand_branch.txt: comparing radicands and taking the square root
            //  only on a new minimum -- with ~250 targets per workgroup some lane of the wave has a new minimum in nearly every step.)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (MODE == 0) {
                    float d = sqrt_rn((2.0f - 2.0f * acc[r]) + 1e-6f);                                  // reference arithmetic, fp32
                    d = d != d ? -INFINITY : d;
                    // strict: the first index among equal distances; the FIRST candidate of a lane is always taken (a row whose
                    // distances are all +inf must still return np.argmin's answer, index 0 -- with `d < best` alone no lane ever wrote a key)
                    const bool take = j < j_end && (d < best || best_j == 0x7fffffff);
                    best = take ? d : best;
                    best_j = take ? j : best_j;
                } else {
                    const float d = acc[r];
                    const bool dn = d != d;
                    const bool take = j < j_end && !best_nan && (dn || d > best || (d == best && j < best_j));
                    if (take) { best = d; best_j = j; best_nan = dn; }
                }
            }
        }
    }
    // merge the two halves of the targets, then the splits
    if (MODE == 0) best_nan = best == -INFINITY;
    unsigned long long key = MODE == 0 ? match_key(best_nan ? NAN : best, best_j)
                                       : (((unsigned long long)ip_bits(best_nan ? NAN : best) << 32) | (unsigned int)best_j);
    const unsigned long long other = __shfl_xor(key, 32, 64);
    key = other < key ? other : key;
    if (h == 0 && s0 + l31 < Ns && best_j != 0x7fffffff) atomicMin(keys + s0 + l31, key);
}

template <int MODE>
__global__ __launch_bounds__(256) void match_decode_kernel(const unsigned long long* __restrict__ keys, int* __restrict__ idx,
                                                           float* __restrict__ dist, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    idx[i] = (int)(unsigned int)(k & 0xffffffffu);
    if (dist) {
        const unsigned int bits = (unsigned int)(k >> 32);
        if (MODE == 0) dist[i] = bits == 0u ? NAN : __uint_as_float(bits);      // bits 0 <=> NaN distance (a true 0 cannot occur: + 1e-6)
        else dist[i] = bits == 0u ? NAN : ip_from_bits(bits);
    }
}

// corr[c] = (i, src2tgt[i]) for the kept i in ascending order (np.where order); *count = number kept.
// mutual == 0 keeps every i.  One workgroup: block-wide exclusive scan in chunks of 1024.
__global__ __launch_bounds__(1024) void corr_select_kernel(const int* __restrict__ src2tgt, const int* __restrict__ tgt2src, int Ns,
                                                           int mutual, int* __restrict__ corr, int* __restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < Ns; i0 += 1024) {
        const int i = i0 + t;
        int keep = 0, j = 0;
        if (i < Ns) {
            j = src2tgt[i];
            keep = mutual ? (tgt2src[j] == i) : 1;
        }
        int incl = keep;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int offs = base;
        for (int w = 0; w < wave; ++w) offs += wave_tot[w];
        if (keep) {
            const int c = offs + incl - 1;
            corr[2 * c] = i;
            corr[2 * c + 1] = j;
        }
        __syncthreads();
        if (t == 1023) base = offs + incl;
        __syncthreads();
    }
    if (t == 0) *count = base;
}

// gather + centre: src_sel[c] = src_kp[corr[c][0]], tgt_sel[c] = tgt_kp[corr[c][1]], corr_pos = concat - column mean.
// One workgroup (the set is at most a few 10^4 rows of 6 floats); column sums in fp64 so that the mean is the
// correctly rounded one whatever the summation order.
__global__ __launch_bounds__(1024) void corr_pos_kernel(const float* __restrict__ src_kp, const float* __restrict__ tgt_kp,
                                                        const int* __restrict__ corr, const int* __restrict__ count,
                                                        float* __restrict__ corr_pos, float* __restrict__ src_sel,
                                                        float* __restrict__ tgt_sel) {
    __shared__ double red[16][6];
    __shared__ float mean[6];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = *count;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int c = t; c < n; c += 1024) {
        const int i = corr[2 * c], j = corr[2 * c + 1];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float a = src_kp[i * 3 + d], b = tgt_kp[j * 3 + d];
            src_sel[c * 3 + d] = a;
            tgt_sel[c * 3 + d] = b;
            s[d] += (double)a;
            s[3 + d] += (double)b;
        }
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) s[d] = wave_sum(s[d]);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) red[wave][d] = s[d];
    }
    __syncthreads();
    if (t < 6) {
        double tot = 0;
        for (int w = 0; w < 16; ++w) tot += red[w][t];
        mean[t] = n > 0 ? (float)(tot / (double)n) : 0.f;
    }
    __syncthreads();
    for (int c = t; c < n; c += 1024) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            corr_pos[c * 6 + d] = src_sel[c * 3 + d] - mean[d];
            corr_pos[c * 6 + 3 + d] = tgt_sel[c * 3 + d] - mean[3 + d];
        }
    }
}

}  // namespace pdsc

using namespace pdsc;

extern "C" size_t pdsc_match_scratch_bytes(int Ns, int Nt) {
    if (Ns <= 0 || Nt <= 0) return 0;
    return (size_t)(Ns > Nt ? Ns : Nt) * sizeof(unsigned long long);
}

static int match_launch(int mode, const float* src_desc, const float* tgt_desc, int Ns, int Nt, int D, int* nn_idx,
                        float* nn_dist, void* scratch, size_t scratch_bytes, void* stream) {
    PDSC_REQUIRE(src_desc && tgt_desc && nn_idx && scratch, "pdsc_match_descriptors: null pointer");
    PDSC_REQUIRE(Ns > 0 && Nt > 0 && D >= 1 && D <= MT_MAXD, "pdsc_match_descriptors: Ns=%d Nt=%d D=%d (D <= %d)", Ns, Nt, D, MT_MAXD);
    if (scratch_bytes < (size_t)Ns * sizeof(unsigned long long)) {
        set_error("pdsc_match_descriptors: scratch %zu < %zu bytes", scratch_bytes, (size_t)Ns * sizeof(unsigned long long));
        return PDSC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* keys = (unsigned long long*)scratch;
    if (const int rc = launch_fill_u32((unsigned int*)keys, 0xFFFFFFFFu, (size_t)Ns * 2, st); rc != PDSC_OK) return rc;
    // split the targets so that ~1024 workgroups exist, in whole staged tiles
    const int src_blocks = ceil_div(Ns, MT_SRC);
    int splits = ceil_div(1024, src_blocks);
    const int max_splits = ceil_div(Nt, MT_TGT);
    if (splits > max_splits) splits = max_splits;
    const int per = ceil_div(ceil_div(Nt, splits), MT_TGT) * MT_TGT;
    splits = ceil_div(Nt, per);
    if (mode == 0) hipLaunchKernelGGL(match_nn_kernel<0>, dim3(src_blocks, splits), dim3(256), 0, st, src_desc, tgt_desc, Ns, Nt, D, per, keys);
    else hipLaunchKernelGGL(match_nn_kernel<1>, dim3(src_blocks, splits), dim3(256), 0, st, src_desc, tgt_desc, Ns, Nt, D, per, keys);
    int rc = check_launch("pdsc_match_descriptors");
    if (rc != PDSC_OK) return rc;
    if (mode == 0) hipLaunchKernelGGL(match_decode_kernel<0>, dim3(ceil_div(Ns, 256)), dim3(256), 0, st, keys, nn_idx, nn_dist, Ns);
    else hipLaunchKernelGGL(match_decode_kernel<1>, dim3(ceil_div(Ns, 256)), dim3(256), 0, st, keys, nn_idx, nn_dist, Ns);
    return check_launch("pdsc_match_descriptors(decode)");
}

extern "C" int pdsc_match_descriptors(const float* src_desc, const float* tgt_desc, int Ns, int Nt, int D, int* nn_idx,
                                      float* nn_dist, void* scratch, size_t scratch_bytes, void* stream) {
    return match_launch(0, src_desc, tgt_desc, Ns, Nt, D, nn_idx, nn_dist, scratch, scratch_bytes, stream);
}

// evaluation/test_3DLoMatch.py:45-46: dists = einsum('ac,bc->ab', src_feats, tgt_feats); source_idx = argmax(dists, -1)
extern "C" int pdsc_match_descriptors_ip(const float* src_desc, const float* tgt_desc, int Ns, int Nt, int D, int* nn_idx,
                                         float* nn_dot, void* scratch, size_t scratch_bytes, void* stream) {
    return match_launch(1, src_desc, tgt_desc, Ns, Nt, D, nn_idx, nn_dot, scratch, scratch_bytes, stream);
}

extern "C" int pdsc_select_correspondences(const int* src2tgt, const int* tgt2src, int Ns, int* corr, int* count, void* stream) {
    PDSC_REQUIRE(src2tgt && corr && count && Ns > 0, "pdsc_select_correspondences: bad argument");
    hipLaunchKernelGGL(corr_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, src2tgt, tgt2src, Ns, tgt2src ? 1 : 0, corr,
                       count);
    return check_launch("pdsc_select_correspondences");
}

extern "C" int pdsc_build_corr_pos(const float* src_keypts, const float* tgt_keypts, const int* corr, const int* count,
                                   float* corr_pos, float* src_sel, float* tgt_sel, void* stream) {
    PDSC_REQUIRE(src_keypts && tgt_keypts && corr && count && corr_pos && src_sel && tgt_sel, "pdsc_build_corr_pos: null pointer");
    hipLaunchKernelGGL(corr_pos_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, src_keypts, tgt_keypts, corr, count, corr_pos,
                       src_sel, tgt_sel);
    return check_launch("pdsc_build_corr_pos");
}
