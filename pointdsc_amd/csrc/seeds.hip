// a-4 (tail), a-5, a-6: per-correspondence head, NMS seed selection, feature-space kNN of the seeds.
//   reference: F.normalize + classification.4 (models/PointDSC.py:156,112,171)
//              pick_seeds                     (models/PointDSC.py:199-217)
//              knn + seed gather              (models/common.py:48-69, models/PointDSC.py:250-252)
// All three are row-parallel N x N predicate / selection problems with no data reuse worth an LDS tile:
// one wavefront per row, lanes stride the columns (coalesced), wave shuffles/ballots do the reductions.
// Ordering rules (ties): equal NMS keys -> ascending index; equal kNN distances -> ascending index
// (DESIGN.md "tie semantics"; torch.argsort / torch.topk leave both unspecified).
#include <math.h>
#include <stdlib.h>
#include "pdsc_common.h"

namespace pdsc {

int knn_dist_rows(const float* normed, const int* seeds, float* dist, long long ldd, int bs, int N, int S,
                  hipStream_t st);

// ---- normalise + final classifier layer: one wave per row ----------------------------------------
__global__ __launch_bounds__(256) void normalize_conf_kernel(const float* __restrict__ feat, const float* __restrict__ h2,
                                                             const float* __restrict__ w3, const float* __restrict__ b3,
                                                             float* __restrict__ normed, float* __restrict__ conf, int M) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float2 v = *reinterpret_cast<const float2*>(feat + row * PDSC_CHANNELS + lane * 2);
    const float ss = wave_sum(fmaf(v.y, v.y, v.x * v.x));
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float2 o;
    o.x = v.x / nrm;
    o.y = v.y / nrm;
    *reinterpret_cast<float2*>(normed + row * PDSC_CHANNELS + lane * 2) = o;
    float p = lane < 32 ? h2[row * 32 + lane] * w3[lane] : 0.f;
    p = wave_sum(p);
    if (lane == 0) conf[row] = p + b3[0];
}

// ---- NMS keys: key[i] = conf[i] * all_j( conf[i] >= conf[j] || dist(i,j) >= R ) ---------------------
// `radius2` = the smallest fp32 x with sqrt_rn(x) >= radius (computed on the host): since the correctly rounded square
// root is monotone, `sqrt(x) >= radius` and `x >= radius2` are the same predicate bit for bit -- without the ~20
// instructions of an IEEE sqrt per pair.
// Row-per-lane formulation: a thread owns TWO rows (i, i + 256), the columns of the
// workgroup's slice come as LDS broadcasts, so a pair evaluation is 3 packed subtracts, 1 packed multiply, 2 packed fmas
// and two compares -- no cross-lane traffic, no per-element bounds checks.  The column range is split over workgroups
// (blockIdx.y); keys start as a copy of conf and a slice that suppresses row i stores conf[i] * 0 (every writer stores the
// same value, so the plain stores need no ordering).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NMS2_ROWS = 512;           // rows per workgroup (2 per thread)
constexpr int NMS2_MAX_SLICE = 2048;     // columns per workgroup slice (32 KiB of LDS)

__global__ __launch_bounds__(256) void nms_flags_kernel(const float* __restrict__ src, const float* __restrict__ conf,
                                                        float radius2, float* __restrict__ keys, int N, int slice) {
    __shared__ __attribute__((aligned(16))) float4 rec[NMS2_MAX_SLICE];
    const int t = threadIdx.x, b = blockIdx.z;
    const float* s = src + (size_t)b * N * 3;
    const float* c = conf + (size_t)b * N;
    const int j_begin = blockIdx.y * slice, j_end = min(N, j_begin + slice);
    for (int j = j_begin + t; j < j_end; j += 256) rec[j - j_begin] = make_float4(s[j * 3], s[j * 3 + 1], s[j * 3 + 2], c[j]);
    __syncthreads();
    const int i0 = blockIdx.x * NMS2_ROWS + t, i1 = i0 + 256;
    const int r0 = min(i0, N - 1), r1 = min(i1, N - 1);
    const f32x2 mx = {s[r0 * 3], s[r1 * 3]}, my = {s[r0 * 3 + 1], s[r1 * 3 + 1]}, mz = {s[r0 * 3 + 2], s[r1 * 3 + 2]};
    const float c0 = c[r0], c1 = c[r1];
    bool ok0 = true, ok1 = true;
    const int n = j_end - j_begin;
    for (int j0 = 0; j0 < n; j0 += 64) {
        if (!__any(ok0 || ok1)) break;                  // every row of the wave is already suppressed
        const int jn = min(64, n - j0);
        for (int jj = 0; jj < jn; ++jj) {
            const float4 o = rec[j0 + jj];              // same address in every lane: LDS broadcast
            const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
            const f32x2 dx = mx - ox, dy = my - oy, dz = mz - oz;
            const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));   // norm3's radicand
            ok0 = ok0 && ((c0 >= o.w) || (d2[0] >= radius2));
            ok1 = ok1 && ((c1 >= o.w) || (d2[1] >= radius2));
        }
    }
    if (!ok0 && i0 < N) keys[(size_t)b * N + i0] = c0 * 0.0f;       // -0.0 for suppressed negatives, like torch
    if (!ok1 && i1 < N) keys[(size_t)b * N + i1] = c1 * 0.0f;
}

// ---- top-S by descending key, equal keys by ascending index: one workgroup per pair ---------------------
//   1. radix select (4 x 8 bits, most significant first) of the S-th value in descending order on monotone key bits
//      (-0.0 and +0.0 compare equal, as in torch.sort);
//   2. the S survivors -- every key above the threshold value plus the lowest-index keys equal to it -- are collected as
//      composites (descending-key bits << 32 | index);
//   3. each survivor's rank among the S composites (S^2 comparisons instead of N^2) places its index into seeds[].
// Replaces an N^2 rank count (158 us at 32 pairs of N = 5000).
constexpr int SEL_THREADS = 1024;
__device__ __forceinline__ unsigned int desc_key_bits(float f) {
    unsigned int u = __float_as_uint(f == 0.0f ? 0.0f : f);                    // -0.0 -> +0.0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                           // ascending float order == ascending unsigned
    return ~u;                                                                // ... descending
}
__global__ __launch_bounds__(SEL_THREADS) void seed_select_kernel(const float* __restrict__ keys, int* __restrict__ seeds, int N,
                                                                  int num_seeds) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long surv[];     // [num_seeds]
    __shared__ int hist[256];
    __shared__ unsigned int sh_prefix;
    __shared__ int sh_remaining, sh_count, sh_eq_base;
    __shared__ int wave_cnt[SEL_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* k = keys + (size_t)b * N;
    unsigned int prefix = 0, mask = 0;
    int remaining = num_seeds;                        // 1-based rank of the wanted value inside the current candidate set
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (t < 256) hist[t] = 0;
        __syncthreads();
        for (int i = t; i < N; i += SEL_THREADS) {
            const unsigned int u = desc_key_bits(k[i]);
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
        }
        __syncthreads();
        if (wave == 0) {                              // digit d with  sum(hist[< d]) < remaining <= sum(hist[<= d])
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            int incl = (h0 + h1) + (h2 + h3);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const unsigned long long reach = __ballot(incl >= remaining);      // never empty: the total is >= remaining
            const int first = __ffsll((long long)reach) - 1;
            if (lane == first) {
                int before = incl - ((h0 + h1) + (h2 + h3)), d = 4 * lane;
                if (before + h0 < remaining) { before += h0; ++d;
                    if (before + h1 < remaining) { before += h1; ++d;
                        if (before + h2 < remaining) { before += h2; ++d; } } }
                sh_prefix = prefix | ((unsigned int)d << shift);
                sh_remaining = remaining - before;
            }
        }
        __syncthreads();
        prefix = sh_prefix;
        remaining = sh_remaining;
        mask |= 255u << shift;
    }
    // prefix = the S-th value (descending); `remaining` of the keys equal to it are wanted: those with the lowest indices
    if (t == 0) { sh_count = 0; sh_eq_base = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += SEL_THREADS) {
        const int i = i0 + t;
        const unsigned int u = i < N ? desc_key_bits(k[i]) : 0xFFFFFFFFu;
        const bool eq = i < N && u == prefix;
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int eq_before = sh_eq_base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) eq_before += wave_cnt[w];
        const bool take = i < N && (u < prefix || (eq && eq_before < remaining));
        if (take) {
            const int slot = atomicAdd(&sh_count, 1);
            surv[slot] = ((unsigned long long)u << 32) | (unsigned int)i;
        }
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < SEL_THREADS / 64; ++w) tot += wave_cnt[w];
            sh_eq_base += tot;
        }
        __syncthreads();
    }
    // exactly num_seeds survivors; rank each among them (composites are unique)
    for (int e = t; e < num_seeds; e += SEL_THREADS) {
        const unsigned long long mine = surv[e];
        int rank = 0;
        for (int f = 0; f < num_seeds; ++f) rank += surv[f] < mine;          // LDS broadcast reads
        seeds[(size_t)b * num_seeds + rank] = (int)(mine & 0xFFFFFFFFull);
    }
}

// ---- kNN selection: one workgroup per seed row, radix select (8-bit digits, most significant first) of the
//      (k+1)-th smallest composite value (monotone(dist) << IDX_BITS | index): ties in distance resolve by ascending
//      index through the same comparison, as the oracle's stable sort does ------------------------------------------
__device__ __forceinline__ unsigned int float_order_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending float order == ascending unsigned order
}

constexpr int KNN_THREADS = 256;
constexpr int KNN_FAST_CAP = 1024;

__global__ __launch_bounds__(KNN_THREADS) void knn_select_kernel(const float* __restrict__ dist, long long ldd,
                                                                 int* __restrict__ knn_idx, int N, int S, int k,
                                                                 int idx_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds_u[];
    unsigned int* keys = lds_u;                     // [N] monotone distance bits
    __shared__ int hist[256];
    __shared__ int wave_tot[KNN_THREADS / 64];
    __shared__ int sel_digit, sel_remaining;
    __shared__ unsigned long long cand[PDSC_MAX_K + 1];
    __shared__ int cand_n;
    const int s = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* row = dist + ((size_t)b * S + s) * ldd;
    const int want = k + 1;
    const unsigned long long idx_mask = (1ULL << idx_bits) - 1ULL;

    // ---- fast path: a cheap upper bound on the (k+1)-th smallest composite, then an exact ranking of the few survivors.
    // Every thread finds the minimum of its strided share; the (k+1)-th smallest of the 256 thread minima bounds the
    // (k+1)-th smallest of the row from above (those k+1 minima are k+1 distinct elements), so only elements <= bound can
    // belong to the answer -- typically ~2(k+1) of them.  Composites are unique (index in the low bits): no ties anywhere.
    __shared__ unsigned long long tmin[KNN_THREADS];
    __shared__ unsigned long long fcand[KNN_FAST_CAP];
    __shared__ unsigned long long bound;
    __shared__ int fcand_n;
    unsigned long long mymin = ~0ULL;
    // the row is read as float4 with four loads in flight per thread (a scalar strided loop waits for every load in turn:
    // 20 dependent HBM round trips per thread at N = 5000 were most of this kernel's 237 us); rows are 256-byte aligned
    // (ldd is a multiple of 64) and only columns < N are looked at
    for (int j0 = 0; j0 < N; j0 += KNN_THREADS * 16) {
        f32x4 part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + (u * KNN_THREADS + t) * 4;
            part[u] = j < N ? *reinterpret_cast<const f32x4*>(row + j) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + (u * KNN_THREADS + t) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (j + e < N) {
                    const unsigned int kb = float_order_bits(part[u][e]);
                    keys[j + e] = kb;
                    const unsigned long long v = ((unsigned long long)kb << idx_bits) | (unsigned)(j + e);
                    mymin = v < mymin ? v : mymin;
                }
            }
        }
    }
    tmin[t] = mymin;
    if (t == 0) { cand_n = 0; fcand_n = 0; }
    __syncthreads();
    if ((N + 3) / 4 >= want && want <= KNN_THREADS) {        // at least k+1 threads own elements (a thread owns float4 groups)
        int r = 0;
        for (int u = 0; u < KNN_THREADS; ++u) r += tmin[u] < mymin;
        if (r == want - 1) bound = mymin;                     // exactly one thread: minima are distinct
        __syncthreads();
        const unsigned long long ub = bound;
        for (int j = t; j < N; j += KNN_THREADS) {
            const unsigned long long v = ((unsigned long long)keys[j] << idx_bits) | (unsigned)j;
            if (v <= ub) {
                const int slot = atomicAdd(&fcand_n, 1);
                if (slot < KNN_FAST_CAP) fcand[slot] = v;
            }
        }
        __syncthreads();
        const int nc = fcand_n;
        if (nc <= KNN_FAST_CAP) {                             // block-uniform
            for (int c = t; c < nc; c += KNN_THREADS) {
                const unsigned long long mine = fcand[c];
                int rank = 0;
                for (int u = 0; u < nc; ++u) rank += fcand[u] < mine;
                if (rank >= 1 && rank < want) knn_idx[((size_t)b * S + s) * k + (rank - 1)] = (int)(mine & idx_mask);
            }
            return;
        }
    }
    // ---- general path (tiny rows, or more than KNN_FAST_CAP survivors -- e.g. thousands of equal distances) -----------
    int remaining = want;
    unsigned long long prefix = 0;                  // digits fixed so far (the bits above `shift + 8`)
    const int top = (32 + idx_bits + 7) / 8 * 8;    // composite width rounded up to whole digits
    for (int shift = top - 8; shift >= 0; shift -= 8) {
        hist[t] = 0;                                // KNN_THREADS == 256 bins
        __syncthreads();                            // (also orders the key writes / the previous pass's reads)
        for (int j0 = 0; j0 < N; j0 += KNN_THREADS) {
            const int j = j0 + t;
            int digit = -1;                                 // -1: not a candidate any more
            if (j < N) {
                const unsigned long long v = ((unsigned long long)keys[j] << idx_bits) | (unsigned)j;
                if ((v >> (shift + 8)) == prefix) digit = (int)(v >> shift) & 255;
            }
            // distances cluster in a few bins of the leading digits: LDS atomics on one address serialise per lane (and 8
            // workgroups share the CU's LDS).  Two rounds of wave-level aggregation (one atomic per distinct digit) take
            // the clustered part; whatever is left is spread out and goes through plain atomics.
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const unsigned long long active = __ballot(digit >= 0);
                if (active == 0) break;
                const int d0 = __builtin_amdgcn_readlane(digit, __ffsll((long long)active) - 1);
                const unsigned long long same = __ballot(digit == d0);
                if (digit == d0) {
                    if (lane == __ffsll((long long)same) - 1) atomicAdd(&hist[d0], __popcll(same));
                    digit = -1;
                }
            }
            if (digit >= 0) atomicAdd(&hist[digit], 1);
        }
        __syncthreads();
        // inclusive scan of the 256 bins: wave scan + wave totals
        const int h = hist[t];
        int incl = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        for (int w = 0; w < wave; ++w) incl += wave_tot[w];
        if (incl >= remaining && incl - h < remaining) {   // exactly one bin: the digit of the target
            sel_digit = t;
            sel_remaining = remaining - (incl - h);
        }
        __syncthreads();
        prefix = (prefix << 8) | (unsigned long long)sel_digit;
        remaining = sel_remaining;
    }
    // prefix == the (k+1)-th smallest composite; collect everything <= prefix (exactly k+1 values)
    for (int j = t; j < N; j += KNN_THREADS) {
        const unsigned long long v = ((unsigned long long)keys[j] << idx_bits) | (unsigned)j;
        if (v <= prefix) {
            const int slot = atomicAdd(&cand_n, 1);
            if (slot <= PDSC_MAX_K) cand[slot] = v;
        }
    }
    __syncthreads();
    // rank the candidates (ascending) and drop rank 0 (the reference's `[:, :, 1:]`)
    if (t < want) {
        const unsigned long long mine = cand[t];
        int rank = 0;
        for (int u = 0; u < want; ++u) rank += cand[u] < mine;
        if (rank >= 1) knn_idx[((size_t)b * S + s) * k + (rank - 1)] = (int)(mine & ((1ULL << idx_bits) - 1ULL));
    }
}

}  // namespace pdsc

extern "C" int pdsc_normalize_confidence(const float* feat, const float* h2, const float* w3, const float* b3,
                                         float* normed, float* conf, int M, void* stream) {
    PDSC_REQUIRE(feat && h2 && w3 && b3 && normed && conf, "pdsc_normalize_confidence: null pointer");
    PDSC_REQUIRE(M > 0, "pdsc_normalize_confidence: M=%d", M);
    hipLaunchKernelGGL(pdsc::normalize_conf_kernel, dim3(pdsc::ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, feat, h2,
                       w3, b3, normed, conf, M);
    return pdsc::check_launch("pdsc_normalize_confidence");
}

extern "C" int pdsc_nms_keys(const float* src, const float* conf, float radius, float* keys, int bs, int N, void* stream) {
    PDSC_REQUIRE(src && conf && keys, "pdsc_nms_keys: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_nms_keys: bs=%d N=%d", bs, N);
    // smallest x with sqrtf(x) >= radius (host sqrtf is correctly rounded, like the device's)
    float radius2 = 0.f;
    if (radius > 0.f) {
        radius2 = radius * radius;
        while (radius2 > 0.f && sqrtf(nextafterf(radius2, 0.f)) >= radius) radius2 = nextafterf(radius2, 0.f);
        while (sqrtf(radius2) < radius) radius2 = nextafterf(radius2, INFINITY);
    } else if (radius != radius) {
        radius2 = radius;                                                    // NaN radius: every comparison is false
    }
    {
        hipStream_t st = (hipStream_t)stream;
        if (hipMemcpyAsync(keys, conf, (size_t)bs * N * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return pdsc::check_launch("pdsc_nms_keys(copy)");
        const int row_blocks = pdsc::ceil_div(N, pdsc::NMS2_ROWS);
        int splits = pdsc::ceil_div(1024, row_blocks * bs);             // ~1024 workgroups
        const int min_splits = pdsc::ceil_div(N, pdsc::NMS2_MAX_SLICE);
        if (splits < min_splits) splits = min_splits;
        if (splits > N) splits = N;
        const int slice = pdsc::ceil_div(N, splits);
        splits = pdsc::ceil_div(N, slice);
        hipLaunchKernelGGL(pdsc::nms_flags_kernel, dim3(row_blocks, splits, bs), dim3(256), 0, st, src, conf, radius2, keys, N, slice);
        return pdsc::check_launch("pdsc_nms_keys");
    }
}

extern "C" int pdsc_rank_select(const float* keys, int* seeds, int bs, int N, int num_seeds, void* stream) {
    PDSC_REQUIRE(keys && seeds, "pdsc_rank_select: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && num_seeds >= 0 && num_seeds <= N, "pdsc_rank_select: bs=%d N=%d S=%d", bs, N, num_seeds);
    if (num_seeds == 0) return PDSC_OK;
    const size_t lds_bytes = (size_t)num_seeds * sizeof(unsigned long long);
    PDSC_REQUIRE(lds_bytes <= 128 * 1024, "pdsc_rank_select: num_seeds=%d exceeds the single-workgroup LDS list (16384)", num_seeds);
    const int rc = pdsc::ensure_dynamic_lds(reinterpret_cast<const void*>(&pdsc::seed_select_kernel), lds_bytes > 65536 ? 128 * 1024 : 65536,
                                            "pdsc_rank_select(dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(pdsc::seed_select_kernel, dim3(bs), dim3(pdsc::SEL_THREADS), lds_bytes, (hipStream_t)stream, keys, seeds, N,
                       num_seeds);
    return pdsc::check_launch("pdsc_rank_select");
}

extern "C" int pdsc_knn_seeds(const float* normed, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N,
                              int S, int k, void* stream) {
    PDSC_REQUIRE(normed && seeds && dist_scratch && knn_idx, "pdsc_knn_seeds: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 1 && S > 0, "pdsc_knn_seeds: bs=%d N=%d S=%d", bs, N, S);
    PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K && k <= N - 1, "pdsc_knn_seeds: k=%d (N=%d, max %d)", k, N, PDSC_MAX_K);
    PDSC_REQUIRE((size_t)N * 4 <= 144 * 1024, "pdsc_knn_seeds: N=%d exceeds the single-workgroup LDS row (36864)", N);
    hipStream_t st = (hipStream_t)stream;
    const long long ldd = pdsc_compat_ld(N);
    int rc = pdsc::knn_dist_rows(normed, seeds, dist_scratch, ldd, bs, N, S, st);
    if (rc != PDSC_OK) return rc;
    int idx_bits = 1;
    while ((1 << idx_bits) < N) ++idx_bits;
    const size_t lds_bytes = (size_t)N * sizeof(unsigned int);
    {   // + ~12 KiB static <= 160 KiB
        const int rc_lds = pdsc::ensure_dynamic_lds(reinterpret_cast<const void*>(&pdsc::knn_select_kernel), 144 * 1024, "pdsc_knn_seeds(dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
    }
    hipLaunchKernelGGL(pdsc::knn_select_kernel, dim3(S, bs), dim3(pdsc::KNN_THREADS), lds_bytes, st, dist_scratch, ldd,
                       knn_idx, N, S, k, idx_bits);
    return pdsc::check_launch("pdsc_knn_seeds");
}
