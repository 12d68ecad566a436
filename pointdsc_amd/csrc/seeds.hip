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
#include "ragged.h"

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

// ... the forward's variant (r05): a workgroup owns one 32-row tile of one pair and ALSO leaves the normalised rows in point-fragment
// order (split_layout.h: tile = [q = 0..15][lane = 0..63][4 floats], lane (l31, h) = channels 8q + 4h .. + 3 of row l31) -- the B
// operand of the fused kNN kernel below, which then loads 1 KiB of consecutive memory per instruction instead of 32 pieces of 32
// bytes.  Same per-row arithmetic as normalize_conf_kernel (same bits); rows past N inside the last tile are zero in the image.
__global__ __launch_bounds__(256) void normalize_conf_pf_kernel(const float* __restrict__ feat, const float* __restrict__ h2,
                                                                const float* __restrict__ w3, const float* __restrict__ b3,
                                                                float* __restrict__ normed, float* __restrict__ normed_pf,
                                                                float* __restrict__ conf, int N, int Npf) {
    __shared__ __attribute__((aligned(16))) float img[32 * PDSC_CHANNELS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int q = lane >> 2, hh = (lane >> 1) & 1, e = (lane & 1) * 2;       // this lane's channels 2 lane, 2 lane + 1 = 8q + 4hh + e, + 1
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr, i = tile * 32 + r;
        float2 o = make_float2(0.f, 0.f);
        if (i < N) {                                                         // (wave-uniform)
            const long long row = (long long)b * N + i;
            const float2 v = *reinterpret_cast<const float2*>(feat + row * PDSC_CHANNELS + lane * 2);
            const float ss = wave_sum(fmaf(v.y, v.y, v.x * v.x));
            const float nrm = fmaxf(sqrtf(ss), 1e-12f);
            o.x = v.x / nrm;
            o.y = v.y / nrm;
            *reinterpret_cast<float2*>(normed + row * PDSC_CHANNELS + lane * 2) = o;
            float p = lane < 32 ? h2[row * 32 + lane] * w3[lane] : 0.f;
            p = wave_sum(p);
            if (lane == 0) conf[row] = p + b3[0];
        }
        *reinterpret_cast<float2*>(img + q * 256 + (r + 32 * hh) * 4 + e) = o;
    }
    __syncthreads();
    float* dst = normed_pf + ((size_t)b * Npf + (size_t)tile * 32) * PDSC_CHANNELS;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        *reinterpret_cast<f32x4*>(dst + (t + 256 * u) * 4) = *reinterpret_cast<const f32x4*>(img + (t + 256 * u) * 4);
}

int launch_normalize_conf_pf(const float* feat, const float* h2, const float* w3, const float* b3, float* normed, float* normed_pf,
                             float* conf, int bs, int N, hipStream_t st) {
    PDSC_REQUIRE(feat && h2 && w3 && b3 && normed && normed_pf && conf, "pdsc_normalize_confidence(pf): null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_normalize_confidence(pf): bs=%d N=%d", bs, N);
    hipLaunchKernelGGL(normalize_conf_pf_kernel, dim3(ceil_div(N, 32), bs), dim3(256), 0, st, feat, h2, w3, b3, normed, normed_pf, conf, N,
                       (int)round_up(N, 32));
    return check_launch("pdsc_normalize_confidence(pf)");
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
    // (scalar fp32 on purpose: packed fp32 with broadcast operand selects is not used anywhere in this library, pointdsc_amd/build.py)
    const float mx0 = s[r0 * 3], my0 = s[r0 * 3 + 1], mz0 = s[r0 * 3 + 2], mx1 = s[r1 * 3], my1 = s[r1 * 3 + 1], mz1 = s[r1 * 3 + 2];
    const float c0 = c[r0], c1 = c[r1];
    bool ok0 = true, ok1 = true;
    const int n = j_end - j_begin;
    for (int j0 = 0; j0 < n; j0 += 64) {
        if (!__any(ok0 || ok1)) break;                  // every row of the wave is already suppressed
        const int jn = min(64, n - j0);
        for (int jj = 0; jj < jn; ++jj) {
            const float4 o = rec[j0 + jj];              // same address in every lane: LDS broadcast
            const float dx0 = mx0 - o.x, dy0 = my0 - o.y, dz0 = mz0 - o.z, dx1 = mx1 - o.x, dy1 = my1 - o.y, dz1 = mz1 - o.z;
            const float d20 = fmaf(dz0, dz0, fmaf(dy0, dy0, dx0 * dx0)), d21 = fmaf(dz1, dz1, fmaf(dy1, dy1, dx1 * dx1));   // norm3's radicand
            ok0 = ok0 && ((c0 >= o.w) || (d20 >= radius2));
            ok1 = ok1 && ((c1 >= o.w) || (d21 >= radius2));
        }
    }
    if (!ok0 && i0 < N) keys[(size_t)b * N + i0] = c0 * 0.0f;       // -0.0 for suppressed negatives, like torch
    if (!ok1 && i1 < N) keys[(size_t)b * N + i1] = c1 * 0.0f;
}

// ---- NMS keys through a 2-D cell grid: the same predicate on ~1 % of the pairs --------------------------------------
// A point j can only suppress i when ||s_i - s_j|| < R, hence |dx| < R and |dy| < R.  nms_grid_kernel (one workgroup per
// pair) counting-sorts the points into <= 64 x 64 cells of width >= R (1 + 1e-3) in x and y; nms_window_kernel evaluates the
// UNCHANGED predicate -- same fp32 expression for the squared distance, same radius2 -- against the points of the 3 x 3
// neighbouring cells only.  Every skipped j has |dx| or |dy| >= R (1 + 1e-3), so its computed squared distance is >= radius2
// (margin 1e-3 against rounding of order 1e-7) and the predicate is true: the keys are bit-identical to nms_flags_kernel's.
// The cell of a point is floor((x - xmin) / w) in fp32 with w >= R (1 + 1e-3): for |x_i - x_j| < R the two quotients differ
// by < 0.999 + 64 * 2^-22 < 1, so their floors differ by at most 1 whatever the rounding.
constexpr int GRID_MAX = 64;
struct NmsGridHeader { float xmin, ymin, inv_unused0, inv_unused1, wx, wy; int nbx, nby; };
static_assert(sizeof(NmsGridHeader) == 32, "header layout");
__device__ __forceinline__ int grid_cell_1d(float v, float vmin, float w, int nb) {
    const float q = floorf((v - vmin) / w);
    return q >= 0.f ? (q < (float)nb ? (int)q : nb - 1) : 0;      // NaN -> cell 0 (only converted when finite and in range)
}
static size_t nms_ws_pair_bytes(int N) {
    return (size_t)round_up((long long)N * 16 + (long long)N * 4 + (GRID_MAX * GRID_MAX + 1) * 4 + sizeof(NmsGridHeader), 256);
}
// Ragged batches (nvalid != NULL): NS = rows per pair in src / conf / keys and in the workspace layout (the longest pair), the
// pair's own count N = nvalid[b] bounds every loop.
__global__ __launch_bounds__(1024) void nms_grid_kernel(const float* __restrict__ src, const float* __restrict__ conf, float radius,
                                                        unsigned char* __restrict__ ws, size_t ws_pair, int NS,
                                                        const int* __restrict__ nvalid) {
    const int N = nvalid ? nvalid[blockIdx.x] : NS;
    __shared__ int cells[GRID_MAX * GRID_MAX + 1];
    __shared__ float red[4][16];
    __shared__ int wtot[16];
    __shared__ NmsGridHeader hdr;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* s = src + (size_t)b * NS * 3;
    const float* c = conf + (size_t)b * NS;
    unsigned char* w = ws + (size_t)b * ws_pair;
    float4* rec = reinterpret_cast<float4*>(w);
    int* oidx = reinterpret_cast<int*>(w + (size_t)NS * 16);
    int* cell_start = reinterpret_cast<int*>(w + (size_t)NS * 20);
    NmsGridHeader* hout = reinterpret_cast<NmsGridHeader*>(w + (size_t)NS * 20 + (GRID_MAX * GRID_MAX + 1) * 4);
    float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY;
    int bad = 0;          // a NaN / Inf coordinate: its distances are NaN / Inf for EVERY partner, the window argument does not hold
    for (int i = t; i < N; i += 1024) {
        const float x = s[i * 3], y = s[i * 3 + 1], z = s[i * 3 + 2];
        xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
        bad |= !(fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        xmn = fminf(xmn, __shfl_xor(xmn, off, 64)); xmx = fmaxf(xmx, __shfl_xor(xmx, off, 64));
        ymn = fminf(ymn, __shfl_xor(ymn, off, 64)); ymx = fmaxf(ymx, __shfl_xor(ymx, off, 64));
    }
    if (lane == 0) { red[0][wave] = xmn; red[1][wave] = xmx; red[2][wave] = ymn; red[3][wave] = ymx; }
    for (int i = t; i <= GRID_MAX * GRID_MAX; i += 1024) cells[i] = 0;
    const int any_bad = __syncthreads_or(bad);
    if (t == 0) {
        for (int k = 1; k < 16; ++k) {
            red[0][0] = fminf(red[0][0], red[0][k]); red[1][0] = fmaxf(red[1][0], red[1][k]);
            red[2][0] = fminf(red[2][0], red[2][k]); red[3][0] = fmaxf(red[3][0], red[3][k]);
        }
        const float rm = radius * 1.001f;
        const float rx = red[1][0] - red[0][0], ry = red[3][0] - red[2][0];
        int nbx = (int)fminf(floorf(rx / rm), (float)GRID_MAX), nby = (int)fminf(floorf(ry / rm), (float)GRID_MAX);
        if (!(nbx >= 1) || any_bad) nbx = 1;                 // non-finite input: one cell = every pair evaluated, as the N^2 kernel does
        if (!(nby >= 1) || any_bad) nby = 1;
        hdr.xmin = red[0][0]; hdr.ymin = red[2][0]; hdr.nbx = nbx; hdr.nby = nby;
        hdr.wx = nbx > 1 ? rx / (float)nbx : 1.0f;            // >= rm by construction (nbx <= rx / rm); one cell: any width
        hdr.wy = nby > 1 ? ry / (float)nby : 1.0f;
        hdr.inv_unused0 = 0.f; hdr.inv_unused1 = 0.f;
    }
    __syncthreads();
    const NmsGridHeader h = hdr;
    const int ncell = h.nbx * h.nby;
    // histogram
    for (int i = t; i < N; i += 1024) {
        const int cell = grid_cell_1d(s[i * 3 + 1], h.ymin, h.wy, h.nby) * h.nbx + grid_cell_1d(s[i * 3], h.xmin, h.wx, h.nbx);
        atomicAdd(&cells[cell], 1);
    }
    __syncthreads();
    // exclusive scan of cells[0..ncell) in place (4 consecutive cells per thread), total into cells[ncell]
    {
        int v[4], sum = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (4 * t + e) < ncell ? cells[4 * t + e] : 0; sum += v[e]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int k = 0; k < wave; ++k) base += wtot[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (4 * t + e < ncell) cells[4 * t + e] = base;
            base += v[e];
        }
        if (t == 1023) cells[ncell] = base;                   // == N
    }
    __syncthreads();
    for (int i = t; i <= ncell; i += 1024) cell_start[i] = cells[i];
    if (t == 0) *hout = h;
    __syncthreads();
    // scatter (cells[] now serves as the per-cell cursor)
    for (int i = t; i < N; i += 1024) {
        const float x = s[i * 3], y = s[i * 3 + 1], z = s[i * 3 + 2];
        const int cell = grid_cell_1d(y, h.ymin, h.wy, h.nby) * h.nbx + grid_cell_1d(x, h.xmin, h.wx, h.nbx);
        const int pos = atomicAdd(&cells[cell], 1);
        rec[pos] = make_float4(x, y, z, c[i]);
        oidx[pos] = i;
    }
}

__global__ __launch_bounds__(256) void nms_window_kernel(const unsigned char* __restrict__ ws, size_t ws_pair, float radius2,
                                                         float* __restrict__ keys, int NS, const int* __restrict__ nvalid) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    const int N = nvalid ? nvalid[b] : NS;
    const unsigned char* w = ws + (size_t)b * ws_pair;
    const float4* rec = reinterpret_cast<const float4*>(w);
    const int* oidx = reinterpret_cast<const int*>(w + (size_t)NS * 16);
    const int* cell_start = reinterpret_cast<const int*>(w + (size_t)NS * 20);
    const NmsGridHeader h = *reinterpret_cast<const NmsGridHeader*>(w + (size_t)NS * 20 + (GRID_MAX * GRID_MAX + 1) * 4);
    if (p >= N) return;
    const float4 me = rec[p];
    const int cx = grid_cell_1d(me.x, h.xmin, h.wx, h.nbx), cy = grid_cell_1d(me.y, h.ymin, h.wy, h.nby);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.nbx - 1);
    bool ok = true;
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, h.nby - 1); ++yy) {
        const int j0 = cell_start[yy * h.nbx + x0], j1 = cell_start[yy * h.nbx + x1 + 1];
        for (int j = j0; j < j1; ++j) {
            const float4 o = rec[j];
            const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));                  // norm3's radicand, as nms_flags_kernel
            ok = ok && ((me.w >= o.w) || (d2 >= radius2));
        }
    }
    keys[(size_t)b * NS + oidx[p]] = ok ? me.w : me.w * 0.0f;                       // -0.0 for suppressed negatives, like torch
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
// Ragged batches: NS / SS = rows per pair of keys / slots per pair of seeds (those of the longest pair); this pair ranks its own
// N = nvalid[b] keys and picks num_seeds = svalid[b] of them.  The unused slots [num_seeds, SS) are filled with copies of the
// pair's first seed: every later stage then works on valid indices, a duplicate hypothesis scores exactly like the original
// and, sitting behind it, can never win the first-maximum argmax (models/PointDSC.py:331) -- nor change the AND of the
// per-seed convergence flags.
__global__ __launch_bounds__(SEL_THREADS) void seed_select_kernel(const float* __restrict__ keys, int* __restrict__ seeds, int NS,
                                                                  int SS, const int* __restrict__ nvalid, const int* __restrict__ svalid,
                                                                  unsigned int* __restrict__ conv_mask_init) {
    // (r06, one launch less on the forward's chain) the per-pair convergence mask of the seed solver that follows starts at all-ones:
    // initialised here -- this kernel is one workgroup per pair and runs before it on the same stream -- instead of by a fill launch
    if (conv_mask_init && threadIdx.x == 0) conv_mask_init[blockIdx.x] = 0xFFFFFFFFu;
    const int N = nvalid ? nvalid[blockIdx.x] : NS, num_seeds = svalid ? svalid[blockIdx.x] : SS;
    extern __shared__ __attribute__((aligned(16))) unsigned long long surv[];     // [num_seeds]
    __shared__ int hist[256];
    __shared__ unsigned int sh_prefix;
    __shared__ int sh_remaining, sh_count, sh_eq_base;
    __shared__ int wave_cnt[SEL_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float* k = keys + (size_t)b * NS;
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
        seeds[(size_t)b * SS + rank] = (int)(mine & 0xFFFFFFFFull);
        if (rank == 0)
            for (int f = num_seeds; f < SS; ++f) seeds[(size_t)b * SS + f] = (int)(mine & 0xFFFFFFFFull);     // (ragged batches only)
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
                                                                 int* __restrict__ knn_idx, int NS, int S, int k,
                                                                 int idx_bits, const int* __restrict__ nvalid) {
    const int N = nvalid ? nvalid[blockIdx.y] : NS;      // ragged batches: only this pair's own columns are candidates
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
    __shared__ unsigned long long gmin[64];
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
    // The bound comes from 64 GROUP minima (group l = threads l, l+64, l+128, l+192), ranked by one wave: 64 comparisons for
    // 64 lanes instead of 256 for 256 threads (that O(256^2) ranking was most of this kernel: ~1300 instructions per thread);
    // the (k+1)-th smallest of 64 distinct elements still bounds the (k+1)-th smallest of the row from above, it just lets
    // ~65 instead of ~43 survivors through to the exact ranking below.
    if (min(64, (N + 3) / 4) >= want) {                       // at least k+1 groups own elements (a thread owns float4 groups)
        if (wave == 0) {
            unsigned long long gm = tmin[lane];
#pragma unroll
            for (int q = 1; q < KNN_THREADS / 64; ++q) { const unsigned long long o = tmin[64 * q + lane]; gm = o < gm ? o : gm; }
            gmin[lane] = gm;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int r = 0;
#pragma unroll 8
            for (int u = 0; u < 64; ++u) r += gmin[u] < gm;
            if (r == want - 1) bound = gm;                    // exactly one lane: group minima are distinct elements
        }
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

static float nms_radius2(float radius);
static float nms_radius2_fwd(float radius) { return nms_radius2(radius); }

extern "C" int pdsc_nms_keys(const float* src, const float* conf, float radius, float* keys, int bs, int N, void* stream) {
    PDSC_REQUIRE(src && conf && keys, "pdsc_nms_keys: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_nms_keys: bs=%d N=%d", bs, N);
    const float radius2 = nms_radius2_fwd(radius);
    {
        hipStream_t st = (hipStream_t)stream;
        if (const int rc = pdsc::launch_copy_u32((unsigned int*)keys, (const unsigned int*)conf, (size_t)bs * N, st); rc != PDSC_OK) return rc;
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

static float nms_radius2(float radius) { return pdsc::sqrt_threshold_radicand(radius); }

extern "C" size_t pdsc_nms_workspace_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * pdsc::nms_ws_pair_bytes(N);
}

namespace pdsc {

int launch_nms_keys_grid(const float* src, const float* conf, float radius, float* keys, void* workspace, size_t workspace_bytes,
                         int bs, int N, const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(src && conf && keys, "pdsc_nms_keys_grid: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_nms_keys_grid: bs=%d N=%d", bs, N);
    // a radius that is not a positive finite number has no cell width: the N^2 kernel handles it (every point its own maximum
    // for radius <= 0, none for NaN)
    if (!(radius > 0.f) || !(radius < INFINITY) || !workspace) {
        PDSC_REQUIRE(!nvalid, "pdsc_nms_keys_grid: ragged batches need a positive finite nms_radius and the grid workspace");
        return pdsc_nms_keys(src, conf, radius, keys, bs, N, st);
    }
    if (workspace_bytes < pdsc_nms_workspace_bytes(bs, N)) {
        set_error("pdsc_nms_keys_grid: workspace %zu < %zu bytes", workspace_bytes, pdsc_nms_workspace_bytes(bs, N));
        return PDSC_ERR_WORKSPACE;
    }
    const size_t ws_pair = nms_ws_pair_bytes(N);
    hipLaunchKernelGGL(nms_grid_kernel, dim3(bs), dim3(1024), 0, st, src, conf, radius, (unsigned char*)workspace, ws_pair, N, nvalid);
    int rc = check_launch("pdsc_nms_keys_grid(grid)");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(nms_window_kernel, dim3(ceil_div(N, 256), bs), dim3(256), 0, st, (const unsigned char*)workspace, ws_pair,
                       nms_radius2(radius), keys, N, nvalid);
    return check_launch("pdsc_nms_keys_grid(window)");
}

int launch_rank_select(const float* keys, int* seeds, int bs, int N, int num_seeds, const int* nvalid, const int* svalid, hipStream_t st,
                       unsigned int* conv_mask_init) {
    PDSC_REQUIRE(keys && seeds, "pdsc_rank_select: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && num_seeds >= 0 && num_seeds <= N, "pdsc_rank_select: bs=%d N=%d S=%d", bs, N, num_seeds);
    if (num_seeds == 0) return conv_mask_init ? launch_fill_u32(conv_mask_init, 0xFFFFFFFFu, (size_t)bs, st) : PDSC_OK;
    const size_t lds_bytes = (size_t)num_seeds * sizeof(unsigned long long);
    PDSC_REQUIRE(lds_bytes <= 128 * 1024, "pdsc_rank_select: num_seeds=%d exceeds the single-workgroup LDS list (16384)", num_seeds);
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&seed_select_kernel), lds_bytes > 65536 ? 128 * 1024 : 65536,
                                      "pdsc_rank_select(dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(seed_select_kernel, dim3(bs), dim3(SEL_THREADS), lds_bytes, st, keys, seeds, N, num_seeds, nvalid, svalid, conv_mask_init);
    return check_launch("pdsc_rank_select");
}

// ---- kNN of the seeds WITHOUT the S x N distance matrix (r05): Gram rows on the fp32 matrix cores and the selection in one launch.
// knn_dist_rows + knn_select_kernel write and re-read 4 S N bytes per pair (320 MB at 32 pairs of N = 5000: 1.7 x the Gram's
// algorithmic traffic); here a workgroup owns 32 seeds, its four waves walk the columns in blocks of 32 (wave w: blocks w, w + 4, ..),
// and a distance leaves the registers only if it can still be among its seed's k + 1 smallest:
//   * arithmetic of gram_rows_kernel<1, .> to the bit: A = the seed's feature row (registers), B = 32 columns' rows loaded straight
//     from L2 (the normalised features of a pair are 2.5 MB), 64 x v_mfma_f32_32x32x2_f32 in the same order, dist = 2 - 2 * dot;
//   * candidate = composite (monotone(dist) << idx_bits | column), unique; per seed an LDS list of KF_CAP composites and an upper
//     bound tau on its (k+1)-th smallest composite: a composite enters the list only if < tau (tau starts at "everything");
//   * after every round of 128 columns (one workgroup barrier) a list longer than KF_CAP - 128 is cut back: tau = the (k+1)-th
//     smallest of 64 lane minima over the list (an upper bound of the list's (k+1)-th smallest, hence of the seed's), entries
//     above it are dropped; should more than KF_KEEP survive (many equal distances) the list is cut to exactly the k + 1 smallest;
//   * at the end the list holds every one of the seed's k + 1 smallest composites: exact ranking, ranks 1..k are the neighbours
//     (rank 0 dropped like `[:, :, 1:]`, models/common.py:69) -- the same selection on the same distance bits as knn_select_kernel.
constexpr int KF_ROWS = 32, KF_CAP = 256, KF_KEEP = 96, KF_MAX_WANT = 48;

struct KnnFusedArgs {
    const float* X;          // [bs][NS][128] normalised features (the seeds' rows: A operand)
    const float* Xpf;        // [bs][round_up(NS, 32)][128] the same rows in point-fragment order (B operand), or NULL: gather from X
    const int* seeds;        // [bs][S]
    int* knn_idx;            // [bs][S][k]
    int NS, S, k, idx_bits;
    int bs, nrb;             // pairs, seed blocks of KF_ROWS per pair (grid = bs * nrb workgroups, see the block map in the kernel)
    const int* nvalid;       // ragged batches: [bs] rows per pair, or NULL
};

__global__ __launch_bounds__(256, 2) void knn_fused_kernel(KnnFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long kf_dyn[];
    unsigned long long (*list)[KF_CAP] = reinterpret_cast<unsigned long long (*)[KF_CAP]>(kf_dyn);       // [KF_ROWS][KF_CAP]: 64 KiB (dynamic: past the static limit)
    __shared__ float tau[KF_ROWS];          // upper bound of the seed's (k+1)-th smallest DISTANCE (the distance part of the bound composite)
    __shared__ int cnt[KF_ROWS];
    __shared__ unsigned long long gmin[4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    // block -> (pair, seed block).  Consecutive workgroup ids go round the eight XCDs (each with its own L2), and every workgroup of
    // a pair streams that pair's whole 2.5 MB of normalised rows: with the plain (x, y) grid each XCD fetched every pair (PMC, r05
    // bundle: 815 MB of HBM traffic against 82 MB of operands).  With a batch that is a multiple of 8, XCD x takes the pairs
    // x, x + 8, ...: a pair's seed blocks share one L2 and walk the columns in step.
    int b, rb;
    {
        const int nrb = a.nrb, L = blockIdx.x;
        if ((a.bs & 7) == 0) {
            const int xcd = L & 7, idx = L >> 3;
            b = xcd + 8 * (idx / nrb);
            rb = idx % nrb;
        } else {
            b = L / nrb;
            rb = L % nrb;
        }
    }
    const int r0 = rb * KF_ROWS;
    const int N = a.nvalid ? a.nvalid[b] : a.NS;
    const float* X = a.X + (size_t)b * a.NS * PDSC_CHANNELS;
    const int want = a.k + 1;
    const unsigned long long idx_mask = (1ULL << a.idx_bits) - 1ULL;
    if (t < KF_ROWS) { tau[t] = INFINITY; cnt[t] = 0; }

    // A fragments of this lane's seed row (k-slot (4q+e, half h) <-> channel 8q+4h+e): identical in the four waves
    f32x4 af[16];
    {
        const int row = a.seeds[(size_t)b * a.S + min(r0 + l31, a.S - 1)];
        const float* p = X + (size_t)row * PDSC_CHANNELS + 4 * h;
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
    }
    const int nblocks = (N + 31) >> 5, rounds = (nblocks + 3) >> 2;
    f32x4 bf[16];
    const int Npf = (a.NS + 31) & ~31;
    const float* Xpf = a.Xpf ? a.Xpf + (size_t)b * Npf * PDSC_CHANNELS + lane * 4 : nullptr;
    auto load_b = [&](int cb) {
        if (Xpf) {
            // point-fragment image: instruction q of the wave reads 1 KiB of consecutive memory (columns past N: whatever the image
            // holds there -- those lanes' results are never looked at)
            const float* p = Xpf + (size_t)min(cb, nblocks - 1) * 32 * PDSC_CHANNELS;
#pragma unroll
            for (int q = 0; q < 16; ++q) bf[q] = *reinterpret_cast<const f32x4*>(p + 256 * q);
        } else {
            const int col = min(cb * 32 + l31, N - 1);
            const float* p = X + (size_t)col * PDSC_CHANNELS + 4 * h;
#pragma unroll
            for (int q = 0; q < 16; ++q) bf[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
        }
    };
    load_b(wave);
    __syncthreads();
    for (int rd = 0; rd < rounds; ++rd) {
        const int cb = rd * 4 + wave;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][e], bf[q][e], acc, 0, 0, 0);
        load_b(cb + 4);                                               // next round's columns: in flight under the filter and the barrier
        const int col = cb * 32 + l31;
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = 2.0f - 2.0f * acc[r];                 // == reference `2 - 2*matmul` (gram_rows_kernel MODE 1)
                // one float compare per element (a composite at or below the bound has a distance at or below the bound's: a
                // superset passes, a NaN never does); the 64-bit composite is only built for what enters the list
                if (v <= tau[row] && r0 + row < a.S) {
                    const unsigned long long key = ((unsigned long long)float_order_bits(v) << a.idx_bits) | (unsigned)col;
                    const int slot = atomicAdd(&cnt[row], 1);
                    list[row][slot] = key;                            // slot < KF_CAP: a round adds at most 128 to a list of at most KF_CAP - 128
                }
            }
        }
        __syncthreads();
        // cut back the long lists: wave w looks after rows 8 w .. 8 w + 7
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr;
            const int n = cnt[row];                                    // (wave-uniform)
            if (n <= KF_CAP - 128) continue;
            // lane minima over the list (n > 128: every lane owns at least two entries), the (k+1)-th smallest of them bounds the
            // list's (k+1)-th smallest from above
            unsigned long long e[KF_CAP / 64], mine = ~0ULL;
#pragma unroll
            for (int u = 0; u < KF_CAP / 64; ++u) {
                e[u] = lane + 64 * u < n ? list[row][lane + 64 * u] : ~0ULL;
                mine = e[u] < mine ? e[u] : mine;
            }
            gmin[wave][lane] = mine;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int rk = 0;
#pragma unroll 8
            for (int u = 0; u < 64; ++u) rk += gmin[wave][u] < mine;
            const unsigned long long vote = __ballot(rk == want - 1);       // exactly one lane (minima are distinct composites)
            const int src = __ffsll((long long)vote) - 1;
            unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), src) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)(mine & 0xffffffffULL), src);
            // survivors (<= ub) back to the front of the list, in any order
            int kept = 0;
#pragma unroll
            for (int u = 0; u < KF_CAP / 64; ++u) {
                const bool keep = e[u] <= ub;                              // (padding ~0 is never <= a real composite)
                const unsigned long long bal = __ballot(keep);
                if (keep) list[row][kept + __popcll(bal & ((1ULL << lane) - 1ULL))] = e[u];
                kept += __popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (kept > KF_KEEP) {
                // many entries at or below the bound (equal distances): cut to exactly the k + 1 smallest by exact ranking
                unsigned long long f[KF_CAP / 64], cand = 0ULL;
                int rank[KF_CAP / 64];
#pragma unroll
                for (int u = 0; u < KF_CAP / 64; ++u) { f[u] = lane + 64 * u < kept ? list[row][lane + 64 * u] : ~0ULL; rank[u] = 0; }
                for (int j = 0; j < kept; ++j) {
                    const unsigned long long o = list[row][j];
#pragma unroll
                    for (int u = 0; u < KF_CAP / 64; ++u) rank[u] += o < f[u];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                            // every lane has read the list: now it is rewritten in rank order
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int u = 0; u < KF_CAP / 64; ++u) {
                    const bool real = lane + 64 * u < kept;
                    if (real && rank[u] < want) list[row][rank[u]] = f[u];
                    if (real && rank[u] == want - 1) cand = f[u];           // (composites are never 0: the distance bits have their top bit set)
                }
                const unsigned long long who = __ballot(cand != 0ULL);     // exactly one lane
                const int sl = __ffsll((long long)who) - 1;
                ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cand >> 32), sl) << 32) |
                     (unsigned)__builtin_amdgcn_readlane((int)(cand & 0xffffffffULL), sl);
                kept = want;
            }
            if (lane == 0) {
                cnt[row] = kept;
                const unsigned int ob = (unsigned int)(ub >> a.idx_bits);                          // float_order_bits of the bound's distance, inverted:
                tau[row] = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob);
            }
        }
        __syncthreads();
    }
    // ---- every list now holds its seed's k + 1 smallest composites (and some more): exact ranks, 1 .. k are the neighbours ----
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr;
        if (r0 + row >= a.S) continue;
        const int n = cnt[row];
        unsigned long long f[KF_CAP / 64];
        int rank[KF_CAP / 64];
#pragma unroll
        for (int u = 0; u < KF_CAP / 64; ++u) { f[u] = lane + 64 * u < n ? list[row][lane + 64 * u] : ~0ULL; rank[u] = 0; }
        for (int j = 0; j < n; ++j) {
            const unsigned long long o = list[row][j];
#pragma unroll
            for (int u = 0; u < KF_CAP / 64; ++u) rank[u] += o < f[u];
        }
#pragma unroll
        for (int u = 0; u < KF_CAP / 64; ++u)
            if (lane + 64 * u < n && rank[u] >= 1 && rank[u] < want)
                a.knn_idx[((size_t)b * a.S + r0 + row) * a.k + (rank[u] - 1)] = (int)(f[u] & idx_mask);
    }
}

// form: 0 = the library's choice, 1 = two launches through the S x N matrix (knn_dist_rows + knn_select_kernel), 2 = fused
int launch_knn_seeds(const float* normed, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N, int S, int k,
                     const int* nvalid, hipStream_t st) {
    return launch_knn_seeds_form(normed, nullptr, seeds, dist_scratch, knn_idx, bs, N, S, k, nvalid, 0, st);
}

bool knn_seeds_uses_fused(int bs, int N, int S, int k) {
    return k + 1 <= KF_MAX_WANT && N >= 256 && (long long)bs * ceil_div(S, KF_ROWS) >= 384;
}

// normed_pf (optional): the normalised rows in point-fragment order (normalize_conf_pf_kernel): the fused form's B operand
int launch_knn_seeds_form(const float* normed, const float* normed_pf, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N,
                          int S, int k, const int* nvalid, int form, hipStream_t st) {
    PDSC_REQUIRE(form >= 0 && form <= 2, "pdsc_knn_seeds: form=%d", form);
    // fused: enough workgroups of 32 seeds to fill the chip (else the two-launch form, whose Gram splits the columns too), k + 1
    // within the 64 lane minima's reach, and at least k + 1 columns per pair
    const bool fits = k + 1 <= KF_MAX_WANT && N >= 256;
    PDSC_REQUIRE(form != 2 || fits, "pdsc_knn_seeds: the fused form needs k + 1 <= %d and N >= 256 (k=%d, N=%d)", KF_MAX_WANT, k, N);
    if (form == 2 || (form == 0 && knn_seeds_uses_fused(bs, N, S, k))) {
        PDSC_REQUIRE(normed && seeds && knn_idx, "pdsc_knn_seeds: null pointer");
        PDSC_REQUIRE(bs > 0 && N > 1 && S > 0, "pdsc_knn_seeds: bs=%d N=%d S=%d", bs, N, S);
        PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K && k <= N - 1, "pdsc_knn_seeds: k=%d (N=%d, max %d)", k, N, PDSC_MAX_K);
        KnnFusedArgs a{};
        a.X = normed; a.Xpf = normed_pf; a.seeds = seeds; a.knn_idx = knn_idx; a.NS = N; a.S = S; a.k = k; a.nvalid = nvalid;
        a.idx_bits = 1;
        while ((1 << a.idx_bits) < N) ++a.idx_bits;
        const size_t lds_bytes = (size_t)KF_ROWS * KF_CAP * sizeof(unsigned long long);
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_fused_kernel), lds_bytes, "pdsc_knn_seeds(fused, dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
        a.bs = bs; a.nrb = ceil_div(S, KF_ROWS);
        hipLaunchKernelGGL(knn_fused_kernel, dim3((unsigned)(a.nrb * bs)), dim3(256), lds_bytes, st, a);
        return check_launch("pdsc_knn_seeds(fused)");
    }

    PDSC_REQUIRE(normed && seeds && dist_scratch && knn_idx, "pdsc_knn_seeds: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 1 && S > 0, "pdsc_knn_seeds: bs=%d N=%d S=%d", bs, N, S);
    PDSC_REQUIRE(k >= 1 && k <= PDSC_MAX_K && k <= N - 1, "pdsc_knn_seeds: k=%d (N=%d, max %d)", k, N, PDSC_MAX_K);
    PDSC_REQUIRE((size_t)N * 4 <= 144 * 1024, "pdsc_knn_seeds: N=%d exceeds the single-workgroup LDS row (36864)", N);
    const long long ldd = pdsc_compat_ld(N);
    int rc = knn_dist_rows(normed, seeds, dist_scratch, ldd, bs, N, S, st);
    if (rc != PDSC_OK) return rc;
    int idx_bits = 1;
    while ((1 << idx_bits) < N) ++idx_bits;
    const size_t lds_bytes = (size_t)N * sizeof(unsigned int);
    {   // + ~12 KiB static <= 160 KiB
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&knn_select_kernel), 144 * 1024, "pdsc_knn_seeds(dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
    }
    hipLaunchKernelGGL(knn_select_kernel, dim3(S, bs), dim3(KNN_THREADS), lds_bytes, st, dist_scratch, ldd, knn_idx, N, S, k, idx_bits, nvalid);
    return check_launch("pdsc_knn_seeds");
}

}  // namespace pdsc

extern "C" int pdsc_nms_keys_grid(const float* src, const float* conf, float radius, float* keys, void* workspace,
                                  size_t workspace_bytes, int bs, int N, void* stream) {
    return pdsc::launch_nms_keys_grid(src, conf, radius, keys, workspace, workspace_bytes, bs, N, nullptr, (hipStream_t)stream);
}

extern "C" int pdsc_rank_select(const float* keys, int* seeds, int bs, int N, int num_seeds, void* stream) {
    return pdsc::launch_rank_select(keys, seeds, bs, N, num_seeds, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pdsc_knn_seeds_form(const float* normed, const float* normed_pf, const int* seeds, float* dist_scratch, int* knn_idx,
                                   int bs, int N, int S, int k, int form, void* stream) {
    return pdsc::launch_knn_seeds_form(normed, normed_pf, seeds, dist_scratch, knn_idx, bs, N, S, k, nullptr, form, (hipStream_t)stream);
}

extern "C" int pdsc_normalize_confidence_pf(const float* feat, const float* h2, const float* w3, const float* b3, float* normed,
                                            float* normed_pf, float* conf, int bs, int N, void* stream) {
    return pdsc::launch_normalize_conf_pf(feat, h2, w3, b3, normed, normed_pf, conf, bs, N, (hipStream_t)stream);
}

extern "C" int pdsc_knn_seeds(const float* normed, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N,
                              int S, int k, void* stream) {
    return pdsc::launch_knn_seeds(normed, seeds, dist_scratch, knn_idx, bs, N, S, k, nullptr, (hipStream_t)stream);
}
