// f-3: the spectral-matching baseline, i.e. the N x N power iteration of
//   reference baseline_scripts/baseline_3DMatch.py:19-53 (SM):
//     diff = corr[i] - corr[j];  d = |diff[0:3]| - |diff[3:6]|             (corr = centred corr_pos [N,6])
//     M    = max(0, 4.5 - d^2 / 2 / sigma^2), sigma = inlier_threshold / 3, zero diagonal
//     v    = 1;  10 x { v = M v;  v = v / (|v| + 1e-6) }
//     labels = 1 for the int(N * top_ratio) largest entries of v;  pred_trans = rigid_transform_3d(src, tgt, v * labels)
// (the same matrix-vector power iteration is what models/PointDSC.py:170 (commented) / cal_confidence run on N x N.)
//
// Two kernels, both HBM-bound:
//   sm_matrix_kernel : writes M once, 4 N^2 bytes (tiles of 64 x 256, keypoints from L1/L2; the matrix is symmetric
//                      but is written plainly -- it is read 10 times, written once)
//   sm_matvec_kernel : y = M v_raw * scale, one wave per row block, lanes stride the columns with float4 loads
//                      (1 KiB per wave instruction, fully coalesced), v in LDS, wave-shuffle row reductions; the
//                      normalisation of iteration t is folded into iteration t+1's read of v (per-block partial sums of
//                      y^2 are summed in a fixed order, so the result is deterministic), one launch per iteration.
// 4 N^2 bytes per iteration: 100 MB at N = 5000, i.e. the whole matrix fits the 256 MiB Infinity Cache.
#include <math.h>
#include <mutex>
#include <type_traits>
#include "pdsc_common.h"

namespace pdsc {

constexpr int SMV_ROWS = 16;             // rows per workgroup in the mat-vec (4 waves x 4 rows)
constexpr int SMV_MAX_BLOCKS = 4096;     // partial-sum slots

__global__ __launch_bounds__(256) void sm_matrix_kernel(const float* __restrict__ corr, float sigma2, float* __restrict__ M,
                                                        long long ld, int N) {
    // thread -> 4 consecutive columns of one row; block = 64 rows x 256 columns (4 rows x 64 float4 per pass)
    const int b = blockIdx.z;
    const float* c = corr + (size_t)b * N * 6;
    float* Mb = M + (size_t)b * N * ld;
    const int j4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
    if (j4 >= ld) return;
    float cj[4][6];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = min(j4 + e, N - 1);
#pragma unroll
        for (int d = 0; d < 6; ++d) cj[e][d] = c[j * 6 + d];
    }
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + 4 * r;
        if (i >= N) break;
        float ci[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) ci[d] = c[i * 6 + d];
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ax = ci[0] - cj[e][0], ay = ci[1] - cj[e][1], az = ci[2] - cj[e][2];
            const float bx = ci[3] - cj[e][3], by = ci[4] - cj[e][4], bz = ci[5] - cj[e][5];
            const float ds = sqrtf((ax * ax + ay * ay) + az * az);              // torch.sum(diff ** 2, -1) ** 0.5
            const float dt = sqrtf((bx * bx + by * by) + bz * bz);
            const float d = ds - dt;
            float m = fmaxf(0.0f, 4.5f - ((d * d) / 2.0f) / sigma2);  // 4.5 - M**2 / 2 / sigma**2
            if (j4 + e == i || j4 + e >= N) m = 0.0f;
            out[e] = m;
        }
        *reinterpret_cast<f32x4*>(Mb + (size_t)i * ld + j4) = out;
    }
}

// y[i] = sum_j M[i][j] * (v[j] * scale),  scale = 1 / (sqrt(sum of the previous iteration's partials) + 1e-6) (or 1);
// partial_out[block] = sum of y[i]^2 over the block's rows.
__global__ __launch_bounds__(256) void sm_matvec_kernel(const float* __restrict__ M, long long ld, const float* __restrict__ v,
                                                        const float* __restrict__ partial_in, int n_partial_in,
                                                        float* __restrict__ y, float* __restrict__ partial_out, int N) {
    extern __shared__ __attribute__((aligned(16))) float vs[];      // [ld] scaled v, zero padded
    __shared__ float wsum[4];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float scale = 1.0f;
    if (partial_in) {
        // sum of the previous iteration's per-block |y|^2 in block order (fixed order: deterministic; every workgroup needs it before
        // its first multiply).  Staged through the (still unused) v area by one coalesced pass and summed from LDS in batches of 16
        // -- read from global memory one dependent add at a time it was a fifth of a workgroup's life at N = 10 000.  The padding
        // adds + 0.0f: same bits.
        const int np16 = (n_partial_in + 15) & ~15;                  // <= ld
        for (int p = t; p < np16; p += 256) vs[p] = p < n_partial_in ? partial_in[(size_t)b * SMV_MAX_BLOCKS + p] : 0.f;
        __syncthreads();
        float s = 0.f;
        for (int p0 = 0; p0 < np16; p0 += 16) {
            f32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(vs + p0 + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += q[u][e];
        }
        scale = 1.0f / (sqrtf(s) + 1e-6f);
        __syncthreads();                                             // (the v area is overwritten next)
    }
    const float* vb = v + (size_t)b * N;
    for (int j = t; j < ld; j += 256) vs[j] = j < N ? vb[j] * scale : 0.f;
    __syncthreads();
    const float* Mb = M + (size_t)b * N * ld;
    float sq = 0.f;
    {
        // the wave's 4 rows advance together: 4 independent 1-KiB loads in flight per step
        constexpr int R = SMV_ROWS / 4;
        const int i0 = blockIdx.x * SMV_ROWS + wave * R;
        const float* row[R];
        float acc[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            row[r] = Mb + (size_t)min(i0 + r, N - 1) * ld;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
        }
        // columns >= N never enter the sums: a caller's padding columns [N, ld) may hold anything (0 * NaN would poison a row)
        const int n4 = (N + 3) & ~3;                                // <= ld (ld is a multiple of 4)
        // four column steps (16 loads of 1 KiB per wave) in flight before the first multiply: with one step per trip a wave had 4 KiB
        // outstanding and a CU 16-32 KiB -- a quarter of what 8 TB/s x the HBM round trip asks for.  The fmaf chains still walk the
        // columns in ascending order: same bits.
        constexpr int U = 4;
        for (int j0 = lane * 4; j0 < n4; j0 += 256 * U) {
            f32x4 x[U], m[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 256 * u;
                if (j < n4) {
                    x[u] = *reinterpret_cast<const f32x4*>(vs + j);
#pragma unroll
                    for (int r = 0; r < R; ++r) m[u][r] = *reinterpret_cast<const f32x4*>(row[r] + j);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 256 * u;
                if (j < n4) {
                    if (j + 4 > N) {                                    // the one ragged group of the row
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) m[u][r][e] = j + e < N ? m[u][r][e] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r][e] = fmaf(m[u][r][e], x[u][e], acc[r][e]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float yi = wave_sum((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]));
            if (i0 + r < N) {                                       // wave-uniform
                if (lane == 0) y[(size_t)b * N + i0 + r] = yi;
                sq = fmaf(yi, yi, sq);
            }
        }
    }
    if (lane == 0) wsum[wave] = sq;
    __syncthreads();
    if (t == 0) partial_out[(size_t)b * SMV_MAX_BLOCKS + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ---- the matrix in the register file (r04): single pairs of N <= 5120 ------------------------------------------------
// The reference runs SM() one pair at a time (it asserts bs == 1).  4 N^2 bytes are 100 MB at N = 5000 and the chip's vector
// register files hold 128 MiB (256 CUs x 512 KiB): ONE persistent launch computes the matrix straight into registers -- 20 rows
// per workgroup (one workgroup per CU, one wave per SIMD with all 512 registers: 5 rows x 80 columns per lane), never written to
// HBM -- and runs every power iteration from there; per iteration only y (20 KB) crosses the chip, behind one grid barrier.
// Same arithmetic in the same order as sm_matrix_kernel + sm_matvec_kernel (column mapping of the lanes, fmaf chains, the
// per-16-row partial sums of |y|^2 and their sequential total): y and the partials are bit-identical to the streaming path's.
constexpr int SMR_CPL = 20;                      // f32x4 column groups per lane: columns 4 lane + 256 g .. + 3
constexpr int SMR_RPW = 5;                       // rows per wave
constexpr int SMR_ROWS = 4 * SMR_RPW;            // rows per workgroup
constexpr int SMR_MAXN = SMR_CPL * 256;          // 5120
constexpr int SMR_REPLICAS = 32;                 // copies of y (one wave instruction writes a row's value into all of them)

template <int B, int E, class F>
__device__ __forceinline__ void smr_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        smr_for<B + 1, E>(f);
    }
}

struct SmResidentArgs {
    const float* corr;      // [N][6]
    float sigma2;
    int N, iters, nwg;
    float* ya;              // [N] y of even iterations
    float* yb;              // [N] y of odd iterations
    float* yrep;            // [2][SMR_REPLICAS][SMR_MAXN] the same y, replicated: what the workgroups read back (see the kernel)
    float* partial_out;     // [nblocks] per-16-row partial sums of the LAST y (what sm_finish_kernel reads)
    unsigned int* bar;      // grid barrier flags (256 arrival words + 8 release words 64 B apart), zero at launch
};

// Grid barrier.  Agent-scope accesses bypass the L2s (eight XCDs, eight L2s) and are performed at the memory side, one line at a
// time -- so what a barrier costs is the number of requests that hit the same line.  Measured on this kernel (250 workgroups):
// every workgroup adding to one counter, or to one of eight, and polling it: ~25 us per barrier; every workgroup polling all 250
// arrival flags: the same.  Hence: every workgroup posts the round in ITS OWN arrival word (250 plain stores); workgroup 0 alone
// polls the arrival words (one coalesced 1 KiB load per round) and then posts the round in eight release words, one per XCD and
// 64 bytes apart; the other workgroups poll only their XCD's release word (~31 pollers per line).
// No fences: an agent-scope release / acquire fence is an L2-wide write-back / invalidate, and 4 waves x 31 workgroups per XCD
// issuing one each per barrier were the 35 us per iteration this kernel first showed -- whatever the flag protocol.  Instead the
// data that crosses workgroups (y: 20 KB per iteration) is itself stored and loaded at agent scope (write-through / L2 bypass),
// and a workgroup arrives only after its own stores have been acknowledged (s_waitcnt vmcnt(0) in every wave, then the
// workgroup barrier).  All workgroups are resident (one per CU, checked by the launcher), so the waits cannot starve.
constexpr int SMR_REL = 256, SMR_REL_STRIDE = 16;     // flag words: [0, 256) arrivals, 256 + 16 x: release word of XCD x
constexpr int SMR_ABORT = SMR_REL + SMR_REL_STRIDE * 8;      // one more word: set by whoever gives up waiting
// Every wait is BOUNDED (r05): the launcher asks the runtime for a cooperative launch (all workgroups co-resident or the launch
// fails), but a wait that could spin forever turns any broken assumption -- a CU mask, a pre-empted queue -- into a dead GPU.  A
// waiter that sees no progress for SMR_SPIN_BUDGET ticks of the 100 MHz constant clock raises the abort word; every waiter also polls
// it, so the whole grid leaves within one more poll and the kernel poisons its outputs (NaN) instead of hanging.
constexpr long long SMR_SPIN_BUDGET = 50ll * 1000 * 1000;     // 0.5 s at 100 MHz: ~40 000 x a healthy barrier
// returns false when the barrier was abandoned (workgroup-uniform)
__device__ __forceinline__ bool smr_grid_barrier(unsigned int* flags, int nwg, unsigned int round /* 1, 2, ... */) {
    __shared__ int s_abort;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    const long long t0 = (long long)wall_clock64();
    if (blockIdx.x == 0) {
        for (;;) {
            bool ok = true;
            if (threadIdx.x > 0 && (int)threadIdx.x < nwg) ok = __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= round;
            if (__syncthreads_and(ok)) break;
            if (threadIdx.x == 0 && ((long long)wall_clock64() - t0 > SMR_SPIN_BUDGET ||
                                     __hip_atomic_load(flags + SMR_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(flags + SMR_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
            }
            __syncthreads();
            if (s_abort) return false;
            __builtin_amdgcn_s_sleep(4);
        }
        if (threadIdx.x < 8) __hip_atomic_store(flags + SMR_REL + SMR_REL_STRIDE * threadIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (threadIdx.x == 0) {
            __hip_atomic_store(flags + blockIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int* rel = flags + SMR_REL + SMR_REL_STRIDE * (blockIdx.x & 7);
            unsigned int polls = 0;
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
                if ((++polls & 63u) == 0u && ((long long)wall_clock64() - t0 > SMR_SPIN_BUDGET ||
                                              __hip_atomic_load(flags + SMR_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(flags + SMR_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        if (s_abort) return false;
    }
    return true;
}

__global__ __launch_bounds__(256, 1) void sm_resident_kernel(SmResidentArgs a) {
    __shared__ __attribute__((aligned(16))) float vs[SMR_MAXN];      // scaled v, zero beyond N
    __shared__ __attribute__((aligned(16))) float parts[SMR_MAXN / SMV_ROWS];      // 320 = a multiple of 16; [nblocks, 320) stays 0
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int N = a.N;
    const int row0 = blockIdx.x * SMR_ROWS + wave * SMR_RPW;
    const float* c = a.corr;

    // ---- the matrix rows of this wave, computed once (arithmetic of sm_matrix_kernel) ----
    // (every index below is a compile-time constant -- smr_for unrolls in the front end -- so the 400 matrix values of a lane are
    //  registers from the start; with `#pragma unroll` loops the array went to scratch)
    f32x4 m[SMR_RPW][SMR_CPL];
    float ci[SMR_RPW][6];
    smr_for<0, SMR_RPW>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        smr_for<0, 6>([&](auto dc) { ci[r][decltype(dc)::value] = c[(size_t)min(row0 + r, N - 1) * 6 + decltype(dc)::value]; });
    });
    smr_for<0, SMR_CPL>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        const int j4 = lane * 4 + 256 * g;
        if (256 * g >= N) {                      // (workgroup-uniform) a column group entirely past N: never read by the mat-vec
            smr_for<0, SMR_RPW>([&](auto rc) { m[decltype(rc)::value][g] = f32x4{0.f, 0.f, 0.f, 0.f}; });
            return;
        }
        float cj[4][6];
        smr_for<0, 4>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const int j = min(j4 + e, N - 1);
            smr_for<0, 6>([&](auto dc) { cj[e][decltype(dc)::value] = c[(size_t)j * 6 + decltype(dc)::value]; });
        });
        smr_for<0, SMR_RPW>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int i = row0 + r;
            smr_for<0, 4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                const float ax = ci[r][0] - cj[e][0], ay = ci[r][1] - cj[e][1], az = ci[r][2] - cj[e][2];
                const float bx = ci[r][3] - cj[e][3], by = ci[r][4] - cj[e][4], bz = ci[r][5] - cj[e][5];
                const float ds = sqrtf((ax * ax + ay * ay) + az * az);
                const float dt = sqrtf((bx * bx + by * by) + bz * bz);
                const float d = ds - dt;
                float mm = fmaxf(0.0f, 4.5f - ((d * d) / 2.0f) / a.sigma2);
                if (j4 + e == i || j4 + e >= N) mm = 0.0f;
                m[r][g][e] = mm;
            });
        });
    });

    const int nblocks = ceil_div_dev(N, SMV_ROWS);
    for (int j = t; j < SMR_MAXN; j += 256) vs[j] = j < N ? 1.0f : 0.f;        // v0 = 1 (x scale 1)
    for (int p = t; p < SMR_MAXN / SMV_ROWS; p += 256) parts[p] = 0.f;
    __syncthreads();
    for (int it = 0; it < a.iters; ++it) {
        float* ycur = (it & 1) ? a.yb : a.ya;
        // y = M v: the lane's columns ascending, one fmaf chain per element, (a0 + a1) + (a2 + a3), wave_sum -- as sm_matvec_kernel
        float acc[SMR_RPW][4];
        smr_for<0, SMR_RPW>([&](auto rc) { smr_for<0, 4>([&](auto ec) { acc[decltype(rc)::value][decltype(ec)::value] = 0.f; }); });
        smr_for<0, SMR_CPL>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (256 * g < N && lane * 4 + 256 * g < ((N + 3) & ~3)) {       // the streaming kernel's loop bound (columns beyond contribute m = 0 anyway)
                const f32x4 x = *reinterpret_cast<const f32x4*>(vs + lane * 4 + 256 * g);
                smr_for<0, SMR_RPW>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    smr_for<0, 4>([&](auto ec) { constexpr int e = decltype(ec)::value; acc[r][e] = fmaf(m[r][g][e], x[e], acc[r][e]); });
                });
            }
        });
        smr_for<0, SMR_RPW>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const float yi = wave_sum((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]));
            // y crosses the chip through memory (agent scope: no L2 in between).  250 workgroups reading the same 20 KB back from
            // the handful of HBM channels it lives in took ~10 us per iteration; so a row's value is written into 32 copies by one
            // wave instruction (lane l -> copy l) and workgroup w reads copy w % 32: 8 readers per copy.  Lane 32 writes the plain
            // vector the finish kernel reads.
            if (row0 + r < N) {
                if (lane < SMR_REPLICAS)
                    __hip_atomic_store(a.yrep + ((size_t)(it & 1) * SMR_REPLICAS + lane) * SMR_MAXN + row0 + r, yi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (lane == SMR_REPLICAS)
                    ycur[row0 + r] = yi;
            }
        });
        if (!smr_grid_barrier(a.bar, a.nwg, (unsigned)(it + 1))) {
            // abandoned (see smr_grid_barrier): no hang, and nothing that could pass for a result -- the finish kernel's scale and
            // every output derived from it become NaN
            if (t == 0) a.partial_out[0] = __builtin_nanf("");
            return;
        }
        // every workgroup: y into LDS (one coalesced pass), |y|^2 in the streaming kernel's grouping (blocks of 16 rows: per wave a
        // chain over its 4 rows, then (w0 + w1) + (w2 + w3)), the blocks summed in order -> the same scale bits
        {
            // agent-scope (sc0 sc1: past both caches) 16-byte loads; rows past N come back as the zeros the copies were filled with
            const float* ysrc = a.yrep + ((size_t)(it & 1) * SMR_REPLICAS + (blockIdx.x & (SMR_REPLICAS - 1))) * SMR_MAXN;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ysrc, 0, SMR_MAXN * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < SMR_MAXN / 1024; ++u) {
                const int j = 4 * t + 1024 * u;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, j * 4, 0, 1 | 16));
                *reinterpret_cast<f32x4*>(vs + j) = v;
            }
        }
        __syncthreads();
        for (int p = t; p < nblocks; p += 256) {
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sq = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = p * SMV_ROWS + 4 * q + r;
                    if (i < N) sq = fmaf(vs[i], vs[i], sq);
                }
                w[q] = sq;
            }
            parts[p] = (w[0] + w[1]) + (w[2] + w[3]);
        }
        __syncthreads();
        if (it + 1 == a.iters) {
            if (blockIdx.x == 0)
                for (int p = t; p < nblocks; p += 256) a.partial_out[p] = parts[p];
            break;
        }
        float s = 0.f;                                       // the blocks in order, as the streaming kernel sums them; the padding adds
        for (int p0 = 0; p0 < nblocks; p0 += 16) {           // + 0.0f (exact), so a batch is 4 x ds_read_b128 and 16 dependent adds
            f32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(parts + p0 + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += q[u][e];
        }
        const float scale = 1.0f / (sqrtf(s) + 1e-6f);
        for (int j = t; j < N; j += 256) vs[j] = vs[j] * scale;       // (same product as the streaming kernel's vb[j] * scale)
        __syncthreads();
    }
}

// final normalisation + selection mask: eig = y * scale; weights = eig * [rank(eig) < num_top]  (stable descending
// rank by counting, ties by ascending index -- torch.argsort is unspecified there)
__global__ __launch_bounds__(256) void sm_finish_kernel(const float* __restrict__ y, const float* __restrict__ partial_in,
                                                        int n_partial_in, int num_top, float* __restrict__ eig,
                                                        float* __restrict__ labels, float* __restrict__ weights, int N) {
    __shared__ __attribute__((aligned(16))) float parts[SMV_MAX_BLOCKS];      // the partials, zero padded to a multiple of 16
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int np16 = (n_partial_in + 15) & ~15;
    for (int p = threadIdx.x; p < np16; p += 256) parts[p] = p < n_partial_in ? partial_in[(size_t)b * SMV_MAX_BLOCKS + p] : 0.f;
    __syncthreads();
    if (i >= N) return;
    float s = 0.f;                                                   // block order, as sm_matvec_kernel sums them (+ 0.0f padding: same bits)
    for (int p0 = 0; p0 < np16; p0 += 16) {
        f32x4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(parts + p0 + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += q[u][e];
    }
    const float scale = 1.0f / (sqrtf(s) + 1e-6f);
    const float* yb = y + (size_t)b * N;
    const float ei = yb[i] * scale;
    int cnt = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        if (j < N) {
            const float ej = yb[j] * scale;
            cnt += (ej > ei) || (ej == ei && j < i);
        }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) {
        const float lab = cnt < num_top ? 1.0f : 0.0f;
        eig[(size_t)b * N + i] = ei;
        labels[(size_t)b * N + i] = lab;
        weights[(size_t)b * N + i] = ei * lab;
    }
}


// ---- cal_confidence (reference models/PointDSC.py:366-401): confidence of a spectral-matching solution -------------
// One 1024-thread workgroup per pair for the O(N) vector steps (fixed-order reductions: deterministic), sm_matvec_kernel
// for every M x.  B = M - lambda1 v v^T is never formed: B x = M x - lambda1 v (v . x).
constexpr int CC_THREADS = 1024;
template <int NV>
__device__ __forceinline__ void cc_block_sum(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < CC_THREADS / 64; ++w) t += red[w * NV + i];
        v[i] = t;
    }
}
// step 0: lambda1 = (v . Mv) / (v . v); x <- 1.   method 0 / 2 finish here (conf = lambda1, or v . Mv / N)
__global__ __launch_bounds__(CC_THREADS) void cc_rayleigh_kernel(const float* __restrict__ v, const float* __restrict__ Mv, int method,
                                                                 float* __restrict__ lambda1, float* __restrict__ x,
                                                                 float* __restrict__ conf, int N) {
    __shared__ double red[CC_THREADS / 64 * 2];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mv + (size_t)b * N;
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        acc[0] += (double)vb[i] * (double)yb[i];
        acc[1] += (double)vb[i] * (double)vb[i];
        if (x) x[(size_t)b * N + i] = 1.0f;
    }
    cc_block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        const float l1 = (float)acc[0] / (float)acc[1];
        lambda1[b] = l1;
        if (method == 0) conf[b] = l1;
        if (method == 2) conf[b] = (float)acc[0] / (float)N;
    }
}
// one deflated power step: z = Mx - lambda1 v (v . x);  x <- z / (|z| + 1e-6)
__global__ __launch_bounds__(CC_THREADS) void cc_deflate_kernel(const float* __restrict__ v, const float* __restrict__ Mx,
                                                                const float* __restrict__ lambda1, float* __restrict__ x, int N) {
    __shared__ double red[CC_THREADS / 64];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mx + (size_t)b * N;
    float* xb = x + (size_t)b * N;
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) s[0] += (double)vb[i] * (double)xb[i];
    cc_block_sum<1>(s, red);
    const float coef = lambda1[b] * (float)s[0];
    double n2[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        const float z = yb[i] - coef * vb[i];
        n2[0] += (double)z * (double)z;
    }
    cc_block_sum<1>(n2, red);
    const float inv = 1.0f / ((float)sqrt(n2[0]) + 1e-6f);
    for (int i = threadIdx.x; i < N; i += CC_THREADS) xb[i] = (yb[i] - coef * vb[i]) * inv;
}
// lambda2 = (x . Bx) / (x . x);  conf = lambda1 / lambda2
__global__ __launch_bounds__(CC_THREADS) void cc_ratio_kernel(const float* __restrict__ v, const float* __restrict__ Mx,
                                                              const float* __restrict__ lambda1, const float* __restrict__ x,
                                                              float* __restrict__ conf, int N) {
    __shared__ double red[CC_THREADS / 64 * 3];
    const int b = blockIdx.x;
    const float* vb = v + (size_t)b * N;
    const float* yb = Mx + (size_t)b * N;
    const float* xb = x + (size_t)b * N;
    double acc[3] = {0.0, 0.0, 0.0};                 // v.x, x.Mx, x.x
    for (int i = threadIdx.x; i < N; i += CC_THREADS) {
        acc[0] += (double)vb[i] * (double)xb[i];
        acc[1] += (double)xb[i] * (double)yb[i];
        acc[2] += (double)xb[i] * (double)xb[i];
    }
    cc_block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        const double l1 = (double)lambda1[b];
        const float l2 = (float)((acc[1] - l1 * acc[0] * acc[0]) / acc[2]);     // x.Bx = x.Mx - lambda1 (v.x)^2
        conf[b] = lambda1[b] / l2;
    }
}

}  // namespace pdsc

using namespace pdsc;

extern "C" size_t pdsc_sm_workspace_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    const size_t ld = (size_t)pdsc_compat_ld(N);
    // matrix (streaming form) | 3 vectors | 2 x partial sums / barrier flags | y copies of the register-resident form
    return (size_t)bs * N * ld * 4 + (size_t)bs * N * 4 * 3 + (size_t)bs * SMV_MAX_BLOCKS * 4 * 2 +
           (N <= SMR_MAXN ? (size_t)bs * 2 * SMR_REPLICAS * SMR_MAXN * 4 : 0) + 1024;
}

// Two register-resident launches must never share the chip: each needs (nearly) every compute unit for its grid barrier, and two
// half-dispatched grids would wait for each other.  Three guards (r05):
//   1. the launch is COOPERATIVE (hipLaunchCooperativeKernel): the runtime refuses a grid it cannot make co-resident and orders
//      cooperative launches of one device among themselves -- the launch fails instead of hanging;
//   2. launches of different streams of this process are additionally chained through one event PER DEVICE (the next launch waits
//      for the previous one's end, whatever its stream) -- state keyed by the device the stream belongs to, not by the first device
//      this process happened to use;
//   3. every wait inside the kernel is bounded (smr_grid_barrier): what the two guards above cannot see -- another PROCESS with
//      the same kernel on the same GPU, a CU-masked queue -- ends in NaN outputs after 0.5 s, not in a dead GPU.
// The form is opt-in (`form = 2`); `form = 0` never picks it.  It is refused under stream capture: a replayed graph bypasses guard 2.
constexpr int SMR_MAX_DEVICES = 64;
struct SmrDeviceState {
    hipEvent_t done = nullptr;
    bool recorded = false;
    int cus = -1;
};
static std::mutex g_smr_mutex;
static SmrDeviceState g_smr_dev[SMR_MAX_DEVICES];

static SmrDeviceState* smr_device_state() {      // (caller holds g_smr_mutex) state of the CURRENT device, nullptr when it cannot be identified
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SMR_MAX_DEVICES) return nullptr;
    SmrDeviceState* s = &g_smr_dev[dev];
    if (s->cus < 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        s->cus = v;
    }
    return s;
}

static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
}

static int sm_device_cus() {
    std::lock_guard<std::mutex> lock(g_smr_mutex);
    SmrDeviceState* s = smr_device_state();
    return s ? s->cus : 0;
}

// form: 0 = the library's choice (= 1, always), 1 = HBM-streaming form, 2 = register-resident matrix, opt-in (N <= 5120 and 20 rows per CU, else an error)
static int sm_baseline_impl(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                            int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                            void* workspace, size_t workspace_bytes, int bs, int N, void* stream, int form, const char* who) {
    PDSC_REQUIRE(corr_pos && src_keypts && tgt_keypts && pred_trans && pred_labels && workspace, "%s: null pointer", who);
    PDSC_REQUIRE(bs > 0 && N > 1 && num_top >= 0 && num_top <= N && num_iterations >= 1, "%s: bs=%d N=%d top=%d iters=%d", who,
                 bs, N, num_top, num_iterations);
    PDSC_REQUIRE(ceil_div(N, SMV_ROWS) <= SMV_MAX_BLOCKS, "%s: N=%d too large (max %d)", who, N, SMV_ROWS * SMV_MAX_BLOCKS);
    if (workspace_bytes < pdsc_sm_workspace_bytes(bs, N)) {
        set_error("%s: workspace %zu < %zu bytes", who, workspace_bytes, pdsc_sm_workspace_bytes(bs, N));
        return PDSC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long ld = pdsc_compat_ld(N);
    float* M = (float*)workspace;
    float* va = M + (size_t)bs * N * ld;
    float* vb = va + (size_t)bs * N;
    float* wts = vb + (size_t)bs * N;
    float* pa = wts + (size_t)bs * N;
    float* pb = pa + (size_t)bs * SMV_MAX_BLOCKS;
    // sigma = inlier_threshold / 3 in double (python floats), sigma ** 2 in double, then the fp32 divisor of a tensor op
    const double sigma = (double)inlier_threshold / 3.0;
    const float sigma2 = (float)(sigma * sigma);
    const int nblocks = ceil_div(N, SMV_ROWS);
    int rc;
    float *vin, *pin, *vout;
    const int nwg = ceil_div(N, SMR_ROWS);
    const bool fits = N <= SMR_MAXN && nwg <= sm_device_cus();
    PDSC_REQUIRE(form != 2 || fits, "%s: the register-resident form needs N <= %d and %d rows per compute unit (N=%d, %d CUs)", who,
                 SMR_MAXN, SMR_ROWS, N, sm_device_cus());
    // form 0 = the streaming form, always (r05): the register-resident form is faster above N ~ 3500 but needs the whole chip to itself
    // (see the guards above), which a library cannot promise on behalf of its caller -- it is opt-in.
    // measured per call, 10 iterations (tools/sm_resident_probe.py, tools/sm_bench.py, profiles/r04_z_sm_resident.txt):
    // one pair of N = 1000 120 us resident / 87 streaming, 2048: 149 / 127, 3000: 182 / 173, 5000: 252 / 342; 8 pairs of N = 5000
    // 1776 / 1898 (consecutive pairs' launches overlap).  The resident form's iteration is a grid barrier and a y round trip
    // (8-13 us whatever N), the streaming form's a pass over 4 N^2 bytes (5.5-25 us); they cross a little above N = 3000.
    if (form == 2) {
        // the matrix never leaves the register file: one persistent launch per pair (pairs one after the other on the stream; the
        // reference itself runs one pair per call).  pb (unused by this form) holds each pair's grid-barrier counters.
        PDSC_REQUIRE(!stream_is_capturing(st), "%s: the register-resident form cannot be captured into a graph (its launches are ordered "
                     "through a per-device event a replay would bypass); use form 1", who);
        std::lock_guard<std::mutex> lock(g_smr_mutex);
        SmrDeviceState* ds = smr_device_state();
        PDSC_REQUIRE(ds, "%s: current device not identified", who);
        if (!ds->done && hipEventCreateWithFlags(&ds->done, hipEventDisableTiming) != hipSuccess) {
            set_error("%s: hipEventCreate failed", who);
            return PDSC_ERR_LAUNCH;
        }
        if (ds->recorded && hipStreamWaitEvent(st, ds->done, 0) != hipSuccess) {
            set_error("%s: hipStreamWaitEvent failed", who);
            return PDSC_ERR_LAUNCH;
        }
        rc = launch_fill_u32((unsigned int*)pb, 0u, (size_t)bs * SMV_MAX_BLOCKS + (size_t)bs * 2 * SMR_REPLICAS * SMR_MAXN, st);      // barrier flags + the y copies (zero past N)
        if (rc != PDSC_OK) return rc;
        for (int b = 0; b < bs; ++b) {
            SmResidentArgs a{};
            a.corr = corr_pos + (size_t)b * N * 6; a.sigma2 = sigma2; a.N = N; a.iters = num_iterations; a.nwg = nwg;
            a.ya = va + (size_t)b * N; a.yb = vb + (size_t)b * N;
            a.yrep = pb + (size_t)bs * SMV_MAX_BLOCKS + (size_t)b * 2 * SMR_REPLICAS * SMR_MAXN;
            a.partial_out = pa + (size_t)b * SMV_MAX_BLOCKS;
            a.bar = (unsigned int*)pb + (size_t)b * SMV_MAX_BLOCKS;
            void* kargs[] = {(void*)&a};
            const hipError_t le = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&sm_resident_kernel), dim3(nwg), dim3(256), kargs, 0, st);
            if (le != hipSuccess) {
                (void)hipGetLastError();
                set_error("%s(resident): cooperative launch of %d workgroups refused: %s", who, nwg, hipGetErrorString(le));
                return PDSC_ERR_LAUNCH;
            }
            rc = check_launch("pdsc_sm_baseline(resident)");
            if (rc != PDSC_OK) return rc;
        }
        if (hipEventRecord(ds->done, st) == hipSuccess) ds->recorded = true;
        else {
            (void)hipGetLastError();
            set_error("%s(resident): hipEventRecord failed", who);
            return PDSC_ERR_LAUNCH;
        }
        vin = ((num_iterations - 1) & 1) ? vb : va;      // the last y (unnormalised)
        vout = ((num_iterations - 1) & 1) ? va : vb;
        pin = pa;
    } else {
        hipLaunchKernelGGL(sm_matrix_kernel, dim3(ceil_div((int)ld, 256), ceil_div(N, 64), bs), dim3(256), 0, st, corr_pos, sigma2, M, ld, N);
        rc = check_launch("pdsc_sm_baseline(matrix)");
        if (rc != PDSC_OK) return rc;
        // v0 = 1 (bit pattern of 1.0f)
        rc = launch_fill_u32((unsigned int*)va, 0x3f800000u, (size_t)bs * N, st);      // a kernel, like every other fill of the library (no hipMemsetAsync)
        if (rc != PDSC_OK) return rc;
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sm_matvec_kernel), 160 * 1024 - 64, "pdsc_sm_baseline(dynamic LDS)");
        if (rc != PDSC_OK) return rc;
        float* pout = pa;
        vin = va; vout = vb; pin = nullptr;
        for (int it = 0; it < num_iterations; ++it) {
            hipLaunchKernelGGL(sm_matvec_kernel, dim3(nblocks, bs), dim3(256), (size_t)ld * sizeof(float), st, M, ld, vin, pin, nblocks, vout,
                               pout, N);
            rc = check_launch("pdsc_sm_baseline(matvec)");
            if (rc != PDSC_OK) return rc;
            float* tv = vin; vin = vout; vout = tv;
            pin = pout; pout = (pout == pa) ? pb : pa;
        }
    }
    // vin = last y (unnormalised), pin = its partial sums
    float* eig = leading_eig ? leading_eig : vout;
    hipLaunchKernelGGL(sm_finish_kernel, dim3(ceil_div(N, 4), bs), dim3(256), 0, st, vin, pin, nblocks, num_top, eig, pred_labels, wts, N);
    rc = check_launch("pdsc_sm_baseline(finish)");
    if (rc != PDSC_OK) return rc;
    return pdsc_rigid_transform_3d(src_keypts, tgt_keypts, wts, 0.0f, pred_trans, bs, N, stream);
}

extern "C" int pdsc_sm_baseline(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                                int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                                void* workspace, size_t workspace_bytes, int bs, int N, void* stream) {
    return sm_baseline_impl(corr_pos, src_keypts, tgt_keypts, inlier_threshold, num_top, num_iterations, pred_trans, pred_labels,
                            leading_eig, workspace, workspace_bytes, bs, N, stream, 0, "pdsc_sm_baseline");
}

extern "C" int pdsc_sm_baseline_form(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                                     int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                                     void* workspace, size_t workspace_bytes, int bs, int N, int form, void* stream) {
    PDSC_REQUIRE(form >= 0 && form <= 2, "pdsc_sm_baseline_form: form=%d (0 pick, 1 streaming, 2 register-resident)", form);
    return sm_baseline_impl(corr_pos, src_keypts, tgt_keypts, inlier_threshold, num_top, num_iterations, pred_trans, pred_labels,
                            leading_eig, workspace, workspace_bytes, bs, N, stream, form, "pdsc_sm_baseline_form");
}

extern "C" size_t pdsc_cal_confidence_workspace_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * N * 4 * 2 + (size_t)bs * SMV_MAX_BLOCKS * 4 + (size_t)bs * 16 + 1024;
}

extern "C" int pdsc_cal_confidence(const float* M, long long ld, const float* leading_eig, int method, int num_iterations,
                                   float* confidence, void* workspace, size_t workspace_bytes, int bs, int N, void* stream) {
    PDSC_REQUIRE(M && leading_eig && confidence && workspace, "pdsc_cal_confidence: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && ld >= N && ld % 4 == 0, "pdsc_cal_confidence: bs=%d N=%d ld=%lld (ld >= N, multiple of 4)", bs, N, ld);
    PDSC_REQUIRE(method >= 0 && method <= 2 && num_iterations >= 0, "pdsc_cal_confidence: method=%d iterations=%d", method, num_iterations);
    PDSC_REQUIRE(ceil_div(N, SMV_ROWS) <= SMV_MAX_BLOCKS && (size_t)ld * sizeof(float) <= 160 * 1024 - 64,
                 "pdsc_cal_confidence: N=%d too large", N);
    if (workspace_bytes < pdsc_cal_confidence_workspace_bytes(bs, N)) {
        set_error("pdsc_cal_confidence: workspace %zu < %zu bytes", workspace_bytes, pdsc_cal_confidence_workspace_bytes(bs, N));
        return PDSC_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* y = (float*)workspace;
    float* x = y + (size_t)bs * N;
    float* part = x + (size_t)bs * N;
    float* lam = part + (size_t)bs * SMV_MAX_BLOCKS;
    {
        const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&sm_matvec_kernel), 160 * 1024 - 64, "pdsc_cal_confidence(dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
    }
    const int nblocks = ceil_div(N, SMV_ROWS);
    auto matvec = [&](const float* in) {
        hipLaunchKernelGGL(sm_matvec_kernel, dim3(nblocks, bs), dim3(256), (size_t)ld * sizeof(float), st, M, ld, in, (const float*)nullptr,
                           0, y, part, N);
        return check_launch("pdsc_cal_confidence(matvec)");
    };
    int rc = matvec(leading_eig);
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(cc_rayleigh_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, method, lam, method == 1 ? x : nullptr,
                       confidence, N);
    rc = check_launch("pdsc_cal_confidence(rayleigh)");
    if (rc != PDSC_OK || method != 1) return rc;
    for (int it = 0; it < num_iterations; ++it) {
        rc = matvec(x);
        if (rc != PDSC_OK) return rc;
        hipLaunchKernelGGL(cc_deflate_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, lam, x, N);
        rc = check_launch("pdsc_cal_confidence(deflate)");
        if (rc != PDSC_OK) return rc;
    }
    rc = matvec(x);
    if (rc != PDSC_OK) return rc;
    hipLaunchKernelGGL(cc_ratio_kernel, dim3(bs), dim3(CC_THREADS), 0, st, leading_eig, y, lam, x, confidence, N);
    return check_launch("pdsc_cal_confidence(ratio)");
}
