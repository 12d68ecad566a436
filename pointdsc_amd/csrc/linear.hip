// a-2: point-wise layers = NT GEMMs  Y[M][Nout] = act(X[M][K] . W[Nout][K]^T + b) (+R), exact fp32 on
// v_mfma_f32_32x32x2_f32 (reference: every Conv1d(k=1)[+BN][+ReLU], models/PointDSC.py:12-23,54-61,107-113).
// The same kernel (MODE 1, gathered rows) produces the seed rows of the kNN distance matrix
// 2 - 2 <x_s, x_j> (reference models/common.py:58-60).
//
// Workgroup = 4 waves, tile 64 rows x (64*NT) columns, whole K (<=128) staged once in LDS with a +4-float
// row pad (16 distinct 16-B bank slots for the 16 rows of a ds_read_b128 lane group).  MFMA operand
// convention used throughout this library: k-slot (step t = 4q+e, half h = lane>>5) <-> channel 8q+4h+e,
// so one ds_read_b128 per lane feeds 4 MFMA steps for both A and B.
#include "pdsc_common.h"
#include "ragged.h"

namespace pdsc {

constexpr int LIN_BM = 64;

struct LinearArgs {
    const float* X; long long ldx; long long x_batch;       // X rows, batch stride (floats)
    const int* row_idx; long long idx_batch;                 // optional gather of X rows
    const float* W; long long w_batch;                       // [Nout][K]
    const float* bias;
    const float* R; long long ldr;
    float* Y; long long ldy; long long y_batch;
    int M, K, Nout, relu;
    const float* sigma;                                      // MODE 2: learned feature-compatibility sigma (device)
};

template <int NT, int MODE>
__global__ __launch_bounds__(256) void linear_kernel(LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = a.K, KP = K + 4, K4 = K >> 2;
    float* Xs = lds;                       // [64][KP]
    float* Ws = lds + LIN_BM * KP;         // [64*NT][KP]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * LIN_BM;
    const int n0 = blockIdx.x * (64 * NT);
    const float* X = a.X + (size_t)b * a.x_batch;
    const float* W = a.W + (size_t)b * a.w_batch;
    const int* ridx = a.row_idx ? a.row_idx + (size_t)b * a.idx_batch : nullptr;

    for (int idx = t; idx < LIN_BM * K4; idx += 256) {
        const int row = idx / K4, c = idx - row * K4;
        int m = m0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < a.M) {
            if (ridx) m = ridx[m];
            v = *reinterpret_cast<const f32x4*>(X + (size_t)m * a.ldx + c * 4);
        }
        *reinterpret_cast<f32x4*>(Xs + row * KP + c * 4) = v;
    }
    for (int idx = t; idx < 64 * NT * K4; idx += 256) {
        const int row = idx / K4, c = idx - row * K4;
        const int n = n0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < a.Nout) v = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + c * 4);
        *reinterpret_cast<f32x4*>(Ws + row * KP + c * 4) = v;
    }
    __syncthreads();

    const int wm = wave & 1, wn = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    const float* xa = Xs + (wm * 32 + l31) * KP + 4 * h;
    const float* wb = Ws + ((wn * NT) * 32 + l31) * KP + 4 * h;
    const int nq = K >> 3;
    for (int q = 0; q < nq; ++q) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(xa + 8 * q);
        f32x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const f32x4*>(wb + nt * 32 * KP + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[nt][e], acc[nt], 0, 0, 0);
    }

    float* Y = a.Y + (size_t)b * a.y_batch;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + l31;
        if (n >= a.Nout) continue;
        const float bval = (MODE == 0 && a.bias) ? a.bias[n] : 0.f;
        float sig2 = 1.f;
        if (MODE == 2) { const float sg = a.sigma[0]; sig2 = sg * sg; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 32 + i;
            if (m >= a.M) continue;
            float v = acc[nt][r];
            if (MODE == 0) {
                v = v + bval;
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.R) v = a.R[(size_t)m * a.ldr + n] + v;
            } else if (MODE == 1) {
                v = 2.0f - 2.0f * v;       // == reference `2 - 2*matmul` (one rounding)
            } else {
                // feature similarity matrix of the training/validation forward (models/PointDSC.py:158-163):
                // clamp(1 - (1 - <f_m, f_n>) / sigma^2, 0, 1), zero diagonal
                v = fminf(fmaxf(1.0f - (1.0f - v) / sig2, 0.0f), 1.0f);
                if (m == n) v = 0.0f;
            }
            Y[(size_t)m * a.ldy + n] = v;
        }
    }
}

// encoder.layer0: in_dim (<= 16: the reference's data loaders build 6-, 9- and 12-column inputs, datasets/ThreeDMatch.py:299-312)
// inputs -> C channels, pure VALU (K = 6 is too thin for MFMA); bound by the 512 B per point it writes.  A thread keeps its 4
// channels' weights (4 x KD + bias, rows of the packed matrix are 16 floats apart) in registers and walks its share of the
// rows: per row in_dim broadcast loads, 4 KD FMAs, one 16-byte store (a wave instruction stores 2 whole rows).
constexpr int L0_MAX_BLOCKS = 2048;
constexpr int L0_LD = 16;            // floats per row of PDSC_W_LAYER0_W
template <int KD>
__global__ __launch_bounds__(256) void layer0_kernel(const float* __restrict__ corr, int in_dim,
                                                     const float* __restrict__ W0, const float* __restrict__ b0,
                                                     float* __restrict__ feat, int M, unsigned int* __restrict__ range_flag, int bs) {
    // (r06) the forward's range sentinel starts clean: this is the first launch of the encoder
    if (range_flag && blockIdx.x == 0)
        for (int i = threadIdx.x; i < bs; i += 256) range_flag[i] = 0u;
    const int c4 = (threadIdx.x & 31) * 4, rl = threadIdx.x >> 5;        // 32 channel groups x 8 row lanes, grid-stride over rows
    float w[4][KD], bias[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int d4 = 0; d4 < KD; d4 += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(W0 + (c4 + c) * L0_LD + d4);
#pragma unroll
            for (int d = 0; d < 4; ++d) w[c][d4 + d] = v[d];
        }
        bias[c] = b0[c4 + c];
    }
    for (long long row = (long long)blockIdx.x * 8 + rl; row < M; row += (long long)gridDim.x * 8) {
        float x[KD];
#pragma unroll
        for (int d = 0; d < KD; ++d) x[d] = d < in_dim ? corr[row * in_dim + d] : 0.f;
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < KD; ++d) s = fmaf(x[d], w[c][d], s);     // one chain in column order (KD = 8: the r02 results, bit for bit)
            o[c] = s + bias[c];
        }
        *reinterpret_cast<f32x4*>(feat + row * PDSC_CHANNELS + c4) = o;
    }
}

// ---- a-4, first two layers of the confidence head in ONE launch (r04): h2 = relu(W2 relu(W1 x + b1) + b2), x [M][128] -> h2 [M][32]
//      reference: classification.0 .. classification.3 (models/PointDSC.py:107-111, :171).
// The forward used to issue two pdsc_linear launches for this (64 x 64 output tiles of which half is padding at Nout = 32, W staged
// again by each of the 2500 workgroups, the [M][32] hidden layer through HBM, no overlap of loads and MFMAs: 2 x 47 us at 32 pairs).
// Here a persistent workgroup keeps W1, W2 and the biases in LDS, walks 128-row tiles with the next tile's loads in flight under
// the MFMAs, and chains the two GEMMs in registers: the operand roles are swapped (A = W, B = x: accumulator lane = point, register
// r = output channel (r&3) + 8(r>>2) + 4h), so relu(h1 + b1) IS the B operand of the second GEMM (k-slot (4q+e, h) <-> channel
// 8q + 4h + e, the convention of this file).  Same instruction (v_mfma_f32_32x32x2_f32), same k order, a*b = b*a per product: every
// output element goes through the fma chain of the two pdsc_linear launches, bit for bit (test_classifier_hidden_equals_two_linears).
constexpr int CH_ROWS = 128, CH_K = PDSC_CHANNELS, CH_H = 32;
constexpr int CH_XLD = CH_K + 4, CH_W2LD = CH_H + 4;
struct ClassifierLds {
    float W1[CH_H * CH_XLD];
    float W2[CH_H * CH_W2LD];
    float b1[CH_H], b2[CH_H];
    float X[CH_ROWS * CH_XLD];
};

__global__ __launch_bounds__(256) void classifier_hidden_kernel(const float* __restrict__ X, const float* __restrict__ W1,
                                                               const float* __restrict__ b1, const float* __restrict__ W2,
                                                               const float* __restrict__ b2, float* __restrict__ H2, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ClassifierLds& sh = *reinterpret_cast<ClassifierLds*>(lds_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int num_tiles = ceil_div_dev(M, CH_ROWS);
    for (int idx = t; idx < CH_H * (CH_K / 4); idx += 256) {
        const int row = idx / (CH_K / 4), c = idx - row * (CH_K / 4);
        *reinterpret_cast<f32x4*>(sh.W1 + row * CH_XLD + 4 * c) = *reinterpret_cast<const f32x4*>(W1 + (size_t)row * CH_K + 4 * c);
    }
    for (int idx = t; idx < CH_H * (CH_H / 4); idx += 256) {
        const int row = idx / (CH_H / 4), c = idx - row * (CH_H / 4);
        *reinterpret_cast<f32x4*>(sh.W2 + row * CH_W2LD + 4 * c) = *reinterpret_cast<const f32x4*>(W2 + (size_t)row * CH_H + 4 * c);
    }
    if (t < CH_H) { sh.b1[t] = b1[t]; sh.b2[t] = b2[t]; }
    f32x4 stage[16];                              // this thread's share of a 128 x 128 tile: rows (t >> 5) + 8 i, columns 4 (t & 31) ..
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = tile * CH_ROWS + (t >> 5) + 8 * i;
            stage[i] = m < M ? *reinterpret_cast<const f32x4*>(X + (size_t)m * CH_K + 4 * (t & 31)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    int tile = blockIdx.x;
    if (tile < num_tiles) load_tile(tile);
    for (; tile < num_tiles; tile += gridDim.x) {
        __syncthreads();                          // the previous tile's readers are done (first pass: the weights are staged)
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(sh.X + ((t >> 5) + 8 * i) * CH_XLD + 4 * (t & 31)) = stage[i];
        __syncthreads();
        if (tile + gridDim.x < num_tiles) load_tile(tile + gridDim.x);       // in flight under the MFMAs below
        // GEMM 1: acc[r] = sum_k W1[chan(r)][k] x[point][k], lane = point wave * 32 + l31
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* wa = sh.W1 + l31 * CH_XLD + 4 * h;
        const float* xb = sh.X + (wave * 32 + l31) * CH_XLD + 4 * h;
#pragma unroll
        for (int q = 0; q < CH_K / 8; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(wa + 8 * q);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(xb + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
        // bias + relu: register r = channel (r&3) + 8(r>>2) + 4h; as B operand of GEMM 2: x[q][e] = h1[4q + e]
        f32x4 h1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) h1[q][e] = fmaxf(acc[4 * q + e] + sh.b1[e + 8 * q + 4 * h], 0.f);
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        const float* wa2 = sh.W2 + l31 * CH_W2LD + 4 * h;
#pragma unroll
        for (int q = 0; q < CH_H / 8; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(wa2 + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], h1[q][e], acc2, 0, 0, 0);
        }
        const int m = tile * CH_ROWS + wave * 32 + l31;
        if (m < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc2[4 * g + e] + sh.b2[e + 8 * g + 4 * h], 0.f);
                *reinterpret_cast<f32x4*>(H2 + (size_t)m * CH_H + 8 * g + 4 * h) = o;
            }
        }
    }
}

int launch_classifier_hidden(const float* X, const float* W1, const float* b1, const float* W2, const float* b2, float* H2, int M,
                             hipStream_t st) {
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&classifier_hidden_kernel), sizeof(ClassifierLds), "pdsc_classifier_hidden(dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    const int tiles = ceil_div(M, CH_ROWS);
    const int grid = tiles < 256 ? tiles : 256;        // one persistent workgroup per CU (89 KiB of LDS each)
    hipLaunchKernelGGL(classifier_hidden_kernel, dim3(grid), dim3(256), sizeof(ClassifierLds), st, X, W1, b1, W2, b2, H2, M);
    return check_launch("pdsc_classifier_hidden");
}

static size_t linear_lds_bytes(int NT, int K) { return (size_t)(LIN_BM + 64 * NT) * (K + 4) * sizeof(float); }

template <int NT, int MODE>
static int launch_linear(const LinearArgs& a, int batches, hipStream_t st) {
    const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&linear_kernel<NT, MODE>), linear_lds_bytes(NT, 128), "pdsc_linear(dynamic LDS)");
    if (rc_lds != PDSC_OK) return rc_lds;
    dim3 grid(ceil_div(a.Nout, 64 * NT), ceil_div(a.M, LIN_BM), batches);
    hipLaunchKernelGGL((linear_kernel<NT, MODE>), grid, dim3(256), linear_lds_bytes(NT, a.K), st, a);
    return check_launch("pdsc_linear");
}

// ---- row-block x all-columns Gram of L2-normalised features: the S x N distance rows of the seeds' kNN
//      (MODE 1: 2 - 2 <x_s, x_j>, models/common.py:58-60) and the N x N feature similarity matrix of the validation
//      forward (MODE 2: clamp(1 - (1 - <x_i, x_j>) / sigma^2, 0, 1), zero diagonal, models/PointDSC.py:158-163).
// A workgroup keeps 128 rows (32 per wave, gathered through `row_idx` in MODE 1) in registers as MFMA A fragments for its
// whole life and streams a range of 64-column tiles through a double-buffered LDS stage: the row operand is read once,
// the column operand once per row block (the generic linear_kernel re-reads both per 64 x 128 output tile and fits one
// workgroup per CU).  Accumulator lane = column, so each of the 16 stores of a tile writes 128 contiguous bytes per half
// wave.  Exact fp32 MFMA; bound: MFMA (256 flop per output element) with the 4-byte output stream behind it.
constexpr int GR_ROWS = 128, GR_LD = PDSC_CHANNELS + 4;

struct GramArgs {
    const float* X;          // [bs][N][128]
    const int* row_idx;      // MODE 1: [bs][R] rows of X; MODE 2: NULL (rows = 0..N-1)
    const float* sigma;      // MODE 2
    float* Y; long long ldy;
    int R, N, tiles_per_split;
};

// GR_COLS = columns per LDS stage: 64 (67.5 KiB, 2 workgroups per CU) or 32 (33.8 KiB, 4 per CU); same results bit for bit
template <int MODE, int GR_COLS>
__global__ __launch_bounds__(256, 2) void gram_rows_kernel(GramArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x [GR_COLS][GR_LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const float* X = a.X + (size_t)b * a.N * PDSC_CHANNELS;
    const int r0 = blockIdx.y * GR_ROWS + wave * 32;
    const int num_tiles = ceil_div_dev(a.N, GR_COLS);
    const int t0 = blockIdx.x * a.tiles_per_split, t1 = min(num_tiles, t0 + a.tiles_per_split);
    if (t0 >= t1) return;

    // A fragments of this lane's row: k-slot (4q+e, half h) <-> channel 8q+4h+e
    f32x4 af[16];
    {
        int row = min(r0 + l31, a.R - 1);
        if (MODE == 1) row = a.row_idx[(size_t)b * a.R + row];
        const float* p = X + (size_t)row * PDSC_CHANNELS + 4 * h;
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] = *reinterpret_cast<const f32x4*>(p + 8 * q);
    }
    float sig2 = 1.f;
    if (MODE == 2) { const float sg = a.sigma[0]; sig2 = sg * sg; }

    // stage loader: thread -> NST float4 of a GR_COLS x 128 tile
    constexpr int NST = GR_COLS * 32 / 256;
    f32x4 stage[NST];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int f = t + 256 * i, r = f >> 5, c4 = (f & 31) * 4;
            const int col = min(tile * GR_COLS + r, a.N - 1);
            stage[i] = *reinterpret_cast<const f32x4*>(X + (size_t)col * PDSC_CHANNELS + c4);
        }
    };
    auto store_tile = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int f = t + 256 * i, r = f >> 5, c4 = (f & 31) * 4;
            *reinterpret_cast<f32x4*>(buf + r * GR_LD + c4) = stage[i];
        }
    };
    load_tile(t0);
    store_tile(lds);
    __syncthreads();
    float* Yb = a.Y + (size_t)b * a.R * a.ldy;
    for (int tile = t0; tile < t1; ++tile) {
        const float* cur = lds + ((tile - t0) & 1) * GR_COLS * GR_LD;
        if (tile + 1 < t1) load_tile(tile + 1);                     // in flight under the MFMAs
#pragma unroll
        for (int sub = 0; sub < GR_COLS / 32; ++sub) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* brow = cur + (32 * sub + l31) * GR_LD + 4 * h;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 bf = *reinterpret_cast<const f32x4*>(brow + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][e], bf[e], acc, 0, 0, 0);
            }
            const int col = tile * GR_COLS + 32 * sub + l31;
            if (col < a.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.R) {
                        float v;
                        if (MODE == 1) v = 2.0f - 2.0f * acc[r];      // == reference `2 - 2*matmul` (one rounding)
                        else {
                            v = fminf(fmaxf(1.0f - (1.0f - acc[r]) / sig2, 0.0f), 1.0f);
                            if (row == col) v = 0.0f;
                        }
                        Yb[(size_t)row * a.ldy + col] = v;
                    }
                }
            }
        }
        if (tile + 1 < t1) {
            store_tile(lds + ((tile + 1 - t0) & 1) * GR_COLS * GR_LD);   // the other buffer: its readers finished a barrier ago
            __syncthreads();
        }
    }
}

template <int MODE, int GR_COLS>
static int launch_gram_cols(GramArgs a, int bs, hipStream_t st) {
    const size_t lds_bytes = 2 * (size_t)GR_COLS * GR_LD * sizeof(float);     // 67 584 B (64 columns)
    const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_rows_kernel<MODE, GR_COLS>), lds_bytes, "gram_rows(dynamic LDS)");
    if (rc_lds != PDSC_OK) return rc_lds;
    const int row_blocks = ceil_div(a.R, GR_ROWS), tiles = ceil_div(a.N, GR_COLS);
    int splits = ceil_div(GR_COLS == 32 ? 1536 : 768, row_blocks * bs);                   // ~3 workgroups per CU in flight
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    a.tiles_per_split = ceil_div(tiles, splits);
    splits = ceil_div(tiles, a.tiles_per_split);
    hipLaunchKernelGGL((gram_rows_kernel<MODE, GR_COLS>), dim3(splits, row_blocks, bs), dim3(256), lds_bytes, st, a);
    return check_launch("gram_rows_kernel");
}
template <int MODE>
static int launch_gram(GramArgs a, int bs, hipStream_t st) {
    return env_int("PDSC_GRAM_COLS", 64) == 32 ? launch_gram_cols<MODE, 32>(a, bs, st) : launch_gram_cols<MODE, 64>(a, bs, st);     // A/B knob
}

int knn_dist_rows(const float* normed, const int* seeds, float* dist, long long ldd, int bs, int N, int S,
                  hipStream_t st) {
    GramArgs a{};
    a.X = normed; a.row_idx = seeds; a.Y = dist; a.ldy = ldd; a.R = S; a.N = N;
    return launch_gram<1>(a, bs, st);
}

}  // namespace pdsc

extern "C" int pdsc_feature_compat(const float* normed, const float* sigma, float* Mout, long long ld, int bs, int N,
                                   void* stream) {
    PDSC_REQUIRE(normed && sigma && Mout, "pdsc_feature_compat: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0 && ld >= N, "pdsc_feature_compat: bs=%d N=%d ld=%lld", bs, N, ld);
    pdsc::GramArgs a{};
    a.X = normed; a.sigma = sigma; a.Y = Mout; a.ldy = ld; a.R = N; a.N = N;
    return pdsc::launch_gram<2>(a, bs, (hipStream_t)stream);
}

extern "C" int pdsc_linear(const float* X, long long ldx, const float* W, const float* bias, const float* residual,
                           long long ldr, float* Y, long long ldy, int M, int K, int Nout, int relu, void* stream) {
    PDSC_REQUIRE(X && W && Y, "pdsc_linear: null pointer");
    PDSC_REQUIRE(M > 0 && Nout > 0, "pdsc_linear: M=%d Nout=%d", M, Nout);
    PDSC_REQUIRE(K >= 8 && K <= 128 && K % 8 == 0, "pdsc_linear: K=%d must be a multiple of 8 in [8,128]", K);
    PDSC_REQUIRE(ldx >= K && ldx % 4 == 0, "pdsc_linear: ldx=%lld", ldx);
    PDSC_REQUIRE(ldy >= Nout, "pdsc_linear: ldy=%lld < Nout", ldy);
    PDSC_REQUIRE(!residual || ldr >= Nout, "pdsc_linear: ldr=%lld < Nout", ldr);
    pdsc::LinearArgs a{};
    a.X = X; a.ldx = ldx; a.W = W; a.bias = bias; a.R = residual; a.ldr = ldr; a.Y = Y; a.ldy = ldy;
    a.M = M; a.K = K; a.Nout = Nout; a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    if (Nout > 64) return pdsc::launch_linear<2, 0>(a, 1, st);
    return pdsc::launch_linear<1, 0>(a, 1, st);
}

namespace pdsc {
int launch_layer0(const float* corr_pos, int in_dim, const float* W0, const float* b0, float* feat, int M, unsigned int* range_flag, int bs,
                  hipStream_t st) {
    PDSC_REQUIRE(corr_pos && W0 && b0 && feat, "pdsc_layer0: null pointer");
    PDSC_REQUIRE(in_dim >= 1 && in_dim <= 16 && M > 0, "pdsc_layer0: in_dim=%d (1..16) M=%d", in_dim, M);
    const int blocks = ceil_div(M, 8) < L0_MAX_BLOCKS ? ceil_div(M, 8) : L0_MAX_BLOCKS;
    if (in_dim <= 8)
        hipLaunchKernelGGL(layer0_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, corr_pos, in_dim, W0, b0, feat, M, range_flag, bs);
    else
        hipLaunchKernelGGL(layer0_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, corr_pos, in_dim, W0, b0, feat, M, range_flag, bs);
    return check_launch("pdsc_layer0");
}
}  // namespace pdsc

extern "C" int pdsc_layer0(const float* corr_pos, int in_dim, const float* W0, const float* b0, float* feat, int M,
                           void* stream) {
    return pdsc::launch_layer0(corr_pos, in_dim, W0, b0, feat, M, nullptr, 0, (hipStream_t)stream);
}

extern "C" int pdsc_classifier_hidden(const float* feat, const float* W1, const float* b1, const float* W2, const float* b2, float* h2,
                                      int M, void* stream) {
    PDSC_REQUIRE(feat && W1 && b1 && W2 && b2 && h2 && M > 0, "pdsc_classifier_hidden: bad argument");
    return pdsc::launch_classifier_hidden(feat, W1, b1, W2, b2, h2, M, (hipStream_t)stream);
}
