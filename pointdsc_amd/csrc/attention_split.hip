// a-3, split-precision variant: the same spatial-consistency guided attention as attention.hip
// (reference models/PointDSC.py:39-42), but the two contractions run on the bf16 matrix cores with every fp32
// operand carried as hi + lo bf16 parts and three MFMAs per operand pair (split_layout.h): 3/16 of the
// matrix-pipe time of the exact fp32 MFMA at ~2^-16 relative error per product, which keeps the 12-layer
// features within 5e-6 and R/t within 1e-5 of the fp32 path (SURVEY.md Appendix B measured that ONE bf16 term
// is not enough: 6e-4 on R/t).  Softmax, accumulation and the output stay fp32.
//
// Bound: MFMA (bf16) with the compat stream (4 N^2 bytes per layer per pair) close behind on HBM.
//   executed flops = 3 x algorithmic (4 C N^2 per layer per pair) on v_mfma_f32_32x32x16_bf16.
//
// Decomposition: workgroup = NW waves (8, or 4 for small problems) = NW*32 queries x one contiguous range of
// 32-key tiles.  One wave = 32 queries:
//   S^T = K Q^T      A = K tile rows from LDS (ds_read_b128, XOR-swizzled image), B = Q hi/lo held in 64 VGPRs
//                    -> lane (query = lane&31, half h) holds the 16 keys (r&3)+8(r>>2)+4h of its query: the
//                       online softmax is lane-local (one cross-half shuffle per tile)
//   O^T += V^T P^T   A = V^T rows (channel-major image, keys in exactly the order the S^T accumulator holds them),
//                    B = P hi/lo converted in registers -> accumulator lane = query, rescale is per-lane scalar.
// K/V tiles are 32 KiB blocks stored in HBM as the exact LDS image (split_layout.h), fetched by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction), double buffered, one barrier per tile.  Workgroups that
// share a K/V range (same pair, same key split) are placed on the same XCD so the tiles are served by one L2.
#include <stdlib.h>
#include "attention_common.h"
#include "split_layout.h"

namespace pdsc {

struct AttSplitArgs {
    const __bf16* qs;            // [bs*N][256]  (hi | lo), q pre-scaled by log2(e)/sqrt(C)
    const unsigned char* kv;     // [bs][num_tiles][32 KiB]
    const float* compat;         // [bs][N][ld]
    long long ld;
    float* msg;                  // [bs*N][128]
    float* part_o;               // [bs][nsplit][Npad][128]
    float* part_ml;              // [bs][nsplit][Npad][2]
    int N, Npad, nsplit, num_tiles, nq, bs;
};

template <int NW>
__device__ __forceinline__ void issue_tile(const unsigned char* __restrict__ tile, unsigned char* buf, int wave, int lane) {
    constexpr int PPW = 32 / NW;                 // 1-KiB pieces per wave
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int i = wave * PPW + u;
        __builtin_amdgcn_global_load_lds((gptr_t)(tile + i * 1024 + lane * 16), (lptr_t)(buf + i * 1024), 16, 0, 0);
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void sc_attention_split_kernel(AttSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x 32 KiB tile images
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int N = a.N;

    // block -> (query block, key split, pair).  Blocks b, b+8, b+16.. run on the same XCD: give every
    // (pair, split) group -- the blocks that stream the same K/V tiles -- to one XCD when the group count allows.
    int qb, grp;
    {
        const int id = blockIdx.x, G = a.nsplit * a.bs;
        if ((G & 7) == 0) {
            const int i = id >> 3;
            grp = (id & 7) + 8 * (i / a.nq);
            qb = i % a.nq;
        } else {
            qb = id % a.nq;
            grp = id / a.nq;
        }
    }
    const int sp = grp % a.nsplit, b = grp / a.nsplit;

    const int per = a.num_tiles / a.nsplit, rem = a.num_tiles % a.nsplit;
    const int kt0 = sp * per + min(sp, rem);
    const int kt1 = kt0 + per + (sp < rem ? 1 : 0);

    const unsigned char* kvb = a.kv + (size_t)b * a.num_tiles * SPL_TILE_BYTES;
    const int qrow = min(qb * (NW * 32) + wave * 32 + l31, N - 1);
    const float* crow = a.compat + ((size_t)b * N + qrow) * a.ld + 4 * h;

    // prologue: first tile in flight, then this lane's Q fragments (hi and lo) and the first compat slice
    issue_tile<NW>(kvb + (size_t)kt0 * SPL_TILE_BYTES, lds, wave, lane);
    bf16x8 qh[8], ql[8];
    {
        const __bf16* qsrc = a.qs + ((size_t)b * N + qrow) * SPL_Q_LD + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qh[j] = *reinterpret_cast<const bf16x8*>(qsrc + 16 * j);
            ql[j] = *reinterpret_cast<const bf16x8*>(qsrc + PDSC_CHANNELS + 16 * j);
        }
    }
    f32x4 cc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cc[g] = *reinterpret_cast<const f32x4*>(crow + kt0 * SPL_BK + 8 * g);

    f32x16 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    const int koff = l31 * 256;                  // K image row of this lane (key l31)
    const int ksw = l31 & 15;
    const int voff = l31 * 64;                   // V^T image row of this lane (channel 32c + l31)
    const int vsw = (l31 >> 2) & 3;

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const unsigned char* T = lds + buf * SPL_TILE_BYTES;
        // tile kt landed (own LDS-DMA pieces) + everyone finished reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < kt1) issue_tile<NW>(kvb + (size_t)(kt + 1) * SPL_TILE_BYTES, lds + (buf ^ 1) * SPL_TILE_BYTES, wave, lane);

        // ---- S^T = K Q^T : hi*hi on one accumulator, the two cross terms on another --------------------
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int co = koff + (((2 * j + h) ^ ksw) << 4);
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(T + SPL_KH + co);
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(T + SPL_KL + co);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[j], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[j], s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[j], s1, 0, 0, 0);
        }

        // ---- online softmax (log2 domain), lane-local: this lane = query l31, keys (r&3)+8(r>>2)+4h --------
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = cc[r >> 2][r & 3] * (s0[r] + s1[r]);
        if (kt + 1 < kt1) {   // compat of the NEXT tile into the registers just consumed
#pragma unroll
            for (int g = 0; g < 4; ++g) cc[g] = *reinterpret_cast<const f32x4*>(crow + (kt + 1) * SPL_BK + 8 * g);
        }
        if ((kt + 1) * SPL_BK > N) {   // tail tile (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * SPL_BK + (r & 3) + 8 * (r >> 2) + 4 * h;
                x[r] = key < N ? x[r] : -INFINITY;
            }
        }
        float mloc = x[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, x[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(x[r] - m_run);
            psum += p;
            __bf16 hi, lo;
            split_bf16(p, hi, lo);
            ph[r >> 3][r & 7] = hi;
            pl[r >> 3][r & 7] = lo;
        }
        l_run += psum;

        // ---- O^T += V^T P^T : small terms first ------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int vo = c * 2048 + voff + (((2 * j + h) ^ vsw) << 4);
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(T + SPL_VH + vo);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(T + SPL_VL + vo);
                o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[j], o[c], 0, 0, 0);
                o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[j], o[c], 0, 0, 0);
                o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[j], o[c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: o[c][4g+e] = O^T[channel 32c + 8g + 4h + e][query l31] ---------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int query = qb * (NW * 32) + wave * 32 + l31;
    if (query < N) {
        if (a.nsplit == 1) {
            float* dst = a.msg + ((size_t)b * N + query) * PDSC_CHANNELS + 4 * h;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[c][4 * g] / l_tot, o[c][4 * g + 1] / l_tot, o[c][4 * g + 2] / l_tot, o[c][4 * g + 3] / l_tot};
                    *reinterpret_cast<f32x4*>(dst + 32 * c + 8 * g) = v;
                }
        } else {
            const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + query;
            float* dst = a.part_o + slot * PDSC_CHANNELS + 4 * h;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dst + 32 * c + 8 * g) = v;
                }
            if (h == 0) {
                a.part_ml[slot * 2 + 0] = m_run;
                a.part_ml[slot * 2 + 1] = l_tot;
            }
        }
    }
}

// ---- fp32 (q|k|v) rows -> split streams (the layer kernel's head epilogue does this in place; this stand-alone
//      packer serves the stage tests and callers that bring their own projections) ---------------------------
__global__ __launch_bounds__(256) void pack_qkv_split_kernel(const float* __restrict__ qkv, __bf16* __restrict__ qs,
                                                             unsigned char* __restrict__ kv, int N, int num_tiles) {
    const int tile = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float* rows = qkv + (size_t)b * N * 3 * PDSC_CHANNELS;
    unsigned char* img = kv + ((size_t)b * num_tiles + tile) * SPL_TILE_BYTES;
    const int k0 = tile * SPL_BK;
    // Q: thread -> (row, 4 channels)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
        if (k0 + row < N) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rows + (size_t)(k0 + row) * 3 * PDSC_CHANNELS + c4);
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { __bf16 a, c; split_bf16(v[e], a, c); hi[e] = a; lo[e] = c; }
            __bf16* dst = qs + ((size_t)b * N + k0 + row) * SPL_Q_LD + c4;
            *reinterpret_cast<bf16x4*>(dst) = hi;
            *reinterpret_cast<bf16x4*>(dst + PDSC_CHANNELS) = lo;
        }
    }
    // K: thread -> (key, chunk of 8 channels)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i, key = f >> 4, chunk = f & 15;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = k0 + key < N ? rows[(size_t)(k0 + key) * 3 * PDSC_CHANNELS + PDSC_CHANNELS + 8 * chunk + e] : 0.f;
            __bf16 a, c; split_bf16(v, a, c); hi[e] = a; lo[e] = c;
        }
        *reinterpret_cast<bf16x8*>(img + SPL_KH + spl_k_offset(key, chunk)) = hi;
        *reinterpret_cast<bf16x8*>(img + SPL_KL + spl_k_offset(key, chunk)) = lo;
    }
    // V^T: thread -> (channel, key chunk jh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i, ch = f & 127, jh = f >> 7;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = k0 + spl_v_key(jh, e);
            const float v = key < N ? rows[(size_t)key * 3 * PDSC_CHANNELS + 2 * PDSC_CHANNELS + ch] : 0.f;
            __bf16 a, c; split_bf16(v, a, c); hi[e] = a; lo[e] = c;
        }
        *reinterpret_cast<bf16x8*>(img + SPL_VH + spl_v_offset(ch, jh)) = hi;
        *reinterpret_cast<bf16x8*>(img + SPL_VL + spl_v_offset(ch, jh)) = lo;
    }
}

// waves per workgroup and key split for (bs, N): fill the 256 CUs (one 8-wave or two 4-wave workgroups each)
// with as few rounds x tiles-per-round as possible; prefer group counts that are multiples of 8 (XCD mapping)
static void split_plan(int bs, int N, int* nw_out, int* nsplit_out) {
    const int tiles = spl_num_tiles(N);
    const int nw = (long long)bs * ceil_div(N, 256) >= 48 ? 8 : 4;
    const int nq = ceil_div(N, nw * 32);
    const int slots = nw == 8 ? 256 : 512;
    const int cap = tiles / 4 > 1 ? tiles / 4 : 1;
    int best = 1;
    double best_cost = 1e30;
    for (int ns = 1; ns <= cap && ns <= 64; ++ns) {
        const int wgs = nq * bs * ns;
        const int rounds = ceil_div(wgs, slots);
        // per workgroup: its tiles + prologue/epilogue (~3 tiles' worth, more when partials are written)
        double cost = (double)rounds * (ceil_div(tiles, ns) + (ns > 1 ? 4.0 : 3.0));
        if (((ns * bs) & 7) != 0) cost *= 1.03;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
    }
    *nw_out = nw;
    *nsplit_out = best;
}

}  // namespace pdsc

using namespace pdsc;

extern "C" size_t pdsc_split_q_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * N * SPL_Q_LD * sizeof(__bf16);
}
extern "C" size_t pdsc_split_kv_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * spl_num_tiles(N) * SPL_TILE_BYTES;
}

extern "C" int pdsc_attention_split_default_split(int bs, int N) {
    if (bs <= 0 || N <= 0) return -1;
    int nw, ns;
    split_plan(bs, N, &nw, &ns);
    return ns;
}

extern "C" size_t pdsc_attention_split_scratch_bytes(int bs, int N, int nsplit) {
    if (bs <= 0 || N <= 0) return 0;
    int nw, ns;
    split_plan(bs, N, &nw, &ns);
    if (nsplit <= 0) nsplit = ns;
    if (nsplit == 1) return 0;
    const size_t slots = (size_t)bs * nsplit * round_up(N, 256);
    return slots * (PDSC_CHANNELS + 2) * sizeof(float);
}

extern "C" int pdsc_pack_qkv_split(const float* qkv, void* q_split, void* kv_tiles, int bs, int N, void* stream) {
    PDSC_REQUIRE(qkv && q_split && kv_tiles, "pdsc_pack_qkv_split: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_pack_qkv_split: bs=%d N=%d", bs, N);
    const int tiles = spl_num_tiles(N);
    hipLaunchKernelGGL(pack_qkv_split_kernel, dim3(tiles, bs), dim3(256), 0, (hipStream_t)stream, qkv, (__bf16*)q_split,
                       (unsigned char*)kv_tiles, N, tiles);
    return check_launch("pdsc_pack_qkv_split");
}

extern "C" int pdsc_sc_attention_split(const void* q_split, const void* kv_tiles, const float* compat, long long ld,
                                       float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                                       void* stream) {
    PDSC_REQUIRE(q_split && kv_tiles && compat && msg, "pdsc_sc_attention_split: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_sc_attention_split: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= round_up(N, SPL_BK) && ld % 4 == 0,
                 "pdsc_sc_attention_split: ld=%lld must be a multiple of 4 and >= N rounded up to 32", ld);
    const int tiles = spl_num_tiles(N);
    int nw, ns;
    split_plan(bs, N, &nw, &ns);
    static int force_nw = -1;
    if (force_nw < 0) {
        const char* env = getenv("PDSC_ATT_SPLIT_NW");      // tuning/A-B knob
        force_nw = env ? atoi(env) : 0;
    }
    if (force_nw == 4 || force_nw == 8) nw = force_nw;
    if (nsplit <= 0) nsplit = ns;
    if (nsplit > tiles) nsplit = tiles;
    const size_t need = nsplit == 1 ? 0 : (size_t)bs * nsplit * round_up(N, 256) * (PDSC_CHANNELS + 2) * sizeof(float);
    if (need > 0 && (!scratch || scratch_bytes < need)) {
        set_error("pdsc_sc_attention_split: scratch %zu < %zu bytes", scratch_bytes, need);
        return PDSC_ERR_WORKSPACE;
    }
    AttSplitArgs a{};
    a.qs = (const __bf16*)q_split; a.kv = (const unsigned char*)kv_tiles; a.compat = compat; a.ld = ld; a.msg = msg;
    a.N = N; a.Npad = (int)round_up(N, 256); a.nsplit = nsplit; a.num_tiles = tiles; a.bs = bs;
    a.nq = ceil_div(N, nw * 32);
    a.part_o = (float*)scratch;
    a.part_ml = a.part_o ? a.part_o + (size_t)bs * nsplit * a.Npad * PDSC_CHANNELS : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = 2 * SPL_TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sc_attention_split_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sc_attention_split_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const unsigned grid = (unsigned)(a.nq * nsplit * bs);
    profile_mark_begin(PDSC_PROF_ATTENTION, st);
    if (nw == 8)
        hipLaunchKernelGGL(sc_attention_split_kernel<8>, dim3(grid), dim3(512), lds_bytes, st, a);
    else
        hipLaunchKernelGGL(sc_attention_split_kernel<4>, dim3(grid), dim3(256), lds_bytes, st, a);
    profile_mark_end(PDSC_PROF_ATTENTION, st);
    int rc = check_launch("pdsc_sc_attention_split");
    if (rc != PDSC_OK) return rc;
    if (nsplit > 1) {
        AttArgs c{};
        c.msg = msg; c.part_o = a.part_o; c.part_ml = a.part_ml;
        c.N = N; c.Npad = a.Npad; c.nsplit = nsplit; c.num_tiles = tiles;
        rc = launch_attention_combine(c, bs, st);
    }
    return rc;
}
