// a-3, split-precision variant: the same spatial-consistency guided attention as attention.hip
// (reference models/PointDSC.py:39-42), but the two contractions run on the f16 matrix cores with every fp32
// operand carried as fp16 hi + fp16 lo parts and three MFMAs per operand pair (split_layout.h): 3/16 of the
// matrix-pipe time of the exact fp32 MFMA at ~2^-21 relative error per product (rounds 1-4: bf16 pairs, 2^-16, which
// held the seeded checkpoints' features within 5e-6 of the fp32 path but left trained-like KITTI logits 0.27 out;
// SURVEY.md Appendix B measured that ONE 16-bit term is not enough: 6e-4 on R/t).  Softmax, accumulation and the
// output stay fp32.
//
// Bound: MFMA (f16) with the compat stream (4 N^2 bytes per layer per pair) close behind on HBM.
//   executed flops = 3 x algorithmic (4 C N^2 per layer per pair) on v_mfma_f32_32x32x16_f16.
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
#include <type_traits>
#include "attention_common.h"
#include "split_layout.h"
#include "ragged.h"

namespace pdsc {


// LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave instruction, descriptor + scalar offset + one 32-bit lane
// offset -- no 64-bit address registers) of one part of a tile.  The K part (16 KiB) and the V^T part (16 KiB) of a
// 32 KiB tile image (split_layout.h) are linear copies, pieces dealt round-robin to the waves.
template <int NW, int BYTES>
__device__ __forceinline__ void issue_linear(__amdgpu_buffer_rsrc_t rsrc, int src_off, unsigned char* dst, int wave, unsigned lane16) {
    constexpr int PIECES = BYTES / 1024;
    static_assert(BYTES % 1024 == 0, "tile parts are whole KiB");
#pragma unroll
    for (int u = 0; u < (PIECES + NW - 1) / NW; ++u) {
        const int i = wave + NW * u;
        if ((u + 1) * NW <= PIECES || i < PIECES)        // wave-uniform; only the last round can be partial
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + i * 1024), 16, lane16, src_off + i * 1024, 0, 0);
    }
}
// o *= alpha in place: one asm statement per accumulator register keeps the 64 registers where they are (a plain
// `o[c][r] *= alpha` inside a branch made the register allocator keep two copies of O and move one per tile)
__device__ __forceinline__ f32x16 scale_acc(f32x16 o, float alpha) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = o[r];
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(alpha));
        o[r] = v;
    }
    return o;
}

constexpr float ATT_RESCALE_THR = 8.0f;   // running max is only moved when a logit exceeds it by 2^8 (log2 domain)
// The reference exponent m_run of a query sits ATT_P_BIAS BELOW its running row maximum: p = exp2(logit - m_run) then spans
// (0, 2^(BIAS + THR)] = (0, 32768] instead of (0, 256].  p is split into fp16 hi + lo for the P V product (split_layout.h), and
// fp16's denormal floor is absolute (2^-25 rounds to 0): with the maximum at 1 every key more than 17.3 nats below it would drop
// out of the sum (up to N 2^-25 = 6e-4 of the row sum at N = 20000, which fp32 keeps); with the maximum at 2^7 the floor is
// 2^-32 of the largest term.  The bias is a common factor of o and l (and of every leaf's partials) and cancels in o / l.
constexpr float ATT_P_BIAS = 7.0f;
constexpr int SPL_K_BYTES = 2 * SPL_K_PLANE;    // Kh | Kl   16 KiB
constexpr int SPL_V_BYTES = 2 * SPL_V_PLANE;    // Vh | Vl   16 KiB

// Software pipeline (per wave, per key tile t):  phase A = QK^T of tile t+1 on the matrix pipe WHILE the VALU turns
// tile t's logits into P (exp2, hi/lo split);  phase B = P V of tile t WHILE the VALU forms tile t+1's logits.  Both
// waves of a SIMD therefore always have matrix and vector work to interleave (with the plain QK -> softmax -> PV order
// the two waves, released by the same barrier, do their softmax at the same time and the matrix pipe idles).
// K and compat consequently run one tile ahead of V: per array two LDS stages, 128 KiB in all.
#define PDSC_TRACE_STAMP(k)                                                      \
    if (TRACE) {                                                                 \
        const long long now__ = __builtin_readcyclecounter();                    \
        tr[k] += now__ - tlast;                                                  \
        tlast = now__;                                                           \
    }

// C16: the compat stream is unorm16 (pdsc_spatial_compat_u16: value = u / 65535, 0 and 1 exact, |error| <= 2^-17; inside
// every 32-key group the 16 keys a lane half holds are contiguous: position 16 h + 4 g + e for key 8 g + 4 h + e), i.e.
// 64 B per query row per tile instead of 128: half the HBM stream, half the compat LDS stage and DMA instructions.
// CM (compat mode): 0 = fp32 matrix, slices staged in LDS by LDS-DMA (default);  1 = unorm16 matrix, staged in LDS (C16);
// 2 = fp32 matrix, every lane loads the 16 values of its query straight into registers (CREG: no compat LDS stage -- 64 KiB
// per workgroup, so two 4-wave workgroups fit a CU and one's prologue / epilogue overlaps the other's main loop).
// PS (persistent): one workgroup per CU walks several (pair, key split, query block) items.  The loads that run ahead of an
// item's last tiles fetch the NEXT item's first K / compat / V tiles (today they fetch tiles nobody reads), so the
// prologue's HBM round trip and the first tile's wait (5 % of a workgroup's life, tools/attention_trace.py) sit behind the
// previous item's last tiles.  Point-fragment partials only: that epilogue needs no LDS, the stages stay live across items.
// PEEL: the split's last tile runs as peeled tail code (see tile_iteration); false = the r01-r03 straight-line loop (A/B record,
// experiments builds: PDSC_ATT_PEEL=0)
// MG (r05, leaf form): the key range of a pair is cut into a.nleaf LEAVES (a function of the pair's own tile count and of a.nleaf
// only -- with the canonical leaf count of attention_leaf_count(N) a function of N alone); every leaf is accumulated from a fresh
// online-softmax state, whoever computes it, and leaves a partial (O, m, l) in point-fragment order in part_o / part_ml
// [pair][leaf][Npad]; a workgroup owns the consecutive leaves [sp C / nsplit, (sp + 1) C / nsplit) and streams through them without
// a break in its K / V / compat pipeline (a leaf that ends inside its range costs 17 store instructions per wave and a reset of the
// accumulators).  The fused layer kernel merges the C leaf partials in leaf order while it loads them (merge_partials.h), exactly as
// it merges key-split partials: with canonical leaves the bits of a pair do not depend on how many pairs share its launch.
// (Measured and dropped, profiles/r05_b_ab_leaves_in_kernel_merge.txt: merging inside this launch -- the last wave to finish a
//  query tile, found through a ticket, loads the other partials past the caches -- puts 20-30 us of dependent memory round trips at
//  the end of every workgroup: +14 % per launch at 32 pairs of N = 5000, against -5 % for the layer launch that then loads one message.)
// Energy ablation (diagnostic builds only: PDSC_HIPCC_EXTRA=-DPDSC_ATT_ABLATE=<mask>, WRONG results, timing / power only;
// tools/attention_power.py, profiles/r06_attention_energy_budget.txt): bit 0 drops the V_hi * P_lo MFMAs (1/6 of the launch's MFMAs),
// bit 1 the two other low-order terms (K_lo * Q_hi, V_lo * P_hi: 1/3), bit 2 the last one (K_hi * Q_lo: 1/6) -- mask 7 leaves the
// hi * hi products alone (1/3 of the MFMAs).  The product and experiments libraries are built without the macro.
#ifndef PDSC_ATT_ABLATE
#define PDSC_ATT_ABLATE 0
#endif
#define PDSC_MFMA_IF(bit, a_, b_, c_) (((PDSC_ATT_ABLATE) & (bit)) ? (c_) : PDSC_MFMA_X3(a_, b_, c_, 0, 0, 0))

template <int NW, int CM = 0, bool TRACE = false, bool PS = false, bool PEEL = true, bool MG = false>
__global__ __launch_bounds__(NW * 64, 2) void sc_attention_split_kernel(AttSplitArgs a) {
    constexpr bool C16 = CM == 1, CREG = CM == 2;
    static_assert(!PS || (!CREG && !TRACE), "the persistent form exists for the LDS-staged compat formats, untraced");
    static_assert(!MG || (!PS && !TRACE && !CREG && PEEL), "the merged form: one-item, untraced, LDS-staged compat, peeled last tile");
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = TRACE ? (long long)__builtin_readcyclecounter() : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const Ks = lds;                                  // 2 x 16 KiB
    unsigned char* const Vs = lds + 2 * SPL_K_BYTES;                // 2 x 16 KiB
    unsigned char* const Cs = Vs + 2 * SPL_V_BYTES;                 // 2 x NW*4 KiB
    constexpr int CROW = C16 ? 64 : 128;         // bytes of compat per query row per tile
    constexpr int CSTAGE = CREG ? 0 : NW * 32 * CROW;
    constexpr int NCS = CREG ? 0 : (C16 ? 2 : 4);     // compat DMA pieces (1 KiB) per wave per tile
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int NS = a.N;              // rows per pair in every buffer (ragged batches: the longest pair)

    // block -> (query block, key split, pair).  Blocks i, i+8, i+16.. run on the same XCD (own L2): hand every XCD
    // a contiguous run of the (pair, split, query block) list, so that the workgroups streaming the same K/V tiles
    // -- same pair, same split -- sit on one or two XCDs instead of all eight.
    int qb, grp;
    int item = 0, item_end = 0, item_step = 0;   // PS: this workgroup's items = item, item + item_step, ... < item_end
    if constexpr (PS) {
        // XCD x (= blockIdx.x & 7) owns the contiguous run [x * items / 8, (x + 1) * items / 8) of the item list (same pair,
        // same split -> same L2, as in the one-item form); its workgroups take every (gridDim.x / 8)-th item of it
        const int id = blockIdx.x, per_xcd = a.items >> 3;
        item = (id & 7) * per_xcd + (id >> 3);
        item_end = (id & 7) * per_xcd + per_xcd;
        item_step = gridDim.x >> 3;
        if (item >= item_end) return;
        grp = item / a.nq;
        qb = item % a.nq;
    } else {
        const int id = blockIdx.x, W = gridDim.x;
        const int rank = (W & 7) == 0 ? (id & 7) * (W >> 3) + (id >> 3) : id;
        grp = rank / a.nq;
        qb = rank % a.nq;
    }
    int sp = grp % a.nsplit, b = grp / a.nsplit;
    // ragged batches (pdsc_forward_testing_ragged): this pair's own correspondence count -- its queries, its key tiles and the
    // mask of its last tile; buffer strides stay those of the longest pair.  (Not in the persistent form: one N per launch.)
    const int N = (!PS && a.nvalid) ? a.nvalid[b] : NS;
    if (!PS && qb * (NW * 32) >= N) return;      // query block entirely past this pair's rows (workgroup-uniform)
    const int ntiles = (!PS && a.nvalid) ? ceil_div_dev(N, SPL_BK) : a.num_tiles;

    const int per = ntiles / a.nsplit, rem = ntiles % a.nsplit;
    int kt0 = sp * per + min(sp, rem);
    int kt1 = kt0 + per + (sp < rem ? 1 : 0);
    // MG: leaf c = tiles [c lper + min(c, lrem), ...) (a.nleaf <= ntiles); this workgroup owns leaves [leaf, lf1)
    const int lper = MG ? ntiles / a.nleaf : 0, lrem = MG ? ntiles % a.nleaf : 0;
    int leaf = 0, leaf_end = 0;
    if constexpr (MG) {
        const int lf1 = (sp + 1) * a.nleaf / a.nsplit;
        leaf = sp * a.nleaf / a.nsplit;
        kt0 = leaf * lper + min(leaf, lrem);
        kt1 = lf1 * lper + min(lf1, lrem);
        leaf_end = (leaf + 1) * lper + min(leaf + 1, lrem);
    }

    // buffer descriptor of this pair's K/V tile stream (< 4 GiB)
    constexpr unsigned CEL = C16 ? 2u : 4u;      // bytes per compat element
    // (macros, not lambdas: with a lambda that returns a buffer resource hipcc 7.2's host pass silently drops every launch stub
    //  of this template)
#define PDSC_KV_RSRC(pair) \
    __builtin_amdgcn_make_buffer_rsrc((void*)(a.kv + (size_t)(pair) * a.num_tiles * SPL_TILE_STRIDE), 0, a.num_tiles * SPL_TILE_STRIDE, 0x00020000)
    // compat: the descriptor covers only this workgroup's query rows, so every offset fits 32 bits whatever N is
#define PDSC_C_RSRC(pair, qblock)                                                                                                          \
    __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.compat + ((size_t)(pair) * NS + (qblock) * (NW * 32)) * a.ld * CEL), 0, \
                                      (int)((unsigned)min(NW * 32, N - (qblock) * (NW * 32)) * (unsigned)a.ld * CEL), 0x00020000)
    __amdgpu_buffer_rsrc_t kv_rsrc = PDSC_KV_RSRC(b);
    const int q_first = qb * (NW * 32);
    int q_rows = min(NW * 32, N - q_first);
    __amdgpu_buffer_rsrc_t c_rsrc = PDSC_C_RSRC(b, qb);
    // PS: kv_rsrc / c_rsrc address the K and compat tiles kt + dK, kv_v the V tile kt + dV of the run-ahead loads -- this
    // item's streams (dK = 2, dV = 1) until its last two iterations, then the next item's (few scalar registers: with one
    // descriptor set per item in flight the kernel spilled 49 of them into vector lanes, read back inside the tile loop)
    __amdgpu_buffer_rsrc_t kv_v = kv_rsrc;
    int dK = 2, dV = 1, kt0n = kt1, q_rows_n = q_rows, b_n = b, sp_n = sp, qb_n = qb, u0 = 0;
    bool more = false;
    const unsigned lane16 = lane * 16;
    // (sized 4, not NCS: with a template-dependent bound hipcc 7.2's host pass silently drops the kernel's launch stub)
    unsigned coff[4] = {0u, 0u, 0u, 0u};         // byte offset of this lane's 16-B compat chunk per DMA piece, tile 0
    // CREG: byte offset of this lane's first 16-B chunk (keys 4h..4h+3 of tile 0) inside the descriptor; chunk g at + 32 g
    const unsigned creg_off = (unsigned)min(wave * 32 + (lane & 31), q_rows - 1) * (unsigned)a.ld * 4u + 16u * (unsigned)(lane >> 5);
    auto creg_load = [&](int kt, f32x4 (&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dst[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rsrc, creg_off + 32u * g, kt * (SPL_BK * 4), 0));
    };
    auto set_coff = [&](int rows) {
#pragma unroll
        for (int u = 0; u < NCS; ++u) {
            if (C16) {
                // a wave instruction = 16 rows x 64 B; LDS row = 4 chunks of 16 B, logical chunk c stored at c ^ ((row >> 2) & 3)
                const int row = wave * 32 + 16 * u + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                coff[u] = (unsigned)min(row, rows - 1) * (unsigned)a.ld * 2u + 16u * c;
            } else {
                const int row = wave * 32 + 8 * u + (lane >> 3);             // row inside the workgroup's query block
                const int c = (lane & 7) ^ ((row >> 1) & 7);                 // logical 16-B chunk (4 keys) this lane fetches
                coff[u] = (unsigned)min(row, rows - 1) * (unsigned)a.ld * 4u + 16u * c;
            }
        }
    };
    set_coff(q_rows);
    // DMA work of one wave for one loop iteration kt, as 9 slots that are issued BETWEEN the MFMA groups (an LDS-DMA
    // instruction costs its wave ~100 cycles of issue; bunched after the barrier that is ~1000 cycles per tile during
    // which neither wave of a SIMD feeds the matrix pipe):
    //   slots 0-3: compat_{kt+2} pieces (HBM latency: first), slots 4-8: K_{kt+2} / V_{kt+1} pieces i = wave + NW*u.
    // Straight-line: tiles past the end of the split are fetched like any other (<= 2 wasted tiles per workgroup; past
    // the end of the buffer the descriptor's bounds check returns zeros); they land in stages nobody reads any more.
    constexpr int KPIECES = SPL_K_BYTES / 1024, PIECES = SPL_TILE_BYTES / 1024, KV_SLOTS = (PIECES + NW - 1) / NW;
    auto dma_slot = [&](int kt, int st, int slot) {
        if (slot < NCS) {
            // the fp32 compat slices are read once per launch: streamed with the non-temporal policy (aux = 2) they leave
            // the L2 to the K/V tiles the other workgroups of the XCD re-read: +1.6 % pairs/s; the unorm16 stream measured
            // faster without (tools/ab_forward.py, profiles/r02_c_ab_forward_compat_format_b32.txt)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lptr_t)(Cs + st * CSTAGE + (wave * NCS + slot) * 1024), 16, coff[slot],
                                                     (PS ? kt + dK : kt + 2) * (SPL_BK * (int)CEL), 0, C16 ? 0 : 2);
        } else {
            const int i = min(wave + NW * (slot - NCS), PIECES - 1);    // surplus slots repeat the last piece
            const bool isk = i < KPIECES;                                // wave-uniform
            unsigned char* dst = isk ? Ks + st * SPL_K_BYTES + i * 1024 : Vs + (st ^ 1) * SPL_V_BYTES + (i - KPIECES) * 1024;
            const int src = (isk ? (PS ? kt + dK : kt + 2) : (PS ? kt + dV : kt + 1)) * SPL_TILE_STRIDE + i * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(PS ? (isk ? kv_rsrc : kv_v) : kv_rsrc, (lptr_t)dst, 16, lane16, src, 0, 0);
        }
    };
    constexpr int DMA_SLOTS = NCS + KV_SLOTS;    // fp32 compat: 9 (NW = 8) or 14 (NW = 4): 8 go after the QK steps, the rest after PV steps
    static_assert(DMA_SLOTS <= 16, "16 places per iteration");
    auto dma_k = [&](int kt) { issue_linear<NW, SPL_K_BYTES>(kv_rsrc, kt * SPL_TILE_STRIDE + SPL_KH, Ks + ((kt - kt0) & 1) * SPL_K_BYTES, wave, lane16); };
    auto dma_v = [&](int kt) { issue_linear<NW, SPL_V_BYTES>(kv_rsrc, kt * SPL_TILE_STRIDE + SPL_VH, Vs + ((kt - kt0) & 1) * SPL_V_BYTES, wave, lane16); };
    auto dma_c = [&](int kt) {
#pragma unroll
        for (int u = 0; u < NCS; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lptr_t)(Cs + ((kt - kt0) & 1) * CSTAGE + (wave * NCS + u) * 1024), 16,
                                                     coff[u], kt * (SPL_BK * (int)CEL), 0, 0);
    };

    // A wave whose 32 queries ALL lie past the pair's last row has nothing to compute: its accumulators would hold copies of
    // query N-1 in rows no consumer reads (the layer kernels walk ceil(N / 32) tiles, the combine kernel N rows).  N = 5000: waves
    // 5-7 of every pair's last query block = 1.9 % of all waves; N = 10 000: 2.2 %.  Such a wave keeps its share of the K / V
    // loads and every barrier of the protocol below, and skips the matrix and vector work (and its own compat rows): the launch
    // runs at the board's power limit, so MFMAs not executed are time, not only energy.  Wave-uniform, decided
    // once; the waves that compute run exactly the code they ran before.
    if (!PS && !TRACE && !a.compute_all_waves && q_first + wave * 32 >= N) {
        dma_k(kt0);
        if (kt0 + 1 < kt1) dma_k(kt0 + 1);
        dma_v(kt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // the first tile's barrier
        for (int kt = kt0; kt < kt1; ++kt) {
            const int st = (kt - kt0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                         // every wave is past iteration kt - 1: stages st (K) / st ^ 1 (V) are free
            if (kt + 1 < kt1) {                                     // (the last iteration has nothing left to fetch)
#pragma unroll
                for (int slot = NCS; slot < DMA_SLOTS; ++slot) dma_slot(kt, st, slot);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(a.nsplit != 1 && a.part_frag)) __syncthreads();        // the row-order epilogue's barrier
        return;
    }

    // prologue: K, compat of the first two tiles and V of the first in flight, then this lane's Q fragments
    f32x4 ccur[4], cnext[4];                     // CREG: compat of the tile whose logits are formed next / the one after
    dma_k(kt0); dma_c(kt0);
    if (CREG) creg_load(kt0, ccur);
    if (kt0 + 1 < kt1) { dma_k(kt0 + 1); dma_c(kt0 + 1); }
    if (CREG) creg_load(kt0 + 1, cnext);
    dma_v(kt0);
    sp16x8 qh[8], ql[8];
    auto load_q = [&](int pair, int qblock) {
        const int qrow = min(qblock * (NW * 32) + wave * 32 + l31, N - 1);
        const sp16* qsrc = a.qs + ((size_t)pair * NS + qrow) * SPL_Q_LD + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qh[j] = *reinterpret_cast<const sp16x8*>(qsrc + 16 * j);
            ql[j] = *reinterpret_cast<const sp16x8*>(qsrc + PDSC_CHANNELS + 16 * j);
        }
    };
    load_q(b, qb);
    // PS: look one item ahead (descriptors and first tile of the item the run-ahead loads of the last two tiles address)
    auto look_ahead = [&]() {
        more = item + item_step < item_end;
        if (more) {
            const int nx = item + item_step, g2 = nx / a.nq;
            qb_n = nx % a.nq; sp_n = g2 % a.nsplit; b_n = g2 / a.nsplit;
            kt0n = sp_n * per + min(sp_n, rem);
            q_rows_n = min(NW * 32, N - qb_n * (NW * 32));
        }             // (nothing follows: the run-ahead loads keep walking this item's streams -- bounds-checked, never read)
    };
    if constexpr (PS) look_ahead();

    f32x16 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;              // m_run: reference exponent of this query's p values (set by the first tile)
    // PS: the finished item's partials leave in the next item's first loop iteration
    float m_prev = 0.f, l_prev = 0.f;
    size_t slot_prev = 0;
    bool pend = false;
    auto store_pending = [&]() {
        // point-fragment order (see the one-item epilogue below): the accumulator registers are the layer kernel's operands
        float* base = a.part_o + slot_prev * PDSC_CHANNELS + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(base + pf_offset_floats(4 * c + g)) = f32x4{o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
        if (h == 0) {
            a.part_ml[(slot_prev + l31) * 2 + 0] = m_prev;
            a.part_ml[(slot_prev + l31) * 2 + 1] = l_prev;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    };

    // MG: the partial of a finished leaf -> part_o / part_ml [pair][leaf][Npad] in point-fragment order (the accumulator registers of
    // lane (query l31, half h) ARE the 16-byte pieces the fused layer kernel's lane loads: 1 KiB of consecutive memory per store
    // instruction).  The stores of a leaf that ends inside the workgroup's range leave at the top of the next loop iteration (after
    // its barrier, like the persistent form's): a whole tile of time before the next s_waitcnt vmcnt(0) meets them.
    const int q0w = qb * (NW * 32) + wave * 32;                       // first query of this wave inside the pair
    auto leaf_store = [&](int lf, float m_st, float l_st) {
        const size_t slot = ((size_t)b * a.nleaf + lf) * a.Npad + q0w;
        float* base = a.part_o + slot * PDSC_CHANNELS + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(base + pf_offset_floats(4 * c + g)) = f32x4{o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
        if (h == 0) {
            a.part_ml[(slot + l31) * 2 + 0] = m_st;
            a.part_ml[(slot + l31) * 2 + 1] = l_st;
        }
    };
    int leaf_prev = 0;                          // MG: the leaf whose partial is pending (m_prev, l_prev, pend)

    const int koff = l31 * 16 + 512 * h;            // K image (chunk-major): chunk 2j+h of key l31 -> + 1024 j   (immediates)
    const int voff = l31 * 16 + 2048 * h;           // V^T image: chunk 2j+h of channel 32c + l31 -> + 4096 j + 512 c
    const int crow_off = (wave * 32 + l31) * CROW;  // compat row of this lane's query in a compat stage
    const int csw = C16 ? (l31 >> 2) & 3 : ((wave * 32 + l31) >> 1) & 7;
    constexpr float C16_INV = 1.0f / 65535.0f;      // 65535 * fl(1/65535) == 1.0f exactly (and 0 stays 0)
    // unorm16 -> f32 of the 4 keys of group g (registers 4g..4g+3): words 2(g&1), 2(g&1)+1 of 16-B chunk 2h + (g>>1)
    // (two values per v_pk_mul_f32, the scale in a real register pair: packed fp32 with DEFAULT operand selects only -- the
    //  broadcast-select form is not used anywhere in this library, pointdsc_amd/build.py; the pairing is worth 1-2 % of the launch)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 c16_inv2 = {C16_INV, C16_INV};
    asm volatile("" : "+v"(c16_inv2));
    auto c16_group = [&](const unsigned (&w)[8], int g, float (&cc)[4]) {
        const unsigned w0 = w[2 * g], w1 = w[2 * g + 1];
        const f32x2 a = f32x2{(float)(w0 & 0xffffu), (float)(w0 >> 16)} * c16_inv2;
        const f32x2 b = f32x2{(float)(w1 & 0xffffu), (float)(w1 >> 16)} * c16_inv2;
        cc[0] = a[0]; cc[1] = a[1]; cc[2] = b[0]; cc[3] = b[1];
    };
    auto c16_load = [&](const unsigned char* crow, int half, unsigned (&w)[8]) {   // half 0: groups 0,1; half 1: groups 2,3
        const u32x4 v = *reinterpret_cast<const u32x4*>(crow + (((2 * h + half) ^ csw) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) w[4 * half + e] = v[e];
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // logits of one tile relative to the current reference: tl = compat * s - m_run, keys >= N masked
    auto mask_tail = [&](int kt, float (&tl)[16]) {
        if ((kt + 1) * SPL_BK > N) {   // tail tile (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * SPL_BK + (r & 3) + 8 * (r >> 2) + 4 * h;
                tl[r] = key < N ? tl[r] : -INFINITY;
            }
        }
    };
    // maximum over the two lane halves of a query: one v_permlane32_swap (VALU) instead of a ds_bpermute round trip
    auto half_max = [&](float m) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    };
    auto row_max = [&](const float (&tl)[16]) {
        float m = tl[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, tl[r]);
        return half_max(m);
    };

    float tl[16];                                // logits of the tile whose P is formed next
    float mx_next = -INFINITY;
    f32x16 sacc;
    PDSC_TRACE_STAMP(0)                          // 0: prologue issue + Q loads
  for (;;) {                                     // (PS: one pass per item; else exactly one pass)
    // ---- tile kt0: S^T = K Q^T, logits, reference = row maximum ---------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const unsigned char* K = PS ? Ks + (u0 & 1) * SPL_K_BYTES : Ks;
        const unsigned char* C0 = PS ? Cs + (u0 & 1) * CSTAGE : Cs;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const sp16x8 fh = *reinterpret_cast<const sp16x8*>(K + SPL_KH + koff + 1024 * j);
            const sp16x8 fl = *reinterpret_cast<const sp16x8*>(K + SPL_KL + koff + 1024 * j);
            sacc = PDSC_MFMA_IF(2, fl, qh[j], j == 0 ? zero16 : sacc);
            sacc = PDSC_MFMA_IF(4, fh, ql[j], sacc);
            sacc = PDSC_MFMA_X3(fh, qh[j], sacc, 0, 0, 0);
        }
        if (C16) {
            unsigned w[8];
            c16_load(C0 + crow_off, 0, w);
            c16_load(C0 + crow_off, 1, w);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float cc[4];
                c16_group(w, g, cc);
#pragma unroll
                for (int e = 0; e < 4; ++e) tl[4 * g + e] = cc[e] * sacc[4 * g + e];
            }
        } else if (CREG) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) tl[4 * g + e] = ccur[g][e] * sacc[4 * g + e];
                ccur[g] = cnext[g];              // compat of tile kt0 + 1 for the first phase B
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cc = *reinterpret_cast<const f32x4*>(C0 + crow_off + (((2 * g + h) ^ csw) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) tl[4 * g + e] = cc[e] * sacc[4 * g + e];
            }
        }
        mask_tail(kt0, tl);
        m_run = row_max(tl) - ATT_P_BIAS;
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[r] -= m_run;
    }

    PDSC_TRACE_STAMP(1)                          // 1: first tile (wait + QK + logits)
    if (a.prio_mode == 1 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);       // static priority, no per-segment flips
    if (a.prio_mode == 2 && wave < NW / 2) __builtin_amdgcn_s_setprio(1);
    // One iteration of the tile loop.  LAST (one-item form only): the split's last tile has no successor -- no QK^T of a tile
    // kt + 1, no logits, no run-ahead loads: r01-r03 ran the same straight-line body there and threw the 24 MFMAs away (0.6 % of a
    // launch's MFMAs at 32 pairs of N = 5000, 1-2.4 % with the finer key splits of 1-4 pairs); peeled AFTER the loop the tail is
    // straight-line code of its own, the accumulators flow loop -> tail -> epilogue and nothing is copied.
    // kind_c: 0 = an ordinary tile, 1 = LAST (see above).  Leaf form (MG) only: 2 = BOUNDARY, the last tile of a leaf that ends inside
    // the workgroup's range -- tile kt + 1 starts a new leaf, so its logits are formed against a fresh reference (raw products, then
    // their row maximum: bit for bit what the first-tile code above does for a workgroup that STARTS at that leaf) and the finished
    // leaf's state is parked; 3 = the first tile of such a later leaf: the parked partial is stored after the barrier (a whole tile of
    // time before the next s_waitcnt vmcnt(0) meets the stores) and the accumulators restart from zero.  The ordinary iteration is
    // the key-split form's, instruction for instruction: r05c measured +7.5 % per launch when the two leaf cases were run-time
    // branches inside it (profiles/r05_c_ab_leaves_layer_merge.txt, "2 leaves" against "per_launch": same arithmetic).
    auto tile_iteration = [&](const int kt, auto kind_c) __attribute__((always_inline)) {
        constexpr int KIND = (int)decltype(kind_c)::value;
        constexpr bool LAST = KIND == 1, BOUNDARY = KIND == 2, RESTART = KIND == 3;
        static_assert(MG || KIND <= 1, "leaf boundaries exist in the leaf form only");
        const int st = (PS ? u0 + kt - kt0 : kt - kt0) & 1;           // stage of V_kt; K_{kt+1} and compat_{kt+1} live in stage st ^ 1
        const bool has_next = !LAST && kt + 1 < kt1;
        if constexpr (PS) {
            // what this iteration's run-ahead loads address: this item's tiles kt + 2 (K, compat) / kt + 1 (V), or, past its
            // end, the next item's first tiles.  The compat offsets of the lanes follow the next item's row count from the
            // iteration that issues its first compat tile on (this item's last compat tile went out an iteration ago).
            if (more && kt + 2 == kt1) {
                kv_rsrc = PDSC_KV_RSRC(b_n);
                c_rsrc = PDSC_C_RSRC(b_n, qb_n);
                dK = kt0n - kt1 + 2;
                if (q_rows_n != q_rows) set_coff(q_rows_n);
            }
            if (more && kt + 1 == kt1) {
                kv_v = kv_rsrc;
                dV = kt0n - kt1 + 1;
            }
        }
        // K_{kt+1}, compat_{kt+1}, V_kt landed (own LDS-DMA pieces) + everyone finished the previous iteration
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PDSC_TRACE_STAMP(2)                      // 2: wait for own DMA
        __syncthreads();
        PDSC_TRACE_STAMP(3)                      // 3: barrier
        if constexpr (PS) {
            if (pend) { store_pending(); pend = false; }      // (phase A does not touch the accumulators)
        }
        if constexpr (RESTART) {
            leaf_store(leaf_prev, m_prev, l_prev);              // the leaf that ended with the previous tile
            // zero IN PLACE (one asm statement per register, as scale_acc: plain assignments made the register allocator keep two
            // copies of O and move one per tile when this sat behind a run-time branch, r05a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = o[c][r];
                    asm volatile("v_mov_b32 %0, 0" : "+v"(v));
                    o[c][r] = v;
                }
        }
        const float m_sub = BOUNDARY ? 0.f : m_run;
        if (CREG) {
            if (kt != kt0) {                     // (the loads issued one iteration ago have landed: vmcnt(0) above)
#pragma unroll
                for (int g = 0; g < 4; ++g) ccur[g] = cnext[g];
            }
            creg_load(kt + 2, cnext);            // a whole iteration to land
        }
        PDSC_TRACE_STAMP(4)                      // 4: (unused)

        // ---- phase A: S^T(kt+1) = K Q^T on the matrix pipe | P(kt) = exp2(tl), hi/lo split on the VALU ---------
        // (24 MFMAs on one accumulator: dependent fp16 MFMAs issue back to back at full rate, tools/mfma_chain_probe.hip.
        //  Straight-line on purpose: on the last tile the "next" K stage holds stale data and the 24 MFMAs + logits are
        //  wasted work that nothing reads -- cheaper than a second code path, which makes the register allocator keep
        //  copies of the 64 accumulator registers.)
        float psum = 0.f;
        u32x4 phw[2], plw[2];                    // P hi / lo as packed fp16 pairs: word w of operand j = keys (2w, 2w+1) of its 8
        {
            const unsigned char* K = Ks + (st ^ 1) * SPL_K_BYTES;
            // fragments one step ahead of their MFMAs, and the steps pinned in source order: with the chunk-major image every
            // read is `base + immediate`, and left to itself the scheduler hoists all sixteen to the top of the iteration and
            // clusters the MFMAs behind them (measured: +7 % per launch)
            sp16x8 fh = {}, fl = {};
            if constexpr (!LAST) {
                fh = *reinterpret_cast<const sp16x8*>(K + SPL_KH + koff);
                fl = *reinterpret_cast<const sp16x8*>(K + SPL_KL + koff);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sp16x8 nh = fh, nl = fl;
                if constexpr (!LAST) {
                    if (j + 1 < 8) {
                        nh = *reinterpret_cast<const sp16x8*>(K + SPL_KH + koff + 1024 * (j + 1));
                        nl = *reinterpret_cast<const sp16x8*>(K + SPL_KL + koff + 1024 * (j + 1));
                    }
                    sacc = PDSC_MFMA_IF(2, fl, qh[j], j == 0 ? zero16 : sacc);
                    sacc = PDSC_MFMA_IF(4, fh, ql[j], sacc);
                    sacc = PDSC_MFMA_X3(fh, qh[j], sacc, 0, 0, 0);
                    if (j < DMA_SLOTS) dma_slot(kt, st, j);
                }
                {
                    // p values 2j, 2j+1 = one 32-bit word of the P operands: split as a PAIR (split_layout.h split_sp16 arithmetic:
                    // hi = fp16(p), lo = fp16(p - hi), round to nearest even) -- one v_cvt_pk_f16_f32 per plane and pair, where the
                    // element-wise form converted every hi twice (16 of the loop's ~130 vector instructions per tile)
                    const float p0 = __builtin_amdgcn_exp2f(tl[2 * j]), p1 = __builtin_amdgcn_exp2f(tl[2 * j + 1]);
                    psum += p0;
                    psum += p1;
                    unsigned hw, lw;
                    split_sp16_pair(p0, p1, hw, lw);
                    phw[j >> 2][j & 3] = hw;
                    plw[j >> 2][j & 3] = lw;
                }
                fh = nh; fl = nl;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        l_run += psum;
        PDSC_TRACE_STAMP(5)                      // 5: phase A

        // ---- phase B: O^T += V^T P^T (8 steps: channel block c, key half j) | logits of tile kt+1 ---------------
        {
            const unsigned char* V = Vs + st * SPL_V_BYTES;
            const unsigned char* Cn = Cs + (st ^ 1) * CSTAGE + crow_off;
            unsigned cw[8];
            mx_next = -INFINITY;                 // row maximum of tile kt+1's logits, gathered as they are formed
            sp16x8 vh = *reinterpret_cast<const sp16x8*>(V + voff);
            sp16x8 vl = *reinterpret_cast<const sp16x8*>(V + SPL_V_PLANE + voff);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = u >> 1, j = u & 1;
                sp16x8 nvh = vh, nvl = vl;
                if (u + 1 < 8) {
                    const int vo = ((u + 1) >> 1) * 512 + voff + 4096 * ((u + 1) & 1);
                    nvh = *reinterpret_cast<const sp16x8*>(V + vo);
                    nvl = *reinterpret_cast<const sp16x8*>(V + SPL_V_PLANE + vo);
                }
                const sp16x8 phj = __builtin_bit_cast(sp16x8, phw[j]), plj = __builtin_bit_cast(sp16x8, plw[j]);
                o[c] = PDSC_MFMA_IF(2, vl, phj, o[c]);
                o[c] = PDSC_MFMA_IF(1, vh, plj, o[c]);
                o[c] = PDSC_MFMA_X3(vh, phj, o[c], 0, 0, 0);
                if (!LAST && 8 + u < DMA_SLOTS) dma_slot(kt, st, 8 + u);
                if (LAST) {
                    // (no tile kt + 1: no logits to form)
                } else if (C16) {
                    if (u == 0) c16_load(Cn, 0, cw);
                    if (u == 4) c16_load(Cn, 1, cw);
                    if (u & 1) {
                        const int g = u >> 1;
                        float cc[4];
                        c16_group(cw, g, cc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tl[4 * g + e] = fmaf(cc[e], sacc[4 * g + e], -m_sub);
                        mx_next = fmaxf(fmaxf(mx_next, fmaxf(tl[4 * g], tl[4 * g + 1])), fmaxf(tl[4 * g + 2], tl[4 * g + 3]));
                    }
                } else if (CREG) {
                    if (u & 1) {
                        const int g = u >> 1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) tl[4 * g + e] = fmaf(ccur[g][e], sacc[4 * g + e], -m_sub);
                        mx_next = fmaxf(fmaxf(mx_next, fmaxf(tl[4 * g], tl[4 * g + 1])), fmaxf(tl[4 * g + 2], tl[4 * g + 3]));
                    }
                } else if (u & 1) {              // compat chunk 2g+h = keys 8g+4h..+3 = accumulator registers 4g..4g+3
                    const int g = u >> 1;
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(Cn + (((2 * g + h) ^ csw) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) tl[4 * g + e] = fmaf(cc[e], sacc[4 * g + e], -m_sub);
                    mx_next = fmaxf(fmaxf(mx_next, fmaxf(tl[4 * g], tl[4 * g + 1])), fmaxf(tl[4 * g + 2], tl[4 * g + 3]));
                }
                vh = nvh; vl = nvl;
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        PDSC_TRACE_STAMP(6)                      // 6: phase B
        // ---- does tile kt+1 move the reference exponent of any query?  (rare after the first tiles) -------------
        if constexpr (BOUNDARY) {
            // the finished leaf's state waits for the next iteration's barrier (leaf_store); the next leaf starts from nothing
            m_prev = m_run;
            l_prev = l_run + __shfl_xor(l_run, 32, 64);
            leaf_prev = leaf;
            if ((kt + 2) * SPL_BK > N) mask_tail(kt + 1, tl);
            m_run = row_max(tl) - ATT_P_BIAS;
#pragma unroll
            for (int r = 0; r < 16; ++r) tl[r] -= m_run;
            l_run = 0.f;
            ++leaf;
            leaf_end = (leaf + 1) * lper + min(leaf + 1, lrem);
        } else if (has_next) {
            float mloc;
            if ((kt + 2) * SPL_BK > N) {         // tile kt+1 is the ragged last tile of the pair (wave-uniform, once per pair)
                mask_tail(kt + 1, tl);
                mloc = row_max(tl);
            } else
                mloc = half_max(mx_next);
            if (!__all(mloc <= ATT_RESCALE_THR + ATT_P_BIAS)) {
                const float delta = mloc > ATT_RESCALE_THR + ATT_P_BIAS ? mloc - ATT_P_BIAS : 0.f;      // per query; 0 = stays
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run *= alpha;
                m_run += delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) tl[r] -= delta;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = scale_acc(o[c], alpha);
            }
        }
        PDSC_TRACE_STAMP(7)
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    if constexpr (MG) {
        // leaf by leaf (every leaf has at least two tiles: leaf_plan): [restart] ordinary ... ordinary, then boundary or last
        int kt = kt0;
        bool later = false;
        for (;;) {
            const bool final_leaf = leaf_end == kt1;
            const int tail = (final_leaf ? kt1 : leaf_end) - 1;          // the tile the boundary / last iteration handles
            if (later) { tile_iteration(kt, std::integral_constant<int, 3>{}); ++kt; }
            for (; kt < tail; ++kt) tile_iteration(kt, K0{});
            if (final_leaf) { tile_iteration(kt, K1{}); break; }
            tile_iteration(kt, std::integral_constant<int, 2>{});
            ++kt;
            later = true;
        }
    } else if constexpr (PS || !PEEL) {
        for (int kt = kt0; kt < kt1; ++kt) tile_iteration(kt, K0{});
    } else {
        for (int kt = kt0; kt + 1 < kt1; ++kt) tile_iteration(kt, K0{});
        tile_iteration(kt1 - 1, K1{});
    }

    if constexpr (!PS) break;
    else {
        // ---- item done: partials straight from the accumulators (point-fragment order, see below), no LDS, no wait for
        //      the loads in flight -- they carry the next item's first tiles
        // The stores themselves wait until the next item's first loop iteration (store_pending): issued here, the next
        // s_waitcnt vmcnt(0) -- the first tile's -- would sit out their whole round trip.
        l_prev = l_run + __shfl_xor(l_run, 32, 64);
        m_prev = m_run;
        slot_prev = ((size_t)b * a.nsplit + sp) * a.Npad + qb * (NW * 32) + wave * 32;
        if (!more) { store_pending(); break; }
        pend = true;
        u0 += kt1 - kt0;                         // stage parity carries over: the next item's tile 0 sits where tile kt1 would
        item += item_step;
        b = b_n; sp = sp_n; qb = qb_n;
        kt0 = kt0n;
        kt1 = kt0 + per + (sp < rem ? 1 : 0);
        q_rows = q_rows_n;
        dK = 2; dV = 1;                          // (the descriptors switched with the run-ahead loads)
        load_q(b, qb);
        l_run = 0.f;
        look_ahead();
    }
  }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead DMA of the last iterations targets this workgroup's LDS
    if constexpr (PS) return;
    // ---- epilogue: o[c][4g+e] = O^T[channel 32c + 8g + 4h + e][query l31] ---------------------------------
    // In this layout a lane owns 16 bytes of 32 different output rows: a direct store would touch 64 cache lines per
    // instruction.  The K/V stages are dead now, so every wave transposes its 32 x 128 tile through its own 16.5 KiB of
    // LDS (row pitch 528 B) and stores whole rows: one instruction = 2 rows = 8 full lines.
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int q0 = qb * (NW * 32) + wave * 32;
    if constexpr (MG) {
        leaf_store(leaf, m_run, l_tot);          // the last leaf of this workgroup's range (lanes past N: copies of query N-1, the tile's padding)
        return;
    }
    if (a.nsplit != 1 && a.part_frag) {
        // point-fragment order (split_layout.h): the accumulator registers of lane (query l31, half h) ARE the 16-byte
        // pieces the fused layer kernel's lane loads -- 1 KiB of consecutive memory per store instruction, no LDS
        // transposition.  Lanes past N hold copies of query N-1 and fill the padding of the pair's last tile.
        float* base = a.part_o + (((size_t)b * a.nsplit + sp) * a.Npad + q0) * PDSC_CHANNELS + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(base + pf_offset_floats(4 * c + g)) = f32x4{o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
        if (h == 0 && q0 + l31 < a.Npad) {
            const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + q0 + l31;
            a.part_ml[slot * 2 + 0] = m_run;
            a.part_ml[slot * 2 + 1] = l_tot;
        }
        if (TRACE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PDSC_TRACE_STAMP(7)
            if (lane == 0 && a.trace) {
                long long* dst = a.trace + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[k] = tr[k];
            }
        }
        return;
    }
    __syncthreads();                                    // the other waves are done reading K / V
    constexpr int OPITCH = PDSC_CHANNELS * 4 + 16;
    unsigned char* const patch = lds + wave * (32 * OPITCH);
    {
        // un-split: normalised here (o / l, as before); split: the partials stay raw
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]};
                if (a.nsplit == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[c][4 * g + e] / l_tot;
                }
                *reinterpret_cast<f32x4*>(patch + l31 * OPITCH + 128 * c + 32 * g + 16 * h) = v;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        float* const base = a.nsplit == 1 ? a.msg + (size_t)b * NS * PDSC_CHANNELS
                                          : a.part_o + ((size_t)b * a.nsplit + sp) * a.Npad * PDSC_CHANNELS;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = 2 * it + h, piece = l31;
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + r * OPITCH + 16 * piece);
            if (q0 + r < N) *reinterpret_cast<f32x4*>(base + (size_t)(q0 + r) * PDSC_CHANNELS + 4 * piece) = v;
        }
    }
    if (a.nsplit != 1 && h == 0 && q0 + l31 < N) {
        const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + q0 + l31;
        a.part_ml[slot * 2 + 0] = m_run;
        a.part_ml[slot * 2 + 1] = l_tot;
    }
    if (TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PDSC_TRACE_STAMP(7)                      // 7: rescale decisions (accumulated below) + epilogue
        if (lane == 0 && a.trace) {
            long long* dst = a.trace + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = tr[k];
        }
    }
}

// ---- fp32 (q|k|v) rows -> split streams (the layer kernel's head epilogue does this in place; this stand-alone
//      packer serves the stage tests and callers that bring their own projections) ---------------------------
__global__ __launch_bounds__(256) void pack_qkv_split_kernel(const float* __restrict__ qkv, sp16* __restrict__ qs,
                                                             unsigned char* __restrict__ kv, int N, int num_tiles) {
    const int tile = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const float* rows = qkv + (size_t)b * N * 3 * PDSC_CHANNELS;
    unsigned char* img = kv + ((size_t)b * num_tiles + tile) * SPL_TILE_STRIDE;
    const int k0 = tile * SPL_BK;
    // Q: thread -> (row, 4 channels)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i, row = f >> 5, c4 = (f & 31) * 4;
        if (k0 + row < N) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rows + (size_t)(k0 + row) * 3 * PDSC_CHANNELS + c4);
            sp16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { sp16 a, c; split_sp16(v[e], a, c); hi[e] = a; lo[e] = c; }
            sp16* dst = qs + ((size_t)b * N + k0 + row) * SPL_Q_LD + c4;
            *reinterpret_cast<sp16x4*>(dst) = hi;
            *reinterpret_cast<sp16x4*>(dst + PDSC_CHANNELS) = lo;
        }
    }
    // K: thread -> (key, chunk of 8 channels)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i, key = f >> 4, chunk = f & 15;
        sp16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = k0 + key < N ? rows[(size_t)(k0 + key) * 3 * PDSC_CHANNELS + PDSC_CHANNELS + 8 * chunk + e] : 0.f;
            sp16 a, c; split_sp16(v, a, c); hi[e] = a; lo[e] = c;
        }
        *reinterpret_cast<sp16x8*>(img + SPL_KH + spl_k_offset(key, chunk)) = hi;
        *reinterpret_cast<sp16x8*>(img + SPL_KL + spl_k_offset(key, chunk)) = lo;
    }
    // V^T: thread -> (channel, key chunk jh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = t + 256 * i, ch = f & 127, jh = f >> 7;
        sp16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = k0 + spl_v_key(jh, e);
            const float v = key < N ? rows[(size_t)key * 3 * PDSC_CHANNELS + 2 * PDSC_CHANNELS + ch] : 0.f;
            sp16 a, c; split_sp16(v, a, c); hi[e] = a; lo[e] = c;
        }
        *reinterpret_cast<sp16x8*>(img + SPL_VH + spl_v_offset(ch, jh)) = hi;
        *reinterpret_cast<sp16x8*>(img + SPL_VL + spl_v_offset(ch, jh)) = lo;
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
    const int force_nw = env_int("PDSC_ATT_SPLIT_NW", 0), force_ns = env_int("PDSC_ATT_SPLIT_NS", 0);   // tuning/A-B knobs
    *nw_out = (force_nw == 4 || force_nw == 8) ? force_nw : nw;
    *nsplit_out = force_ns > 0 ? (force_ns < tiles ? force_ns : tiles) : best;
}

// ---- leaf form: leaves and plan ----------------------------------------------------------------------------------------
// Canonical leaf count: a function of N ALONE (never of the batch), so that a pair's summation tree -- every leaf from a fresh
// online-softmax state, leaves merged in leaf order by the layer kernel -- and hence its bits do not depend on how many pairs share
// the launch.  At most MERGE_MAX_SPLIT_H3 = 8 (what the layer kernel merges while loading), leaves of at least 4 tiles:
//   N <= 1504 (< 48 tiles): as many leaves as the per-launch planner's finest split for one pair (N = 1000: 8);
//   larger N: 4 -- every leaf partial is 528 B per point that the attention writes and the layer launch reads back, whatever the
//   batch: at 32 pairs of N = 5000 the price of 4 leaves (two per workgroup) is measured in profiles/r05_*ab_leaves*.txt.
int attention_leaf_count(int N) {
    const int t = spl_num_tiles(N);
    return t >= 48 ? 4 : t >= 32 ? 8 : t >= 16 ? 4 : t >= 8 ? 2 : 1;
}

// leaves_mode (pdsc_config.att_leaves): PDSC_LEAVES_CANONICAL = attention_leaf_count(N) leaves, the key split a divisor of it;
// >= 2: that many leaves (tuning, at most PDSC_ATT_MAX_LEAVES).  (PDSC_LEAVES_PER_LAUNCH does not come here: it is the key-split form.)
void leaf_plan(int bs, int N, int leaves_mode, int* nw_out, int* nsplit_out, int* nleaf_out) {
    int nw, ns;
    split_plan(bs, N, &nw, &ns);
    const int tiles = spl_num_tiles(N);
    int C = leaves_mode == PDSC_LEAVES_CANONICAL ? attention_leaf_count(N) : leaves_mode;
    if (C > tiles / 2) C = tiles / 2;            // every leaf at least two tiles (the kernel's boundary and restart iterations are distinct)
    if (C > PDSC_ATT_MAX_LEAVES) C = PDSC_ATT_MAX_LEAVES;
    if (C < 1) C = 1;
    // the per-launch cost model over the divisors of C; a leaf that ends inside a workgroup's range costs about a third of a tile
    const int nq = ceil_div(N, nw * 32), slots = nw == 8 ? 256 : 512;
    int best = 1;
    double best_cost = 1e30;
    for (int d = 1; d <= C; ++d) {
        if (C % d) continue;
        const int wgs = nq * bs * d, rounds = ceil_div(wgs, slots);
        double cost = (double)rounds * (ceil_div(tiles, d) + 4.0 + 0.35 * (C / d));
        if (((d * bs) & 7) != 0) cost *= 1.03;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = d; }
    }
    *nw_out = nw; *nsplit_out = best; *nleaf_out = C;
}

}  // namespace pdsc

using namespace pdsc;

extern "C" int pdsc_attention_leaf_count(int N) { return N > 0 ? attention_leaf_count(N) : -1; }

extern "C" int pdsc_attention_leaf_plan(int bs, int N, int leaves_mode, int* nsplit, int* nleaf) {
    PDSC_REQUIRE(bs > 0 && N > 0 && leaves_mode >= PDSC_LEAVES_CANONICAL && leaves_mode <= PDSC_ATT_MAX_LEAVES && nsplit && nleaf,
                 "pdsc_attention_leaf_plan: bs=%d N=%d leaves_mode=%d", bs, N, leaves_mode);
    int nw;
    leaf_plan(bs, N, leaves_mode, &nw, nsplit, nleaf);
    return PDSC_OK;
}

extern "C" size_t pdsc_attention_leaf_scratch_bytes(int bs, int N, int leaves_mode) {
    if (bs <= 0 || N <= 0 || leaves_mode < PDSC_LEAVES_CANONICAL) return 0;
    int nw, ns, C;
    leaf_plan(bs, N, leaves_mode, &nw, &ns, &C);
    return (size_t)bs * C * round_up(N, 256) * (PDSC_CHANNELS + 2) * sizeof(float);
}

// One attention launch in the leaf form: C leaf partials per query in scratch ([bs][C][Npad][128] then [bs][C][Npad][2], point-
// fragment order) for the H3 layer kernel to merge (C <= MERGE_MAX_SPLIT_H3).
int pdsc::launch_attention_leaves(const void* q_split, const void* kv_tiles, const void* compat, int compat_format, long long ld,
                                  void* scratch, size_t scratch_bytes, int bs, int N, int leaves_mode, const int* nvalid, int n_min,
                                  hipStream_t st) {
    PDSC_REQUIRE(q_split && kv_tiles && compat && scratch, "pdsc_sc_attention_leaves: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_sc_attention_leaves: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(compat_format == PDSC_COMPAT_F32 || compat_format == PDSC_COMPAT_U16, "pdsc_sc_attention_leaves: compat_format=%d", compat_format);
    const bool c16 = compat_format == PDSC_COMPAT_U16;
    PDSC_REQUIRE(ld >= round_up(N, SPL_BK) && ld % (c16 ? 8 : 4) == 0,
                 "pdsc_sc_attention_leaves: ld=%lld must be a multiple of %d and >= N rounded up to 32", ld, c16 ? 8 : 4);
    int nw, ns, C;
    leaf_plan(bs, N, leaves_mode, &nw, &ns, &C);
    // ragged batches: every pair cuts ITS OWN tiles into C leaves -- the shortest pair needs at least C of them
    PDSC_REQUIRE(!nvalid || (n_min + 31) / 32 >= 2 * C, "pdsc_sc_attention_leaves: the shortest pair (%d correspondences) has fewer than two "
                 "32-key tiles for each of the %d leaves planned for bs=%d, N=%d", n_min, C, bs, N);
    const size_t need = (size_t)bs * C * round_up(N, 256) * (PDSC_CHANNELS + 2) * sizeof(float);
    if (scratch_bytes < need) {
        set_error("pdsc_sc_attention_leaves: scratch %zu < %zu bytes", scratch_bytes, need);
        return PDSC_ERR_WORKSPACE;
    }
    const int tiles = spl_num_tiles(N);
    AttSplitArgs a{};
    a.qs = (const sp16*)q_split; a.kv = (const unsigned char*)kv_tiles; a.compat = compat; a.ld = ld; a.msg = nullptr;
    a.N = N; a.Npad = (int)round_up(N, 256); a.nsplit = ns; a.num_tiles = tiles; a.bs = bs;
    a.nq = ceil_div(N, nw * 32);
    a.nleaf = C;
    a.part_o = (float*)scratch;
    a.part_ml = a.part_o + (size_t)bs * C * a.Npad * PDSC_CHANNELS;
    a.nvalid = nvalid;
    a.part_frag = 1;
    a.compat_nt = c16 ? 0 : 1;
    const size_t lds_bytes = 2 * (size_t)(SPL_TILE_BYTES + nw * 32 * (c16 ? 64 : 128));
    const unsigned grid = (unsigned)(a.nq * ns * bs);
    a.items = (int)grid;
    int rc = PDSC_OK;
#define PDSC_ATT_LAUNCH_MG(NWV, CMV)                                                                                                        \
    do {                                                                                                                                    \
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_split_kernel<NWV, CMV, false, false, true, true>), lds_bytes,   \
                                "pdsc_sc_attention_leaves(dynamic LDS)");                                                                   \
        if (rc != PDSC_OK) return rc;                                                                                                       \
        profile_mark_begin(PDSC_PROF_ATTENTION, st);                                                                                        \
        hipLaunchKernelGGL((sc_attention_split_kernel<NWV, CMV, false, false, true, true>), dim3(grid), dim3(NWV * 64), lds_bytes, st, a);  \
        profile_mark_end(PDSC_PROF_ATTENTION, st);                                                                                          \
    } while (0)
    if (nw == 8 && c16) PDSC_ATT_LAUNCH_MG(8, 1);
    else if (nw == 8) PDSC_ATT_LAUNCH_MG(8, 0);
    else if (c16) PDSC_ATT_LAUNCH_MG(4, 1);
    else PDSC_ATT_LAUNCH_MG(4, 0);
#undef PDSC_ATT_LAUNCH_MG
    return check_launch("pdsc_sc_attention_leaves");
}

extern "C" size_t pdsc_split_q_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * N * SPL_Q_LD * sizeof(sp16);
}
extern "C" size_t pdsc_split_kv_bytes(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    return (size_t)bs * spl_num_tiles(N) * SPL_TILE_STRIDE;
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
    hipLaunchKernelGGL(pack_qkv_split_kernel, dim3(tiles, bs), dim3(256), 0, (hipStream_t)stream, qkv, (sp16*)q_split,
                       (unsigned char*)kv_tiles, N, tiles);
    return check_launch("pdsc_pack_qkv_split");
}

static long long* g_att_trace = nullptr;
extern "C" int pdsc_attention_trace(long long* device_buffer) {   // diagnostics: see include/pointdsc_hip.h
    g_att_trace = device_buffer;
    return PDSC_OK;
}

static int launch_attention_split(const void* q_split, const void* kv_tiles, const void* compat, bool c16, long long ld,
                                  float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit, void* stream,
                                  int partial_layout = PDSC_PARTIALS_ROWS, const int* nvalid = nullptr) {
    PDSC_REQUIRE(q_split && kv_tiles && compat, "pdsc_sc_attention_split: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_sc_attention_split: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= round_up(N, SPL_BK) && ld % (c16 ? 8 : 4) == 0,
                 "pdsc_sc_attention_split: ld=%lld must be a multiple of %d and >= N rounded up to 32", ld, c16 ? 8 : 4);
    const int tiles = spl_num_tiles(N);
    int nw, ns;
    split_plan(bs, N, &nw, &ns);
    if (nsplit <= 0) nsplit = ns;
    if (nsplit > tiles) nsplit = tiles;
    PDSC_REQUIRE(msg || nsplit > 1, "pdsc_sc_attention_split: msg == NULL needs a key split > 1 (partials stay in scratch)");
    const size_t need = nsplit == 1 ? 0 : (size_t)bs * nsplit * round_up(N, 256) * (PDSC_CHANNELS + 2) * sizeof(float);
    if (need > 0 && (!scratch || scratch_bytes < need)) {
        set_error("pdsc_sc_attention_split: scratch %zu < %zu bytes", scratch_bytes, need);
        return PDSC_ERR_WORKSPACE;
    }
    AttSplitArgs a{};
    a.qs = (const sp16*)q_split; a.kv = (const unsigned char*)kv_tiles; a.compat = compat; a.ld = ld; a.msg = msg;
    a.N = N; a.Npad = (int)round_up(N, 256); a.nsplit = nsplit; a.num_tiles = tiles; a.bs = bs;
    a.nq = ceil_div(N, nw * 32);
    a.part_o = (float*)scratch;
    a.part_ml = a.part_o ? a.part_o + (size_t)bs * nsplit * a.Npad * PDSC_CHANNELS : nullptr;
    a.trace = g_att_trace;
    a.nvalid = nvalid;
    a.prio_mode = env_int("PDSC_ATT_PRIO", 0);
    a.compute_all_waves = env_int("PDSC_ATT_ALL_WAVES", 0);
    a.compat_nt = c16 ? 0 : 1;                 // (the wide variant still takes it as an argument)
    PDSC_REQUIRE(partial_layout == PDSC_PARTIALS_ROWS || partial_layout == PDSC_PARTIALS_PF, "pdsc_sc_attention_split: partial_layout=%d", partial_layout);
    PDSC_REQUIRE(partial_layout == PDSC_PARTIALS_ROWS || (!msg && nsplit > 1), "pdsc_sc_attention_split: point-fragment partials are not merged here (msg must be NULL, key split > 1)");
    a.part_frag = partial_layout == PDSC_PARTIALS_PF;
    hipStream_t st = (hipStream_t)stream;
    // 2 stages x (K 16 KiB + V 16 KiB + compat of the workgroup's nw*32 queries: 128 B (fp32) or 64 B (unorm16) per row)
    // ... and at least the epilogue's transposition patches (one 32 x 132-float patch per wave)
    // A/B knob PDSC_ATT_CREG = 1: fp32 compat values straight into registers (no LDS stage)
    const bool creg = !c16 && env_int("PDSC_ATT_CREG", 0) != 0;
    const size_t stage_bytes = 2 * (size_t)(SPL_TILE_BYTES + (creg ? 0 : nw * 32 * (c16 ? 64 : 128)));
    const size_t patch_bytes = (size_t)nw * 32 * (PDSC_CHANNELS * 4 + 16);
    // (point-fragment partials leave straight from the accumulators: no transposition patches, so the workgroup asks for its
    //  two stages only -- 96 KiB with the unorm16 matrix -- and leaves the rest of the CU's 160 KiB to other kernels)
    const size_t lds_bytes = (a.part_frag || stage_bytes > patch_bytes) ? stage_bytes : patch_bytes;
    const unsigned grid = (unsigned)(a.nq * nsplit * bs);
    int rc = PDSC_OK;
    const bool trace = nw == 8 && a.trace;
    (void)trace;
    a.items = (int)grid;
#ifdef PDSC_EXPERIMENTS
    // A/B knob PDSC_ATT_PERSIST = 1: one workgroup per CU walking its items (point-fragment partials, 8-wave plan, whole
    // multiples of 8 items, at least two per workgroup)
    const bool persist = nw == 8 && !creg && !trace && !nvalid && a.part_frag && (grid & 7) == 0 && grid >= 512 && env_int("PDSC_ATT_PERSIST", 0) != 0;
    if (persist) {
        unsigned pgrid = (unsigned)env_int("PDSC_ATT_PERSIST_GRID", 256) & ~7u;      // A/B knob: workgroups (multiple of 8)
        if (pgrid < 8 || pgrid > grid) pgrid = 256;
        if (c16) {
            rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_split_kernel<8, 1, false, true>), lds_bytes, "pdsc_sc_attention_split(dynamic LDS)");
            if (rc != PDSC_OK) return rc;
            profile_mark_begin(PDSC_PROF_ATTENTION, st);
            hipLaunchKernelGGL((sc_attention_split_kernel<8, 1, false, true>), dim3(pgrid), dim3(512), lds_bytes, st, a);
        } else {
            rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_split_kernel<8, 0, false, true>), lds_bytes, "pdsc_sc_attention_split(dynamic LDS)");
            if (rc != PDSC_OK) return rc;
            profile_mark_begin(PDSC_PROF_ATTENTION, st);
            hipLaunchKernelGGL((sc_attention_split_kernel<8, 0, false, true>), dim3(pgrid), dim3(512), lds_bytes, st, a);
        }
        profile_mark_end(PDSC_PROF_ATTENTION, st);
        return check_launch("pdsc_sc_attention_split(persistent)");
    }
#endif
#define PDSC_ATT_LAUNCH(NWV, CMV, TRV)                                                                                    \
    do {                                                                                                                    \
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_split_kernel<NWV, CMV, TRV>), lds_bytes,        \
                                "pdsc_sc_attention_split(dynamic LDS)");                                                    \
        if (rc != PDSC_OK) return rc;                                                                                       \
        profile_mark_begin(PDSC_PROF_ATTENTION, st);                                                                        \
        hipLaunchKernelGGL((sc_attention_split_kernel<NWV, CMV, TRV>), dim3(grid), dim3(NWV * 64), lds_bytes, st, a);       \
        profile_mark_end(PDSC_PROF_ATTENTION, st);                                                                          \
    } while (0)
#ifdef PDSC_EXPERIMENTS
    if (trace && c16) PDSC_ATT_LAUNCH(8, 1, true);
    else if (trace) PDSC_ATT_LAUNCH(8, 0, true);
    else
#endif
#ifdef PDSC_EXPERIMENTS
    if (nw == 8 && c16 && env_int("PDSC_ATT_PEEL", 1) == 0) {
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_split_kernel<8, 1, false, false, false>), lds_bytes, "pdsc_sc_attention_split(dynamic LDS)");
        if (rc != PDSC_OK) return rc;
        profile_mark_begin(PDSC_PROF_ATTENTION, st);
        hipLaunchKernelGGL((sc_attention_split_kernel<8, 1, false, false, false>), dim3(grid), dim3(512), lds_bytes, st, a);
        profile_mark_end(PDSC_PROF_ATTENTION, st);
    } else
#endif
    if (nw == 8 && c16) PDSC_ATT_LAUNCH(8, 1, false);
#ifdef PDSC_EXPERIMENTS
    else if (nw == 8 && creg) PDSC_ATT_LAUNCH(8, 2, false);
    else if (nw != 8 && !c16 && creg) PDSC_ATT_LAUNCH(4, 2, false);
#endif
    else if (nw == 8) PDSC_ATT_LAUNCH(8, 0, false);
    else if (c16) PDSC_ATT_LAUNCH(4, 1, false);
    else PDSC_ATT_LAUNCH(4, 0, false);
#undef PDSC_ATT_LAUNCH
    rc = check_launch("pdsc_sc_attention_split");
    if (rc != PDSC_OK) return rc;
    if (nsplit > 1 && msg) {
        AttArgs c{};
        c.msg = msg; c.part_o = a.part_o; c.part_ml = a.part_ml;
        c.N = N; c.Npad = a.Npad; c.nsplit = nsplit; c.num_tiles = tiles;
        rc = launch_attention_combine(c, bs, st);
    }
    return rc;
}

extern "C" int pdsc_sc_attention_split(const void* q_split, const void* kv_tiles, const float* compat, long long ld,
                                       float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                                       void* stream) {
    return launch_attention_split(q_split, kv_tiles, compat, false, ld, msg, scratch, scratch_bytes, bs, N, nsplit, stream);
}

int pdsc::launch_attention_split_ex(const void* q_split, const void* kv_tiles, const void* compat, int compat_format, long long ld,
                                    float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit, int partial_layout,
                                    const int* nvalid, hipStream_t st) {
    PDSC_REQUIRE(compat_format == PDSC_COMPAT_F32 || compat_format == PDSC_COMPAT_U16, "pdsc_sc_attention_split: compat_format=%d", compat_format);
    return launch_attention_split(q_split, kv_tiles, compat, compat_format == PDSC_COMPAT_U16, ld, msg, scratch, scratch_bytes, bs, N,
                                  nsplit, st, partial_layout, nvalid);
}

extern "C" int pdsc_sc_attention_split_partials(const void* q_split, const void* kv_tiles, const void* compat, int compat_format,
                                                long long ld, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                                                int partial_layout, void* stream) {
    return pdsc::launch_attention_split_ex(q_split, kv_tiles, compat, compat_format, ld, nullptr, scratch, scratch_bytes, bs, N, nsplit,
                                           partial_layout, nullptr, (hipStream_t)stream);
}

extern "C" int pdsc_sc_attention_split_u16(const void* q_split, const void* kv_tiles, const unsigned short* compat_u16,
                                           long long ld, float* msg, void* scratch, size_t scratch_bytes, int bs, int N,
                                           int nsplit, void* stream) {
    return launch_attention_split(q_split, kv_tiles, compat_u16, true, ld, msg, scratch, scratch_bytes, bs, N, nsplit, stream);
}
