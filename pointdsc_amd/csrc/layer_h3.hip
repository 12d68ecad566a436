// a-2 fused, wavefront-resident, every GEMM on the 16-bit matrix cores: the chain of layer_wave.hip
//   tail of layer i   : feat  = featB + fc3( relu(fc2'( relu(fc1'(msg)) )) )         (reference models/PointDSC.py:43-45)
//   head of layer i+1 : featB = relu(pcn'(feat)) ; (q|k|v) = Wqkv featB + b            (models/PointDSC.py:75, :36-38)
// for one 32-point tile per wavefront with all activations in registers, but with fc1..fc3 and PointCN in the H3
// arithmetic (fp16 hi + scaled-lo operands, three v_mfma_f32_32x32x16_f16 per operand pair, ~2^-21 per product:
// layer_wave.h) instead of v_mfma_f32_32x32x2_f32 -- 504 MFMAs x 32 = 16.1 k matrix-pipe cycles per tile instead of
// 46 k -- and with the software pipeline that only pays off once the GEMMs are that short:
//   * the epilogue of output tile t (relu / residual / operand split for the next GEMM, staging through the wave's LDS
//     patch, global stores) is cut into 8 steps that are issued BETWEEN the MFMA groups of tile t+1, so the vector, LDS
//     and store work of one tile runs in the shadow of the next tile's matrix work.  Only the last tile of a stage,
//     whose result the next stage's first MFMA needs, keeps its epilogue in line (5 of 24 tiles);
//   * LDS round trips never stall the wave: a step reads the patch into a staging register and the NEXT step stores it.
// Weights come as the PDSC_LAYER_GEMM_H3 fragment streams (pdsc_wfrag_build_*_fmt), one 8 KiB chunk (12 MFMAs) ahead of
// use in two register buffers; eight wavefronts per CU each stream the whole 344 KiB per tile through the CU's vector L1.
// Bound at 32 pairs: HBM (3.76 KB per point) and that L1 stream (64 B/clk/CU: 1024 clocks per chunk for 8 wavefronts).
#include <stdlib.h>
#include <type_traits>
#include "pdsc_common.h"
#include "split_layout.h"
#include "merge_partials.h"
#include "layer_args.h"
#include "layer_wave.h"
#include "ragged.h"

namespace pdsc {

// launches with at most this many 32-point tiles take the four-wavefronts-per-tile kernel (layer_coop.hip).  Measured, whole
// forward, interleaved A/B (profiles/r03_j / r03_k / r03_l_ab_coop.txt): 32 tiles (N=1000 x 1) -25 %, 157 (N=5000 x 1) -11 %,
// 313-314 (1 pair of N=10000, 2 of N=5000) -3 ... -5 %, 625 -0.5 %, 1250 -1.5 %, 2500 (16 pairs) -0.9 %, 5000 (32 pairs) +0.9 %
constexpr int PDSC_H3_COOP_TILES = 2560;

#define LH_STAMP(k) \
    if (TRACE && lane == 0) a.trace[(size_t)gw * 64 + (k)] = __builtin_readcyclecounter();

// tiles per stage and whether `d` is the last output tile of its stage (its epilogue feeds the next stage's operands)
constexpr int stage_tiles(int stage) { return stage == ST_FC1 || stage == ST_FC2 ? 2 : stage == ST_FC3 || stage == ST_PCN ? 4 : 12; }

// Contract (launch_layer_h3_fits): head => the split streams are the only q|k|v output (qs, kv given, qkv_out NULL);
// tail + head => feat_out NULL; tail only => feat_out given.  With that the chunk loop has no branch at all: every chunk is
// one basic block the scheduler can interleave freely.  Anything else takes layer_wave.hip's kernel (same arithmetic).
// PIPE: vector instructions the scheduler is asked to place after every MFMA of a chunk (sched_group_barrier; 0 = its own choice)
// EXP (diagnostics, wrong results; instantiated only with -DPDSC_LAYER_DIAG): knock-outs that tell which unit bounds the
// launch -- 1: weights loaded once, 2: no global stores (8 / 16 / 32 / 64: no Q / K / V / featB stores), 4: no partial /
// residual loads.  profiles/r02_j_layer_knockout*.txt: all three off = 30 us of 181; loads 69, stores 54, weights 21.
// FB_PF: featB leaves in point-fragment order (split_layout.h) instead of rows
// NWV / NBUF: wavefronts per workgroup and depth of the weight-chunk ring.  The product launches (4, 2) only: several
// wavefronts per SIMD hide each other's L2 round trips, which needs more tiles than the chip has SIMDs.  Launches of at most
// PDSC_H3_COOP_TILES tiles go to layer_coop.hip instead (a lone wavefront per tile is one 42-chunk dependency chain, 28 us
// whatever the tile count); the r03 small-launch shapes (1-2 waves, ring of 3-4: -5 % at 313-625 tiles, superseded by the
// four-wavefront kernel) are instantiated in experiments builds only (A/B knob PDSC_LAYER_H3_SHAPE).
template <bool T, bool H, bool TRACE = false, int PIPE = 6, int EXP = 0, bool FB_PF = false, int NWV = LW_WAVES, int NBUF = 2>
__global__ __launch_bounds__(64 * NWV, NBUF > 2 ? 1 : 2) void layer_h3_kernel(LayerArgs a) {
    __shared__ __attribute__((aligned(16))) float Vs_all[NWV][32 * LW_VLD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tpp = ceil_div_dev(a.N, 32);                          // tiles per pair
    const int gw = blockIdx.x * NWV + wave;                         // one wave = one tile
    if (gw >= a.bs * tpp) return;                                    // (no workgroup barriers anywhere below)
    const int b = gw / tpp, tile = gw - b * tpp;
    const int m0 = b * a.N + tile * 32;
    const int valid = min(32, (a.nvalid ? a.nvalid[b] : a.N) - tile * 32);
    if (valid <= 0) return;                                          // (ragged batches: tile past the pair's own rows)
    const bool live = l31 < valid;
    const size_t row = (size_t)m0 + min(l31, valid - 1);
    float* Vs = Vs_all[wave];
    unsigned char* patch = reinterpret_cast<unsigned char*>(Vs);     // the wave's 32 x 144 B staging patch, as bytes

    if (a.stagger_cycles > 0 && blockIdx.x < 512) {
        // every phase of this kernel leans on a different unit (partials: HBM reads; fc / pcn chunks: the L1 weight stream;
        // q|k|v: HBM writes) and the wavefronts of the first round start together, so they walk the phases in lockstep.
        // Delaying half of them puts one group's memory phases beside the other's matrix phases.
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        const bool late = a.stagger_mode == 1 ? (hwid & 1u) : a.stagger_mode == 2 ? (blockIdx.x & 1) : ((blockIdx.x >> 8) & 1);
        if (late) {
            const long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < a.stagger_cycles) __builtin_amdgcn_s_sleep(32);
        }
    }
    LH_STAMP(0)
    float rmax = 0.f;            // fp16 range sentinel: largest |activation| this lane converts to an fp16 hi / lo pair
    constexpr int NCH = num_chunks<T, H>();
    WChunk w[NBUF];
    static_for<0, NBUF - 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NCH) load_chunk<T, true, true>(w[j], a, j, lane);
    });

    // B operands (k-step kk: channels 16kk + 8h .. +7 of this lane's point) of fc1 | fc2 | fc3 | pcn (H3) and q|k|v (fp16)
    u32x4 a0h[8], a0l[8], a1h[4], a1l[4], a2h[4], a2l[4], ayh[8], ayl[8], xqh[8], xql[8];
    f32x4 y3[16];                                                    // residual rows, then feat (fp32)
    if (T) {
        f32x4 x0[16];
        if (a.msg) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x0[q] = *reinterpret_cast<const f32x4*>(a.msg + row * PDSC_CHANNELS + 8 * q + 4 * h);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) make_kstep<true>(x0[2 * kk], x0[2 * kk + 1], a0h[kk], a0l[kk], rmax);
        } else {
            // merge of the attention's key-split partials, the arithmetic of merge_partials_finish (merge_partials.h)
            auto run = [&](auto ns_tag) {
                constexpr int NS = decltype(ns_tag)::value;
                constexpr int GQ = NS <= 2 ? 16 : NS <= 4 ? 8 : 4;   // k-steps per batch of loads (up to 128 registers in flight)
                const size_t slot0 = (size_t)b * NS * a.Npad + (row - (size_t)b * a.N);
                // element q of split sp: rows order = row `slot`, floats 8q + 4h; point-fragment order = tile base + 256 q + 4 lane
                const bool pf = a.io_flags & PDSC_IO_PARTIALS_PF;
                const size_t e0 = pf ? ((size_t)b * NS * a.Npad + (size_t)tile * 32) * PDSC_CHANNELS + lane * 4 : slot0 * PDSC_CHANNELS + 4 * h;
                const int eq = pf ? 256 : 8;
                float wsp[NS], ls[NS];
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    const float2 ml = *reinterpret_cast<const float2*>(a.part_ml + (slot0 + (size_t)sp * a.Npad) * 2);
                    wsp[sp] = ml.x; ls[sp] = ml.y;
                }
                float mmax = wsp[0];
#pragma unroll
                for (int sp = 1; sp < NS; ++sp) mmax = fmaxf(mmax, wsp[sp]);
                float den = 0.f;
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    wsp[sp] = __builtin_amdgcn_exp2f(wsp[sp] - mmax);
                    den = fmaf(ls[sp], wsp[sp], den);
                }
                const float rden = 1.0f / den;
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += GQ) {
                    f32x4 pv[GQ][NS];
#pragma unroll
                    for (int q = 0; q < GQ; ++q)
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
                            pv[q][sp] = (EXP & 4) ? f32x4{1.f, 2.f, 3.f, (float)lane}
                                                  : *reinterpret_cast<const f32x4*>(a.part_o + e0 + (size_t)sp * a.Npad * PDSC_CHANNELS + eq * (q0 + q));
#pragma unroll
                    for (int q = 0; q < GQ; ++q) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] = fmaf(pv[q][sp][e], wsp[sp], acc[e]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x0[q0 + q][e] = acc[e] * rden;
                            asm volatile("" : "+v"(x0[q0 + q][e]));     // materialise here (else the compiler sinks the arithmetic
                        }                                               // to the first MFMA and keeps every batch of loads live)
                    }
#pragma unroll
                    for (int kk = q0 / 2; kk < (q0 + GQ) / 2; ++kk) make_kstep<true>(x0[2 * kk], x0[2 * kk + 1], a0h[kk], a0l[kk], rmax);
                    __builtin_amdgcn_sched_barrier(0);               // keep the next batch's loads behind this batch's use
                }
            };
            switch (a.nsplit) {
                case 1: run(std::integral_constant<int, 1>{}); break;
                case 2: run(std::integral_constant<int, 2>{}); break;
                case 3: run(std::integral_constant<int, 3>{}); break;
                case 4: run(std::integral_constant<int, 4>{}); break;
                case 5: run(std::integral_constant<int, 5>{}); break;
                case 6: run(std::integral_constant<int, 6>{}); break;
                case 7: run(std::integral_constant<int, 7>{}); break;
                default: run(std::integral_constant<int, 8>{}); break;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) y3[q] = *reinterpret_cast<const f32x4*>(a.feat_in + row * PDSC_CHANNELS + 8 * q + 4 * h);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) make_kstep<true>(y3[2 * kk], y3[2 * kk + 1], ayh[kk], ayl[kk], rmax);
    }
    LH_STAMP(1)

    unsigned char* img = H ? a.kv + (size_t)gw * SPL_TILE_STRIDE : nullptr;
    f32x16 acc, cross;
    f32x4 vp[4];                 // values of the output tile whose epilogue is pending
    u32x4 ev = {0u, 0u, 0u, 0u}; // patch -> global staging register of the epilogue pipeline
    float vt[8];                 // ... of the V^T transposition (8 scalar reads per store pair)

    // ---- one step (0..7) of the epilogue of output tile `d` on its values v (lane = point l31, v[g][e] = channel n0 + 8g + 4h + e)
    auto epi = [&](auto stage_c, auto tile_c, auto sc, f32x4 (&v)[4]) {
        constexpr ChunkDesc d = {decltype(stage_c)::value, decltype(tile_c)::value, 0, 0};
        constexpr int s = decltype(sc)::value;
        constexpr int n0 = 32 * d.tile;
        if constexpr (d.stage == ST_FC1 || d.stage == ST_FC2) {
            // relu, then straight into the next GEMM's k-steps 2*tile, 2*tile + 1 (this tile's 32 channels)
            if constexpr (s < 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[s][e] = fmaxf(v[s][e], 0.f);
                if constexpr (s & 1)
                    make_kstep<true>(v[s - 1], v[s], (d.stage == ST_FC1 ? a1h : a2h)[2 * d.tile + (s >> 1)],
                                     (d.stage == ST_FC1 ? a1l : a2l)[2 * d.tile + (s >> 1)], rmax);
            }
        } else if constexpr (d.stage == ST_FC3) {
            if constexpr (s < 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y3[4 * d.tile + s][e] = y3[4 * d.tile + s][e] + v[s][e];
                // tail-only launches return feat; lanes beyond the pair's last point hold copies of its last row (their
                // inputs were loaded from it), so their stores repeat that row's bytes: no predicate, no branch
                if constexpr (!H && !(EXP & 2)) *reinterpret_cast<f32x4*>(a.feat_out + row * PDSC_CHANNELS + n0 + 8 * s + 4 * h) = y3[4 * d.tile + s];
                if constexpr (H && (s & 1))
                    make_kstep<true>(y3[4 * d.tile + s - 1], y3[4 * d.tile + s], ayh[2 * d.tile + (s >> 1)], ayl[2 * d.tile + (s >> 1)], rmax);
            }
        } else if constexpr (d.stage == ST_PCN) {
            // featB = relu: fp32 rows leave through the patch as whole 128-byte lines; fp16 hi / lo operands of q|k|v
            if constexpr (s < 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[s][e] = fmaxf(v[s][e], 0.f);
                if constexpr (FB_PF) {
                    // point-fragment order: this lane's registers are the next launch's residual registers (1 KiB per instruction)
                    if constexpr (!(EXP & (2 | 64))) *reinterpret_cast<f32x4*>(a.featB_out + (size_t)gw * PF_TILE_FLOATS + pf_offset_floats(4 * d.tile + s) + lane * 4) = v[s];
                } else
                    *reinterpret_cast<f32x4*>(patch + l31 * LW_PROW + 32 * s + 16 * h) = v[s];
            }
            if constexpr (!FB_PF) {
                if constexpr (s == 3) wave_lds_sync();
                if constexpr (s >= 4) {           // 8 points x 128 B per store instruction
                    const int pt = min(8 * (s - 4) + (lane >> 3), valid - 1), piece = lane & 7;     // (rows >= valid: copies of the last row)
                    if constexpr (!(EXP & 2)) *reinterpret_cast<u32x4*>(a.featB_out + ((size_t)m0 + pt) * PDSC_CHANNELS + n0 + 4 * piece) = ev;
                    else asm volatile("" :: "v"(ev));
                }
                if constexpr (s >= 3 && s < 7) {
                    const int pt = 8 * (s - 3) + (lane >> 3), piece = lane & 7;
                    ev = *reinterpret_cast<const u32x4*>(patch + pt * LW_PROW + 16 * piece);
                }
            }
            if constexpr (s == 4 || s == 5)
                make_kstep<false>(v[2 * (s - 4)], v[2 * (s - 4) + 1], xqh[2 * d.tile + (s - 4)], xql[2 * d.tile + (s - 4)], rmax);
        } else {
            // q | k | v, output tile d.tile of 12: tiles 0..3 = q, 4..7 = k, 8..11 = v
            if constexpr (s < 4) range_note(rmax, v[s]);
            {
                if constexpr (d.tile >= 4 && d.tile < 8) {
                    // K image, chunk-major (split_layout.h): after the half swap lane (key l31, half h) holds chunk 4(t-4)+s of its
                    // key for the hi (h = 0) / lo (h = 1) plane -- the 32 lanes of a half store 512 consecutive, aligned bytes
                    if constexpr (s < 4) {
                        f32x4 z = v[s];
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[e] = live ? z[e] : 0.f;                    // keys beyond N are zero
                        unsigned hi[2], lo[2];
                        split4(z, hi, lo);
                        const u32x4 ck = chunk_for_store(hi, lo);
                        if constexpr (!(EXP & (2 | 16))) *reinterpret_cast<u32x4*>(img + (h ? SPL_KL : SPL_KH) + spl_k_offset(l31, 4 * (d.tile - 4) + s)) = ck;
                        else asm volatile("" :: "v"(ck));
                    }
                } else if constexpr (d.tile < 4) {
                    // Q rows (hi[128] | lo[128]) fp16 through the patch: row = (hi 64 B | lo 64 B) of this tile's 32 channels
                    if constexpr (s < 4) {
                        unsigned hi[2], lo[2];
                        split4(v[s], hi, lo);
                        *reinterpret_cast<u32x4*>(patch + l31 * LW_PROW + 64 * h + 16 * s) = chunk_for_store(hi, lo);
                    }
                    if constexpr (s == 3) wave_lds_sync();
                    if constexpr (s >= 4) {
                        const int pt = 8 * (s - 4) + (lane >> 3), piece = lane & 7;
                        sp16* dst = a.qs + ((size_t)m0 + min(pt, valid - 1)) * SPL_Q_LD + (piece >> 2) * PDSC_CHANNELS + n0 + 8 * (piece & 3);
                        if constexpr (!(EXP & (2 | 8))) *reinterpret_cast<u32x4*>(dst) = ev;
                        else asm volatile("" :: "v"(ev), "v"(dst));
                    }
                    if constexpr (s >= 3 && s < 7) {
                        const int pt = 8 * (s - 3) + (lane >> 3), piece = lane & 7;
                        ev = *reinterpret_cast<const u32x4*>(patch + pt * LW_PROW + 16 * piece);
                    }
                } else {
                    // V^T image: transpose 32 keys x 32 channels through the wave-private LDS patch
                    if constexpr (s < 4) {
                        f32x4 z = v[s];
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[e] = live ? z[e] : 0.f;
                        *reinterpret_cast<f32x4*>(Vs + l31 * LW_VLD + 8 * s + 4 * h) = z;
                    }
                    if constexpr (s == 3) wave_lds_sync();
                    if constexpr (s == 4 || s == 6) {     // lane = (channel 16*it + lane/4, key chunk lane%4)
                        const int cl = 16 * ((s - 4) >> 1) + (lane >> 2), jh = lane & 3;
#pragma unroll
                        for (int e = 0; e < 8; ++e) vt[e] = Vs[spl_v_key(jh, e) * LW_VLD + cl];
                    }
                    if constexpr (s == 5 || s == 7) {     // 256-byte runs: 16 channels of one key chunk
                        const int cl = 16 * ((s - 5) >> 1) + (lane >> 2), jh = lane & 3;
                        unsigned chi[4], clo[4];
#pragma unroll
                        for (int e = 0; e < 8; e += 2) split2(vt[e], vt[e + 1], chi[e / 2], clo[e / 2]);
                        const int off = spl_v_offset(32 * (d.tile - 8) + cl, jh);
                        if constexpr (!(EXP & (2 | 32))) {
                            *reinterpret_cast<u32x4*>(img + SPL_VH + off) = u32x4{chi[0], chi[1], chi[2], chi[3]};
                            *reinterpret_cast<u32x4*>(img + SPL_VL + off) = u32x4{clo[0], clo[1], clo[2], clo[3]};
                        } else asm volatile("" :: "v"(chi[0]), "v"(chi[1]), "v"(chi[2]), "v"(chi[3]), "v"(clo[0]), "v"(clo[1]), "v"(clo[2]), "v"(clo[3]), "v"(off));
                    }
                }
            }
        }
    };

    static_for<0, NCH>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr ChunkDesc d = chunk_desc<T>(i);
        if constexpr (i + NBUF - 1 < NCH && !((EXP & 1) && i >= 1)) load_chunk<T, true, true>(w[(i + NBUF - 1) % NBUF], a, i + NBUF - 1, lane);
        if constexpr (T && d.stage == ST_FC2 && d.tile == 0 && !(EXP & 4)) {
            // residual rows for fc3's epilogue (the fc1 operand is dead, its registers are free): they come from HBM
            const bool pf = a.io_flags & PDSC_IO_RES_PF;
            const float* r0 = a.res + (pf ? (size_t)gw * PF_TILE_FLOATS + lane * 4 : row * PDSC_CHANNELS + 4 * h);
            const int eq = pf ? 256 : 8;
#pragma unroll
            for (int q = 0; q < 16; ++q) y3[q] = *reinterpret_cast<const f32x4*>(r0 + eq * q);
        }
        const WChunk& wc = w[i % NBUF];
        if constexpr (d.chunk == 0) {
            // accumulator := bias, on the matrix pipe: one extra k-step whose A operand is the bias fragment (zero in the
            // second k slot) and whose B operand is 1 -- no VALU work, and C = 0 is an inline constant
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.bias, 1.0f, zero, 0, 0, 0);
        }
        // the previous output tile of the same stage left its epilogue pending: its 8 steps go between this tile's MFMA groups
        constexpr bool pending = d.tile > 0;
        using StageC = std::integral_constant<int, d.stage>;
        using PrevC = std::integral_constant<int, d.tile - 1>;
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;                   // k-step g of this chunk: three MFMAs
            if constexpr (d.stage == ST_QKV) {
                const sp16x8 wh = __builtin_bit_cast(sp16x8, wc.v[2 * g]), wl = __builtin_bit_cast(sp16x8, wc.v[2 * g + 1]);
                const sp16x8 bh = __builtin_bit_cast(sp16x8, xqh[4 * d.chunk + g]), bl = __builtin_bit_cast(sp16x8, xql[4 * d.chunk + g]);
                acc = PDSC_MFMA_X3(wl, bh, acc, 0, 0, 0);
                acc = PDSC_MFMA_X3(wh, bl, acc, 0, 0, 0);
                acc = PDSC_MFMA_X3(wh, bh, acc, 0, 0, 0);
            } else {
                const u32x4* oh = d.stage == ST_FC1 ? a0h : d.stage == ST_FC2 ? a1h : d.stage == ST_FC3 ? a2h : ayh;
                const u32x4* ol = d.stage == ST_FC1 ? a0l : d.stage == ST_FC2 ? a1l : d.stage == ST_FC3 ? a2l : ayl;
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const f16x8 wh = __builtin_bit_cast(f16x8, wc.v[2 * g]), wl = __builtin_bit_cast(f16x8, wc.v[2 * g + 1]);
                const f16x8 bh = __builtin_bit_cast(f16x8, oh[4 * d.chunk + g]), bl = __builtin_bit_cast(f16x8, ol[4 * d.chunk + g]);
                cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, (d.chunk == 0 && g == 0) ? zero : cross, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc, 0, 0, 0);
                cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, cross, 0, 0, 0);
            }
            if constexpr (pending) {
                if constexpr (d.nchunks == 2) {
                    epi(StageC{}, PrevC{}, std::integral_constant<int, 4 * d.chunk + g>{}, vp);
                } else {
                    epi(StageC{}, PrevC{}, std::integral_constant<int, 2 * g>{}, vp);
                    epi(StageC{}, PrevC{}, std::integral_constant<int, 2 * g + 1>{}, vp);
                }
            }
        });
        if constexpr (d.chunk == d.nchunks - 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) vp[g][e] = d.stage != ST_QKV ? fmaf(cross[4 * g + e], H3_INV, acc[4 * g + e]) : acc[4 * g + e];
            if constexpr (d.tile == stage_tiles(d.stage) - 1) {
                // the next stage's first MFMA reads what this epilogue produces (or the kernel ends): in line
                static_for<0, 8>([&](auto sc) { epi(StageC{}, std::integral_constant<int, d.tile>{}, sc, vp); });
            }
        }
        // in-order issue: the order to aim for is one MFMA, then the ~7 vector instructions that fit in its shadow
        if constexpr (PIPE > 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PIPE, 0);
            }
        }
        LH_STAMP(2 + i)
        __builtin_amdgcn_sched_barrier(0);      // chunks are the unit of the software pipeline: no code motion across them
    });

    LH_STAMP(63)
    range_report(a.range_flag, b, rmax);
}

bool launch_layer_h3_fits(const LayerArgs& a, bool tail, bool head) {
    if (a.gemm_format != PDSC_LAYER_GEMM_H3) return false;
    if (head && (!a.qs || !a.kv || a.qkv_out)) return false;
    if (tail && head && a.feat_out) return false;
    if (tail && !head && !a.feat_out) return false;
    return a.wf_tail || !tail;
}

// launch shape (waves per workgroup, weight-ring depth), see the kernel's NWV / NBUF note
static void h3_launch_shape(int tiles, int* nwv, int* nbuf) {
    *nwv = LW_WAVES; *nbuf = 2;
    (void)tiles;
#ifdef PDSC_EXPERIMENTS
    const int force = env_int("PDSC_LAYER_H3_SHAPE", 0);          // A/B knob (experiments builds): 10 * waves + depth, e.g. 23
    if (force == 42 || force == 23 || force == 13 || force == 24 || force == 14 || force == 22) { *nwv = force / 10; *nbuf = force % 10; }
#endif
}

int launch_layer_h3(const LayerArgs& a, bool tail, bool head, hipStream_t st) {
    const int waves = a.bs * ceil_div(a.N, 32);
    int nwv, nbuf;
    h3_launch_shape(waves, &nwv, &nbuf);
    const dim3 grid(ceil_div(waves, nwv)), block(64 * nwv);
    const bool fb_pf = a.io_flags & PDSC_IO_FEATB_PF;
    const bool timed = tail && head;
    int coop_tiles = PDSC_H3_COOP_TILES;
#ifdef PDSC_EXPERIMENTS
    coop_tiles = env_int("PDSC_LAYER_H3_COOP", coop_tiles);        // A/B knob (experiments builds): tile-count threshold, 0 = never
#endif
    if (waves <= coop_tiles && !a.trace) {
        if (timed) profile_mark_begin(PDSC_PROF_LAYER, st);
        const int rc = launch_layer_h3_coop(a, tail, head, st);
        if (timed) profile_mark_end(PDSC_PROF_LAYER, st);
        return rc;
    }
    if (timed) profile_mark_begin(PDSC_PROF_LAYER, st);
    // every (tail, head, featB order) form in the default shape and in the small-launch shape (+ the A/B shapes)
#ifdef PDSC_EXPERIMENTS
#define PDSC_H3_LAUNCH_EXTRA(TT, HH, PF)                                                                                            \
        if (nwv == 1 && nbuf == 4) hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF, 1, 4>), grid, block, 0, st, a);      \
        else if (nwv == 2 && nbuf == 3) hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF, 2, 3>), grid, block, 0, st, a); \
        else if (nwv == 1 && nbuf == 3) hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF, 1, 3>), grid, block, 0, st, a); \
        else if (nwv == 2 && nbuf == 4) hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF, 2, 4>), grid, block, 0, st, a); \
        else if (nwv == 2 && nbuf == 2) hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF, 2, 2>), grid, block, 0, st, a); \
        else
#else
#define PDSC_H3_LAUNCH_EXTRA(TT, HH, PF)
#endif
#define PDSC_H3_LAUNCH(TT, HH, PF)                                                                                                  \
    do {                                                                                                                            \
        PDSC_H3_LAUNCH_EXTRA(TT, HH, PF)                                                                                            \
        hipLaunchKernelGGL((layer_h3_kernel<TT, HH, false, 6, 0, PF>), grid, block, 0, st, a);                                       \
    } while (0)
    if (tail && head && fb_pf) {
#ifdef PDSC_LAYER_DIAG      // knock-out build (PDSC_HIPCC_EXTRA=-DPDSC_LAYER_DIAG python -m pointdsc_amd.build --force; tools/layer_bench.py)
        const int ex = env_int("PDSC_LAYER_H3_EXP", 0);
        if (ex == 1) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 1, true>), grid, block, 0, st, a);
        else if (ex == 2) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 2, true>), grid, block, 0, st, a);
        else if (ex == 4) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 4, true>), grid, block, 0, st, a);
        else if (ex == 7) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 7, true>), grid, block, 0, st, a);
        else if (ex == 8) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 8, true>), grid, block, 0, st, a);
        else if (ex == 16) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 16, true>), grid, block, 0, st, a);
        else if (ex == 32) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 32, true>), grid, block, 0, st, a);
        else if (ex == 64) hipLaunchKernelGGL((layer_h3_kernel<true, true, false, 6, 64, true>), grid, block, 0, st, a);
        else
#endif
        PDSC_H3_LAUNCH(true, true, true);
    } else if (head && !tail && fb_pf) {
        PDSC_H3_LAUNCH(false, true, true);
    } else if (tail && head) {
#ifdef PDSC_EXPERIMENTS       // (the shader-clock trace form: experiments builds, tools/layer_trace.py)
        if (a.trace) hipLaunchKernelGGL((layer_h3_kernel<true, true, true>), dim3(ceil_div(waves, LW_WAVES)), dim3(64 * LW_WAVES), 0, st, a);
        else
#endif
        PDSC_H3_LAUNCH(true, true, false);
    } else if (tail) {
        PDSC_H3_LAUNCH(true, false, false);
    } else {
        PDSC_H3_LAUNCH(false, true, false);
    }
#undef PDSC_H3_LAUNCH
#undef PDSC_H3_LAUNCH_EXTRA
    if (timed) profile_mark_end(PDSC_PROF_LAYER, st);
    return check_launch("pdsc_layer_fused_frag(h3)");
}

}  // namespace pdsc

using namespace pdsc;

extern long long* pdsc_layer_trace_buffer(void);

extern "C" int pdsc_layer_h3_uses_coop(int bs, int N) {
    if (bs <= 0 || N <= 0) return 0;
    int coop_tiles = PDSC_H3_COOP_TILES;
#ifdef PDSC_EXPERIMENTS
    coop_tiles = env_int("PDSC_LAYER_H3_COOP", coop_tiles);
#endif
    return (long long)bs * ceil_div(N, 32) <= coop_tiles ? 1 : 0;
}

extern "C" int pdsc_layer_fused_frag_io(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                                        const float* res, const float* feat_in, float* feat_out, float* featB_out,
                                        void* q_split, void* kv_tiles, const void* wfrag_tail, const void* wfrag_head,
                                        int gemm_format, int io_flags, int bs, int N, void* stream) {
    if (io_flags == 0)
        return pdsc_layer_fused_frag_fmt(msg, part_o, part_ml, nsplit, Npad, res, feat_in, feat_out, featB_out, nullptr, q_split, kv_tiles,
                                         wfrag_tail, wfrag_head, gemm_format, bs, N, stream);
    const bool tail = msg != nullptr || part_o != nullptr, head = featB_out != nullptr;
    PDSC_REQUIRE(tail || head, "pdsc_layer_fused_frag_io: neither tail (msg / partials) nor head (featB_out) requested");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_layer_fused_frag_io: bs=%d N=%d", bs, N);
    PDSC_REQUIRE((io_flags & ~(PDSC_IO_PARTIALS_PF | PDSC_IO_RES_PF | PDSC_IO_FEATB_PF)) == 0, "pdsc_layer_fused_frag_io: io_flags=%d", io_flags);
    PDSC_REQUIRE(gemm_format == PDSC_LAYER_GEMM_H3, "pdsc_layer_fused_frag_io: point-fragment hand-offs need gemm_format = PDSC_LAYER_GEMM_H3");
    if (tail) {
        PDSC_REQUIRE(res && wfrag_tail, "pdsc_layer_fused_frag_io: tail needs res and the tail stream");
        if (!msg) PDSC_REQUIRE(part_ml && nsplit >= 1 && nsplit <= MERGE_MAX_SPLIT_H3 && Npad >= N,
                               "pdsc_layer_fused_frag_io: partials need part_ml, 1 <= nsplit <= %d, Npad >= N", MERGE_MAX_SPLIT_H3);
        PDSC_REQUIRE(!(io_flags & PDSC_IO_PARTIALS_PF) || (!msg && Npad % 32 == 0), "pdsc_layer_fused_frag_io: PF partials come un-merged (msg NULL), Npad a multiple of 32");
    } else {
        PDSC_REQUIRE(feat_in, "pdsc_layer_fused_frag_io: head-only needs feat_in");
        PDSC_REQUIRE(!(io_flags & (PDSC_IO_PARTIALS_PF | PDSC_IO_RES_PF)), "pdsc_layer_fused_frag_io: head-only takes feat_in in row order");
    }
    if (head) PDSC_REQUIRE(q_split && kv_tiles && wfrag_head, "pdsc_layer_fused_frag_io: head needs the split streams and the head stream");
    else PDSC_REQUIRE(feat_out && !(io_flags & PDSC_IO_FEATB_PF), "pdsc_layer_fused_frag_io: tail-only needs feat_out (row order)");
    LayerArgs a{};
    a.msg = msg; a.part_o = part_o; a.part_ml = part_ml; a.nsplit = nsplit; a.Npad = Npad;
    a.res = res; a.feat_in = feat_in; a.feat_out = feat_out; a.featB_out = featB_out;
    a.qs = (sp16*)q_split; a.kv = (unsigned char*)kv_tiles;
    a.N = N; a.bs = bs;
    a.wf_tail = (const unsigned char*)wfrag_tail; a.wf_head = (const unsigned char*)wfrag_head;
    a.gemm_format = gemm_format;
    a.io_flags = io_flags;
    a.stagger_cycles = env_int("PDSC_LAYER_STAGGER", 0);
    a.stagger_mode = env_int("PDSC_LAYER_STAGGER_MODE", 1);
    a.trace = nullptr;
    a.nvalid = layer_nvalid_slot();
    a.range_flag = range_flag_slot();
    PDSC_REQUIRE(launch_layer_h3_fits(a, tail, head), "pdsc_layer_fused_frag_io: output set not served by the point-fragment kernel "
                                                     "(tail + head: no feat_out; tail only: feat_out)");
    return launch_layer_h3(a, tail, head, (hipStream_t)stream);
}
