// a-2 fused layer for launches of at most PDSC_H3_COOP_TILES tiles: FOUR wavefronts per 32-point tile (layer_h3.hip has one).
//   tail of layer i   : feat  = featB + fc3( relu(fc2'( relu(fc1'(msg)) )) )         (reference models/PointDSC.py:43-45)
//   head of layer i+1 : featB = relu(pcn'(feat)) ; (q|k|v) = Wqkv featB + b            (models/PointDSC.py:75, :36-38)
// With few tiles (N = 1000 x 1: 32 tiles on 256 CUs) a launch of layer_h3_kernel is ONE wavefront's dependency chain: 42
// weight chunks x 12 MFMAs issued in order behind the chunk loads, 28 us per launch, 11 launches per forward = 60 % of the
// forward.  Here the output tiles of every stage are dealt to the four wavefronts of a workgroup (fc1 / fc2: tiles 0..1 to
// waves 0..1; fc3 / pcn: one tile each; q|k|v: tiles w, 4+w, 8+w = one Q, one K, one V tile each), so the chain per wavefront
// is 12 chunks instead of 42, each wavefront streams a quarter of the weights, and the stages hand their result -- already
// split into the next GEMM's B operands -- to each other through 16 KiB of LDS and a workgroup barrier (5 per launch).
// The arithmetic of every output element is that of layer_h3_kernel (same MFMA order per output tile: bias step, then per
// k-step cross / main / cross), so the two kernels agree bit for bit; tests/test_gpu_parity.py holds them to that.
#include <type_traits>
#include "pdsc_common.h"
#include "split_layout.h"
#include "merge_partials.h"
#include "layer_args.h"
#include "layer_wave.h"
#include "ragged.h"

namespace pdsc {

constexpr int LC_WAVES = 4;

struct CoopOps { u32x4 v[2][8][2][64]; };      // [set][k-step][hi | lo][lane]: the B operands of the next stage, 32 KiB

// one weight chunk on the matrix cores, the instruction order of layer_h3_kernel
template <bool QKV>
__device__ __forceinline__ void coop_chunk(f32x16& acc, f32x16& cross, const WChunk& wc, const u32x4* oh, const u32x4* ol, bool first) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (first) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.bias, 1.0f, zero, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (QKV) {
            const sp16x8 wh = __builtin_bit_cast(sp16x8, wc.v[2 * g]), wl = __builtin_bit_cast(sp16x8, wc.v[2 * g + 1]);
            const sp16x8 bh = __builtin_bit_cast(sp16x8, oh[g]), bl = __builtin_bit_cast(sp16x8, ol[g]);
            acc = PDSC_MFMA_X3(wl, bh, acc, 0, 0, 0);
            acc = PDSC_MFMA_X3(wh, bl, acc, 0, 0, 0);
            acc = PDSC_MFMA_X3(wh, bh, acc, 0, 0, 0);
        } else {
            const f16x8 wh = __builtin_bit_cast(f16x8, wc.v[2 * g]), wl = __builtin_bit_cast(f16x8, wc.v[2 * g + 1]);
            const f16x8 bh = __builtin_bit_cast(f16x8, oh[g]), bl = __builtin_bit_cast(f16x8, ol[g]);
            if (first && g == 0) cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, zero, 0, 0, 0);
            else cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, cross, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc, 0, 0, 0);
            cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, cross, 0, 0, 0);
        }
    }
}

template <bool QKV>
__device__ __forceinline__ void coop_finish(const f32x16& acc, const f32x16& cross, f32x4 (&v)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[g][e] = QKV ? acc[4 * g + e] : fmaf(cross[4 * g + e], H3_INV, acc[4 * g + e]);
}

// NB: depth of the wavefront's weight-chunk ring.  4, with the operands re-read from LDS per chunk and the partials loaded in
// batches, fits 256 registers (236), so two workgroups share a CU: launches of 257..512 tiles (the per-GPU shares of the 8-GPU
// configurations: 2 pairs of N = 5000, 1 pair of N = 10000) run in one round.  A ring of 6 (382 registers, one workgroup per
// CU) measured 1-2 % SLOWER even at 32 tiles (profiles/r03_k_ab_coop.txt) and is not instantiated.
template <bool T, bool H, bool FB_PF, int NB>
__global__ __launch_bounds__(64 * LC_WAVES, NB <= 4 ? 2 : 1) void layer_h3_coop_kernel(LayerArgs a) {
    __shared__ __attribute__((aligned(16))) CoopOps ops;
    __shared__ __attribute__((aligned(16))) float Vs_all[LC_WAVES][32 * LW_VLD];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tpp = ceil_div_dev(a.N, 32);                          // tiles per pair
    const int gw = blockIdx.x;                                       // one workgroup = one tile
    const int b = gw / tpp, tile = gw - b * tpp;
    const int m0 = b * a.N + tile * 32;
    const int valid = min(32, (a.nvalid ? a.nvalid[b] : a.N) - tile * 32);
    if (valid <= 0) return;                                          // (ragged batches; uniform over the workgroup)
    const bool live = l31 < valid;
    const size_t row = (size_t)m0 + min(l31, valid - 1);
    float* Vs = Vs_all[w];
    unsigned char* patch = reinterpret_cast<unsigned char*>(Vs);
    float rmax = 0.f;            // fp16 range sentinel (pdsc_common.h): largest |activation| this lane converts to an fp16 pair

    // ---- this wavefront's weight chunks in the order it uses them: sequence position k = 0..11 --------------------------------
    //   0, 1: fc1 tile w (waves 0, 1 only)   2: fc2 tile w (waves 0, 1 only)   3: fc3 tile w   4, 5: pcn tile w
    //   6, 7: q|k|v tile w (Q)   8, 9: tile 4 + w (K)   10, 11: tile 8 + w (V)
    // tail stream: fc1 tile t = chunks 2t, 2t+1; fc2 tile t = 4 + t; fc3 tile t = 6 + t.  head stream: pcn tile t = 2t, 2t+1;
    // q|k|v tile t = 8 + 2t, 9 + 2t.  Bias fragments by tile ordinal (layer_wave.h).  Position k lives in ring slot k % NB and
    // is requested as soon as position k - NB has been consumed.
    constexpr int K0 = T ? 0 : 4, K1 = H ? 12 : 4;
    WChunk W[NB];
    const bool low = w < 2;                                          // fc1 / fc2 have two output tiles: waves 0 and 1
    auto issue = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k >= K0 && k < K1) {
            WChunk& dst = W[k % NB];
            if constexpr (k < 2) { if (low) load_chunk_frag(dst, a.wf_tail, 2 * w + k, LW_TAIL_CHUNKS, k == 0 ? w : -1, lane); }
            else if constexpr (k == 2) { if (low) load_chunk_frag(dst, a.wf_tail, 4 + w, LW_TAIL_CHUNKS, 2 + w, lane); }
            else if constexpr (k == 3) load_chunk_frag(dst, a.wf_tail, 6 + w, LW_TAIL_CHUNKS, 4 + w, lane);
            else if constexpr (k < 6) load_chunk_frag(dst, a.wf_head, 2 * w + (k - 4), LW_HEAD_CHUNKS, k == 4 ? w : -1, lane);
            else {
                const int t = w + 4 * ((k - 6) >> 1);
                load_chunk_frag(dst, a.wf_head, 8 + 2 * t + (k & 1), LW_HEAD_CHUNKS, (k & 1) ? -1 : 4 + t, lane);
            }
        }
    };
    static_for<K0, K0 + NB>([&](auto kc) { issue(kc); });
    // chunk at position k on the matrix cores (operands: k-steps 4c .. 4c+3 of LDS set `set`), then the request for k + NB
    f32x16 acc, cross;
    auto run_chunk = [&](auto kc, int set, int c, bool first, auto qkv_c) {
        constexpr int k = decltype(kc)::value;
        u32x4 oh[4], ol[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { oh[g] = ops.v[set][4 * c + g][0][lane]; ol[g] = ops.v[set][4 * c + g][1][lane]; }
        coop_chunk<decltype(qkv_c)::value>(acc, cross, W[k % NB], oh, ol, first);
        issue(std::integral_constant<int, k + NB>{});
    };
    using std::false_type; using std::true_type;
    auto put = [&](int set, int kk, const u32x4& oh, const u32x4& ol) { ops.v[set][kk][0][lane] = oh; ops.v[set][kk][1][lane] = ol; };
    f32x4 y3[4];            // this wave's 32 channels (32w + 8s + 4h + e) of the residual rows, then of feat
    f32x4 v[4];

    if constexpr (T) {
        // ---- stage 0: operand of fc1 = the attention's message; this wave converts channels 32w .. 32w+31 (k-steps 2w, 2w+1)
        f32x4 x0[4];
        if (a.msg) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x0[q] = *reinterpret_cast<const f32x4*>(a.msg + row * PDSC_CHANNELS + 8 * (4 * w + q) + 4 * h);
        } else {
            // merge of the attention's key-split partials, the arithmetic of merge_partials_finish (merge_partials.h)
            auto run = [&](auto ns_tag) {
                constexpr int NS = decltype(ns_tag)::value;
                const size_t slot0 = (size_t)b * NS * a.Npad + (row - (size_t)b * a.N);
                const bool pf = a.io_flags & PDSC_IO_PARTIALS_PF;
                const size_t e0 = pf ? ((size_t)b * NS * a.Npad + (size_t)tile * 32) * PDSC_CHANNELS + lane * 4 : slot0 * PDSC_CHANNELS + 4 * h;
                const int eq = pf ? 256 : 8;
                // q per batch of loads: everything at once in the latency form; <= 32 registers of partials in flight in the
                // two-workgroups-per-CU form (256 registers)
                constexpr int GQ = NB > 4 ? 4 : NS <= 2 ? 4 : NS <= 4 ? 2 : 1;
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
                for (int q0 = 0; q0 < 4; q0 += GQ) {
                    f32x4 pv[GQ][NS];
#pragma unroll
                    for (int q = 0; q < GQ; ++q)
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
                            pv[q][sp] = *reinterpret_cast<const f32x4*>(a.part_o + e0 + (size_t)sp * a.Npad * PDSC_CHANNELS + (size_t)eq * (4 * w + q0 + q));
#pragma unroll
                    for (int q = 0; q < GQ; ++q) {
                        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s[e] = fmaf(pv[q][sp][e], wsp[sp], s[e]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x0[q0 + q][e] = s[e] * rden;
                            if constexpr (GQ < 4) asm volatile("" : "+v"(x0[q0 + q][e]));    // materialise: the next batch's loads reuse the registers
                        }
                    }
                    if constexpr (GQ < 4) __builtin_amdgcn_sched_barrier(0);
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
        {
            u32x4 ph, pl;
            make_kstep<true>(x0[0], x0[1], ph, pl, rmax); put(0, 2 * w, ph, pl);
            make_kstep<true>(x0[2], x0[3], ph, pl, rmax); put(0, 2 * w + 1, ph, pl);
        }
        // residual rows of this wave's fc3 tile (needed three stages on: the loads fly under fc1 / fc2)
        {
            const bool pf = a.io_flags & PDSC_IO_RES_PF;
            const float* r0 = a.res + (pf ? (size_t)gw * PF_TILE_FLOATS + lane * 4 : row * PDSC_CHANNELS + 4 * h);
            const int eq = pf ? 256 : 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) y3[s] = *reinterpret_cast<const f32x4*>(r0 + eq * (4 * w + s));
        }
        __syncthreads();

        // ---- fc1: 128 -> 64, output tile w on waves 0 and 1 --------------------------------------------------------------
        if (low) {
            run_chunk(std::integral_constant<int, 0>{}, 0, 0, true, false_type{});
            run_chunk(std::integral_constant<int, 1>{}, 0, 1, false, false_type{});
            coop_finish<false>(acc, cross, v);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[s][e] = fmaxf(v[s][e], 0.f);
            u32x4 ph, pl;
            make_kstep<true>(v[0], v[1], ph, pl, rmax); put(1, 2 * w, ph, pl);
            make_kstep<true>(v[2], v[3], ph, pl, rmax); put(1, 2 * w + 1, ph, pl);
        } else {
            issue(std::integral_constant<int, NB>{});                 // (waves 2, 3: positions 0..2 are empty, their slots free)
            issue(std::integral_constant<int, NB + 1>{});
        }
        __syncthreads();

        // ---- fc2: 64 -> 64, output tile w on waves 0 and 1 ---------------------------------------------------------------
        if (low) {
            run_chunk(std::integral_constant<int, 2>{}, 1, 0, true, false_type{});
            coop_finish<false>(acc, cross, v);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[s][e] = fmaxf(v[s][e], 0.f);
            u32x4 ph, pl;
            make_kstep<true>(v[0], v[1], ph, pl, rmax); put(0, 2 * w, ph, pl);
            make_kstep<true>(v[2], v[3], ph, pl, rmax); put(0, 2 * w + 1, ph, pl);
        } else {
            issue(std::integral_constant<int, NB + 2>{});
        }
        __syncthreads();

        // ---- fc3: 64 -> 128, output tile w; feat = residual + fc3 --------------------------------------------------------
        run_chunk(std::integral_constant<int, 3>{}, 0, 0, true, false_type{});
        coop_finish<false>(acc, cross, v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y3[s][e] = y3[s][e] + v[s][e];
            // tail-only launches return feat; lanes beyond the pair's last point hold copies of its last row
            if constexpr (!H) *reinterpret_cast<f32x4*>(a.feat_out + row * PDSC_CHANNELS + 32 * w + 8 * s + 4 * h) = y3[s];
        }
    } else {
        // head only (first layer): feat comes in as rows
#pragma unroll
        for (int s = 0; s < 4; ++s) y3[s] = *reinterpret_cast<const f32x4*>(a.feat_in + row * PDSC_CHANNELS + 32 * w + 8 * s + 4 * h);
    }

    if constexpr (H) {
        {
            u32x4 ph, pl;
            make_kstep<true>(y3[0], y3[1], ph, pl, rmax); put(1, 2 * w, ph, pl);
            make_kstep<true>(y3[2], y3[3], ph, pl, rmax); put(1, 2 * w + 1, ph, pl);
        }
        __syncthreads();

        // ---- pcn: 128 -> 128, output tile w; featB = relu -----------------------------------------------------------------
        run_chunk(std::integral_constant<int, 4>{}, 1, 0, true, false_type{});
        run_chunk(std::integral_constant<int, 5>{}, 1, 1, false, false_type{});
        coop_finish<false>(acc, cross, v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[s][e] = fmaxf(v[s][e], 0.f);
            if constexpr (FB_PF) *reinterpret_cast<f32x4*>(a.featB_out + (size_t)gw * PF_TILE_FLOATS + pf_offset_floats(4 * w + s) + lane * 4) = v[s];
            else *reinterpret_cast<f32x4*>(patch + l31 * LW_PROW + 32 * s + 16 * h) = v[s];
        }
        if constexpr (!FB_PF) {
            wave_lds_sync();
#pragma unroll
            for (int it = 0; it < 4; ++it) {      // 8 points x 128 B per store instruction (rows >= valid: copies of the last row)
                const int pt = 8 * it + (lane >> 3), piece = lane & 7;
                const u32x4 ev = *reinterpret_cast<const u32x4*>(patch + pt * LW_PROW + 16 * piece);
                *reinterpret_cast<u32x4*>(a.featB_out + ((size_t)m0 + min(pt, valid - 1)) * PDSC_CHANNELS + 32 * w + 4 * piece) = ev;
            }
            wave_lds_sync();
        }
        {
            u32x4 ph, pl;
            make_kstep<false>(v[0], v[1], ph, pl, rmax); put(0, 2 * w, ph, pl);
            make_kstep<false>(v[2], v[3], ph, pl, rmax); put(0, 2 * w + 1, ph, pl);
        }
        __syncthreads();

        // ---- q | k | v: 128 -> 384; this wave's Q tile w, K tile w, V tile w ---------------------------------------------------
        unsigned char* img = a.kv + (size_t)gw * SPL_TILE_STRIDE;
        const int n0 = 32 * w;

        // Q rows (hi[128] | lo[128]) fp16 through the patch: row = (hi 64 B | lo 64 B) of this tile's 32 channels
        run_chunk(std::integral_constant<int, 6>{}, 0, 0, true, true_type{});
        run_chunk(std::integral_constant<int, 7>{}, 0, 1, false, true_type{});
        coop_finish<true>(acc, cross, v);
#pragma unroll
        for (int s = 0; s < 4; ++s) range_note(rmax, v[s]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned hi[2], lo[2];
            split4(v[s], hi, lo);
            *reinterpret_cast<u32x4*>(patch + l31 * LW_PROW + 64 * h + 16 * s) = chunk_for_store(hi, lo);
        }
        wave_lds_sync();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pt = 8 * it + (lane >> 3), piece = lane & 7;
            const u32x4 ev = *reinterpret_cast<const u32x4*>(patch + pt * LW_PROW + 16 * piece);
            sp16* dst = a.qs + ((size_t)m0 + min(pt, valid - 1)) * SPL_Q_LD + (piece >> 2) * PDSC_CHANNELS + n0 + 8 * (piece & 3);
            *reinterpret_cast<u32x4*>(dst) = ev;
        }

        // K image, chunk-major (split_layout.h): after the half swap lane (key l31, half h) holds chunk 4w + s of its key for the
        // hi (h = 0) / lo (h = 1) plane
        run_chunk(std::integral_constant<int, 8>{}, 0, 0, true, true_type{});
        run_chunk(std::integral_constant<int, 9>{}, 0, 1, false, true_type{});
        coop_finish<true>(acc, cross, v);
#pragma unroll
        for (int s = 0; s < 4; ++s) range_note(rmax, v[s]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 z = v[s];
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = live ? z[e] : 0.f;                    // keys beyond N are zero
            unsigned hi[2], lo[2];
            split4(z, hi, lo);
            *reinterpret_cast<u32x4*>(img + (h ? SPL_KL : SPL_KH) + spl_k_offset(l31, 4 * w + s)) = chunk_for_store(hi, lo);
        }

        // V^T image: transpose 32 keys x 32 channels through the wave-private LDS patch
        run_chunk(std::integral_constant<int, 10>{}, 0, 0, true, true_type{});
        run_chunk(std::integral_constant<int, 11>{}, 0, 1, false, true_type{});
        coop_finish<true>(acc, cross, v);
#pragma unroll
        for (int s = 0; s < 4; ++s) range_note(rmax, v[s]);
        wave_lds_sync();                                                              // (the Q passes have read the patch)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 z = v[s];
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = live ? z[e] : 0.f;
            *reinterpret_cast<f32x4*>(Vs + l31 * LW_VLD + 8 * s + 4 * h) = z;
        }
        wave_lds_sync();
#pragma unroll
        for (int it = 0; it < 2; ++it) {          // lane = (channel 16*it + lane/4, key chunk lane%4): 256-byte runs
            const int cl = 16 * it + (lane >> 2), jh = lane & 3;
            float vt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[e] = Vs[spl_v_key(jh, e) * LW_VLD + cl];
            unsigned chi[4], clo[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) split2(vt[e], vt[e + 1], chi[e / 2], clo[e / 2]);
            const int off = spl_v_offset(n0 + cl, jh);
            *reinterpret_cast<u32x4*>(img + SPL_VH + off) = u32x4{chi[0], chi[1], chi[2], chi[3]};
            *reinterpret_cast<u32x4*>(img + SPL_VL + off) = u32x4{clo[0], clo[1], clo[2], clo[3]};
        }
    }
    range_report(a.range_flag, b, rmax);
}

int launch_layer_h3_coop(const LayerArgs& a, bool tail, bool head, hipStream_t st) {
    const dim3 grid(a.bs * ceil_div(a.N, 32)), block(64 * LC_WAVES);
    const bool fb_pf = a.io_flags & PDSC_IO_FEATB_PF;
    if (tail && head && fb_pf) hipLaunchKernelGGL((layer_h3_coop_kernel<true, true, true, 4>), grid, block, 0, st, a);
    else if (head && !tail && fb_pf) hipLaunchKernelGGL((layer_h3_coop_kernel<false, true, true, 4>), grid, block, 0, st, a);
    else if (tail && head) hipLaunchKernelGGL((layer_h3_coop_kernel<true, true, false, 4>), grid, block, 0, st, a);
    else if (tail) hipLaunchKernelGGL((layer_h3_coop_kernel<true, false, false, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((layer_h3_coop_kernel<false, true, false, 4>), grid, block, 0, st, a);
    return check_launch("pdsc_layer_fused_frag(h3, four wavefronts per tile)");
}

}  // namespace pdsc
