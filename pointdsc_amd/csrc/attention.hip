// a-3: spatial-consistency guided non-local attention, flash style, exact fp32 on the matrix cores
// (reference models/PointDSC.py:39-42: einsum QK^T / sqrt(C), softmax(compat * .), einsum with V).
//
//   msg[o][:] = sum_i softmax_i( compat[o][i] * <Q_o,K_i>/sqrt(C) ) V_i          (dense softmax: compat==0
//   entries still contribute exp(0), so nothing can be skipped)
//
// Bound: MFMA.  4*C*N^2 flop per layer per pair on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate --
// bf16/fp16 inputs break the 1e-4 R/t budget).  The N x N score matrix never exists;
// the only N^2 traffic is one streamed read of compat (4*N^2 bytes per layer).
//
// Decomposition: workgroup = 4 waves = 128 queries x one contiguous range of 32-key tiles (key ranges are
// split across workgroups when bs*N/128 alone cannot fill 256 CUs; partials are merged by
// attention_combine_kernel).  Each wave owns 32 queries:
//   S^T = K . Q^T   (A = K tile from LDS, B = Q held in 64 VGPRs for the whole kernel)
//         -> lane (query = lane&31, half = lane>>5) holds 16 keys of ITS query: softmax state is lane-local
//   O^T += V^T . P^T (A = V tile from LDS, B = the P registers exactly as the softmax left them)
//         -> accumulator lane = query again, so the online-softmax rescale is a per-lane scalar multiply.
// k-slot convention (pdsc linear.hip): QK step (q,e), half h <-> channel 8q+4h+e;
//                                      PV step r,     half h <-> key (r&3)+8(r>>2)+4h  (= S^T's C/D row map).
// K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, double buffered); the K
// image is XOR-swizzled through the SOURCE address (chunk ^= row&15) so the column-slice ds_read_b128 of
// 16 different rows hits 16 different bank slots.  Q is pre-scaled by log2(e)/sqrt(C) in the packed weights,
// so p = exp2(compat*s - m).
#include <stdlib.h>
#include "pdsc_common.h"
#include "attention_common.h"

namespace pdsc {

#define PDSC_ATT_DEFAULT_VARIANT 1
constexpr int ATT_BQ = 128;
constexpr int ATT_BK = 32;
constexpr int ATT_C = PDSC_CHANNELS;
constexpr int ATT_QKV_LD = 3 * PDSC_CHANNELS;
constexpr int ATT_TILE_FLOATS = ATT_BK * ATT_C;        // 4096 floats = 16 KiB



// one wave issues its quarter (4 x 1 KiB) of a 32-key K tile and V tile
__device__ __forceinline__ void issue_tile_loads(const float* __restrict__ kbase, const float* __restrict__ vbase,
                                                 int k0, int N, float* Ks, float* Vs, int wave, int lane) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = wave * 4 + u;                      // 1-KiB piece: rows 2i, 2i+1
        const int row = 2 * i + (lane >> 5);
        const int cph = lane & 31;                       // physical 16-B chunk inside the LDS row
        const int krow = min(k0 + row, N - 1);           // clamp: tail keys are masked in the softmax
        const float* ksrc = kbase + (size_t)krow * ATT_QKV_LD + ((cph ^ (row & 15)) << 2);
        const float* vsrc = vbase + (size_t)krow * ATT_QKV_LD + (cph << 2);
        __builtin_amdgcn_global_load_lds((gptr_t)ksrc, (lptr_t)(Ks + i * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)vsrc, (lptr_t)(Vs + i * 256), 16, 0, 0);
    }
}

// PIPE = 0: operand fragments read right before use (compiler-scheduled);
// PIPE = 1: K/V fragment reads run two ds_read_b128 ahead of the MFMAs that consume them (explicit ring).
template <int PIPE>
__global__ __launch_bounds__(256, 2) void sc_attention_kernel(AttArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // K0 K1 V0 V1, 16 KiB each
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x, sp = blockIdx.y, b = blockIdx.z;
    // ragged batches: this pair's own count decides its queries, its keys and its tile range; the strides stay those of the longest pair
    const int NS = a.N;
    const int N = a.nvalid ? a.nvalid[b] : NS;
    if (qb * ATT_BQ >= N) return;                                   // (uniform over the workgroup: no query of this block exists)
    const int ntiles = a.nvalid ? ceil_div_dev(N, ATT_BK) : a.num_tiles;

    // key-tile range of this split: tiles [kt0, kt1) -- possibly empty for a short pair of a ragged batch: its partial is then
    // (m = -1e30, l = 0, O = 0), which the merge weighs with exp2(-1e30 - max) = 0
    const int per = ntiles / a.nsplit, rem = ntiles % a.nsplit;
    const int kt0 = sp * per + min(sp, rem);
    const int kt1 = kt0 + per + (sp < rem ? 1 : 0);

    const float* qkvb = a.qkv + (size_t)b * NS * ATT_QKV_LD;
    const float* kbase = qkvb + ATT_C;
    const float* vbase = qkvb + 2 * ATT_C;
    const int qrow = min(qb * ATT_BQ + wave * 32 + l31, N - 1);
    const float* crow = a.compat + ((size_t)b * NS + qrow) * a.ld + 4 * h;

    // prologue: first K/V tile in flight, then this lane's Q fragment
    if (kt0 < kt1) issue_tile_loads(kbase, vbase, kt0 * ATT_BK, N, lds, lds + 2 * ATT_TILE_FLOATS, wave, lane);
    f32x4 qf[16];
    {
        const float* qsrc = qkvb + (size_t)qrow * ATT_QKV_LD + 4 * h;
#pragma unroll
        for (int q = 0; q < 16; ++q) qf[q] = *reinterpret_cast<const f32x4*>(qsrc + 8 * q);
    }
    f32x4 cc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cc[g] = *reinterpret_cast<const f32x4*>(crow + min(kt0, ntiles - 1) * ATT_BK + 8 * g);

    f32x16 o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m_run = -1.0e30f, l_run = 0.f;

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const float* Kb = lds + buf * ATT_TILE_FLOATS;
        const float* Vb = lds + (2 + buf) * ATT_TILE_FLOATS;
        // tile kt landed (own LDS-DMA pieces) + everyone finished reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < kt1)
            issue_tile_loads(kbase, vbase, (kt + 1) * ATT_BK, N, lds + (buf ^ 1) * ATT_TILE_FLOATS,
                             lds + (2 + (buf ^ 1)) * ATT_TILE_FLOATS, wave, lane);

        // ---- S^T = K Q^T -------------------------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* krow_p = Kb + l31 * ATT_C;
        const int ksw = l31 & 15;
        if (PIPE == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(krow_p + (((2 * q + h) ^ ksw) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[q][e], s, 0, 0, 0);
            }
        } else {
            f32x4 kr[4];
            kr[0] = *reinterpret_cast<const f32x4*>(krow_p + (((0 + h) ^ ksw) << 2));
            kr[1] = *reinterpret_cast<const f32x4*>(krow_p + (((2 + h) ^ ksw) << 2));
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                if (q + 2 < 16) {
                    kr[(q + 2) & 3] = *reinterpret_cast<const f32x4*>(krow_p + (((2 * (q + 2) + h) ^ ksw) << 2));
                    kr[(q + 3) & 3] = *reinterpret_cast<const f32x4*>(krow_p + (((2 * (q + 3) + h) ^ ksw) << 2));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[q & 3][e], qf[q][e], s, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[(q + 1) & 3][e], qf[q + 1][e], s, 0, 0, 0);
            }
        }

        // ---- online softmax (log2 domain), lane-local: this lane = query l31, keys (r&3)+8(r>>2)+4h ----
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = cc[r >> 2][r & 3] * s[r];
        if (kt + 1 < kt1) {   // compat of the NEXT tile into the registers just consumed: a whole PV + QK phase to land
#pragma unroll
            for (int g = 0; g < 4; ++g) cc[g] = *reinterpret_cast<const f32x4*>(crow + (kt + 1) * ATT_BK + 8 * g);
        }
        if ((kt + 1) * ATT_BK > N) {   // tail tile (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * ATT_BK + (r & 3) + 8 * (r >> 2) + 4 * h;
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
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = __builtin_amdgcn_exp2f(x[r] - m_run);
            psum += x[r];
        }
        l_run += psum;

        // ---- O^T += V^T P^T ----------------------------------------------------------------------
        const float* vcol = Vb + 4 * l31;
        if (PIPE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x4 va = *reinterpret_cast<const f32x4*>(vcol + key * ATT_C);
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[c], x[r], o[c], 0, 0, 0);
            }
        } else {
            const float* vh = vcol + 4 * h * ATT_C;
            f32x4 vr[4];
            vr[0] = *reinterpret_cast<const f32x4*>(vh + 0 * ATT_C);
            vr[1] = *reinterpret_cast<const f32x4*>(vh + 1 * ATT_C);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (r + 2 < 16) {
                    vr[(r + 2) & 3] = *reinterpret_cast<const f32x4*>(vh + (((r + 2) & 3) + 8 * ((r + 2) >> 2)) * ATT_C);
                    vr[(r + 3) & 3] = *reinterpret_cast<const f32x4*>(vh + (((r + 3) & 3) + 8 * ((r + 3) >> 2)) * ATT_C);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[r & 3][c], x[r], o[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[(r + 1) & 3][c], x[r + 1], o[c], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: o[c][r] = O^T[channel 4*i+c][query l31], i = (r&3)+8(r>>2)+4h ------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int query = qb * ATT_BQ + wave * 32 + l31;
    if (query < N) {
        if (a.nsplit == 1) {
            float* dst = a.msg + ((size_t)b * NS + query) * ATT_C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                f32x4 v = {o[0][r] / l_tot, o[1][r] / l_tot, o[2][r] / l_tot, o[3][r] / l_tot};
                *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
            }
        } else {
            const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + query;
            float* dst = a.part_o + slot * ATT_C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                f32x4 v = {o[0][r], o[1][r], o[2][r], o[3][r]};
                *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
            }
            if (h == 0) {
                a.part_ml[slot * 2 + 0] = m_run;
                a.part_ml[slot * 2 + 1] = l_tot;
            }
        }
    }
}

// merge the per-split partials: thread -> (query, 4 channels)
__global__ __launch_bounds__(256) void attention_combine_kernel(AttArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const long long query = gid >> 5;
    const int c4 = (int)(gid & 31) * 4;
    if (query >= (a.nvalid ? a.nvalid[b] : a.N)) return;
    float mmax = -INFINITY;
    for (int sp = 0; sp < a.nsplit; ++sp) {
        const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + query;
        mmax = fmaxf(mmax, a.part_ml[slot * 2]);
    }
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < a.nsplit; ++sp) {
        const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + query;
        const float w = __builtin_amdgcn_exp2f(a.part_ml[slot * 2] - mmax);
        L = fmaf(a.part_ml[slot * 2 + 1], w, L);
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.part_o + slot * ATT_C + c4);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[c], w, acc[c]);
    }
    const float rL = 1.0f / L;             // as merge_partials.h
    f32x4 out = {acc[0] * rL, acc[1] * rL, acc[2] * rL, acc[3] * rL};
    *reinterpret_cast<f32x4*>(a.msg + ((size_t)b * a.N + query) * ATT_C + c4) = out;
}

static int attention_npad(int N) { return (int)round_up(N, ATT_BQ); }

int launch_attention_combine(const AttArgs& a, int bs, hipStream_t st) {
    const long long threads = (long long)a.N * 32;
    hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((threads + 255) / 256), bs), dim3(256), 0, st, a);
    return check_launch("pdsc_sc_attention(combine)");
}

}  // namespace pdsc

extern "C" int pdsc_attention_default_split(int bs, int N) {
    if (bs <= 0 || N <= 0) return -1;
    const int qblocks = pdsc::ceil_div(N, pdsc::ATT_BQ) * bs;
    const int tiles = pdsc::ceil_div(N, pdsc::ATT_BK);
    // two 4-wave workgroups per CU are co-resident (2 waves/SIMD): aim for ~512 workgroups, keep >= 4 tiles each
    int ns = (512 + qblocks / 2) / qblocks;
    if (ns < 1) ns = 1;
    const int cap = tiles / 4 > 1 ? tiles / 4 : 1;
    if (ns > cap) ns = cap;
    return ns;
}

extern "C" size_t pdsc_attention_scratch_bytes(int bs, int N, int nsplit) {
    if (bs <= 0 || N <= 0) return 0;
    if (nsplit <= 0) nsplit = pdsc_attention_default_split(bs, N);
    if (nsplit == 1) return 0;
    const size_t slots = (size_t)bs * nsplit * pdsc::attention_npad(N);
    return slots * (pdsc::ATT_C + 2) * sizeof(float);
}

namespace pdsc {
int launch_attention_fp32(const float* qkv, const float* compat, long long ld, float* msg, void* scratch, size_t scratch_bytes, int bs, int N,
                          int nsplit, const int* nvalid, hipStream_t st_in) {
    void* stream = (void*)st_in;

    PDSC_REQUIRE(qkv && compat && msg, "pdsc_sc_attention: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 0, "pdsc_sc_attention: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(ld >= pdsc::round_up(N, pdsc::ATT_BK) && ld % 4 == 0,
                 "pdsc_sc_attention: ld=%lld must be a multiple of 4 and >= N rounded up to 32", ld);
    const int tiles = pdsc::ceil_div(N, pdsc::ATT_BK);
    if (nsplit <= 0) nsplit = pdsc_attention_default_split(bs, N);
    if (nsplit > tiles) nsplit = tiles;
    const size_t need = pdsc_attention_scratch_bytes(bs, N, nsplit);
    if (need > 0 && (!scratch || scratch_bytes < need)) {
        pdsc::set_error("pdsc_sc_attention: scratch %zu < %zu bytes", scratch_bytes, need);
        return PDSC_ERR_WORKSPACE;
    }
    pdsc::AttArgs a{};
    a.qkv = qkv; a.compat = compat; a.ld = ld; a.msg = msg;
    a.N = N; a.Npad = pdsc::attention_npad(N); a.nsplit = nsplit; a.num_tiles = tiles;
    a.nvalid = nvalid;
    a.part_o = (float*)scratch;
    a.part_ml = a.part_o ? a.part_o + (size_t)bs * nsplit * a.Npad * pdsc::ATT_C : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = 4 * pdsc::ATT_TILE_FLOATS * sizeof(float);   // 64 KiB
    {
        int rc_lds = pdsc::ensure_dynamic_lds(reinterpret_cast<const void*>(&pdsc::sc_attention_kernel<1>), lds_bytes, "pdsc_sc_attention(dynamic LDS)");
        if (rc_lds != PDSC_OK) return rc_lds;
    }
    dim3 grid(pdsc::ceil_div(N, pdsc::ATT_BQ), nsplit, bs);
    pdsc::profile_mark_begin(PDSC_PROF_ATTENTION, st);
#ifdef PDSC_EXPERIMENTS
    const char* env_variant = pdsc::env_str("PDSC_ATT_VARIANT");       // A/B knob; default = shipped variant
    if ((env_variant ? atoi(env_variant) : PDSC_ATT_DEFAULT_VARIANT) == 0) {
        const int rc0 = pdsc::ensure_dynamic_lds(reinterpret_cast<const void*>(&pdsc::sc_attention_kernel<0>), lds_bytes, "pdsc_sc_attention(dynamic LDS)");
        if (rc0 != PDSC_OK) return rc0;
        hipLaunchKernelGGL(pdsc::sc_attention_kernel<0>, grid, dim3(256), lds_bytes, st, a);
    } else
#endif
    hipLaunchKernelGGL(pdsc::sc_attention_kernel<1>, grid, dim3(256), lds_bytes, st, a);
    pdsc::profile_mark_end(PDSC_PROF_ATTENTION, st);
    int rc = pdsc::check_launch("pdsc_sc_attention");
    if (rc != PDSC_OK) return rc;
    if (nsplit > 1) {
        rc = pdsc::launch_attention_combine(a, bs, st);
    }
    return rc;
}
}  // namespace pdsc

extern "C" int pdsc_sc_attention(const float* qkv, const float* compat, long long ld, float* msg, void* scratch,
                                 size_t scratch_bytes, int bs, int N, int nsplit, void* stream) {
    return pdsc::launch_attention_fp32(qkv, compat, ld, msg, scratch, scratch_bytes, bs, N, nsplit, nullptr, (hipStream_t)stream);
}
