// a-3, split-precision attention, "wide" variant: ONE wavefront per SIMD, 64 queries per wavefront.
//
// Same arithmetic, operand streams, LDS images, key-split partials and softmax bookkeeping as sc_attention_split_kernel<8>
// (attention_split.hip) -- a workgroup is still 256 queries x a range of 32-key tiles, one workgroup per CU -- but the 256
// queries are held by 4 waves of 64 (two 32-query groups per wave) instead of 8 waves of 32:
//   * every K / V^T fragment read from LDS feeds the MFMAs of BOTH groups: half the ds_read_b128 per MFMA;
//   * the two groups are independent instruction streams inside one wave: group 0's softmax VALU work sits between
//     group 1's MFMAs (and vice versa) by construction, instead of relying on two co-resident waves drifting apart;
//   * 4 waves instead of 8 meet at the per-tile barrier.
// Registers: O^T 2 x 64, Q hi/lo 2 x 64, S^T 2 x 16, logits 2 x 16, P hi/lo 2 x 16 ... ~400 of the 512-entry unified
// VGPR/AGPR file (one wave per SIMD).
#ifdef PDSC_EXPERIMENTS      // opt-in record (13 % slower than the shipped kernel): experiments builds only
#include <stdlib.h>
#include "attention_common.h"
#include "split_layout.h"

namespace pdsc {

constexpr int WD_NW = 4, WD_G = 2;                       // waves per workgroup, 32-query groups per wave
constexpr int WD_QROWS = WD_NW * WD_G * 32;              // 256 queries per workgroup
constexpr int WD_K_BYTES = 2 * SPL_K_PLANE;        // Kh | Kl   16 KiB
constexpr int WD_V_BYTES = 2 * SPL_V_PLANE;        // Vh | Vl   16 KiB
constexpr int WD_CSTAGE = WD_QROWS * 128;                // compat slice of the 256 queries for one tile: 32 KiB
constexpr float WD_RESCALE_THR = 8.0f;

template <int BYTES>
__device__ __forceinline__ void wd_issue_linear(__amdgpu_buffer_rsrc_t rsrc, int src_off, unsigned char* dst, int wave, unsigned lane16) {
    constexpr int PIECES = BYTES / 1024;
#pragma unroll
    for (int u = 0; u < (PIECES + WD_NW - 1) / WD_NW; ++u) {
        const int i = wave + WD_NW * u;
        if ((u + 1) * WD_NW <= PIECES || i < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + i * 1024), 16, lane16, src_off + i * 1024, 0, 0);
    }
}

__global__ __launch_bounds__(WD_NW * 64, 1) void sc_attention_wide_kernel(AttSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const Ks = lds;                                  // 2 x 17 KiB
    unsigned char* const Vs = lds + 2 * WD_K_BYTES;                 // 2 x 20 KiB
    unsigned char* const Cs = Vs + 2 * WD_V_BYTES;                  // 2 x 32 KiB
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int N = a.N;

    int qb, grp;
    {
        const int id = blockIdx.x, W = gridDim.x;
        const int rank = (W & 7) == 0 ? (id & 7) * (W >> 3) + (id >> 3) : id;     // same XCD placement as the 8-wave kernel
        grp = rank / a.nq;
        qb = rank % a.nq;
    }
    const int sp = grp % a.nsplit, b = grp / a.nsplit;
    const int per = a.num_tiles / a.nsplit, rem = a.num_tiles % a.nsplit;
    const int kt0 = sp * per + min(sp, rem);
    const int kt1 = kt0 + per + (sp < rem ? 1 : 0);

    const __amdgpu_buffer_rsrc_t kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.kv + (size_t)b * a.num_tiles * SPL_TILE_STRIDE), 0, a.num_tiles * SPL_TILE_STRIDE, 0x00020000);
    const int q_first = qb * WD_QROWS;
    const int q_rows = min(WD_QROWS, N - q_first);
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const float*)a.compat + ((size_t)b * N + q_first) * a.ld), 0, (int)((unsigned)q_rows * (unsigned)a.ld * 4u), 0x00020000);
    const unsigned lane16 = lane * 16;
    // compat slice: this wave fetches its own 64 rows, 8 pieces of 8 rows x 128 B; chunk c of a row stored at c ^ ((row >> 1) & 7)
    unsigned coff[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int row = wave * 64 + 8 * u + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        coff[u] = (unsigned)min(row, q_rows - 1) * (unsigned)a.ld * 4u + 16u * c;
    }
    constexpr int KPIECES = WD_K_BYTES / 1024, PIECES = SPL_TILE_BYTES / 1024;       // 17, 37
    constexpr int KV_SLOTS = (PIECES + WD_NW - 1) / WD_NW;                           // 10
    constexpr int DMA_SLOTS = 8 + KV_SLOTS;                                          // 18 per wave per iteration
    const int compat_nt = a.compat_nt;
    auto dma_slot = [&](int kt, int st, int slot) {
        if (slot < 8) {
            unsigned char* dst = Cs + st * WD_CSTAGE + (wave * 8 + slot) * 1024;
            if (compat_nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lptr_t)dst, 16, coff[slot], (kt + 2) * (SPL_BK * 4), 0, 2);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lptr_t)dst, 16, coff[slot], (kt + 2) * (SPL_BK * 4), 0, 0);
        } else {
            const int i = min(wave + WD_NW * (slot - 8), PIECES - 1);
            const bool isk = i < KPIECES;
            unsigned char* dst = isk ? Ks + st * WD_K_BYTES + i * 1024 : Vs + (st ^ 1) * WD_V_BYTES + (i - KPIECES) * 1024;
            const int src = (isk ? kt + 2 : kt + 1) * SPL_TILE_STRIDE + i * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(kv_rsrc, (lptr_t)dst, 16, lane16, src, 0, 0);
        }
    };
    auto dma_k = [&](int kt) { wd_issue_linear<WD_K_BYTES>(kv_rsrc, kt * SPL_TILE_STRIDE + SPL_KH, Ks + ((kt - kt0) & 1) * WD_K_BYTES, wave, lane16); };
    auto dma_v = [&](int kt) { wd_issue_linear<WD_V_BYTES>(kv_rsrc, kt * SPL_TILE_STRIDE + SPL_VH, Vs + ((kt - kt0) & 1) * WD_V_BYTES, wave, lane16); };
    auto dma_c = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lptr_t)(Cs + ((kt - kt0) & 1) * WD_CSTAGE + (wave * 8 + u) * 1024), 16, coff[u],
                                                     kt * (SPL_BK * 4), 0, 0);
    };

    dma_k(kt0); dma_c(kt0);
    if (kt0 + 1 < kt1) { dma_k(kt0 + 1); dma_c(kt0 + 1); }
    dma_v(kt0);
    bf16x8 qh[WD_G][8], ql[WD_G][8];
#pragma unroll
    for (int g = 0; g < WD_G; ++g) {
        const int qrow = min(q_first + wave * 64 + g * 32 + l31, N - 1);
        const __bf16* qsrc = a.qs + ((size_t)b * N + qrow) * SPL_Q_LD + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qh[g][j] = *reinterpret_cast<const bf16x8*>(qsrc + 16 * j);
            ql[g][j] = *reinterpret_cast<const bf16x8*>(qsrc + PDSC_CHANNELS + 16 * j);
        }
    }
    f32x16 o[WD_G][4];
#pragma unroll
    for (int g = 0; g < WD_G; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][c][r] = 0.f;
    float m_run[WD_G] = {0.f, 0.f}, l_run[WD_G] = {0.f, 0.f};

    const int koff = l31 * 16 + 512 * h;             // chunk-major images (split_layout.h)
    const int voff = l31 * 16 + 2048 * h;
    const int csw = (l31 >> 1) & 7;                       // (row >> 1) & 7 with row = 32-aligned base + l31
    int crow_off[WD_G];
#pragma unroll
    for (int g = 0; g < WD_G; ++g) crow_off[g] = (wave * 64 + g * 32 + l31) * 128;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto mask_tail = [&](int kt, float (&tl)[16]) {
        if ((kt + 1) * SPL_BK > N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * SPL_BK + (r & 3) + 8 * (r >> 2) + 4 * h;
                tl[r] = key < N ? tl[r] : -INFINITY;
            }
        }
    };
    auto row_max = [&](const float (&tl)[16]) {
        float m = tl[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, tl[r]);
        return fmaxf(m, __shfl_xor(m, 32, 64));
    };

    float tl[WD_G][16];
    f32x16 sacc[WD_G];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 fh = *reinterpret_cast<const bf16x8*>(Ks + SPL_KH + koff + 1024 * j);
            const bf16x8 fl = *reinterpret_cast<const bf16x8*>(Ks + SPL_KL + koff + 1024 * j);
#pragma unroll
            for (int g = 0; g < WD_G; ++g) {
                sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, qh[g][j], j == 0 ? zero16 : sacc[g], 0, 0, 0);
                sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, ql[g][j], sacc[g], 0, 0, 0);
                sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, qh[g][j], sacc[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < WD_G; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 cc = *reinterpret_cast<const f32x4*>(Cs + crow_off[g] + (((2 * q + h) ^ csw) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) tl[g][4 * q + e] = cc[e] * sacc[g][4 * q + e];
            }
            mask_tail(kt0, tl[g]);
            m_run[g] = row_max(tl[g]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tl[g][r] -= m_run[g];
        }
    }

    for (int kt = kt0; kt < kt1; ++kt) {
        const int st = (kt - kt0) & 1;
        const bool has_next = kt + 1 < kt1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- phase A: S^T(kt+1) for both groups | P(kt) = exp2(tl), hi/lo split ------------------------------------
        float psum[WD_G] = {0.f, 0.f};
        bf16x8 ph[WD_G][2], pl[WD_G][2];
        {
            const unsigned char* K = Ks + (st ^ 1) * WD_K_BYTES;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8 fh = *reinterpret_cast<const bf16x8*>(K + SPL_KH + koff + 1024 * j);
                const bf16x8 fl = *reinterpret_cast<const bf16x8*>(K + SPL_KL + koff + 1024 * j);
#pragma unroll
                for (int g = 0; g < WD_G; ++g) {
                    sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, qh[g][j], j == 0 ? zero16 : sacc[g], 0, 0, 0);
                    sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, ql[g][j], sacc[g], 0, 0, 0);
                    sacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, qh[g][j], sacc[g], 0, 0, 0);
                    dma_slot(kt, st, 2 * j + g);
#pragma unroll
                    for (int r = 2 * j; r < 2 * j + 2; ++r) {
                        const float p = __builtin_amdgcn_exp2f(tl[g][r]);
                        psum[g] += p;
                        __bf16 hi, lo;
                        split_bf16(p, hi, lo);
                        ph[g][r >> 3][r & 7] = hi;
                        pl[g][r >> 3][r & 7] = lo;
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < WD_G; ++g) l_run[g] += psum[g];

        // ---- phase B: O^T += V^T P^T for both groups | logits of tile kt+1 -------------------------------------------
        {
            const unsigned char* V = Vs + st * WD_V_BYTES;
            const unsigned char* Cn = Cs + (st ^ 1) * WD_CSTAGE;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = u >> 1, j = u & 1;
                const int vo = c * 512 + voff + 4096 * j;
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(V + vo);
                const bf16x8 vl = *reinterpret_cast<const bf16x8*>(V + SPL_V_PLANE + vo);
#pragma unroll
                for (int g = 0; g < WD_G; ++g) {
                    o[g][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[g][j], o[g][c], 0, 0, 0);
                    o[g][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[g][j], o[g][c], 0, 0, 0);
                    o[g][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[g][j], o[g][c], 0, 0, 0);
                    // logits of group g: chunk q of its compat row, one chunk per (u, g) slot where q = (u >> 1) and the slot
                    // parity matches the group (u even -> group 0, u odd -> group 1)
                    if ((u & 1) == g) {
                        const int q = u >> 1;
                        const f32x4 cc = *reinterpret_cast<const f32x4*>(Cn + crow_off[g] + (((2 * q + h) ^ csw) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) tl[g][4 * q + e] = fmaf(cc[e], sacc[g][4 * q + e], -m_run[g]);
                    }
                }
                if (16 + u < DMA_SLOTS) dma_slot(kt, st, 16 + u);
            }
        }
        if (has_next) {
#pragma unroll
            for (int g = 0; g < WD_G; ++g) {
                mask_tail(kt + 1, tl[g]);
                const float mloc = row_max(tl[g]);
                if (!__all(mloc <= WD_RESCALE_THR)) {
                    const float delta = mloc > WD_RESCALE_THR ? mloc : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    l_run[g] *= alpha;
                    m_run[g] += delta;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tl[g][r] -= delta;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[g][c][r] *= alpha;
                }
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int OPITCH = PDSC_CHANNELS * 4 + 16;
#pragma unroll
    for (int g = 0; g < WD_G; ++g) {
        const float l_tot = l_run[g] + __shfl_xor(l_run[g], 32, 64);
        const int q0 = q_first + wave * 64 + g * 32;
        unsigned char* const patch = lds + (wave * WD_G + g) * (32 * OPITCH);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {o[g][c][4 * q], o[g][c][4 * q + 1], o[g][c][4 * q + 2], o[g][c][4 * q + 3]};
                if (a.nsplit == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[g][c][4 * q + e] / l_tot;
                }
                *reinterpret_cast<f32x4*>(patch + l31 * OPITCH + 128 * c + 32 * q + 16 * h) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* const base = a.nsplit == 1 ? a.msg + (size_t)b * N * PDSC_CHANNELS
                                          : a.part_o + ((size_t)b * a.nsplit + sp) * a.Npad * PDSC_CHANNELS;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = 2 * it + h, piece = l31;
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + r * OPITCH + 16 * piece);
            if (q0 + r < N) *reinterpret_cast<f32x4*>(base + (size_t)(q0 + r) * PDSC_CHANNELS + 4 * piece) = v;
        }
        if (a.nsplit != 1 && h == 0 && q0 + l31 < N) {
            const size_t slot = ((size_t)b * a.nsplit + sp) * a.Npad + q0 + l31;
            a.part_ml[slot * 2 + 0] = m_run[g];
            a.part_ml[slot * 2 + 1] = l_tot;
        }
    }
}

int launch_attention_wide(const AttSplitArgs& a, unsigned grid, hipStream_t st) {
    const size_t stage_bytes = 2 * (size_t)(SPL_TILE_BYTES + WD_CSTAGE);                       // 138 KiB
    const size_t patch_bytes = (size_t)WD_NW * WD_G * 32 * (PDSC_CHANNELS * 4 + 16);           // 132 KiB
    const size_t lds_bytes = stage_bytes > patch_bytes ? stage_bytes : patch_bytes;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sc_attention_wide_kernel), lds_bytes, "pdsc_sc_attention_split(wide, dynamic LDS)");
    if (rc != PDSC_OK) return rc;
    profile_mark_begin(PDSC_PROF_ATTENTION, st);
    hipLaunchKernelGGL(sc_attention_wide_kernel, dim3(grid), dim3(WD_NW * 64), lds_bytes, st, a);
    profile_mark_end(PDSC_PROF_ATTENTION, st);
    return check_launch("pdsc_sc_attention_split(wide)");
}

}  // namespace pdsc
#endif  // PDSC_EXPERIMENTS
