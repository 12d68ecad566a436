/*
 * pointdsc_hip.h -- C ABI of libpointdsc_hip.so (gfx950 / MI355X).
 *
 * The reference (XuyangBai/PointDSC) has no FFI of its own: its hot path is the Python method
 * PointDSC.forward(data) in testing mode (reference models/PointDSC.py:128-197) built from stock ATen
 * calls.  This header is the boundary a maintainer binds with ctypes: one entry
 * point per reference stage (so each stage can be parity-checked on its own) plus one whole-path call.
 * Every entry point cites the reference lines it replaces.
 *
 * Two tiers (r05):
 *   PRODUCT BOUNDARY -- what a caller of the reference's module needs: pdsc_forward_testing (+ _ragged, _streams),
 *     pdsc_forward_validation; the weight packers pdsc_wpack_floats / _offset, pdsc_wsplit_bytes / _offset / _build; the workspace
 *     queries pdsc_workspace_bytes / _offset; pdsc_encoder_range_probe; pdsc_version / pdsc_last_error; and, for the callers either
 *     side of the path (SURVEY.md section 8 f-2 .. f-4), pdsc_match_* / pdsc_select_correspondences / pdsc_build_corr_pos,
 *     pdsc_sm_baseline*, pdsc_cal_confidence, pdsc_eval_stats.
 *   STAGE LEVEL -- one entry point per reference stage (sections a-1 .. a-11 below), the plan / size queries that go with them,
 *     pdsc_selftest_* and the diagnostic hooks.  The forward does not go through them (it calls the same launchers directly); they
 *     exist so that every stage can be parity-checked on its own (tests/test_gpu_parity.py), for the tools, and for a maintainer
 *     who keeps the reference's module and swaps single stages.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP, fp32 unless stated), caller-owned, never retained;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it: no host
 *     synchronisation, no allocation (graph-capturable).  Results depend on the arguments only.  Process-global
 *     state is limited to: the per-(kernel, device) dynamic-LDS opt-in table (mutex-protected), the per-device event that orders
 *     opt-in register-resident spectral-matching launches (mutex-protected), the error string
 *     (thread-local), and the opt-in DIAGNOSTIC hooks -- pdsc_profile_* event timing, pdsc_attention_trace,
 *     pdsc_layer_trace -- which are process-wide switches and not thread-safe: use them from one thread.
 *     The product library reads NO environment variable: everything that selects a kernel or changes arithmetic is a field
 *     of pdsc_config or an argument.  The PDSC_* tuning / A-B knobs DESIGN.md lists exist in experiments builds only
 *     (-DPDSC_EXPERIMENTS, `python -m pointdsc_amd.build --experiments` -> libpointdsc_hip_exp.so; pdsc_experiments_enabled());
 *   - tensors are dense row-major; `bs` = number of correspondence sets (pairs), `N` = correspondences
 *     per pair, `C` = 128 channels, `S` = number of seeds, `k` = neighbours per seed;
 *   - bs > 1 means bs independent pairs, i.e. the reference called once per pair (the reference's
 *     testing mode asserts bs == 1, models/PointDSC.py:210,414);
 *   - return value 0 = enqueued, < 0 = rejected (pdsc_last_error() tells why); nothing is enqueued
 *     on rejection.
 */
#ifndef POINTDSC_HIP_H
#define POINTDSC_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDSC_VERSION 8
#define PDSC_CHANNELS 128        /* num_channels of every released PointDSC model */
#define PDSC_MAX_K 64            /* neighbours per seed handled by one wavefront   */
#define PDSC_MAX_POWER_ITERS 32

enum pdsc_status {
    PDSC_OK = 0,
    PDSC_ERR_ARG = -1,           /* bad argument (null pointer, unsupported size) */
    PDSC_ERR_WORKSPACE = -2,     /* workspace too small                           */
    PDSC_ERR_LAUNCH = -3         /* HIP launch error                              */
};

/* Constructor arguments of reference PointDSC.__init__ (models/PointDSC.py:81-91) that the path uses. */
typedef struct pdsc_config {
    int in_dim;              /* 6 (1..16: datasets/ThreeDMatch.py:299-312 builds 6 / 9 / 12) */
    int num_layers;          /* 12 in the released snapshots                   */
    int num_channels;        /* must be PDSC_CHANNELS                          */
    int num_iterations;      /* power-iteration cap, 10                        */
    int k;                   /* neighbours per seed, 40                        */
    int refine_iters;        /* 20 (models/PointDSC.py:416-418)                */
    float inlier_threshold;  /* hypothesis-scoring threshold (:328,:335)       */
    float nms_radius;        /* NMS radius R (:174)                            */
    float refine_threshold;  /* 0.10 if inlier_threshold == 0.10 else 1.2 (:415-418) */
    int attention_precision; /* enum pdsc_attention_precision: how the two N x N x C contractions are evaluated */
    int compat_format;       /* enum pdsc_compat_format: how the forward stores the N x N spatial-consistency matrix  */
    int layer_gemm;          /* enum pdsc_layer_gemm: arithmetic of the fc_message / PointCN GEMMs in the fused layer kernel */
    int att_leaves;          /* enum pdsc_att_leaves: summation tree of the attention's key dimension (split-precision modes) */
} pdsc_config;

/* How the N keys of a query are summed (models/PointDSC.py:41-42: one softmax-weighted sum per query) when the key range is
 * cut so that one pair can fill the chip.  A LEAF is a run of 32-key tiles accumulated from a fresh online-softmax state;
 * the fused layer kernel merges the partials (O, m, l) in order (weights exp2(m - max m), one reciprocal) while it loads them.
 *   PER_LAUNCH (0): one partial per key split, the split planned per launch from (batch, N): the bits of a pair depend on how
 *                   many pairs share its launch (all within the contract; the r01-r04 results).
 *   CANONICAL  (1): pdsc_attention_leaf_count(N) leaves, a function of N alone; the launch plan only decides WHICH workgroup
 *                   computes a leaf (the key split is a divisor of the leaf count), never the arithmetic -- a pair's result is
 *                   bit-identical at every batch size, on one GPU or sharded over eight.  Split-precision attention with the
 *                   H3 layer kernel only (the other arithmetic modes keep PER_LAUNCH).        [default of the Python module]
 *   2 .. 8        : that many leaves (tuning). */
enum pdsc_att_leaves { PDSC_LEAVES_PER_LAUNCH = 0, PDSC_LEAVES_CANONICAL = 1 };

/* Arithmetic of the point-wise GEMMs whose results land on the residual stream (fc1..fc3 of fc_message, PointCN;
 * models/PointDSC.py:12-23,56-61) inside the fused layer kernel (split-precision attention modes).  H3 runs on
 * layer_h3_kernel, or on the bit-identical layer_h3_coop_kernel for launches of at most 2560 tiles
 * (pdsc_layer_h3_uses_coop(bs, N) == 1); F32 on layer_wave_kernel, or on the workgroup-per-tile kernel for small problems
 * (pdsc_layer_prefers_block(bs, N) == 1):
 *   F32: v_mfma_f32_32x32x2_f32, exact fp32 products (600 MFMAs x 64 matrix-pipe cycles per 32-point tile).
 *   H3 : every fp32 operand as fp16 hi + fp16 lo' (lo' = (x - hi) * 2048), product = hi*hi + (hi*lo' + lo'*hi) / 2048 on
 *        v_mfma_f32_32x32x16_f16 with fp32 accumulation: ~2^-21 relative error per product (as the fp16 hi/lo split of the
 *        attention operands), 216 MFMAs x 32 cycles per tile.  Operands must stay inside the fp16 range
 *        (|x| < 65504); the network's activations and weights are O(1). */
enum pdsc_layer_gemm { PDSC_LAYER_GEMM_F32 = 0, PDSC_LAYER_GEMM_H3 = 1 };

/* Order of the [rows][C] fp32 matrices handed from the attention to the fused layer kernel (key-split partials) and from
 * one fused layer launch to the next (featB = the residual): plain rows, or point-fragment order (PF) -- rows in tiles of 32,
 * a tile = [q = 0..15][lane = 0..63][4 floats] with lane (l31 = lane & 31, h = lane >> 5) holding channels 8q + 4h .. + 3 of
 * row l31: exactly the registers of the producing and of the consuming wavefront, so every store / load instruction moves
 * 1 KiB of consecutive memory and neither side transposes through LDS.  PF buffers are padded to whole tiles per pair
 * (ceil(N / 32) * 32 rows); padding rows hold copies of the pair's last row.  Only the H3 layer kernel reads / writes PF. */
enum pdsc_partial_layout { PDSC_PARTIALS_ROWS = 0, PDSC_PARTIALS_PF = 1 };
enum pdsc_layer_io { PDSC_IO_PARTIALS_PF = 1, PDSC_IO_RES_PF = 2, PDSC_IO_FEATB_PF = 4 };

/* Storage of the spatial-consistency matrix between its build and the 12 attention launches that stream it
 * (the split-precision modes only; PDSC_ATT_FP32 always uses fp32 storage):
 *   U16: unorm16, value = round(c * 65535) / 65535 with c evaluated on the hardware's 1-ulp square root (r03: the exact
 *        sqrt / divide of the fp32 matrix made the build instruction-bound) -- within 2 units (3e-5) of the fp32 matrix,
 *        the diagonal exactly 1, symmetric bit for bit (r05: with the attention operands as fp16 pairs this is the largest
 *        arithmetic difference left between the default and the exact-fp32 configuration); half the HBM stream (2 N^2 instead of 4 N^2 bytes per layer per pair), half the workspace; +4.6 % pairs/s
 *        at N=5000 (tools/ab_forward.py).
 *   F32: the fp32 matrix of pdsc_spatial_compat, bit-identical to the reference's.
 * The Python module defaults to U16 (parity census equal to F32's); the C struct has no default. */
enum pdsc_compat_format { PDSC_COMPAT_F32 = 0, PDSC_COMPAT_U16 = 1 };

/* Arithmetic of the attention contractions (models/PointDSC.py:39,42).  Softmax, accumulation, outputs: fp32 in both.
 *   FP16X3: every fp32 operand split into fp16 hi + fp16 lo (lo = x - hi, unscaled: 22 significant bits down to fp16's
 *           denormal floor, which the gfx950 f16 MFMA honours), three v_mfma_f32_32x32x16_f16 per operand pair
 *           (hi*hi + hi*lo + lo*hi) into one fp32 accumulator: ~2^-21 relative error per product.  Operands must stay inside the
 *           fp16 range (|q|, |k|, |v| < 65504; the softmax weights are kept in (0, 32768] by construction): the Python module
 *           probes the first forward of a checkpoint and falls back to FP32 outside it (pdsc_encoder_range_probe).
 *           Also used for the q|k|v projection (its results only feed the attention); the GEMMs whose results
 *           land on the residual stream (PointCN, fc_message) follow pdsc_config.layer_gemm.           [default]
 *           Rounds 1-4 split into bf16 pairs (2^-16 per product, fp32's range): on trained-like KITTI weights, whose
 *           confidence logits reach 33, that left the logits up to 0.27 from the reference's; the fp16 pairs leave 0.04
 *           (exact fp32: 0.017) for 1.7 % of the pairs/s (same-box A/B, profiles/r05_j_ab_split16_summary.txt).
 *   FP32  : v_mfma_f32_32x32x2_f32, exact fp32 products, 16/3 x the matrix-pipe time.
 *   FP16X3_ALL: the point-wise GEMMs too (pdsc_layer_fused_x3): their error lands on the residual stream un-averaged.
 *           A/B record: accepted by experiments builds only.
 * BREAKING in PDSC_VERSION 8: the names PDSC_ATT_BF16X3 / PDSC_ATT_BF16X3_ALL of rounds 1-4 are GONE (they were aliases of values 0 / 2
 * in version 7).  Those modes split into bf16 pairs and had fp32's range; values 0 / 2 split into fp16 pairs and do not -- a C caller
 * that still spells the old name must not compile into different arithmetic silently.  Range contract of FP16X3: every forward
 * carries a device-side sentinel (workspace entry "range_flag", [bs] u32, pdsc_workspace_offset): a pair any of whose activations
 * reached 65504 on its way into an fp16 pair has a non-zero word there after the call AND its final_trans is returned as NaN. */
enum pdsc_attention_precision { PDSC_ATT_FP16X3 = 0, PDSC_ATT_FP32 = 1, PDSC_ATT_FP16X3_ALL = 2 };
/* Optional: where the testing forwards enqueued by THIS thread from now on also leave their range words -- host_words = [bs] u32 of
 * pinned, device-mapped host memory (hipHostMalloc; checked), written by the forward's last launch: a caller that waits for the
 * stream anyway reads them without a device-to-host copy of its own.  NULL switches it off.  Thread-local, like pdsc_last_error. */
int pdsc_set_range_report(unsigned int* host_words);

/* ---- packed weights --------------------------------------------------------------------------
 * One flat fp32 buffer holding the model with BatchNorm (eval) folded into the preceding conv and
 * log2(e)/sqrt(C) folded into the Q projection.  Offsets (in floats) come from pdsc_wpack_offset so
 * that the host packer (pointdsc_amd/model.py) and the kernels cannot disagree.
 * All matrices are [out][in] row-major exactly like Conv1d.weight[:, :, 0]. */
enum pdsc_wsection {
    PDSC_W_LAYER0_W = 0,  /* [C][16]  in_dim (<= 16) zero-padded to 16 (encoder.layer0)       */
    PDSC_W_LAYER0_B,      /* [C]                                                             */
    PDSC_W_PCN_W,         /* per layer [C][C]     PointCN conv + BN folded                   */
    PDSC_W_PCN_B,         /* per layer [C]                                                   */
    PDSC_W_QKV_W,         /* per layer [3C][C]    rows: q (pre-scaled), k, v                 */
    PDSC_W_QKV_B,         /* per layer [3C]                                                  */
    PDSC_W_FC1_W,         /* per layer [C/2][C]   fc_message.0 + BN .1 folded                */
    PDSC_W_FC1_B,         /* per layer [C/2]                                                 */
    PDSC_W_FC2_W,         /* per layer [C/2][C/2] fc_message.3 + BN .4 folded                */
    PDSC_W_FC2_B,         /* per layer [C/2]                                                 */
    PDSC_W_FC3_W,         /* per layer [C][C/2]   fc_message.6                               */
    PDSC_W_FC3_B,         /* per layer [C]                                                   */
    PDSC_W_CLS1_W,        /* [32][C]   classification.0                                      */
    PDSC_W_CLS1_B,        /* [32]                                                            */
    PDSC_W_CLS2_W,        /* [32][32]  classification.2                                      */
    PDSC_W_CLS2_B,        /* [32]                                                            */
    PDSC_W_CLS3_W,        /* [32]      classification.4                                      */
    PDSC_W_CLS3_B,        /* [1]                                                             */
    PDSC_W_SIGMA,         /* [1]  learned feature-compat sigma   (models/PointDSC.py:97)     */
    PDSC_W_SIGMA_SPAT,    /* [1]  spatial sigma_d                (models/PointDSC.py:98)     */
    PDSC_W_NUM_SECTIONS
};

int         pdsc_version(void);
const char* pdsc_last_error(void);
/* 1 in experiments builds (A/B knobs and the opt-in record kernels compiled in), 0 in the product library */
int         pdsc_experiments_enabled(void);

/* total floats of the packed buffer / offset of a section (layer ignored for per-model sections);
 * returns -1 on bad arguments */
long long pdsc_wpack_floats(const pdsc_config* cfg);
long long pdsc_wpack_offset(const pdsc_config* cfg, int section, int layer);

/* leading dimension (floats) the compat matrix rows must have for N correspondences */
long long pdsc_compat_ld(int N);

/* bytes of scratch pdsc_forward_testing needs (compat matrix included) */
size_t pdsc_workspace_bytes(const pdsc_config* cfg, int bs, int N, int num_seeds);

/* ---- a-1  spatial-consistency matrix -----------------------------------------------------------
 * replaces models/PointDSC.py:150-153.
 *   src_dist[i][j] = ||src_i - src_j||_2 ,  compat[i][j] = max(0, 1 - (src_dist - tgt_dist)^2 / sigma_spat^2)
 * compat: [bs][N][ld] (ld >= N, multiple of 4; columns N..ld-1 are written as 0).
 * src_dist: optional (may be NULL) [bs][N][ld]; the fused path never materialises it. */
int pdsc_spatial_compat(const float* src_keypts, const float* tgt_keypts, const float* sigma_spat,
                        float* compat, float* src_dist, long long ld, int bs, int N, void* stream);

/* Partials only (no merge: the fused layer kernel merges while it loads), compat in either storage format, partials in
 * either order (enum pdsc_partial_layout); nsplit as above (0 = the plan's), must come out > 1. */
int pdsc_sc_attention_split_partials(const void* q_split, const void* kv_tiles, const void* compat, int compat_format,
                                     long long ld, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                                     int partial_layout, void* stream);

/* unorm16 variant (enum pdsc_compat_format): compat_u16 [bs][N][ld] uint16, ld = pdsc_compat_ld(N) (multiple of 32),
 * value u = round(compat * 65535); inside every group of 32 columns, column 8g + 4h + e is stored at position
 * 16h + 4g + e (the order the attention kernel's accumulator holds the keys); columns >= N are 0.
 * Consumed by pdsc_sc_attention_split_u16 only. */
int pdsc_spatial_compat_u16(const float* src_keypts, const float* tgt_keypts, const float* sigma_spat,
                            unsigned short* compat_u16, long long ld, int bs, int N, void* stream);

/* Self-test hook for the two hand-rolled exact primitives of the compat kernel (correctly rounded sqrt, division
 * by a loop-invariant): sqrt_out[i] = sqrt(x[i]), div_out[i] = x[i] / divisor, both must equal the IEEE results. */
int pdsc_selftest_exact_math(const float* x, float divisor, float* sqrt_out, float* div_out, long long n, void* stream);
/* Test infrastructure: launches `workgroups` x 256 threads of a synthetic kernel that interleaves bf16 MFMAs with ordinary vector
 * work for `iters` loop iterations (2500 ~ 1 ms) -- the co-resident neighbour beside which packed fp32 instructions with operand
 * selects return wrong lanes (tools/pk_f32_repro.hip "mix").  The library ships no such instruction; the GPU
 * tests keep this neighbour on the chip while they check the entry points bit for bit.  sink: >= 1 float, never written. */
int pdsc_selftest_mfma_valu_neighbour(float* sink, int workgroups, int iters, void* stream);

/* ---- a-2  point-wise layers --------------------------------------------------------------------
 * replaces every Conv1d(kernel_size=1)[+BatchNorm1d(eval)][+ReLU] of models/PointDSC.py:12-23,54-61,107-113.
 *   Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] ) (+ residual[m][n])
 * X [M][ldx], W [Nout][K], Y [M][ldy]; K multiple of 8 and <= 128; ldx, ldy, ldr multiples of 4.
 * bias / residual may be NULL.  relu applies before the residual add (fc_message has no final ReLU). */
int pdsc_linear(const float* X, long long ldx, const float* W, const float* bias,
                const float* residual, long long ldr, float* Y, long long ldy,
                int M, int K, int Nout, int relu, void* stream);

/* encoder.layer0 (models/PointDSC.py:54,73): feat[m][c] = sum_d corr_pos[m][d] * W0[c][d] + b0[c];
 * corr_pos [M][in_dim] dense, W0 [C][8] (zero-padded), feat [M][C]. */
int pdsc_layer0(const float* corr_pos, int in_dim, const float* W0, const float* b0, float* feat,
                int M, void* stream);

/* Fused point-wise chain between two attention calls (one launch instead of five pdsc_linear calls):
 *   tail (msg != NULL):      feat = res + fc3(relu(fc2(relu(fc1(msg)))))            models/PointDSC.py:43-45
 *   head (featB_out != NULL): featB = relu(pcn(feat)); qkv = Wqkv featB + bqkv       models/PointDSC.py:75,36-38
 * tail only -> feat_out required; head only -> feat_in required.  All matrices [out][in], BN folded
 * (sections PDSC_W_FC1..FC3 of layer i, PDSC_W_PCN/QKV of layer i+1).  Buffers must not alias. */
int pdsc_layer_fused(const float* msg, const float* res, const float* feat_in, float* feat_out,
                     float* featB_out, float* qkv_out,
                     const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, const float* wp, const float* bp,
                     const float* wq, const float* bq, int M, void* stream);

/* Same chain, with the head additionally (or instead of qkv_out, which may then be NULL) emitting the fp16 hi/lo
 * operand streams of the split-precision attention (layout: pointdsc_amd/csrc/split_layout.h):
 *   q_split  [bs*N][256] fp16 (hi | lo), pdsc_split_q_bytes(bs, N) bytes;
 *   kv_tiles [bs][ceil(N/32)][37 KiB] (a 32 KiB image per tile of 32 keys, images 37 KiB apart), pdsc_split_kv_bytes(bs, N) bytes.
 * Rows are bs pairs of N points (a 32-point tile never straddles two pairs).
 * The tail input is either `msg` (merged rows) or the un-merged key-split partials (`part_o`, `part_ml`, nsplit, Npad)
 * exactly as pdsc_sc_attention_split leaves them in its scratch when called with msg == NULL: the merge then happens
 * while the tile is loaded (no combine launch, no round trip of msg through HBM).
 * wq_split (optional): the q|k|v weights as fp16 hi [3C][C] | lo [3C][C] (section PDSC_W_QKV_W of the split-weight
 * buffer, pdsc_wsplit_build below) -> that one GEMM runs in split precision; q, k, v only feed the attention, whose
 * own operand split has an error of the same order, and never touch the residual stream. */
int pdsc_layer_fused_split(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                           const float* res, const float* feat_in, float* feat_out,
                           float* featB_out, float* qkv_out, void* q_split, void* kv_tiles,
                           const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, const float* wp, const float* bp,
                           const float* wq, const float* bq, const void* wq_split /* optional */, int bs, int N,
                           void* stream);

/* Split-precision variant of the whole chain (PDSC_ATT_FP16X3_ALL: every GEMM as three f16 MFMAs per operand pair);
 * same tail-input convention.  Weights come from the split-weight buffer:
 *   pdsc_wsplit_bytes(cfg) bytes, filled once per model by pdsc_wsplit_build(cfg, wpack, wsplit, stream);
 *   matrix of `section` (PDSC_W_PCN_W, _QKV_W, _FC1_W, _FC2_W, _FC3_W) of `layer` starts at 16-bit element
 *   pdsc_wsplit_offset(cfg, section, layer): [out][in] hi, then [out][in] lo.  Biases stay fp32 (packed buffer). */
size_t    pdsc_wsplit_bytes(const pdsc_config* cfg);
long long pdsc_wsplit_offset(const pdsc_config* cfg, int section, int layer);
int       pdsc_wsplit_build(const pdsc_config* cfg, const float* wpack, void* wsplit, void* stream);
int pdsc_layer_fused_x3(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                        const float* res, const float* feat_in, float* feat_out, float* featB_out, float* qkv_out,
                        void* q_split, void* kv_tiles,
                        const void* w1, const float* b1, const void* w2, const float* b2, const void* w3, const float* b3,
                        const void* wp, const float* bp, const void* wq, const float* bq, int bs, int N, void* stream);

/* The same chain (tail of layer i, head of layer i+1, q|k|v projection in split precision) with the weights supplied as
 * MFMA-fragment-ordered streams: 8 KiB chunks in exactly the order the wavefront-resident kernel consumes them, so every
 * weight load of a wave is 1 KiB of consecutive memory.  This is the entry pdsc_forward_* uses by default.
 *   tail stream: pdsc_wfrag_tail_bytes() bytes from (fc1 [C/2][C], fc2 [C/2][C/2], fc3 [C][C/2]) fp32 and their biases;
 *   head stream: pdsc_wfrag_head_bytes() bytes from (pcn [C][C] fp32 kept fp32, qkv [3C][C] fp32 -> fp16 hi / lo) and
 *                their biases (the bias of an output tile is one more k-step of its GEMM: A = bias, B = 1).
 * pdsc_wsplit_build also stores both per layer inside the split-weight buffer, at 16-bit element
 * pdsc_wsplit_offset(cfg, PDSC_WS_FRAG_TAIL / PDSC_WS_FRAG_HEAD, layer). */
#define PDSC_WS_FRAG_TAIL 100
#define PDSC_WS_FRAG_HEAD 101
#define PDSC_WS_FRAG_TAIL_H3 102   /* the same streams built with gemm_format = PDSC_LAYER_GEMM_H3 */
#define PDSC_WS_FRAG_HEAD_H3 103
int pdsc_layer_prefers_block(int bs, int N);   /* 1: with layer_gemm = F32, pdsc_forward_* takes the workgroup-per-tile kernel for this size */
int pdsc_layer_h3_uses_coop(int bs, int N);    /* 1: with layer_gemm = H3, a launch over bs x N points takes layer_h3_coop_kernel (four
                                                * wavefronts per 32-point tile: at most 2560 tiles), 0: layer_h3_kernel */
size_t pdsc_wfrag_tail_bytes(void);
size_t pdsc_wfrag_head_bytes(void);
int pdsc_wfrag_build_tail(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                          const float* b3, void* out, void* stream);
int pdsc_wfrag_build_head(const float* wp, const float* bp, const float* wq, const float* bq, void* out, void* stream);
/* ... with the format of the fc1..fc3 / pcn chunks chosen (enum pdsc_layer_gemm; the q|k|v chunks are fp16 hi / lo in
 * both; the un-suffixed entries build PDSC_LAYER_GEMM_F32 streams).  pdsc_layer_fused_frag_fmt must be told the format. */
int pdsc_wfrag_build_tail_fmt(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                              const float* b3, void* out, int gemm_format, void* stream);
int pdsc_wfrag_build_head_fmt(const float* wp, const float* bp, const float* wq, const float* bq, void* out,
                              int gemm_format, void* stream);
int pdsc_layer_fused_frag_fmt(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                              const float* res, const float* feat_in, float* feat_out, float* featB_out, float* qkv_out,
                              void* q_split, void* kv_tiles, const void* wfrag_tail, const void* wfrag_head,
                              int gemm_format, int bs, int N, void* stream);
/* ... with point-fragment hand-offs: io_flags = OR of enum pdsc_layer_io (which of part_o / res / featB_out are PF).
 * Needs gemm_format = PDSC_LAYER_GEMM_H3 and the forward's output set (head: split streams only, no qkv_out; tail + head:
 * no feat_out); PDSC_ERR_ARG otherwise.  PF res / featB_out buffers hold bs * ceil(N / 32) * 32 rows. */
int pdsc_layer_fused_frag_io(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                             const float* res, const float* feat_in, float* feat_out, float* featB_out,
                             void* q_split, void* kv_tiles, const void* wfrag_tail, const void* wfrag_head,
                             int gemm_format, int io_flags, int bs, int N, void* stream);
int pdsc_layer_fused_frag(const float* msg, const float* part_o, const float* part_ml, int nsplit, int Npad,
                          const float* res, const float* feat_in, float* feat_out, float* featB_out, float* qkv_out,
                          void* q_split, void* kv_tiles, const void* wfrag_tail, const void* wfrag_head, int bs, int N,
                          void* stream);

/* ---- a-3  spatial-consistency guided non-local attention ---------------------------------------
 * replaces models/PointDSC.py:39-42 (both einsums and the softmax; N x N scores never materialised).
 *   msg[o][:] = sum_i softmax_i( compat[o][i] * <Q_o, K_i> / sqrt(C) ) * V_i
 * qkv [bs*N][3C] rows = (q | k | v) with q PRE-SCALED by log2(e)/sqrt(C) (see PDSC_W_QKV_W);
 * compat [bs][N][ld]; msg [bs*N][C].  `scratch` holds split-key partials, size from
 * pdsc_attention_scratch_bytes; nsplit <= 0 lets the library choose. */
size_t pdsc_attention_scratch_bytes(int bs, int N, int nsplit);
int    pdsc_attention_default_split(int bs, int N);
int    pdsc_sc_attention(const float* qkv, const float* compat, long long ld, float* msg,
                         void* scratch, size_t scratch_bytes, int bs, int N, int nsplit, void* stream);

/* Split-precision variant (PDSC_ATT_FP16X3): same contract, operands as fp16 hi/lo streams.
 * pdsc_pack_qkv_split converts fp32 (q|k|v) rows [bs*N][3C] into the two streams (the fused layer kernel emits
 * them directly; the packer serves stage tests and callers with their own projections). */
size_t pdsc_split_q_bytes(int bs, int N);
size_t pdsc_split_kv_bytes(int bs, int N);
int    pdsc_pack_qkv_split(const float* qkv, void* q_split, void* kv_tiles, int bs, int N, void* stream);
size_t pdsc_attention_split_scratch_bytes(int bs, int N, int nsplit);
int    pdsc_attention_split_default_split(int bs, int N);
/* leaf form (enum pdsc_att_leaves >= PDSC_LEAVES_CANONICAL): canonical leaf count of N; the plan (key split = workgroups per query
 * block, leaves per pair) the forward uses for (bs, N, leaves_mode); bytes of its scratch (the leaf partials) */
int    pdsc_attention_leaf_count(int N);
int    pdsc_attention_leaf_plan(int bs, int N, int leaves_mode, int* nsplit, int* nleaf);
size_t pdsc_attention_leaf_scratch_bytes(int bs, int N, int leaves_mode);
int    pdsc_sc_attention_split(const void* q_split, const void* kv_tiles, const float* compat, long long ld,
                               float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                               void* stream);
/* same, streaming the unorm16 matrix of pdsc_spatial_compat_u16 (ld: uint16 elements per row) */
int    pdsc_sc_attention_split_u16(const void* q_split, const void* kv_tiles, const unsigned short* compat_u16, long long ld,
                                   float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit,
                                   void* stream);
/* msg == NULL (only when the key split is > 1): the partials are left un-merged in `scratch` for pdsc_layer_fused_x3:
 * part_o = scratch as [bs][nsplit][Npad][C] floats, Npad = N rounded up to 256, part_ml right behind it as
 * [bs][nsplit][Npad][2]. */

/* Diagnostics hook of the split-precision kernel: when a device buffer of (#workgroups * 8 waves * 8) int64 is set,
 * the 8-wave kernel variant is replaced by an instrumented build that accumulates per-wave shader-clock sums of its
 * phases (0 prologue, 1 first tile, 2 own-DMA wait, 3 barrier, 4 DMA issue, 5 phase A, 6 phase B, 7 rest) there.
 * NULL (default) switches it off.  Used by tools/attention_trace.py only. */
int pdsc_attention_trace(long long* device_buffer);

/* Same for the fused layer kernel: buffer of (#workgroups * 4 waves * 16) int64 receiving raw shader-clock stamps at the
 * stage boundaries of layer_fused_kernel (tools/layer_trace.py); NULL switches it off. */
int pdsc_layer_trace(long long* device_buffer);

/* ---- a-4  first two layers of the confidence head in one launch ------------------------------------------------
 * replaces classification.0 .. classification.3 (models/PointDSC.py:107-111, called at :171):
 *   h2[m][0:32] = relu(W2 relu(W1 feat[m][0:128] + b1) + b2),  W1 [32][128], W2 [32][32] row-major fp32.
 * Exact fp32 MFMA; every output element goes through the same fma chain as
 *   pdsc_linear(feat, 128, W1, b1, ..., relu) followed by pdsc_linear(h1, 32, W2, b2, ..., relu)   -- bit-identical --
 * without the [M][32] hidden layer going through HBM.  pdsc_forward_* calls this since r04. */
int pdsc_classifier_hidden(const float* feat, const float* W1, const float* b1, const float* W2, const float* b2, float* h2,
                           int M, void* stream);

/* ---- a-4  L2 normalisation + last classifier layer --------------------------------------------
 * replaces F.normalize (models/PointDSC.py:156) and classification.4 (:112,171).
 *   normed[m][:] = feat[m][:] / max(||feat[m]||_2, 1e-12);  conf[m] = <h2[m][0:32], w3> + b3 */
int pdsc_normalize_confidence(const float* feat, const float* h2, const float* w3, const float* b3,
                              float* normed, float* conf, int M, void* stream);

/* ---- a-5  NMS seed selection -------------------------------------------------------------------
 * replaces pick_seeds (models/PointDSC.py:199-217).
 *   keys[i] = conf[i] * [ for all j: conf[i] >= conf[j]  or  ||src_i - src_j|| >= radius ]
 *   seeds   = first num_seeds indices by descending key, equal keys by ascending index. */
int pdsc_nms_keys(const float* src_keypts, const float* conf, float radius, float* keys,
                  int bs, int N, void* stream);
/* the same keys, bit for bit, from ~1 % of the pair evaluations: points counting-sorted into a 2-D cell grid of width >=
 * radius, the predicate evaluated against the 3 x 3 neighbouring cells only (what the forward calls; workspace from
 * pdsc_nms_workspace_bytes; radius <= 0 / NaN / NULL workspace fall back to pdsc_nms_keys) */
size_t pdsc_nms_workspace_bytes(int bs, int N);
int pdsc_nms_keys_grid(const float* src_keypts, const float* conf, float radius, float* keys, void* workspace,
                       size_t workspace_bytes, int bs, int N, void* stream);
int pdsc_rank_select(const float* keys, int* seeds, int bs, int N, int num_seeds, void* stream);

/* ---- a-6  feature-space kNN of the seeds -------------------------------------------------------
 * replaces knn(..., ignore_self=True, normalized=True) + the seed gather
 * (models/common.py:48-69, models/PointDSC.py:250-252); only the seed rows are computed.
 *   dist[s][j] = 2 - 2 * <normed[seed_s], normed[j]>;  knn_idx[s][0:k] = ranks 1..k of ascending
 *   (dist, index) order (rank 0 dropped exactly like `[:, :, 1:]`).
 * dist_scratch: [bs][S][ldd] floats, ldd = pdsc_compat_ld(N).  knn_idx: [bs][S][k] int32.
 * NOTE: pdsc_knn_seeds = pdsc_knn_seeds_form(form 0): for large batches (bs * ceil(S / 32) >= 384) the library takes the fused
 * form, which never writes dist_scratch -- its contents are then UNDEFINED.  A caller that wants the S x N distances back calls
 * pdsc_knn_seeds_form(..., form = 1). */
int pdsc_knn_seeds(const float* normed, const int* seeds, float* dist_scratch, int* knn_idx,
                   int bs, int N, int S, int k, void* stream);
/* form: 0 = the library's choice, 1 = two launches through the S x N distance matrix (Gram rows, then a selection launch),
 * 2 = fused (r05: 32 seeds per workgroup, the distances never leave the chip -- only candidates below a per-seed bound reach an
 * LDS list; needs k + 1 <= 48 and N >= 256; dist_scratch unused).  Same distance bits, same (dist, index) order: the neighbour
 * indices of the two forms are identical.  form 0 takes the fused form when bs * ceil(S / 32) >= 384 workgroups.
 * normed_pf (optional, fused form): the same normalised rows in point-fragment order, [bs][ceil(N / 32) * 32][128]
 * (pdsc_normalize_confidence_pf writes both) -- the kernel then loads its column operand 1 KiB per instruction; NULL = gathered
 * from `normed` (32 pieces of 32 bytes per instruction: 2 x slower, same results). */
int pdsc_knn_seeds_form(const float* normed, const float* normed_pf, const int* seeds, float* dist_scratch, int* knn_idx,
                        int bs, int N, int S, int k, int form, void* stream);
/* pdsc_normalize_confidence per pair (feat [bs][N][128] ...), additionally leaving the normalised rows in point-fragment order */
int pdsc_normalize_confidence_pf(const float* feat, const float* h2, const float* w3, const float* b3, float* normed,
                                 float* normed_pf, float* conf, int bs, int N, void* stream);

/* ---- a-7/a-8  per-seed compatibility + power iteration ----------------------------------------
 * replaces models/PointDSC.py:257-281 and cal_leading_eigenvector (:347-358).
 * Every iterate is stored: eig_iters [bs][S][num_iterations][PDSC_MAX_K]; conv_mask[b] bit i is set iff
 * every seed of pair b satisfied the allclose test at iteration i (the reference's early exit is global
 * over the S seeds).  seed_M (optional, may be NULL): [bs][S][k][k]. */
int pdsc_seed_power_iteration(const float* normed, const float* src_keypts, const float* tgt_keypts,
                              const int* knn_idx, const float* sigma, const float* sigma_spat,
                              float* eig_iters, unsigned int* conv_mask, float* seed_M,
                              int bs, int N, int S, int k, int num_iterations, void* stream);

/* ---- a-9  seed-wise weighted Procrustes --------------------------------------------------------
 * replaces models/PointDSC.py:282-320 (weight normalisation + rigid_transform_3d on the k neighbours).
 * Picks iterate `first set bit of conv_mask, else num_iterations-1`.  seed_trans [bs][S][16]. */
int pdsc_seed_transforms(const float* src_keypts, const float* tgt_keypts, const int* knn_idx,
                         const float* eig_iters, const unsigned int* conv_mask, float* seed_trans,
                         float* seed_weights /* optional [bs][S][k] */,
                         int bs, int N, int S, int k, int num_iterations, void* stream);

/* ---- a-7 + a-8 + a-9 in one launch (what the forward calls) -------------------------------------
 * pdsc_seed_power_iteration and, for the LAST iterate, pdsc_seed_transforms by the same wavefront that owns the seed;
 * a second, normally empty launch re-solves the seeds of a pair whose global early exit (conv_mask) selected an earlier
 * iterate.  seed_trans / seed_weights / seed_M may be NULL (then this is pdsc_seed_power_iteration). */
int pdsc_seed_solve(const float* normed, const float* src_keypts, const float* tgt_keypts, const int* knn_idx,
                    const float* sigma, const float* sigma_spat, float* eig_iters, unsigned int* conv_mask, float* seed_M,
                    float* seed_trans, float* seed_weights, int bs, int N, int S, int k, int num_iterations, void* stream);

/* rigid_transform_3d(A, B, weights, weight_threshold) (models/common.py:7-45 + utils/SE3.py:73-96):
 * A,B [bs][n][3], weights [bs][n] or NULL (=1), T [bs][16] row-major 4x4 with p_B = R p_A + t.
 * weights are NOT modified (the reference zeroes weights < threshold in place). */
int pdsc_rigid_transform_3d(const float* A, const float* B, const float* weights, float weight_threshold,
                            float* T, int bs, int n, void* stream);

/* ---- a-10  hypothesis scoring ------------------------------------------------------------------
 * replaces models/PointDSC.py:325-335.  counts[b][s] = #{n : ||R_s src_n + t_s - tgt_n|| < thr};
 * best = first argmax; initial_trans = seed_trans[best]; labels[n] = residual_best[n] < thr (0/1 fp32). */
int pdsc_score_hypotheses(const float* seed_trans, const float* src_keypts, const float* tgt_keypts,
                          float inlier_threshold, int* counts, int bs, int N, int S, void* stream);
int pdsc_select_best(const int* counts, const float* seed_trans, const float* src_keypts,
                     const float* tgt_keypts, float inlier_threshold, int* best, float* initial_trans,
                     float* labels, int bs, int N, int S, void* stream);

/* ---- a-11  post refinement ---------------------------------------------------------------------
 * replaces post_refinement (models/PointDSC.py:403-438) incl. transform (utils/SE3.py:43-57): the whole
 * <=max_iters loop runs on the device.  solves (optional) [bs] = number of re-solves performed. */
int pdsc_post_refinement(const float* initial_trans, const float* src_keypts, const float* tgt_keypts,
                         float threshold, int max_iters, float* final_trans, int* solves,
                         int bs, int N, void* stream);

/* ---- whole path --------------------------------------------------------------------------------
 * replaces PointDSC.forward(data) with 'testing' in data (models/PointDSC.py:128-197).
 * corr_pos [bs][N][in_dim], src/tgt [bs][N][3]  ->  final_trans [bs][16], final_labels [bs][N] (0/1).
 * num_seeds = int(N * ratio) computed by the caller in double precision like the reference (:174). */
int pdsc_forward_testing(const pdsc_config* cfg, const float* wpack, const void* wsplit /* NULL iff PDSC_ATT_FP32 */,
                         const float* corr_pos, const float* src_keypts, const float* tgt_keypts,
                         int bs, int N, int num_seeds,
                         float* final_trans, float* final_labels,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- whole path, ragged batch ---------------------------------------------------------------------
 * The reference's real evaluation feeds every pair with its own number of correspondences (evaluation/test_3DMatch.py:126
 * `num_node='all'`, datasets/ThreeDMatch.py:271-276) and therefore one pair per call (models/PointDSC.py:210); its own
 * batching crops every pair to the shortest (datasets/dataloader.py:6-31).  Here a batch may mix sizes: the inputs are
 * padded to the longest pair, [bs][N][.] with N = max_b num_corr[b] (padding rows: any finite values, e.g. zeros), and
 *   num_corr           [bs] int32, DEVICE: correspondences of pair b (2 <= num_corr[b] <= N)
 *   num_seeds_per_pair [bs] int32, DEVICE: int(num_corr[b] * ratio) >= 1, computed by the caller in double precision (:174)
 *   num_seeds          = max_b num_seeds_per_pair[b],   n_min = min_b num_corr[b] (host copies; n_min must leave every pair
 *                        at least one 32-key tile per attention key split: ceil(n_min / 32) >= pdsc_attention_split_default_split(bs, N),
 *                        and n_min > min(cfg->k, N - 1): the reference clamps k per pair, k = min(k, num_corr - 1)
 *                        (models/PointDSC.py:250), one launch has one k -- a pair of at most k rows is its own call;
 *                        PDSC_ERR_ARG otherwise)
 * Pair b's results are those of pdsc_forward_testing on its own num_corr[b] rows (same stages on the same data; only the
 * launch plans, i.e. fp32 summation orders, are the batch's): final_trans [bs][16], final_labels [bs][N] with rows
 * >= num_corr[b] zero.  Workspace: pdsc_workspace_bytes(cfg, bs, N, num_seeds).  Split-precision attention modes only. */
int pdsc_forward_testing_ragged(const pdsc_config* cfg, const float* wpack, const void* wsplit,
                                const float* corr_pos, const float* src_keypts, const float* tgt_keypts,
                                int bs, int N, int num_seeds, const int* num_corr, const int* num_seeds_per_pair, int n_min,
                                float* final_trans, float* final_labels,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- whole path on two streams (throughput loops with several forwards in flight, pointdsc_amd/pipeline.py) --------
 * Same result as pdsc_forward_testing (num_corr == NULL) / pdsc_forward_testing_ragged.  The encoder (compat build, 12 attention
 * + layer launches) is enqueued on `stream`; everything after it -- classifier, NMS, seed ranking, kNN, per-seed solver,
 * scoring, refinement: a sequential chain of ~15 small launches -- on `tail_stream`, which the caller creates with a HIGHER
 * priority: while forward i's tail runs, forward i+1's encoder (another stream) keeps the chip full, and the tail's few
 * workgroups are dispatched ahead of the attention launch's queued ones instead of behind them.  fork_event / join_event:
 * caller-owned hipEvent_t (no timing needed), recorded on stream / tail_stream; on return `stream` waits for join_event, so
 * work enqueued on `stream` afterwards is ordered after the results.  Nothing is allocated; capturable in a hipGraph.
 * Measured with two forwards in flight (profiles/r03_h_inflight_ab.txt, r03_i_inflight_ab.txt): -2.6 % per step at 32 pairs of
 * N = 5000, -6 % at 4 pairs, -13 % for one pair of N = 10000 against one stream; two plain streams without it: +-0. */
int pdsc_forward_testing_streams(const pdsc_config* cfg, const float* wpack, const void* wsplit,
                                 const float* corr_pos, const float* src_keypts, const float* tgt_keypts,
                                 int bs, int N, int num_seeds, const int* num_corr /* NULL: uniform batch */,
                                 const int* num_seeds_per_pair, int n_min,
                                 float* final_trans, float* final_labels,
                                 void* workspace, size_t workspace_bytes,
                                 void* stream, void* tail_stream, void* fork_event, void* join_event);

/* ---- validation forward -------------------------------------------------
 * replaces PointDSC.forward(data) WITHOUT the 'testing' key on a module in eval() mode (libs/trainer.py:158-222
 * calls it so): models/PointDSC.py:158-163 (feature similarity matrix M), :176 (seeds = top int(N*ratio) by
 * confidence, no NMS), :182 (per-seed hypotheses; the power iteration's allclose exit is taken over the whole batch),
 * no post refinement, :190-191 (the returned labels are the confidence logits).  Forward only (no autograd).
 *   final_trans [bs][16] = best seed hypothesis; logits [bs][N]; M [bs][N][ldM]. */
int pdsc_forward_validation(const pdsc_config* cfg, const float* wpack, const void* wsplit,
                            const float* corr_pos, const float* src_keypts, const float* tgt_keypts,
                            int bs, int N, int num_seeds, float* final_trans, float* logits,
                            float* M, long long ldM, void* workspace, size_t workspace_bytes, void* stream);

/* M[b][i][j] = clamp(1 - (1 - <normed_i, normed_j>) / sigma^2, 0, 1), M[b][i][i] = 0   (models/PointDSC.py:158-163);
 * normed [bs*N][C], sigma: device pointer to the learned sigma, M [bs][N][ld >= N]. */
int pdsc_feature_compat(const float* normed, const float* sigma, float* M, long long ld, int bs, int N, void* stream);

/* AND of the per-pair convergence masks of pdsc_seed_power_iteration into every entry: the reference's
 * power-iteration exit (models/PointDSC.py:347-358) is global over all matrices of one call. */
int pdsc_conv_mask_all_pairs(unsigned int* conv_mask, int bs, void* stream);

/* ---- correspondence construction: the step in front of the hot path -----------------
 * replaces datasets/ThreeDMatch.py:283-290,305-308 / demo_registration.py:101-108 (numpy on the host in the reference):
 *   distance = sqrt(2 - 2 * src_desc @ tgt_desc^T + 1e-6);  nn_idx[i] = argmin_j distance[i][j] (first index among equal
 *   distances, NaN first -- np.argmin);  nn_dist[i] (optional) = that distance.  The Ns x Nt matrix is never stored.
 * src_desc [Ns][D], tgt_desc [Nt][D] fp32 (L2-normalised descriptors, D <= 64); scratch: pdsc_match_scratch_bytes. */
size_t pdsc_match_scratch_bytes(int Ns, int Nt);
int pdsc_match_descriptors(const float* src_desc, const float* tgt_desc, int Ns, int Nt, int D, int* nn_idx,
                           float* nn_dist, void* scratch, size_t scratch_bytes, void* stream);
/* the 3DLoMatch caller's form (evaluation/test_3DLoMatch.py:45-46): nn_idx[i] = argmax_j <src_desc_i, tgt_desc_j> (torch.argmax:
 * first index among equal maxima, NaN counts as the maximum); nn_dot[i] (optional) = that inner product.  Same scratch. */
int pdsc_match_descriptors_ip(const float* src_desc, const float* tgt_desc, int Ns, int Nt, int D, int* nn_idx,
                              float* nn_dot, void* scratch, size_t scratch_bytes, void* stream);
/* corr[c] = (i, src2tgt[i]) for i ascending; with tgt2src != NULL only the mutual nearest neighbours
 * (tgt2src[src2tgt[i]] == i, ThreeDMatch.py:286-288) are kept.  corr [Ns][2] (capacity), *count = rows written. */
int pdsc_select_correspondences(const int* src2tgt, const int* tgt2src, int Ns, int* corr, int* count, void* stream);
/* src_sel[c] = src_keypts[corr[c][0]], tgt_sel[c] = tgt_keypts[corr[c][1]], corr_pos[c] = (src_sel | tgt_sel) - column mean
 * over the *count rows (in_dim = 6, ThreeDMatch.py:299-308).  Outputs have capacity for every row of corr. */
int pdsc_build_corr_pos(const float* src_keypts, const float* tgt_keypts, const int* corr, const int* count,
                        float* corr_pos, float* src_sel, float* tgt_sel, void* stream);

/* ---- spectral-matching baseline: the N x N power iteration ------------------------------
 * replaces SM() of baseline_scripts/baseline_3DMatch.py:19-53:  M = max(0, 4.5 - d^2 / 2 / sigma^2) (zero diagonal,
 * d = |corr_i[0:3] - corr_j[0:3]| - |corr_i[3:6] - corr_j[3:6]|, sigma = inlier_threshold / 3);  v = 1, num_iterations x
 * { v = M v; v /= |v| + 1e-6 };  pred_labels = 1 for the num_top = int(N * top_ratio) largest entries of v (equal
 * entries by ascending index);  pred_trans = rigid_transform_3d(src_keypts, tgt_keypts, v * pred_labels).
 * corr_pos [bs][N][6] (the centred coordinates the reference passes as `corr`), src/tgt [bs][N][3];
 * pred_trans [bs][16], pred_labels [bs][N], leading_eig (optional) [bs][N]; workspace: pdsc_sm_workspace_bytes
 * (holds M: 4 N ld bytes per pair).  bs > 1 = independent pairs (the reference asserts bs == 1). */
size_t pdsc_sm_workspace_bytes(int bs, int N);
int pdsc_sm_baseline(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                     int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                     void* workspace, size_t workspace_bytes, int bs, int N, void* stream);
/* Two forms, bit-identical results (same arithmetic in the same order):
 *   streaming (form 1): the 4 N^2-byte matrix is written to the workspace once and streamed from HBM per iteration, the pairs of a
 *                       batch in one launch; any N up to 65 536;
 *   register-resident (form 2): ONE persistent launch per pair computes the matrix straight into the chip's vector registers
 *                       (100 MB at N = 5000 of the 128 MiB the 256 compute units hold) and runs every power iteration from there;
 *                       per iteration only y crosses the chip, behind a grid barrier.  N <= 5120, 20 rows per compute unit.
 * pdsc_sm_baseline (= form 0) ALWAYS runs the streaming form (r05).  The resident form (one pair of N = 5000: 252 us against 342)
 * needs the whole chip to itself for its grid barrier, which only the caller can promise: it is opt-in (form 2), launched
 * cooperatively (the launch fails, PDSC_ERR_LAUNCH, when the runtime cannot make the grid co-resident), refused under stream
 * capture, ordered against this process's other resident launches on the same device, and every in-kernel wait is bounded --
 * a second PROCESS running it on the same GPU at the same time ends in NaN outputs after 0.5 s, never in a hang. */
int pdsc_sm_baseline_form(const float* corr_pos, const float* src_keypts, const float* tgt_keypts, float inlier_threshold,
                          int num_top, int num_iterations, float* pred_trans, float* pred_labels, float* leading_eig,
                          void* workspace, size_t workspace_bytes, int bs, int N, int form, void* stream);

/* cal_confidence (models/PointDSC.py:366-401): confidence of a spectral-matching solution from its compatibility matrix
 * M [bs][N][ld] (ld >= N, multiple of 4) and leading eigenvector [bs][N]:
 *   method 0 'eig_value'      : Rayleigh quotient lambda1 = v^T M v / v^T v
 *   method 1 'eig_value_ratio': lambda1 / lambda2, lambda2 from num_iterations power steps on B = M - lambda1 v v^T
 *                               started at 1 (each step normalised by |.| + 1e-6), B never materialised
 *   method 2 'xMx'            : v^T M v / N
 * confidence [bs]; workspace from pdsc_cal_confidence_workspace_bytes.  One HBM-bound N x N mat-vec per step. */
size_t pdsc_cal_confidence_workspace_bytes(int bs, int N);
int pdsc_cal_confidence(const float* M, long long ld, const float* leading_eig, int method, int num_iterations,
                        float* confidence, void* workspace, size_t workspace_bytes, int bs, int N, void* stream);

/* ---- evaluation row on the device -----------------------------------------------------------
 * replaces libs/loss.py:44-51 (RE / TE / recall of TransformationLoss), :96-100 (precision / recall / F1, sklearn on the
 * host) and the stats row of evaluation/test_3DMatch.py:90-98, per pair, without a device -> host copy:
 *   stats[b][0..8] = success (RE < re_thre && TE < te_thre), RE [deg], TE [cm], #gt inliers, gt inlier ratio,
 *                    #gt inliers among the predicted inliers, precision, recall, F1       (pred > 0 is "predicted inlier")
 * trans, gt_trans [bs][16]; pred_labels, gt_labels [bs][N]; stats [bs][9]. */
int pdsc_eval_stats(const float* trans, const float* gt_trans, const float* pred_labels, const float* gt_labels,
                    float re_thre, float te_thre, float* stats, int bs, int N, void* stream);

/* ---- range probe for layer_gemm = PDSC_LAYER_GEMM_H3 ----------------------------------------------------------------
 * The H3 arithmetic carries every operand of the fc_message / PointCN GEMMs as fp16 hi + lo, so every activation of the
 * 12-layer chain -- hidden ones included -- must stay below 65504.  This entry runs the ENCODER (compat, layer0, 12 x
 * {PointCN, q|k|v, attention, fc1, fc2, fc3 + residual}; models/PointDSC.py:48-77) once with the fp32 GEMMs, one launch per
 * conv, and leaves in absmax[kind] (DEVICE, PDSC_RANGE_NUM_KINDS floats) the largest |value| of each activation kind over
 * all layers (a NaN anywhere reads back as NaN).  The module calls it on the first forward after packing weights and falls
 * back to PDSC_LAYER_GEMM_F32 with a warning when a value is out of range.  Same workspace as pdsc_forward_testing. */
enum pdsc_range_kind { PDSC_RANGE_LAYER0 = 0, PDSC_RANGE_POINTCN = 1, PDSC_RANGE_QKV = 2, PDSC_RANGE_MESSAGE = 3, PDSC_RANGE_FC1 = 4,
                       PDSC_RANGE_FC2 = 5, PDSC_RANGE_FEATURE = 6, PDSC_RANGE_NUM_KINDS = 8 };
int pdsc_encoder_range_probe(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                             const float* src_keypts, const float* tgt_keypts, int bs, int N, int num_seeds, float* absmax,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Named views into the workspace of the last layout computed for (cfg, bs, N, num_seeds): lets the
 * parity tests read intermediates after pdsc_forward_testing.  Returns byte offset or -1. */
long long pdsc_workspace_offset(const pdsc_config* cfg, int bs, int N, int num_seeds, const char* name);

/* ---- opt-in kernel timing (bench.py's roofline leg) ----------------------------------------------
 * When enabled, the two roofline kernels are bracketed by hipEvents recorded on the caller's stream:
 * kind 0 = sc_attention_kernel (MFMA roofline), kind 1 = compat_kernel (HBM roofline).  Disabled by
 * default: then no event is created or recorded.  pdsc_profile_read synchronises on the recorded events
 * (host side, call it after the timed region) and returns total milliseconds + number of launches. */
enum pdsc_profile_kind { PDSC_PROF_ATTENTION = 0, PDSC_PROF_COMPAT = 1, PDSC_PROF_LAYER = 2 /* tail + head launches */,
                         PDSC_PROF_NUM_KINDS = 3 };
int pdsc_profile_enable(int max_records_per_kind);   /* 0 disables and frees the events */
int pdsc_profile_reset(void);
/* bracket only every stride-th launch of `kind` (an event record costs the stream a few microseconds and separates the
 * kernels around it; bench.py samples one attention / layer launch per forward: the launches of a kind do identical work) */
int pdsc_profile_set_stride(int kind, int stride);
int pdsc_profile_read(int kind, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* POINTDSC_HIP_H */
