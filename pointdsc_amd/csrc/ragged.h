// Ragged batches (pdsc_forward_testing_ragged): every pair of a batch has its own correspondence count n_b <= N and its own
// seed count s_b <= S, while every buffer keeps the strides of the longest pair (rows n_b .. N-1 of a pair are padding).
// The stages whose result depends on the count -- attention (keys / queries), NMS, seed ranking, kNN columns, hypothesis
// scoring, best-hypothesis labels, refinement -- take the device arrays `nvalid` / `svalid` ([bs] int32, NULL = uniform
// batch) through these internal launchers; the public stage entry points are the same launchers with NULL.
// Everything row-parallel (layer kernels, classifier, normalisation, Gram rows, per-seed solver) runs unchanged over the
// padded layout: padding rows carry finite copies / unused values that no valid row ever reads.
#pragma once
#include "pdsc_common.h"

namespace pdsc {

int launch_nms_keys_grid(const float* src, const float* conf, float radius, float* keys, void* workspace, size_t workspace_bytes,
                         int bs, int N, const int* nvalid, hipStream_t st);
// conv_mask_init != NULL: the kernel also sets conv_mask_init[b] = 0xFFFFFFFF for every pair (the start value of the seed solver's
// convergence mask: launch_seed_solve_forward then skips its own fill launch)
int launch_rank_select(const float* keys, int* seeds, int bs, int N, int num_seeds, const int* nvalid, const int* svalid, hipStream_t st,
                       unsigned int* conv_mask_init = nullptr);
// solver.hip: pdsc_seed_solve as the forward runs it; mask_ready: conv_mask already holds its start value (no fill launch)
int launch_seed_solve_forward(const float* normed, const float* src, const float* tgt, const int* knn_idx, const float* sigma,
                              const float* sigma_spat, float* eig_iters, unsigned int* conv_mask, float* seed_trans, float* seed_weights,
                              int bs, int N, int S, int k, int num_iterations, bool mask_ready, hipStream_t st);
// score.hip: pdsc_select_best + pdsc_post_refinement in one launch (one workgroup per pair does both; same arithmetic, same bits)
int launch_select_and_refine(const int* counts, const float* seed_trans, const float* src, const float* tgt, float inlier_threshold,
                             float refine_threshold, int max_iters, int* best, float* initial_trans, float* labels, float* final_trans,
                             int* solves, int bs, int N, int S, const int* nvalid, hipStream_t st, int* trace, const unsigned int* range_flag,
                             unsigned int* range_report = nullptr);
int launch_knn_seeds(const float* normed, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N, int S, int k,
                     const int* nvalid, hipStream_t st);
// r05: the fused form (no S x N matrix) and its point-fragment operand
bool knn_seeds_uses_fused(int bs, int N, int S, int k);
int launch_knn_seeds_form(const float* normed, const float* normed_pf, const int* seeds, float* dist_scratch, int* knn_idx, int bs, int N,
                          int S, int k, const int* nvalid, int form, hipStream_t st);
int launch_normalize_conf_pf(const float* feat, const float* h2, const float* w3, const float* b3, float* normed, float* normed_pf,
                             float* conf, int bs, int N, hipStream_t st);
int launch_score_hypotheses_slp(const float* seed_trans, const float* src, const float* tgt, float thr2, int* counts, int bs, int N, int S,
                                const int* nvalid, hipStream_t st);      // score_slp.hip (experiments builds)
float*& score_debug_slot();      // score.hip (experiments builds: diagnostics of the next scoring launch)
int launch_score_hypotheses(const float* seed_trans, const float* src, const float* tgt, float inlier_threshold, int* counts, int bs,
                            int N, int S, const int* nvalid, hipStream_t st);
int launch_select_best(const int* counts, const float* seed_trans, const float* src, const float* tgt, float inlier_threshold, int* best,
                       float* best_trans, float* labels, int bs, int N, int S, const int* nvalid, hipStream_t st);
int launch_post_refinement(const float* initial_trans, const float* src, const float* tgt, float threshold, int max_iters,
                           float* final_trans, int* solves, int bs, int N, const int* nvalid, hipStream_t st, int* trace = nullptr,
                           const unsigned int* range_flag = nullptr);
// linear.hip: encoder.layer0; range_flag != NULL: its first bs words are zeroed by this launch (the forward's first encoder launch)
int launch_layer0(const float* corr_pos, int in_dim, const float* W0, const float* b0, float* feat, int M, unsigned int* range_flag, int bs,
                  hipStream_t st);
constexpr int PDSC_REFINE_TRACE = 24;      // ints per pair in the workspace entry "refine_trace"
// msg != NULL: merged output rows (key split 1, or the combine launch); msg == NULL: the key-split partials stay in `scratch`
int launch_attention_split_ex(const void* q_split, const void* kv_tiles, const void* compat, int compat_format, long long ld,
                              float* msg, void* scratch, size_t scratch_bytes, int bs, int N, int nsplit, int partial_layout,
                              const int* nvalid, hipStream_t st);

// leaf form (attention_split.hip, sc_attention_split_kernel<..., MG = true>): C leaf partials per query, left in `scratch` in
// point-fragment order for the H3 layer kernel to merge
constexpr int PDSC_ATT_MAX_LEAVES = 8;       // = MERGE_MAX_SPLIT_H3 (merge_partials.h): what the layer kernel merges while it loads
int attention_leaf_count(int N);
void leaf_plan(int bs, int N, int leaves_mode, int* nw_out, int* nsplit_out, int* nleaf_out);
int launch_attention_leaves(const void* q_split, const void* kv_tiles, const void* compat, int compat_format, long long ld,
                            void* scratch, size_t scratch_bytes, int bs, int N, int leaves_mode, const int* nvalid, int n_min,
                            hipStream_t st);

// The three fused-layer entry points (pdsc_layer_fused_split / _frag_fmt / _frag_io: 18-24 arguments each) read the count
// array from this thread-local slot when they fill LayerArgs; run_forward sets it for the duration of a ragged call and
// clears it before returning (host-side, per thread: concurrent callers on other threads are unaffected).
const int*& layer_nvalid_slot();

}  // namespace pdsc
