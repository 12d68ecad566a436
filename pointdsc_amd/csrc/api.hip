// Host side of libpointdsc_hip.so: error plumbing, packed-weight layout, workspace layout and the
// whole-path orchestrator pdsc_forward_testing (reference PointDSC.forward in testing mode,
// models/PointDSC.py:128-197).  Pure HIP runtime -- no torch types cross this boundary.  Every stage is
// enqueued on the caller's stream with no host synchronisation, so one forward is hipGraph-capturable.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "pdsc_common.h"
#include "ragged.h"
#include "attention_common.h"

namespace pdsc {

static thread_local char g_err[512] = "";

const int*& layer_nvalid_slot() {
    static thread_local const int* slot = nullptr;
    return slot;
}

unsigned int*& range_flag_slot() {
    static thread_local unsigned int* slot = nullptr;
    return slot;
}

// where the NEXT forward of this thread also reports its range words: pinned, device-mapped host memory ([bs] u32), or NULL
static unsigned int*& range_report_slot() {
    static thread_local unsigned int* slot = nullptr;
    return slot;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PDSC_ERR_LAUNCH;
    }
    return PDSC_OK;
}

// ---- dynamic-LDS opt-in, once per (kernel, device) ----------------------------------------------
int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what) {
    // granted size per (kernel, device): a larger request on one device must not mark the others as served
    struct Entry { const void* fn; size_t granted[64]; };
    static Entry table[64];
    static int used = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return check_launch(what);
    std::lock_guard<std::mutex> lock(mu);
    Entry* e = nullptr;
    for (int i = 0; i < used; ++i)
        if (table[i].fn == fn) { e = &table[i]; break; }
    if (e && dev < 64 && e->granted[dev] >= bytes) return PDSC_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return check_launch(what);
    if (!e && used < 64) { e = &table[used++]; e->fn = fn; memset(e->granted, 0, sizeof(e->granted)); }
    if (e && dev < 64) e->granted[dev] = bytes;
    return PDSC_OK;
}

__global__ void fill_u32_kernel(unsigned int* p, unsigned int value, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = value;
}
__global__ void copy_u32_kernel(unsigned int* __restrict__ dst, const unsigned int* __restrict__ src, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// slot := max(slot, max_i |x[i]|) on the bit patterns (non-negative floats order like unsigned integers; NaN patterns sort above inf,
// so a NaN anywhere shows as "out of range")
__global__ void absmax_kernel(const float* __restrict__ x, size_t count, unsigned int* __restrict__ slot) {
    unsigned int m = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned int u = __float_as_uint(x[i]) & 0x7fffffffu;
        m = u > m ? u : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(slot, m);
}
static int launch_absmax(const float* x, size_t count, unsigned int* slot, hipStream_t st) {
    const size_t blocks = (count + 1023) / 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, count, slot);
    return check_launch("absmax");
}
int launch_copy_u32(unsigned int* dst, const unsigned int* src, size_t count, hipStream_t st) {
    if (count == 0) return PDSC_OK;
    const size_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(copy_u32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, dst, src, count);
    return check_launch("copy");
}
int launch_fill_u32(unsigned int* p, unsigned int value, size_t count, hipStream_t st) {
    if (count == 0) return PDSC_OK;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, p, value, count);
    return check_launch("fill");
}

// ---- opt-in event timing -----------------------------------------------------------------------
// stride: only every stride-th launch of the kind is bracketed by events (an event record costs the stream ~3.5 us and breaks
// the back-to-back issue of the kernels around it: 50 records per forward were 8 % of a 2 ms step)
struct ProfKind { hipEvent_t* start = nullptr; hipEvent_t* stop = nullptr; int cap = 0, n = 0, stride = 1; long long calls = 0; bool open = false; };
static ProfKind g_prof[PDSC_PROF_NUM_KINDS];

void profile_mark_begin(int kind, hipStream_t st) {
    ProfKind& p = g_prof[kind];
    if (p.cap == 0 || p.n >= p.cap) return;
    if ((p.calls++ % p.stride) != 0) return;
    (void)hipEventRecord(p.start[p.n], st);
    p.open = true;
}
void profile_mark_end(int kind, hipStream_t st) {
    ProfKind& p = g_prof[kind];
    if (!p.open) return;
    (void)hipEventRecord(p.stop[p.n], st);
    p.open = false;
    ++p.n;
}

static bool config_ok(const pdsc_config* c) {
    if (!c) { set_error("pdsc_config is null"); return false; }
    if (c->num_channels != PDSC_CHANNELS) { set_error("num_channels=%d (only %d supported)", c->num_channels, PDSC_CHANNELS); return false; }
    if (c->in_dim < 1 || c->in_dim > 16) { set_error("in_dim=%d must be in [1,16]", c->in_dim); return false; }
    if (c->num_layers < 0 || c->num_layers > 64) { set_error("num_layers=%d", c->num_layers); return false; }
    if (c->num_iterations < 0 || c->num_iterations > PDSC_MAX_POWER_ITERS) { set_error("num_iterations=%d", c->num_iterations); return false; }
    if (c->k < 1 || c->k > PDSC_MAX_K) { set_error("k=%d must be in [1,%d]", c->k, PDSC_MAX_K); return false; }
    if (c->refine_iters < 0) { set_error("refine_iters=%d", c->refine_iters); return false; }
    if (c->attention_precision < PDSC_ATT_FP16X3 || c->attention_precision > PDSC_ATT_FP16X3_ALL) {
        set_error("attention_precision=%d", c->attention_precision); return false;
    }
#ifndef PDSC_EXPERIMENTS
    if (c->attention_precision == PDSC_ATT_FP16X3_ALL) {
        set_error("attention_precision=PDSC_ATT_FP16X3_ALL (all-split layer GEMMs) exists in experiments builds only");
        return false;
    }
#endif
    if (c->compat_format != PDSC_COMPAT_U16 && c->compat_format != PDSC_COMPAT_F32) { set_error("compat_format=%d", c->compat_format); return false; }
    if (c->layer_gemm != PDSC_LAYER_GEMM_F32 && c->layer_gemm != PDSC_LAYER_GEMM_H3) { set_error("layer_gemm=%d", c->layer_gemm); return false; }
    if (c->att_leaves < PDSC_LEAVES_PER_LAUNCH || c->att_leaves > PDSC_ATT_MAX_LEAVES) { set_error("att_leaves=%d (enum pdsc_att_leaves, or 2..%d leaves)", c->att_leaves, PDSC_ATT_MAX_LEAVES); return false; }
    return true;
}

static long long section_floats(int section) {
    const long long C = PDSC_CHANNELS, H = C / 2;
    switch (section) {
        case PDSC_W_LAYER0_W: return C * 16;
        case PDSC_W_LAYER0_B: return C;
        case PDSC_W_PCN_W: return C * C;
        case PDSC_W_PCN_B: return C;
        case PDSC_W_QKV_W: return 3 * C * C;
        case PDSC_W_QKV_B: return 3 * C;
        case PDSC_W_FC1_W: return H * C;
        case PDSC_W_FC1_B: return H;
        case PDSC_W_FC2_W: return H * H;
        case PDSC_W_FC2_B: return H;
        case PDSC_W_FC3_W: return C * H;
        case PDSC_W_FC3_B: return C;
        case PDSC_W_CLS1_W: return 32 * C;
        case PDSC_W_CLS1_B: return 32;
        case PDSC_W_CLS2_W: return 32 * 32;
        case PDSC_W_CLS2_B: return 32;
        case PDSC_W_CLS3_W: return 32;
        case PDSC_W_CLS3_B: return 4;   // 1 used, padded to keep every section 16-byte aligned
        case PDSC_W_SIGMA: return 4;
        case PDSC_W_SIGMA_SPAT: return 4;
    }
    return -1;
}
static bool per_layer(int section) { return section >= PDSC_W_PCN_W && section <= PDSC_W_FC3_B; }

static long long layer_block_floats() {
    long long n = 0;
    for (int s = PDSC_W_PCN_W; s <= PDSC_W_FC3_B; ++s) n += section_floats(s);
    return n;
}

static long long wpack_offset(const pdsc_config* c, int section, int layer) {
    if (section < 0 || section >= PDSC_W_NUM_SECTIONS) return -1;
    long long off = 0;
    if (section <= PDSC_W_LAYER0_B) {
        for (int s = 0; s < section; ++s) off += section_floats(s);
        return off;
    }
    off = section_floats(PDSC_W_LAYER0_W) + section_floats(PDSC_W_LAYER0_B);
    if (per_layer(section)) {
        if (layer < 0 || layer >= c->num_layers) return -1;
        off += (long long)layer * layer_block_floats();
        for (int s = PDSC_W_PCN_W; s < section; ++s) off += section_floats(s);
        return off;
    }
    off += (long long)c->num_layers * layer_block_floats();
    for (int s = PDSC_W_CLS1_W; s < section; ++s) off += section_floats(s);
    return off;
}

// ---- workspace layout --------------------------------------------------------------------------
struct WsEntry { const char* name; size_t bytes; size_t offset; };
struct WsLayout {
    WsEntry e[40];
    int n = 0;
    size_t total = 0;
    void add(const char* name, size_t bytes) {
        e[n].name = name; e[n].bytes = bytes; e[n].offset = total;
        total += (size_t)round_up((long long)bytes, 256);
        ++n;
    }
    long long find(const char* name) const {
        for (int i = 0; i < n; ++i) if (strcmp(e[i].name, name) == 0) return (long long)e[i].offset;
        return -1;
    }
    size_t bytes_of(const char* name) const {
        for (int i = 0; i < n; ++i) if (strcmp(e[i].name, name) == 0) return e[i].bytes;
        return 0;
    }
};

static WsLayout make_layout(const pdsc_config* c, int bs, int N, int S) {
    WsLayout L;
    const size_t M = (size_t)bs * N, C = PDSC_CHANNELS, f = sizeof(float);
    const size_t ld = (size_t)pdsc_compat_ld(N);
    const int k = c->k < N - 1 ? c->k : N - 1;
    const int iters = c->num_iterations > 0 ? c->num_iterations : 1;
    const bool compat16 = c->attention_precision != PDSC_ATT_FP32 && c->compat_format == PDSC_COMPAT_U16;
    L.add("compat", (size_t)bs * N * ld * (compat16 ? sizeof(unsigned short) : f));
    L.add("featA", M * C * f);
    const size_t Mpf = (size_t)bs * round_up(N, 32);          // featB / featC may be kept in point-fragment order: whole 32-row tiles per pair
    L.add("featB", Mpf * C * f);
    L.add("featC", Mpf * C * f);
    L.add("qkv", M * 3 * C * f);
    L.add("msg", M * C * f);
    L.add("t64a", M * (C / 2) * f);
    L.add("t64b", M * (C / 2) * f);
    {
        const size_t a32 = pdsc_attention_scratch_bytes(bs, N, 0), a16 = pdsc_attention_split_scratch_bytes(bs, N, 0);
        const size_t amg = c->att_leaves >= PDSC_LEAVES_CANONICAL ? pdsc_attention_leaf_scratch_bytes(bs, N, c->att_leaves) : 0;
        L.add("att_scratch", c->attention_precision == PDSC_ATT_FP32 ? a32 : (a16 > amg ? a16 : amg));
    }
    L.add("q_split", c->attention_precision != PDSC_ATT_FP32 ? pdsc_split_q_bytes(bs, N) : 0);
    L.add("kv_tiles", c->attention_precision != PDSC_ATT_FP32 ? pdsc_split_kv_bytes(bs, N) : 0);
    L.add("normed", M * C * f);
    L.add("normed_pf", knn_seeds_uses_fused(bs, N, S, k) ? Mpf * C * f : 0);      // the fused kNN's B operand (point-fragment order)
    L.add("h1", M * 32 * f);
    L.add("h2", M * 32 * f);
    L.add("conf", M * f);
    L.add("keys", M * f);
    L.add("nms_ws", pdsc_nms_workspace_bytes(bs, N));
    L.add("seeds", (size_t)bs * S * sizeof(int));
    L.add("knn_dist", (size_t)bs * S * ld * f);
    L.add("knn_idx", (size_t)bs * S * (k > 0 ? k : 1) * sizeof(int));
    L.add("eig_iters", (size_t)bs * S * iters * PDSC_MAX_K * f);
    L.add("conv_mask", (size_t)bs * sizeof(unsigned int));
    L.add("seed_trans", (size_t)bs * S * 16 * f);
    L.add("seed_w", (size_t)bs * S * (k > 0 ? k : 1) * f);
    L.add("counts", (size_t)bs * S * sizeof(int));
    L.add("best", (size_t)bs * sizeof(int));
    L.add("initial_trans", (size_t)bs * 16 * f);
    L.add("solves", (size_t)bs * sizeof(int));
    L.add("range_flag", (size_t)bs * sizeof(unsigned int));      // fp16 range sentinel (pdsc_common.h): != 0 = pair b left the fp16 range
    L.add("refine_trace", (size_t)bs * PDSC_REFINE_TRACE * sizeof(int));      // inlier count per refinement iteration, -1 padded (parity census)
#ifdef PDSC_EXPERIMENTS
    L.add("score_dbg", (size_t)bs * S * 16 * f);        // diagnostics of the scoring kernel (score.hip, DBG)
#endif
    return L;
}

}  // namespace pdsc

using namespace pdsc;

extern "C" int pdsc_version(void) { return PDSC_VERSION; }

extern "C" int pdsc_set_range_report(unsigned int* host_words) {
    if (host_words) {
        // must be host memory the device can write (hipHostMalloc / a registered range): anything else would fault inside the kernel
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, host_words) != hipSuccess || at.type != hipMemoryTypeHost || at.devicePointer == nullptr) {
            (void)hipGetLastError();
            set_error("pdsc_set_range_report: %p is not pinned, device-mapped host memory", (void*)host_words);
            return PDSC_ERR_ARG;
        }
        host_words = (unsigned int*)at.devicePointer;
    }
    range_report_slot() = host_words;
    return PDSC_OK;
}
extern "C" int pdsc_experiments_enabled(void) {
#ifdef PDSC_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* pdsc_last_error(void) { return g_err; }

extern "C" long long pdsc_wpack_floats(const pdsc_config* cfg) {
    if (!config_ok(cfg)) return -1;
    return wpack_offset(cfg, PDSC_W_SIGMA_SPAT, 0) + section_floats(PDSC_W_SIGMA_SPAT);
}
extern "C" long long pdsc_wpack_offset(const pdsc_config* cfg, int section, int layer) {
    if (!config_ok(cfg)) return -1;
    return wpack_offset(cfg, section, layer);
}

extern "C" size_t pdsc_workspace_bytes(const pdsc_config* cfg, int bs, int N, int num_seeds) {
    if (!config_ok(cfg) || bs <= 0 || N <= 1 || num_seeds <= 0) return 0;
    return make_layout(cfg, bs, N, num_seeds).total;
}
extern "C" long long pdsc_workspace_offset(const pdsc_config* cfg, int bs, int N, int num_seeds, const char* name) {
    if (!config_ok(cfg) || bs <= 0 || N <= 1 || num_seeds <= 0 || !name) return -1;
    return make_layout(cfg, bs, N, num_seeds).find(name);
}

extern "C" int pdsc_profile_enable(int max_records) {
    for (int k = 0; k < PDSC_PROF_NUM_KINDS; ++k) {
        ProfKind& p = g_prof[k];
        for (int i = 0; i < p.cap; ++i) { (void)hipEventDestroy(p.start[i]); (void)hipEventDestroy(p.stop[i]); }
        delete[] p.start; delete[] p.stop;
        const int keep_stride = p.stride;
        p = ProfKind();
        p.stride = keep_stride;
        if (max_records > 0) {
            p.start = new hipEvent_t[max_records];
            p.stop = new hipEvent_t[max_records];
            for (int i = 0; i < max_records; ++i) {
                if (hipEventCreate(&p.start[i]) != hipSuccess || hipEventCreate(&p.stop[i]) != hipSuccess)
                    return check_launch("pdsc_profile_enable");
            }
            p.cap = max_records;
        }
    }
    return PDSC_OK;
}
extern "C" int pdsc_profile_reset(void) {
    for (int k = 0; k < PDSC_PROF_NUM_KINDS; ++k) { g_prof[k].n = 0; g_prof[k].open = false; g_prof[k].calls = 0; }
    return PDSC_OK;
}
extern "C" int pdsc_profile_set_stride(int kind, int stride) {
    PDSC_REQUIRE(kind >= 0 && kind < PDSC_PROF_NUM_KINDS && stride >= 1, "pdsc_profile_set_stride: kind=%d stride=%d", kind, stride);
    g_prof[kind].stride = stride;
    g_prof[kind].calls = 0;
    return PDSC_OK;
}
extern "C" int pdsc_profile_read(int kind, double* total_ms, int* launches) {
    PDSC_REQUIRE(kind >= 0 && kind < PDSC_PROF_NUM_KINDS && total_ms && launches, "pdsc_profile_read: bad argument");
    ProfKind& p = g_prof[kind];
    double tot = 0.0;
    for (int i = 0; i < p.n; ++i) {
        if (hipEventSynchronize(p.stop[i]) != hipSuccess) return check_launch("pdsc_profile_read");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start[i], p.stop[i]) != hipSuccess) return check_launch("pdsc_profile_read");
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.n;
    return PDSC_OK;
}

#define PDSC_TRY(call)                 \
    do {                               \
        const int rc__ = (call);       \
        if (rc__ != PDSC_OK) return rc__; \
    } while (0)

// mode 0 = testing forward; mode 1 = validation forward (no 'testing' key, module in eval mode): feature similarity
// matrix M, seeds = top-S by confidence (no NMS), batch-wide power-iteration exit, no refinement, labels = logits
// nvalid / svalid (device, [bs]) != NULL: ragged batch (ragged.h) -- N and num_seeds are then those of the longest pair,
// n_min the shortest pair's count (host copy: the attention's key split must leave every pair at least one tile per split).
static int run_forward(int mode, const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                       const float* src, const float* tgt, int bs, int N, int num_seeds,
                       float* final_trans, float* final_labels, float* Mout, long long ldM, void* workspace,
                       size_t workspace_bytes, void* stream, const int* nvalid = nullptr, const int* svalid = nullptr, int n_min = 0,
                       void* tail_stream = nullptr, void* ev_fork = nullptr, void* ev_join = nullptr, unsigned int* probe = nullptr) {
    if (!config_ok(cfg)) return PDSC_ERR_ARG;
    PDSC_REQUIRE(!tail_stream || (ev_fork && ev_join && tail_stream != stream),
                 "pdsc_forward_testing_streams: a tail stream (different from the main stream) needs the fork and join events");
    struct SlotGuard {       // the fused-layer entry points read the count array from the thread-local slot (ragged.h)
        explicit SlotGuard(const int* p) { layer_nvalid_slot() = p; }
        ~SlotGuard() { layer_nvalid_slot() = nullptr; }
    } slot_guard(nvalid);
    struct RangeGuard {      // ... and the range sentinel's flag array (pdsc_common.h)
        ~RangeGuard() { range_flag_slot() = nullptr; }
    } range_guard;
    hipStream_t hst = (hipStream_t)stream;
    if (nvalid) {
        PDSC_REQUIRE(mode == 0 && svalid, "pdsc_forward_testing_ragged: testing forward only, both count arrays needed");
        PDSC_REQUIRE(n_min >= 2 && n_min <= N, "pdsc_forward_testing_ragged: n_min=%d (N=%d)", n_min, N);
        // one launch has one neighbour count k = min(cfg->k, N - 1); the reference clamps per pair, k_b = min(k, num_corr_b - 1)
        // (models/PointDSC.py:250): a pair with fewer than k + 1 correspondences must be its own call
        PDSC_REQUIRE(n_min > (cfg->k < N - 1 ? cfg->k : N - 1), "pdsc_forward_testing_ragged: the shortest pair (%d correspondences) has no "
                     "more than k=%d: the reference clamps k per pair (k = min(k, num_corr - 1)); run such a pair in its own call",
                     n_min, cfg->k < N - 1 ? cfg->k : N - 1);
        PDSC_REQUIRE(cfg->attention_precision == PDSC_ATT_FP32 || cfg->att_leaves >= PDSC_LEAVES_CANONICAL ||
                     (n_min + 31) / 32 >= pdsc_attention_split_default_split(bs, N),
                     "pdsc_forward_testing_ragged: the shortest pair (%d correspondences) has fewer 32-key tiles than the key split "
                     "planned for bs=%d, N=%d (%d): batch pairs of more similar size", n_min, bs, N, pdsc_attention_split_default_split(bs, N));
    }
    PDSC_REQUIRE(wpack && corr_pos && src && tgt && final_trans && final_labels && workspace,
                 "pdsc_forward_testing: null pointer");
    PDSC_REQUIRE(bs > 0 && N > 1, "pdsc_forward_testing: bs=%d N=%d", bs, N);
    PDSC_REQUIRE(num_seeds >= 1 && num_seeds <= N,
                 "pdsc_forward_testing: num_seeds=%d (int(N*ratio) must be >= 1; the reference fails on an empty seed set)",
                 num_seeds);
    const WsLayout L = make_layout(cfg, bs, N, num_seeds);
    if (workspace_bytes < L.total) {
        set_error("pdsc_forward_testing: workspace %zu < %zu bytes", workspace_bytes, L.total);
        return PDSC_ERR_WORKSPACE;
    }
    char* ws = (char*)workspace;
    auto F = [&](const char* n) { return (float*)(ws + L.find(n)); };
    auto I = [&](const char* n) { return (int*)(ws + L.find(n)); };
    auto W = [&](int section, int layer) { return wpack + wpack_offset(cfg, section, layer); };

    const int C = PDSC_CHANNELS, M = bs * N, S = num_seeds;
    const int k = cfg->k < N - 1 ? cfg->k : N - 1;
    const long long ld = pdsc_compat_ld(N);
    float *compat = F("compat"), *featA = F("featA"), *featB = F("featB"), *featC = F("featC"), *qkv = F("qkv"), *msg = F("msg");
    float *t64a = F("t64a"), *t64b = F("t64b"), *normed = F("normed"), *h1 = F("h1"), *h2 = F("h2");
    float *conf = F("conf"), *keys = F("keys"), *knn_dist = F("knn_dist"), *eig = F("eig_iters");
    float *seed_trans = F("seed_trans"), *seed_w = F("seed_w"), *initial = F("initial_trans");
    int *seeds = I("seeds"), *knn_idx = I("knn_idx"), *counts = I("counts"), *best = I("best"), *solves = I("solves");
    unsigned int* conv_mask = (unsigned int*)(ws + L.find("conv_mask"));
    void* att_scratch = ws + L.find("att_scratch");
    const bool split = cfg->attention_precision != PDSC_ATT_FP32;
    const bool x3_gemm = cfg->attention_precision == PDSC_ATT_FP16X3_ALL;
    PDSC_REQUIRE(!split || wsplit, "pdsc_forward_testing: the split-precision modes need the split-weight buffer (pdsc_wsplit_build)");
    auto WS = [&](int section, int layer) { return (const void*)((const unsigned short*)wsplit + pdsc_wsplit_offset(cfg, section, layer)); };
    const size_t att_bytes = L.bytes_of("att_scratch");
    void* q_split = split ? ws + L.find("q_split") : nullptr;
    void* kv_tiles = split ? ws + L.find("kv_tiles") : nullptr;
    // fp16 range sentinel: zeroed by the layer0 launch, set by the layer kernels' conversion sites, read by the refinement launch
    unsigned int* range_flag = split && !probe ? (unsigned int*)(ws + L.find("range_flag")) : nullptr;
    range_flag_slot() = range_flag;

    // Step 1 (models/PointDSC.py:150-155): compat, then the SCNonlocal encoder
    const bool compat16 = split && cfg->compat_format == PDSC_COMPAT_U16;
    if (compat16)
        PDSC_TRY(pdsc_spatial_compat_u16(src, tgt, W(PDSC_W_SIGMA_SPAT, 0), (unsigned short*)compat, ld, bs, N, stream));
    else
        PDSC_TRY(pdsc_spatial_compat(src, tgt, W(PDSC_W_SIGMA_SPAT, 0), compat, nullptr, ld, bs, N, stream));
    auto attention_split = [&](float* msg_out, int nsplit) {
        return launch_attention_split_ex(q_split, kv_tiles, compat, compat16 ? PDSC_COMPAT_U16 : PDSC_COMPAT_F32, ld, msg_out, att_scratch,
                                         att_bytes, bs, N, nsplit, PDSC_PARTIALS_ROWS, nvalid, hst);
    };
    PDSC_TRY(launch_layer0(corr_pos, cfg->in_dim, W(PDSC_W_LAYER0_W, 0), W(PDSC_W_LAYER0_B, 0), featA, M, range_flag, bs, hst));
    // probe != NULL (pdsc_encoder_range_probe): one launch per conv, every intermediate in the workspace, |max| of each kind recorded
    const int fused = probe ? 0 : env_int("PDSC_FUSED_LAYERS", 1);          // tuning/A-B knob: 0 = one pdsc_linear launch per conv
    auto P = [&](int kind, const float* x, size_t count) { return probe ? launch_absmax(x, count, probe + kind, hst) : PDSC_OK; };
    PDSC_TRY(P(PDSC_RANGE_LAYER0, featA, (size_t)M * C));
    if (fused && cfg->num_layers > 0 && split) {
        // split precision: head of layer 0, then per layer attention (partials left un-merged when the keys are split)
        // + ONE launch for the merge, the tail of layer i and the head of layer i+1
        const int ns = pdsc_attention_split_default_split(bs, N);
        const int Npad = (int)round_up(N, 256);
        const int fuse_env = env_int("PDSC_FUSE_MERGE", 1);      // tuning/A-B knob
        // the layer kernels merge the key-split partials while loading (merge_partials.h): up to 4 splits, 8 in the
        // workgroup-per-tile kernel that small problems take
        const char* var = env_str("PDSC_LAYER_VARIANT");          // (experiments builds) b / w: force the workgroup-per-tile / wavefront kernel
        // arithmetic of fc1..fc3 / PointCN (enum pdsc_layer_gemm); A/B knob PDSC_LAYER_GEMM = 0 / 1 overrides
        const int gemm = env_int("PDSC_LAYER_GEMM", cfg->layer_gemm) == PDSC_LAYER_GEMM_H3 ? PDSC_LAYER_GEMM_H3 : PDSC_LAYER_GEMM_F32;
        // Which layer kernel.  H3 GEMMs: layer_h3_kernel, or the bit-identical layer_h3_coop_kernel for launches of at most 2560
        // tiles (launch_layer_h3 decides) -- never the fp32 kernels: N = 1000 x 1 0.384 ms per forward against 0.557 with the
        // workgroup-per-tile fp32 kernel (profiles/r03_b_ab_*.txt, r03_k_ab_coop.txt).  fp32 GEMMs: layer_wave_kernel, or the
        // workgroup-per-tile kernel of layer.hip for small problems (pdsc_layer_prefers_block).
        const bool small = pdsc_layer_prefers_block(bs, N) && gemm != PDSC_LAYER_GEMM_H3;
        const bool block_layer = !x3_gemm && ((var && var[0] == 'b') || (!(var && var[0] == 'w') && small));
        // tuning/A-B knob: PDSC_LAYER_FRAG = 0 = natural-layout weights (pdsc_layer_fused_split)
        const bool frag_env = env_int("PDSC_LAYER_FRAG", 1) && !(var && var[0] == 'b');
        const bool frag = frag_env && !x3_gemm && ((var && var[0] == 'w') || !small);
        // (merge_partials.h: 4 splits in layer_wave.hip, 8 in the workgroup-per-tile kernel and in layer_h3.hip)
        const bool h3_kernel = frag && gemm == PDSC_LAYER_GEMM_H3 && env_int("PDSC_LAYER_PF", 1) != 0 && env_int("PDSC_LAYER_H3_VARIANT", 1) != 0;
        const bool fuse_merge = fuse_env && ns > 1 && ns <= ((block_layer || h3_kernel) ? 8 : 4);
        const float* part_o = fuse_merge ? (const float*)att_scratch : nullptr;
        const float* part_ml = fuse_merge ? part_o + (size_t)bs * ns * Npad * C : nullptr;
        const int ws_tail = gemm == PDSC_LAYER_GEMM_H3 ? PDSC_WS_FRAG_TAIL_H3 : PDSC_WS_FRAG_TAIL;
        const int ws_head = gemm == PDSC_LAYER_GEMM_H3 ? PDSC_WS_FRAG_HEAD_H3 : PDSC_WS_FRAG_HEAD;
        // H3 + fused merge: the hand-offs attention -> layer kernel -> next layer kernel in point-fragment order (split_layout.h);
        // A/B knob PDSC_LAYER_PF = 0: plain rows
        // leaf form (r05, enum pdsc_att_leaves): the key range cut into leaves that depend on N alone; the H3 layer kernel merges the
        // leaf partials exactly as it merges key-split partials (the other layer kernels keep the per-launch key split)
        const bool pf_ok = frag && gemm == PDSC_LAYER_GEMM_H3 && env_int("PDSC_LAYER_PF", 1) != 0 && env_int("PDSC_LAYER_H3_VARIANT", 1) != 0;
        int lf_ns = 0, lf_leaves = 0, lf_nw = 0;
        if (cfg->att_leaves >= PDSC_LEAVES_CANONICAL) leaf_plan(bs, N, cfg->att_leaves, &lf_nw, &lf_ns, &lf_leaves);
        const bool leaves = pf_ok && cfg->att_leaves >= PDSC_LEAVES_CANONICAL && (!nvalid || (n_min + 31) / 32 >= 2 * lf_leaves);
        if (nvalid && !leaves)
            PDSC_REQUIRE((n_min + 31) / 32 >= pdsc_attention_split_default_split(bs, N),
                         "pdsc_forward_testing_ragged: the shortest pair (%d correspondences) has fewer 32-key tiles than the key split "
                         "planned for bs=%d, N=%d (%d): batch pairs of more similar size", n_min, bs, N, pdsc_attention_split_default_split(bs, N));
        const bool pf = leaves || (pf_ok && fuse_merge);
        const float* lf_o = (const float*)att_scratch;
        const float* lf_ml = lf_o + (size_t)bs * lf_leaves * Npad * C;
        if (x3_gemm)
            PDSC_TRY(pdsc_layer_fused_x3(nullptr, nullptr, nullptr, 0, 0, nullptr, featA, nullptr, featB, nullptr, q_split, kv_tiles,
                                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, WS(PDSC_W_PCN_W, 0), W(PDSC_W_PCN_B, 0),
                                         WS(PDSC_W_QKV_W, 0), W(PDSC_W_QKV_B, 0), bs, N, stream));
        else if (pf)
            PDSC_TRY(pdsc_layer_fused_frag_io(nullptr, nullptr, nullptr, 0, 0, nullptr, featA, nullptr, featB, q_split, kv_tiles,
                                              nullptr, WS(ws_head, 0), gemm, PDSC_IO_FEATB_PF, bs, N, stream));
        else if (frag)
            PDSC_TRY(pdsc_layer_fused_frag_fmt(nullptr, nullptr, nullptr, 0, 0, nullptr, featA, nullptr, featB, nullptr, q_split, kv_tiles,
                                               nullptr, WS(ws_head, 0), gemm, bs, N, stream));
        else
            PDSC_TRY(pdsc_layer_fused_split(nullptr, nullptr, nullptr, 0, 0, nullptr, featA, nullptr, featB, nullptr, q_split, kv_tiles,
                                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, W(PDSC_W_PCN_W, 0), W(PDSC_W_PCN_B, 0),
                                            W(PDSC_W_QKV_W, 0), W(PDSC_W_QKV_B, 0), WS(PDSC_W_QKV_W, 0), bs, N, stream));
        float *cur = featB, *nxt = featC;
        for (int i = 0; i < cfg->num_layers; ++i) {
            if (leaves)
                PDSC_TRY(launch_attention_leaves(q_split, kv_tiles, compat, compat16 ? PDSC_COMPAT_U16 : PDSC_COMPAT_F32, ld, att_scratch,
                                                 att_bytes, bs, N, cfg->att_leaves, nvalid, n_min, hst));
            else if (pf)
                PDSC_TRY(launch_attention_split_ex(q_split, kv_tiles, compat, compat16 ? PDSC_COMPAT_U16 : PDSC_COMPAT_F32, ld, nullptr,
                                                   att_scratch, att_bytes, bs, N, ns, PDSC_PARTIALS_PF, nvalid, hst));
            else
                PDSC_TRY(attention_split(fuse_merge ? nullptr : msg, ns));
            const bool last = i + 1 == cfg->num_layers;
            if (pf)
                PDSC_TRY(pdsc_layer_fused_frag_io(nullptr, leaves ? lf_o : part_o, leaves ? lf_ml : part_ml, leaves ? lf_leaves : ns, Npad, cur, nullptr, last ? featA : nullptr,
                                                  last ? nullptr : nxt, last ? nullptr : q_split, last ? nullptr : kv_tiles,
                                                  WS(ws_tail, i), last ? nullptr : WS(ws_head, i + 1), gemm,
                                                  PDSC_IO_PARTIALS_PF | PDSC_IO_RES_PF | (last ? 0 : PDSC_IO_FEATB_PF), bs, N, stream));
            else if (x3_gemm)
                PDSC_TRY(pdsc_layer_fused_x3(fuse_merge ? nullptr : msg, part_o, part_ml, ns, Npad, cur, nullptr, last ? featA : nullptr,
                                             last ? nullptr : nxt, nullptr, last ? nullptr : q_split, last ? nullptr : kv_tiles,
                                             WS(PDSC_W_FC1_W, i), W(PDSC_W_FC1_B, i), WS(PDSC_W_FC2_W, i), W(PDSC_W_FC2_B, i),
                                             WS(PDSC_W_FC3_W, i), W(PDSC_W_FC3_B, i),
                                             last ? nullptr : WS(PDSC_W_PCN_W, i + 1), last ? nullptr : W(PDSC_W_PCN_B, i + 1),
                                             last ? nullptr : WS(PDSC_W_QKV_W, i + 1), last ? nullptr : W(PDSC_W_QKV_B, i + 1),
                                             bs, N, stream));
            else if (frag)
                PDSC_TRY(pdsc_layer_fused_frag_fmt(fuse_merge ? nullptr : msg, part_o, part_ml, ns, Npad, cur, nullptr,
                                                   last ? featA : nullptr, last ? nullptr : nxt, nullptr, last ? nullptr : q_split,
                                                   last ? nullptr : kv_tiles, WS(ws_tail, i),
                                                   last ? nullptr : WS(ws_head, i + 1), gemm, bs, N, stream));
            else
                PDSC_TRY(pdsc_layer_fused_split(fuse_merge ? nullptr : msg, part_o, part_ml, ns, Npad, cur, nullptr,
                                                last ? featA : nullptr, last ? nullptr : nxt, nullptr, last ? nullptr : q_split,
                                                last ? nullptr : kv_tiles,
                                                W(PDSC_W_FC1_W, i), W(PDSC_W_FC1_B, i), W(PDSC_W_FC2_W, i), W(PDSC_W_FC2_B, i),
                                                W(PDSC_W_FC3_W, i), W(PDSC_W_FC3_B, i),
                                                last ? nullptr : W(PDSC_W_PCN_W, i + 1), last ? nullptr : W(PDSC_W_PCN_B, i + 1),
                                                last ? nullptr : W(PDSC_W_QKV_W, i + 1), last ? nullptr : W(PDSC_W_QKV_B, i + 1),
                                                last ? nullptr : WS(PDSC_W_QKV_W, i + 1), bs, N, stream));
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    } else if (fused && cfg->num_layers > 0) {
        // exact fp32: head of layer 0, then per layer: attention + (tail of layer i fused with head of layer i+1)
        PDSC_TRY(pdsc_layer_fused(nullptr, nullptr, featA, nullptr, featB, qkv, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, W(PDSC_W_PCN_W, 0), W(PDSC_W_PCN_B, 0), W(PDSC_W_QKV_W, 0), W(PDSC_W_QKV_B, 0), M,
                                  stream));
        float *cur = featB, *nxt = featC;
        for (int i = 0; i < cfg->num_layers; ++i) {
            PDSC_TRY(launch_attention_fp32(qkv, compat, ld, msg, att_scratch, att_bytes, bs, N, 0, nvalid, hst));      // (r06: ragged batches too)
            const bool last = i + 1 == cfg->num_layers;
            PDSC_TRY(pdsc_layer_fused(msg, cur, nullptr, last ? featA : nullptr, last ? nullptr : nxt, last ? nullptr : qkv,
                                      W(PDSC_W_FC1_W, i), W(PDSC_W_FC1_B, i), W(PDSC_W_FC2_W, i), W(PDSC_W_FC2_B, i),
                                      W(PDSC_W_FC3_W, i), W(PDSC_W_FC3_B, i),
                                      last ? nullptr : W(PDSC_W_PCN_W, i + 1), last ? nullptr : W(PDSC_W_PCN_B, i + 1),
                                      last ? nullptr : W(PDSC_W_QKV_W, i + 1), last ? nullptr : W(PDSC_W_QKV_B, i + 1), M,
                                      stream));
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    } else
    for (int i = 0; i < cfg->num_layers; ++i) {
        PDSC_TRY(pdsc_linear(featA, C, W(PDSC_W_PCN_W, i), W(PDSC_W_PCN_B, i), nullptr, 0, featB, C, M, C, C, 1, stream));
        PDSC_TRY(P(PDSC_RANGE_POINTCN, featB, (size_t)M * C));
        PDSC_TRY(pdsc_linear(featB, C, W(PDSC_W_QKV_W, i), W(PDSC_W_QKV_B, i), nullptr, 0, qkv, 3 * C, M, C, 3 * C, 0, stream));
        PDSC_TRY(P(PDSC_RANGE_QKV, qkv, (size_t)M * 3 * C));
        if (split) {
            PDSC_TRY(pdsc_pack_qkv_split(qkv, q_split, kv_tiles, bs, N, stream));
            PDSC_TRY(attention_split(msg, 0));
        } else
            PDSC_TRY(pdsc_sc_attention(qkv, compat, ld, msg, att_scratch, att_bytes, bs, N, 0, stream));
        PDSC_TRY(P(PDSC_RANGE_MESSAGE, msg, (size_t)M * C));
        PDSC_TRY(pdsc_linear(msg, C, W(PDSC_W_FC1_W, i), W(PDSC_W_FC1_B, i), nullptr, 0, t64a, C / 2, M, C, C / 2, 1, stream));
        PDSC_TRY(P(PDSC_RANGE_FC1, t64a, (size_t)M * (C / 2)));
        PDSC_TRY(pdsc_linear(t64a, C / 2, W(PDSC_W_FC2_W, i), W(PDSC_W_FC2_B, i), nullptr, 0, t64b, C / 2, M, C / 2, C / 2, 1, stream));
        PDSC_TRY(P(PDSC_RANGE_FC2, t64b, (size_t)M * (C / 2)));
        PDSC_TRY(pdsc_linear(t64b, C / 2, W(PDSC_W_FC3_W, i), W(PDSC_W_FC3_B, i), featB, C, featA, C, M, C / 2, C, 0, stream));
        PDSC_TRY(P(PDSC_RANGE_FEATURE, featA, (size_t)M * C));
    }
    if (probe) return PDSC_OK;
    // pdsc_forward_testing_streams: everything after the encoder -- a strictly sequential chain of ~15 small, latency-bound
    // launches -- goes to the caller's second (high-priority) stream: with several forwards in flight its workgroups are then
    // dispatched ahead of the queued workgroups of another forward's attention launch instead of behind them.
    void* const main_stream = stream;
    if (tail_stream) {
        if (hipEventRecord((hipEvent_t)ev_fork, (hipStream_t)main_stream) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)tail_stream, (hipEvent_t)ev_fork, 0) != hipSuccess)
            return check_launch("pdsc_forward_testing_streams(fork)");
        stream = tail_stream;
        hst = (hipStream_t)tail_stream;
    }
    // Step 2.1 (:156,:171,:174): normalise, confidence head, NMS seeds
    if (env_int("PDSC_CLS_FUSED", 1))       // (A/B knob, experiments builds: 0 = the two pdsc_linear launches of r01-r03; same bits)
        PDSC_TRY(launch_classifier_hidden(featA, W(PDSC_W_CLS1_W, 0), W(PDSC_W_CLS1_B, 0), W(PDSC_W_CLS2_W, 0), W(PDSC_W_CLS2_B, 0), h2, M, hst));
    else {
        PDSC_TRY(pdsc_linear(featA, C, W(PDSC_W_CLS1_W, 0), W(PDSC_W_CLS1_B, 0), nullptr, 0, h1, 32, M, C, 32, 1, stream));
        PDSC_TRY(pdsc_linear(h1, 32, W(PDSC_W_CLS2_W, 0), W(PDSC_W_CLS2_B, 0), nullptr, 0, h2, 32, M, 32, 32, 1, stream));
    }
    // (mode 0, large batches: the seeds' kNN runs fused -- knn_fused_kernel -- and takes the normalised rows in point-fragment order too)
    const bool knn_fused = mode == 0 && knn_seeds_uses_fused(bs, N, S, k);
    float* normed_pf = knn_fused ? F("normed_pf") : nullptr;
    if (knn_fused)
        PDSC_TRY(launch_normalize_conf_pf(featA, h2, W(PDSC_W_CLS3_W, 0), W(PDSC_W_CLS3_B, 0), normed, normed_pf, conf, bs, N, hst));
    else
        PDSC_TRY(pdsc_normalize_confidence(featA, h2, W(PDSC_W_CLS3_W, 0), W(PDSC_W_CLS3_B, 0), normed, conf, M, stream));
    if (mode == 0) {
        PDSC_TRY(launch_nms_keys_grid(src, conf, cfg->nms_radius, keys, ws + L.find("nms_ws"), pdsc_nms_workspace_bytes(bs, N), bs, N, nvalid, hst));
        PDSC_TRY(launch_rank_select(keys, seeds, bs, N, S, nvalid, svalid, hst, conv_mask));      // (+ the solver's mask := all-ones)
    } else {
        // models/PointDSC.py:158-163 and :176
        PDSC_TRY(pdsc_feature_compat(normed, W(PDSC_W_SIGMA, 0), Mout, ldM, bs, N, stream));
        PDSC_TRY(pdsc_rank_select(conf, seeds, bs, N, S, stream));
    }
    // Step 3 & 4 (:182 -> :234-336): per-seed hypotheses, scoring, best
    PDSC_TRY(launch_knn_seeds_form(normed, normed_pf, seeds, knn_dist, knn_idx, bs, N, S, k, nvalid, knn_fused ? 2 : 1, hst));
    if (mode == 1 && bs > 1) {
        // validation forward: the early exit is taken over the seeds of ALL pairs of the batch (one torch.allclose over
        // [bs*S, k]) -- the per-pair masks are AND-ed before the iterate is chosen, so the two steps stay apart
        PDSC_TRY(pdsc_seed_power_iteration(normed, src, tgt, knn_idx, W(PDSC_W_SIGMA, 0), W(PDSC_W_SIGMA_SPAT, 0), eig,
                                           conv_mask, nullptr, bs, N, S, k, cfg->num_iterations, stream));
        PDSC_TRY(pdsc_conv_mask_all_pairs(conv_mask, bs, stream));
        PDSC_TRY(pdsc_seed_transforms(src, tgt, knn_idx, eig, conv_mask, seed_trans, seed_w, bs, N, S, k,
                                      cfg->num_iterations, stream));
    } else
        PDSC_TRY(launch_seed_solve_forward(normed, src, tgt, knn_idx, W(PDSC_W_SIGMA, 0), W(PDSC_W_SIGMA_SPAT, 0), eig, conv_mask,
                                           seed_trans, seed_w, bs, N, S, k, cfg->num_iterations, /*mask_ready=*/mode == 0, hst));
#ifdef PDSC_EXPERIMENTS
    score_debug_slot() = env_int("PDSC_SCORE_DEBUG", 0) ? F("score_dbg") : nullptr;
#endif
    PDSC_TRY(launch_score_hypotheses(seed_trans, src, tgt, cfg->inlier_threshold, counts, bs, N, S, nvalid, hst));
#ifdef PDSC_EXPERIMENTS
    score_debug_slot() = nullptr;
#endif
    if (mode == 0) {
        // best hypothesis + its labels, then post refinement (:186 -> :403-438) in the same launch; final_labels stay those of the
        // pre-refinement best hypothesis
        PDSC_TRY(launch_select_and_refine(counts, seed_trans, src, tgt, cfg->inlier_threshold, cfg->refine_threshold, cfg->refine_iters, best, initial,
                                          final_labels, final_trans, solves, bs, N, S, nvalid, hst, I("refine_trace"), range_flag, range_report_slot()));
    } else {
        // best hypothesis is the result (:186 is skipped); the labels of the call are the logits (:190-191)
        PDSC_TRY(pdsc_select_best(counts, seed_trans, src, tgt, cfg->inlier_threshold, best, final_trans, keys /* scratch */,
                                  bs, N, S, stream));
        PDSC_TRY(launch_copy_u32((unsigned int*)final_labels, (const unsigned int*)conf, (size_t)M, (hipStream_t)stream));
    }
    if (tail_stream) {       // join: whatever the caller enqueues on the main stream next is ordered after the results
        if (hipEventRecord((hipEvent_t)ev_join, (hipStream_t)tail_stream) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)main_stream, (hipEvent_t)ev_join, 0) != hipSuccess)
            return check_launch("pdsc_forward_testing_streams(join)");
    }
    return PDSC_OK;
}

extern "C" int pdsc_forward_testing(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                                    const float* src, const float* tgt, int bs, int N, int num_seeds,
                                    float* final_trans, float* final_labels, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return run_forward(0, cfg, wpack, wsplit, corr_pos, src, tgt, bs, N, num_seeds, final_trans, final_labels, nullptr, 0,
                       workspace, workspace_bytes, stream);
}

extern "C" int pdsc_forward_testing_ragged(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                                           const float* src, const float* tgt, int bs, int N, int num_seeds, const int* num_corr,
                                           const int* num_seeds_per_pair, int n_min, float* final_trans, float* final_labels,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    PDSC_REQUIRE(num_corr && num_seeds_per_pair, "pdsc_forward_testing_ragged: the per-pair count arrays (device, [bs] int32) are required");
    return run_forward(0, cfg, wpack, wsplit, corr_pos, src, tgt, bs, N, num_seeds, final_trans, final_labels, nullptr, 0,
                       workspace, workspace_bytes, stream, num_corr, num_seeds_per_pair, n_min);
}

extern "C" int pdsc_forward_testing_streams(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                                            const float* src, const float* tgt, int bs, int N, int num_seeds, const int* num_corr,
                                            const int* num_seeds_per_pair, int n_min, float* final_trans, float* final_labels,
                                            void* workspace, size_t workspace_bytes, void* stream, void* tail_stream, void* fork_event,
                                            void* join_event) {
    PDSC_REQUIRE((num_corr == nullptr) == (num_seeds_per_pair == nullptr), "pdsc_forward_testing_streams: both count arrays or neither");
    PDSC_REQUIRE(tail_stream && fork_event && join_event, "pdsc_forward_testing_streams: tail stream and both events are required");
    return run_forward(0, cfg, wpack, wsplit, corr_pos, src, tgt, bs, N, num_seeds, final_trans, final_labels, nullptr, 0,
                       workspace, workspace_bytes, stream, num_corr, num_seeds_per_pair, n_min, tail_stream, fork_event, join_event);
}

// Range probe for layer_gemm = PDSC_LAYER_GEMM_H3 (fp16 hi/lo operands: every activation of the chain must stay below 65504): the
// encoder once with the fp32 GEMMs, one launch per conv, and the largest |value| of every activation kind over all layers.
extern "C" int pdsc_encoder_range_probe(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                                        const float* src, const float* tgt, int bs, int N, int num_seeds, float* absmax,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    PDSC_REQUIRE(absmax, "pdsc_encoder_range_probe: absmax [PDSC_RANGE_NUM_KINDS] (device) required");
    PDSC_TRY(launch_fill_u32((unsigned int*)absmax, 0u, PDSC_RANGE_NUM_KINDS, (hipStream_t)stream));
    float dummy_T = 0.f;         // (outputs of the tail are not produced in probe mode; the pointers only pass the null checks)
    return run_forward(0, cfg, wpack, wsplit, corr_pos, src, tgt, bs, N, num_seeds, &dummy_T, &dummy_T, nullptr, 0, workspace,
                       workspace_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, nullptr, (unsigned int*)absmax);
}

extern "C" int pdsc_forward_validation(const pdsc_config* cfg, const float* wpack, const void* wsplit, const float* corr_pos,
                                       const float* src, const float* tgt, int bs, int N, int num_seeds,
                                       float* final_trans, float* logits, float* Mout, long long ldM, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    PDSC_REQUIRE(Mout && ldM >= N, "pdsc_forward_validation: M matrix [bs][N][ldM >= N] required");
    return run_forward(1, cfg, wpack, wsplit, corr_pos, src, tgt, bs, N, num_seeds, final_trans, logits, Mout, ldM,
                       workspace, workspace_bytes, stream);
}
