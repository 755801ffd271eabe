// What the translation units of the host side (ppg_front.cpp: context, CNN, pipeline; ppg_match.cpp: distances, matchers,
// frame grid, vocabulary) share: the context itself, the ring of result slots, and the helpers every entry point uses.
// Not part of the C ABI.
#pragma once
#include "../../include/ppgfront.h"
#include "ppg_kernels.h"
#include "ppg_internal.h"
#include "host_pool.h"
#include "lines_tail.h"
#include "vocab.h"

#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

namespace ppgi {

struct Tensor {
    std::vector<uint32_t> dims;
    std::vector<float> v;
};

struct DevConv {
    int ks = 3, nt = 2, cin = 0, cout = 0, nblk = 0;
    float *w = nullptr, *bias = nullptr, *alpha = nullptr, *beta = nullptr;
    bool lowp = false;  // w holds the bf16 hi/lo image of conv_bf16_kernel (reduced-precision modes)
};

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
};

// Everything a batch hands to the host lives kSlots times (a ring), so that four batches can be in
// flight: two in the CNN (one per lane), one in post-processing / matching / copy-out, one in the host's
// line-graph tail
// (ppg_submit_batch* / ppg_wait_batch).  The slot before the current one also is "the previous frame" of
// every stream for frame-to-frame matching.
constexpr int kSlots = 5;      // one more than kMaxInflight: the slot handed out by a wait is not the next one to be reused
constexpr int kMaxInflight = 4;
constexpr int kLanes = 2;
inline int next_slot(int s) { return (s + 1) % kSlots; }
inline int prev_slot(int s) { return (s + kSlots - 1) % kSlots; }
struct Slot {
    // device
    long long uses = 0;           // batches that have been started on this slot
    int* counters = nullptr;      // [B][8]: 0 resid, 1 kept, 2 line candidates, 3 key points
    int32_t* kp_xy = nullptr;
    float *kp_score = nullptr, *kp_un = nullptr, *desc = nullptr, *cand_sc = nullptr;
    uint8_t* kp_out = nullptr;
    uint32_t* cand_ij = nullptr;
    int32_t* d_match = nullptr;   // [B][maxn]
    int* d_mcounts = nullptr;     // [B]
    int* sat = nullptr;           // precision 3: set by the convolutions when an activation saturated the fp16 split
    float *junc = nullptr, *hs2 = nullptr, *descmap = nullptr;  // CNN outputs read by the post stream
    // line graph as the device tail leaves it (and the pinned mirrors the host reads)
    uint32_t *edge_idx = nullptr, *conn_idx = nullptr, *coline_pairs = nullptr;
    float* edge_score = nullptr;
    int32_t *conn_off = nullptr, *coline_off = nullptr;
    uint32_t *h_edge_idx = nullptr, *h_conn_idx = nullptr, *h_coline_pairs = nullptr;
    float* h_edge_score = nullptr;
    int32_t *h_conn_off = nullptr, *h_coline_off = nullptr;
    hipEvent_t cnn_done = nullptr;
    hipEvent_t junc_done = nullptr;    // the junction logits are complete (the key-point kernels start beside the last three convolutions)
    hipEvent_t heat_done = nullptr;    // EdgeHeatmap's convolutions are complete (the line kernels need nothing else)
    hipEvent_t match_done = nullptr;   // the batch's match against the previous one and its copies (match stream)
    // everything the host reads of a batch lies in one device block and one pinned block of the same layout, so a
    // full batch comes back in a single copy (with one frame per call thirteen small copies were a tenth of the call)
    uint8_t *d_res = nullptr, *h_res = nullptr;
    size_t res_bytes = 0;
    // pinned host mirrors
    int* h_counters = nullptr;
    int32_t* h_kp_xy = nullptr;
    float *h_kp_score = nullptr, *h_kp_un = nullptr, *h_desc = nullptr;
    uint8_t* h_kp_out = nullptr;
    int32_t* h_match = nullptr;
    int* h_mcounts = nullptr;
    int* h_sat = nullptr;
    uint8_t* h_img = nullptr;     // pinned staging for host images
    uint8_t* d_img = nullptr;     // device copy of the slot's host images
    std::vector<FrameHost> frames;
    std::vector<int> nkp;         // key points per frame, known after the slot was collected
    hipEvent_t done = nullptr;    // recorded behind the slot's last device-to-host copy
    int n = 0;                    // frames extracted into this slot
    bool inflight = false;        // submitted, not yet waited for
    bool matched = false;         // the submit included the match against the other slot
};

// The CNN intermediates and their stream exist twice: consecutive batches alternate between the two
// lanes, so the work-groups of one batch's convolution fill the partially empty last rounds of the other's.
struct Lane {
    hipStream_t stream = nullptr;
    float *a1 = nullptr, *p1 = nullptr, *a2 = nullptr, *p2 = nullptr, *a3 = nullptr, *p3 = nullptr, *a4 = nullptr,
          *feat = nullptr, *pa_o = nullptr, *da_o = nullptr, *hs0 = nullptr, *hs1 = nullptr;
    int* tile_ctr = nullptr;  // kTileCounters zeroed ints: hand-out counters of the convolutions that take several tiles per work-group
};
constexpr int kTileCounters = 256;


}  // namespace ppgi
using namespace ppgi;

struct ppg_ctx {
    ppg_config cfg;
    std::string weights_path;
    std::string err;
    int W = 0, H = 0, w8 = 0, h8 = 0, B = 0, maxn = 0;
    bool matcher_only = false;   // ppg_create_matcher: streams and the matcher's work arrays only
    Lane lane[kLanes];
    int lane_cur = 0, nlanes = 1;  // PPG_CNN_LANES=2 alternates consecutive batches between two CNN streams
    hipStream_t stream = nullptr;       // CNN: image upload and every convolution (alias of the current lane's)
    hipStream_t post = nullptr;         // everything after the CNN, matching, copies to the host
    hipStream_t copy_stream = nullptr;  // late copies of an already finished slot
    hipStream_t up_stream = nullptr;    // image uploads: the copy of batch k+1 runs beside the CNN of batch k
    // all-pairs match of resident descriptors: beside the post-processing of the next batch instead of in front of it
    // (with one frame per call the claim pass alone is a quarter of the post stream's time)
    hipStream_t match_stream = nullptr;
    hipEvent_t ev_desc_ready = nullptr;
    std::vector<void*> dev_allocs, host_allocs;
    // weights
    DevConv c1b, c2a, c2b, c3a, c3b, c4a, c4b, pa, pb, da, db, h0, h1, h2;
    float *c1a_w = nullptr, *c1a_b = nullptr, *h3_w = nullptr, *h3_b = nullptr;
    float* heads_w = nullptr;  // packed weights of convPa | convDa | EdgeHeatmap block 0, concatenated
    // activations
    float *a1 = nullptr, *p1 = nullptr, *a2 = nullptr, *p2 = nullptr, *a3 = nullptr, *p3 = nullptr, *a4 = nullptr,
          *feat = nullptr, *pa_o = nullptr, *junc = nullptr, *da_o = nullptr, *descmap = nullptr, *hs0 = nullptr,
          *hs1 = nullptr, *hs2 = nullptr, *heat_raw = nullptr, *prob = nullptr, *refined = nullptr, *score = nullptr;
    float *mapx = nullptr, *mapy = nullptr, *remapped = nullptr;  // lens: undistortion maps and the remapped heat map (D[0] != 0 only)
    uint8_t* state = nullptr;
    uint32_t *resid = nullptr, *kept = nullptr;
    unsigned long long* kept_key = nullptr;
    int* counters = nullptr;
    int resid_cap = 0, kept_cap = 0, cand_cap = 0;
    // work arrays of the device line-graph tail (ppg::LineTailArgs)
    int tail_cap = 0, pair_cap = 0, edge_cap = 0;
    float* lt_geo = nullptr;
    int *lt_col_off = nullptr, *lt_col_cur = nullptr;
    float* lt_col_ent = nullptr;
    uint32_t *lt_pairs = nullptr, *lt_kill = nullptr, *lt_pend = nullptr, *lt_cst = nullptr;
    bool lines_on_host = false;           // PPG_LINES_HOST=1: every frame's tail on the host (the statement the device tail is checked against)
    std::vector<uint32_t> fb_ij;          // host tail: one frame's candidates, fetched only for the frames that need it
    std::vector<float> fb_sc;
    long long tail_fallbacks = 0, tail_frames = 0;
    int32_t* kp_xy = nullptr;
    float *kp_score = nullptr, *kp_un = nullptr, *desc = nullptr;
    uint8_t* kp_out = nullptr;
    uint32_t* cand_ij = nullptr;
    float* cand_sc = nullptr;
    int* row_cnt = nullptr;
    unsigned long long* pair_mask = nullptr;
    int mask_words = 0;
    // pinned host mirrors
    uint8_t* h_img = nullptr;
    int* h_counters = nullptr;
    int32_t* h_kp_xy = nullptr;
    float *h_kp_score = nullptr, *h_kp_un = nullptr, *h_desc = nullptr;
    uint8_t* h_kp_out = nullptr;
    std::vector<FrameHost>* frames = nullptr;
    Slot slot[kSlots];
    int cur = 0;  // slot the alias pointers above refer to
    std::vector<TailScratch> tails;
    int host_threads = 1;
    const char* dump_tail = nullptr;
    bool dump_tail_done = false;
    HostPool* pool = nullptr;  // host_threads - 1 workers, created with the first batch that needs them
    bool skip_score = false, skip_heat = false, skip_refine = false;
    bool unfused_conv1a = false;
    int last_n = 0;
    // previous extract call's descriptors / counts stay resident for frame-to-frame matching
    float* desc_prev = nullptr;
    int* counters_prev = nullptr;
    int32_t* d_match = nullptr;
    int* d_mcounts = nullptr;
    // optional per-kernel-class timing with HIP events on the context's stream
    int prof = 0;  // 0 off, 1 every class, 2 device events for the fused first convolution only (+ host timers)
    std::vector<hipEvent_t> ev_all, ev_free;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_pairs;
    double prof_ms[PPG_PROF_N] = {};
    long long prof_launches[PPG_PROF_N] = {};
    double prof_flops[PPG_PROF_N] = {};  // algorithmic FLOPs enqueued per class while profiling
    // matcher scratch (grown on demand)
    DBuf<float> mA, mB, mD, mBest, mSecond, mTopD;
    DBuf<int32_t> mOff, mIdx, mMatch, mTopI;
    DBuf<uint8_t> mSkip, mSkipOut, mRowClaims, xArena;
    uint8_t* xPin = nullptr;         // pinned host image of xArena (ppg_extend_map_matches)
    size_t xPinCap = 0;
    bool extend_on_host = false;     // PPG_EXTEND_HOST=1: the host walk (the statement the device walk is checked against)
    DBuf<uint32_t> mClaim;
    DBuf<float> mNorm;
    DBuf<int32_t> mListCnt, mNearLen;
    DBuf<float> mListRad;
    DBuf<uint16_t> mListCol, mHalf;
    DBuf<float> mBt;
    DBuf<unsigned long long> mExt;
    // device copy of the vocabulary last used with ppg_bow_transform*
    unsigned long long voc_serial = 0;
    float* voc_desc = nullptr;
    int *voc_child_off = nullptr, *voc_child_idx = nullptr, *voc_word = nullptr;
    double* voc_weight = nullptr;
    DBuf<int32_t> bWord, bNode;
    DBuf<double> bWeight;
    // Work-groups per cout block of the fused first / the other fp32 convolutions when a launch has at least twenty
    // times as many tiles (each work-group then walks the tiles with that stride; 0 = one work-group per tile).
    // PPG_CONV_PERSIST="f,o" overrides both.
    int persist_fused = 0, persist_other = 0;
    int xcd_group = 4, xcd_band = 1;   // XCD-aware launch order of the convolutions (ppg::ConvArgs::xcd_group); PPG_XCD_MAP=0 switches it off
    bool persist_forced = false;  // set by PPG_CONV_PERSIST: the numbers apply to every launch, whatever its size
    // Which reduced-precision modes run their large 3x3 layers wave-specialised (conv_ws.hip).  Default: mode 1 only, where
    // the step gains 2-3 %; in mode 3 the step does not move, and in mode 2 (a third of the products) a kernel that takes
    // every CU for itself costs the other streams' kernels more than it saves.  PPG_CONV_WS=1: every reduced-precision
    // mode; =0: none (the one-tile-per-work-group kernel the wave-specialised one is checked against).
    int last_out_slot = -1, last_out_n = 0;   // the slot and frame count the last ppg_extract* / ppg_wait_batch handed out
    long long last_out_use = 0;               // that slot's `uses` then: a later batch on the slot overwrites its device data
    int conv_ws = -1;
    long conv_ws_min_units = 1024;   // (tile, cout block) units a launch needs before it runs wave-specialised (PPG_CONV_WS_MIN)
    bool no_quarter = false;    // PPG_NO_QUARTER_TILES=1: big convolution tiles whatever the batch size (read at ppg_create)
    bool match_matrix = false;  // PPG_MATCH_MATRIX=1: resident matches through the full distance matrix (the path the fused one is checked against)
    int* mCount = nullptr;
    int* h_mcount = nullptr;
    int* h_mcounts = nullptr;    // alias of the current slot's
    int32_t* h_match = nullptr;  // alias of the current slot's
};

#define PPG_HIDDEN __attribute__((visibility("hidden")))
// (defined in ppg_front.cpp)
PPG_HIDDEN int fail(ppg_ctx* c, int code, const char* fmt, ...);   // records the message in the context (or for ppg_last_error(NULL)) and returns code
PPG_HIDDEN hipEvent_t prof_event(ppg_ctx* c);
PPG_HIDDEN void prof_resolve(ppg_ctx* c);
PPG_HIDDEN bool any_inflight(const ppg_ctx* c);
// (defined in ppg_match.cpp)
PPG_HIDDEN int scratch_setup(ppg_ctx* c, int dev);   // where this GPU keeps the work array of ppg_match_nn_noalloc; once per device
// (match_resident: declared in ppg_internal.h)

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail((c), PPG_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// the entry points of the extractor (and those that read resident frames) do not exist on a matcher-only context
#define NEED_EXTRACTOR(c, what)                                                                                  \
    do {                                                                                                         \
        if ((c)->matcher_only) return fail((c), PPG_EINVAL, "%s: this context was made by ppg_create_matcher and has no extractor", what); \
    } while (0)

// Times what is enqueued on the context's stream between construction and destruction.
struct ProfScope {
    ppg_ctx* c;
    int cls;
    hipEvent_t e0 = nullptr;
    hipStream_t st;
    ProfScope(ppg_ctx* ctx, int k, hipStream_t s) : c(ctx), cls(k), st(s) {
        if ((c->prof == 1 || (c->prof == 2 && cls == PPG_PROF_CONV1A)) && (e0 = prof_event(c))) (void)hipEventRecord(e0, st);
    }
    ~ProfScope() {
        if (!e0) return;
        hipEvent_t e1 = prof_event(c);
        if (!e1) { c->ev_free.push_back(e0); return; }
        (void)hipEventRecord(e1, st);
        c->ev_pairs.push_back({cls, {e0, e1}});
    }
};

template <class T>
int dalloc(ppg_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    HIPCHK(c, hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    c->dev_allocs.push_back(q);
    *p = (T*)q;
    return PPG_OK;
}
template <class T>
int halloc(ppg_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    HIPCHK(c, hipHostMalloc(&q, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault));
    c->host_allocs.push_back(q);
    *p = (T*)q;
    return PPG_OK;
}
template <class T>
int grow(ppg_ctx* c, DBuf<T>& b, size_t n) {
    if (b.n >= n && b.p) return PPG_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.n = 0;
    void* q = nullptr;
    const size_t cap = std::max<size_t>(n + n / 2, 256);
    HIPCHK(c, hipMalloc(&q, cap * sizeof(T)));
    b.p = (T*)q;
    b.n = cap;
    return PPG_OK;
}
template <class T>
int upload(ppg_ctx* c, T** p, const std::vector<T>& v) {
    int rc = dalloc(c, p, v.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PPG_OK;
}

// float -> bf16 bits, round to nearest even (finite inputs), and back
