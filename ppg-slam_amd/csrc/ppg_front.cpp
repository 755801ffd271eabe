// Host side of libppgfront: context, weight packing, the extract pipeline, the order-dependent
// tail of detectLines, and the matcher entry points.  C ABI in include/ppgfront.h.
//
// Built with -ffp-contract=off so that the host-side float formulas (line geometry) round exactly
// as the reference's -O3 build without FMA code generation does.
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

namespace {

thread_local std::string g_err;  // reason of a failed ppg_create (no context to hold it)

// Work memory of ppg_match_nn_noalloc: the distance matrix of the largest accepted problem followed by the
// eight-nearest lists (distances, then indices).  A file-scope device array, so that entry point never allocates;
// every GPU the library runs on holds its own copy until the process ends.  Host code must not pass the array's
// name to a kernel (on the host it is a stand-in, not a device address): the per-device address is looked up once
// in scratch_setup() and kept in g_scratch_ptr.
constexpr int kNoAllocMaxRows = PPG_NOALLOC_MAX_ROWS;
constexpr size_t kScratchDistFloats = (size_t)kNoAllocMaxRows * kNoAllocMaxRows;
constexpr size_t kScratchListEntries = (size_t)kNoAllocMaxRows * 8;
constexpr size_t kScratchBytes = (kScratchDistFloats + 2 * kScratchListEntries) * 4;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};
// Order of the calls that share a GPU's copy of `scratch`: host threads take the device's mutex while they enqueue (a
// blocking call keeps it until its results are in place); a call on another stream than its predecessor's first makes
// its stream wait for the event recorded behind that one.
struct ScratchGuard {
    std::mutex mu;
    hipEvent_t done = nullptr;     // recorded behind the last call's kernels
    hipStream_t last = nullptr;    // the stream they went to
    bool used = false;
};
ScratchGuard g_scratch_guard[kMaxDevices];

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

}  // namespace

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
    int conv_ws = -1;
    long conv_ws_min_units = 1024;   // (tile, cout block) units a launch needs before it runs wave-specialised (PPG_CONV_WS_MIN)
    bool no_quarter = false;    // PPG_NO_QUARTER_TILES=1: big convolution tiles whatever the batch size (read at ppg_create)
    bool match_matrix = false;  // PPG_MATCH_MATRIX=1: resident matches through the full distance matrix (the path the fused one is checked against)
    int* mCount = nullptr;
    int* h_mcount = nullptr;
    int* h_mcounts = nullptr;    // alias of the current slot's
    int32_t* h_match = nullptr;  // alias of the current slot's
};

namespace {

int fail(ppg_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail((c), PPG_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Looks up, once per device and with that device current, where this GPU keeps its copy of `scratch`.
int scratch_setup(ppg_ctx* c, int dev) {
    if (dev < 0 || dev >= kMaxDevices) return fail(c, PPG_EINVAL, "device_id %d: at most %d devices are supported", dev, kMaxDevices);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIPCHK(c, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        HIPCHK(c, hipEventCreateWithFlags(&g_scratch_guard[dev].done, hipEventDisableTiming));
        g_scratch_ptr[dev] = p;
    }
    return PPG_OK;
}

// the entry points of the extractor (and those that read resident frames) do not exist on a matcher-only context
#define NEED_EXTRACTOR(c, what)                                                                                  \
    do {                                                                                                         \
        if ((c)->matcher_only) return fail((c), PPG_EINVAL, "%s: this context was made by ppg_create_matcher and has no extractor", what); \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

hipEvent_t prof_event(ppg_ctx* c) {
    if (c->ev_free.empty()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_all.push_back(e);
        return e;
    }
    hipEvent_t e = c->ev_free.back();
    c->ev_free.pop_back();
    return e;
}

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

// Accumulates every timed span whose closing event has completed; later spans stay queued.
void prof_resolve(ppg_ctx* c) {
    size_t keep = 0;
    for (size_t k = 0; k < c->ev_pairs.size(); k++) {
        auto p = c->ev_pairs[k];
        float ms = 0.f;
        if (hipEventQuery(p.second.second) == hipSuccess &&
            hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            c->prof_ms[p.first] += ms;
            c->prof_launches[p.first] += 1;
            c->ev_free.push_back(p.second.first);
            c->ev_free.push_back(p.second.second);
        } else {
            c->ev_pairs[keep++] = p;
        }
    }
    c->ev_pairs.resize(keep);
}

void select_lane(ppg_ctx* c, int li) {
    Lane& l = c->lane[li];
    c->lane_cur = li;
    c->stream = l.stream;
    c->a1 = l.a1; c->p1 = l.p1; c->a2 = l.a2; c->p2 = l.p2; c->a3 = l.a3; c->p3 = l.p3; c->a4 = l.a4;
    c->feat = l.feat; c->pa_o = l.pa_o; c->da_o = l.da_o; c->hs0 = l.hs0; c->hs1 = l.hs1;
}

int inflight_count(const ppg_ctx* c) {
    int k = 0;
    for (const Slot& sl : c->slot) k += sl.inflight;
    return k;
}

bool any_inflight(const ppg_ctx* c) {
    for (const Slot& sl : c->slot)
        if (sl.inflight) return true;
    return false;
}

void select_slot(ppg_ctx* c, int sidx) {
    Slot& a = c->slot[sidx];
    Slot& o = c->slot[prev_slot(sidx)];
    c->cur = sidx;
    c->counters = a.counters; c->kp_xy = a.kp_xy; c->kp_score = a.kp_score; c->kp_un = a.kp_un; c->kp_out = a.kp_out;
    c->desc = a.desc; c->cand_ij = a.cand_ij; c->cand_sc = a.cand_sc; c->d_match = a.d_match; c->d_mcounts = a.d_mcounts;
    c->h_counters = a.h_counters; c->h_kp_xy = a.h_kp_xy; c->h_kp_score = a.h_kp_score; c->h_kp_un = a.h_kp_un;
    c->h_kp_out = a.h_kp_out; c->h_desc = a.h_desc;
    c->h_match = a.h_match; c->h_mcounts = a.h_mcounts; c->frames = &a.frames; c->h_img = a.h_img;
    c->desc_prev = o.desc; c->counters_prev = o.counters;
    c->junc = a.junc; c->hs2 = a.hs2; c->descmap = a.descmap;
}

bool read_blob(const char* path, std::map<std::string, Tensor>& out, std::string& why) {
    FILE* f = fopen(path, "rb");
    if (!f) { why = std::string("cannot open ") + path; return false; }
    char magic[8];
    uint32_t n = 0, rsv = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "PPGW0001", 8) == 0 && fread(&n, 4, 1, f) == 1 &&
              fread(&rsv, 4, 1, f) == 1 && n < 4096;
    struct Entry { char name[56]; uint32_t ndim; uint32_t dims[4]; uint32_t pad; uint64_t off, nbytes; };
    static_assert(sizeof(Entry) == 96, "blob entry is 96 bytes");
    std::vector<Entry> es(ok ? n : 0);
    ok = ok && fread(es.data(), sizeof(Entry), n, f) == n;
    for (size_t i = 0; ok && i < es.size(); i++) {
        const Entry& e = es[i];
        if (e.ndim > 4 || e.nbytes % 4 || e.nbytes > (1u << 28)) { ok = false; break; }
        {   // the entry's shape must account for exactly its bytes
            uint64_t prod = 1;
            for (uint32_t k = 0; k < e.ndim; k++) prod *= e.dims[k];
            if (prod * 4 != e.nbytes) { ok = false; break; }
        }
        Tensor t;
        t.dims.assign(e.dims, e.dims + e.ndim);
        t.v.resize(e.nbytes / 4);
        ok = fseek(f, (long)e.off, SEEK_SET) == 0 && fread(t.v.data(), 1, e.nbytes, f) == e.nbytes;
        char nm[57];
        memcpy(nm, e.name, 56);
        nm[56] = 0;
        out[nm] = std::move(t);
    }
    fclose(f);
    if (!ok) why = std::string("malformed weight blob ") + path;
    return ok;
}

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
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// float -> fp16 bits, round to nearest even, subnormals kept (|x| < 65520), and back
inline uint16_t f16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const float ax = std::fabs(x);
    if (ax < 0x1p-14f) return sign | (uint16_t)std::nearbyint(ax * 0x1p24f);  // subnormal grid (0x0400 = 2^-14 when it rounds up)
    u &= 0x7FFFFFFFu;
    u = (u + 0xFFFu + ((u >> 13) & 1u)) >> 13;       // 10-bit mantissa, exponent still biased by 127
    return sign | (uint16_t)(u - ((127 - 15) << 10));
}
inline float f16_float(uint16_t h) {
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    float v = e == 0 ? (float)m * 0x1p-24f : std::ldexp((float)(m | 1024u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}

// Pack OIHW weights into the LDS image order of conv_mfma_kernel:
//   [cout block][cin chunk][group of 8 cin][tap][half h][cout in block][4 cin]
// with cin = chunk*KC + g*8 + h*4 + s.  `ps2` permutes couts so that a pixel-shuffle block's packed
// cout q*Cps + c (q = 2*dy + dx) is the original channel c*4 + q
// (pixel_shuffle: out[c][2y+dy][2x+dx] = in[c*4 + dy*2 + dx][y][x]).
int make_conv(ppg_ctx* c, const std::map<std::string, Tensor>& t, const std::string& name, int nt, bool ps2,
              const std::string& bn_name, DevConv& dc) {
    auto wi = t.find(name + ".weight"), bi = t.find(name + ".bias");
    if (wi == t.end() || bi == t.end() || wi->second.dims.size() != 4)
        return fail(c, PPG_EIO, "weight blob lacks %s", name.c_str());
    const Tensor& w = wi->second;
    const int cout = w.dims[0], cin = w.dims[1], ks = w.dims[2];
    if ((ks != 1 && ks != 3) || (int)w.dims[3] != ks || (int)bi->second.v.size() != cout)
        return fail(c, PPG_EIO, "%s: unsupported shape", name.c_str());
    const int KC = ppg::conv_chunk(ks), NB = 32 * nt, taps = ks * ks, G = KC / 8;
    if (cin % KC) return fail(c, PPG_EIO, "%s: unsupported shape", name.c_str());
    const int nblk = (cout + NB - 1) / NB, nchunks = cin / KC;
    dc.ks = ks; dc.nt = nt; dc.cin = cin; dc.cout = cout; dc.nblk = nblk;
    auto orig = [&](int cp) -> int {
        if (cp >= cout) return -1;
        if (!ps2) return cp;
        const int cps = cout / 4, q = cp / cps, cc = cp % cps;
        return cc * 4 + q;
    };
    std::vector<float> pk((size_t)nblk * nchunks * taps * KC * NB, 0.f);
    for (int nb = 0; nb < nblk; nb++)
        for (int ch = 0; ch < nchunks; ch++)
            for (int g = 0; g < G; g++)
                for (int tap = 0; tap < taps; tap++)
                    for (int h = 0; h < 2; h++)
                        for (int nn = 0; nn < NB; nn++) {
                            const int o = orig(nb * NB + nn);
                            if (o < 0) continue;
                            for (int s = 0; s < 4; s++) {
                                const int ci = ch * KC + g * 8 + h * 4 + s;
                                const size_t dst = ((((((size_t)nb * nchunks + ch) * G + g) * taps + tap) * 2 + h) * NB + nn) * 4 + s;
                                pk[dst] = w.v[((size_t)o * cin + ci) * taps + tap];
                            }
                        }
    // Reduced-precision modes: the 3x3 layers whose input has a multiple of 16 channels and the 1x1 layers whose input
    // has a multiple of 32 run on the bf16 / fp16 MFMA.  Image order of
    // conv_bf16_kernel: [cout block][chunk of 16 cin][tap][half h][hi | lo][cout in block][8 cin], cin = chunk*16 +
    // h*8 + j; hi = bf16(w) rounded to nearest even, lo = bf16(w - hi) (precision 3: fp16 halves of 2^10 * w).  Same byte
    // count as the fp32 image.
    dc.lowp = c->cfg.precision != 0 && ((ks == 3 && cin % 16 == 0) || (ks == 1 && cin % 32 == 0));
    if (dc.lowp) {
        std::vector<uint16_t> pb(pk.size() * 2, 0);
        const int nch16 = cin / 16;
        for (int nb = 0; nb < nblk; nb++)
            for (int ch = 0; ch < nch16; ch++)
                for (int tap = 0; tap < taps; tap++)
                    for (int h = 0; h < 2; h++)
                        for (int nn = 0; nn < NB; nn++) {
                            const int o = orig(nb * NB + nn);
                            if (o < 0) continue;
                            for (int j = 0; j < 8; j++) {
                                const int ci = ch * 16 + h * 8 + j;
                                float wv = w.v[((size_t)o * cin + ci) * taps + tap];
                                uint16_t hi, lo;
                                if (c->cfg.precision == 3) {  // fp16 halves of the weight scaled by 2^10
                                    if (!(std::fabs(wv) < 63.9f))
                                        return fail(c, PPG_ERANGE, "%s: a weight of magnitude %g leaves the range of the fp16 split (< 63.9): use precision 0 or 1", name.c_str(), (double)wv);
                                    wv = wv * 0x1p10f;
                                    hi = f16_rne(wv);
                                    lo = f16_rne(wv - f16_float(hi));
                                } else {
                                    hi = bf16_rne(wv);
                                    lo = bf16_rne(wv - bf16_float(hi));
                                }
                                const size_t base = (((((size_t)nb * nch16 + ch) * taps + tap) * 2 + h) * 2) * NB;
                                pb[((base + nn) * 8) + j] = hi;
                                pb[((base + NB + nn) * 8) + j] = lo;
                            }
                        }
        memcpy(pk.data(), pb.data(), pk.size() * sizeof(float));
    }
    std::vector<float> bias((size_t)nblk * NB, 0.f), alpha, beta;
    for (int cp = 0; cp < nblk * NB; cp++) {
        const int o = orig(cp);
        if (o >= 0) bias[cp] = bi->second.v[o];
    }
    int rc = upload(c, &dc.w, pk);
    if (!rc) rc = upload(c, &dc.bias, bias);
    if (rc) return rc;
    if (!bn_name.empty()) {
        auto g = t.find(bn_name + ".weight"), b = t.find(bn_name + ".bias"), m = t.find(bn_name + ".running_mean"),
             v = t.find(bn_name + ".running_var");
        if (g == t.end() || b == t.end() || m == t.end() || v == t.end())
            return fail(c, PPG_EIO, "weight blob lacks %s", bn_name.c_str());
        if ((int)g->second.v.size() != cout || (int)b->second.v.size() != cout || (int)m->second.v.size() != cout || (int)v->second.v.size() != cout)
            return fail(c, PPG_EIO, "%s: batch-norm vectors do not match %d output channels", bn_name.c_str(), cout);
        alpha.assign((size_t)nblk * NB, 1.f);
        beta.assign((size_t)nblk * NB, 0.f);
        for (int cp = 0; cp < nblk * NB; cp++) {
            const int o = orig(cp);
            if (o < 0) continue;
            // eval-mode batch norm as ATen evaluates it: alpha = weight/sqrt(var+eps), beta = bias - mean*alpha
            const float invstd = 1.0f / std::sqrt(v->second.v[o] + 1e-5f);
            alpha[cp] = invstd * g->second.v[o];
            beta[cp] = b->second.v[o] - m->second.v[o] * alpha[cp];
        }
        rc = upload(c, &dc.alpha, alpha);
        if (!rc) rc = upload(c, &dc.beta, beta);
    }
    return rc;
}

int load_weights(ppg_ctx* c) {
    std::map<std::string, Tensor> t;
    std::string why;
    if (!read_blob(c->weights_path.c_str(), t, why)) return fail(c, PPG_EIO, "%s", why.c_str());
    int rc = 0;
    auto mk = [&](const char* name, int nt, bool ps2, const char* bn, DevConv& dc) {
        if (!rc) rc = make_conv(c, t, name, nt, ps2, bn, dc);
    };
    mk("backbone.conv1b", 2, false, "", c->c1b);
    mk("backbone.conv2a", 2, false, "", c->c2a);
    mk("backbone.conv2b", 2, false, "", c->c2b);
    mk("backbone.conv3a", 2, false, "", c->c3a);
    mk("backbone.conv3b", 2, false, "", c->c3b);
    mk("backbone.conv4a", 2, false, "", c->c4a);
    mk("backbone.conv4b", 2, false, "", c->c4b);
    mk("point.convPa", 2, false, "", c->pa);
    mk("point.convPb", 1, false, "", c->pb);
    mk("desc.convDa", 2, false, "", c->da);
    mk("desc.convDb", 2, false, "", c->db);
    mk("edge.conv_block_lst.0.0", 2, true, "edge.conv_block_lst.0.1", c->h0);
    mk("edge.conv_block_lst.1.0", 2, true, "edge.conv_block_lst.1.1", c->h1);
    mk("edge.conv_block_lst.2.0", 1, true, "edge.conv_block_lst.2.1", c->h2);
    if (rc) return rc;
    {
        const DevConv* hs[3] = {&c->pa, &c->da, &c->h0};
        size_t tot = 0, off = 0;
        for (const DevConv* d : hs) tot += (size_t)d->nblk * 32 * d->nt * d->cin * 9;
        rc = dalloc(c, &c->heads_w, tot);
        if (rc) return rc;
        for (const DevConv* d : hs) {
            const size_t nfl = (size_t)d->nblk * 32 * d->nt * d->cin * 9;
            HIPCHK(c, hipMemcpy(c->heads_w + off, d->w, nfl * sizeof(float), hipMemcpyDeviceToDevice));
            off += nfl;
        }
    }
    auto w1 = t.find("backbone.conv1a.weight"), b1 = t.find("backbone.conv1a.bias");
    auto w3 = t.find("edge.conv_block_lst.3.weight"), b3 = t.find("edge.conv_block_lst.3.bias");
    if (w1 == t.end() || b1 == t.end() || w3 == t.end() || b3 == t.end() || w1->second.v.size() != 576 ||
        w3->second.v.size() != 8 || b1->second.v.size() != 64 || b3->second.v.size() != 2)
        return fail(c, PPG_EIO, "weight blob lacks conv1a / final heat conv");
    std::vector<float> w1p(9 * 64);
    for (int o = 0; o < 64; o++)
        for (int tap = 0; tap < 9; tap++) w1p[tap * 64 + o] = w1->second.v[o * 9 + tap];
    rc = upload(c, &c->c1a_w, w1p);
    if (!rc) rc = upload(c, &c->c1a_b, b1->second.v);
    if (!rc) rc = upload(c, &c->h3_w, w3->second.v);  // [2][4]
    if (!rc) rc = upload(c, &c->h3_b, b3->second.v);
    return rc;
}

// tile shape with the least padded area; the 16 x 16 tile exists for 3x3 layers with 64-cout blocks and no pooling
int pick_tile(const DevConv& dc, int H, int W, int mode) {
    if (dc.ks != 3 || dc.nt != 2 || mode == ppg::CONV_POOL2) return 32;
    const long a32 = (long)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32), a16 = (long)((H + 15) / 16 * 16) * ((W + 15) / 16 * 16);
    return a16 < a32 ? 16 : 32;
}

// cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap (K, D, R = I, P = K, CV_32F maps) of
// feature/src/PPGExtractor.cpp:65-71, after OpenCV's published algorithm: per destination pixel the normalised
// coordinate (accumulated along the row in double, as OpenCV does), the forward distortion model, back to pixels.
void undistort_rectify_maps(const ppg_config& c, std::vector<float>& mx, std::vector<float>& my) {
    const int W = c.width, H = c.height;
    mx.resize((size_t)W * H); my.resize((size_t)W * H);
    const double fx = c.K[0], fy = c.K[4], cx = c.K[2], cy = c.K[5];
    const double ir0 = 1.0 / fx, ir2 = -cx / fx, ir4 = 1.0 / fy, ir5 = -cy / fy;
    for (int i = 0; i < H; i++) {
        double _x = ir2, _y = i * ir4 + ir5;
        for (int j = 0; j < W; j++, _x += ir0) {
            const double x = _x, y = _y;
            double u, v;
            if (!c.fisheye) {
                const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3];
                const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = 1 + ((0.0 * r2 + k2) * r2 + k1) * r2;
                const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
                const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
                u = fx * xd + cx; v = fy * yd + cy;
            } else {
                const double r = std::sqrt(x * x + y * y), theta = std::atan(r);
                const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
                const double theta_d = theta * (1 + c.D[0] * t2 + c.D[1] * t4 + c.D[2] * t6 + c.D[3] * t8);
                const double scale = r == 0 ? 1.0 : theta_d / r;
                u = fx * x * scale + cx; v = fy * y * scale + cy;
            }
            mx[(size_t)i * W + j] = (float)u;
            my[(size_t)i * W + j] = (float)v;
        }
    }
}

// Tile grid of a launch.  A launch of the big tile (256 pixels x 64 couts per work-group) that would put fewer than
// kQuarterBelow work-groups on the chip (256 CUs take three each) goes out in quarter-size work-groups instead: with
// one or two frames per call the deep layers are 20 to 300 big tiles, and the launch lasts as long as one of them.
constexpr int kQuarterBelow = 600;
void conv_tiling(const ppg_ctx* c, ppg::ConvArgs& a, const DevConv& dc, int mode, bool fused) {
    const int H = a.H, W = a.W;
    a.mw = fused ? 32 : pick_tile(dc, H, W, mode);
    a.tiles_x = (W + a.mw - 1) / a.mw; a.tiles_y = (H + 256 / a.mw - 1) / (256 / a.mw);
    a.quarter = 0;
    // (only for launches of many tiles per work-group: shorter ones gain nothing from it)
    a.persist = (fused ? c->persist_fused : c->persist_other);
    if (!c->persist_forced && (long)a.tiles_x * a.tiles_y * a.B < 20L * a.persist) a.persist = 0;
    a.tile_counter = c->lane[c->lane_cur].tile_ctr;
    a.xcd_group = c->xcd_group; a.xcd_band = c->xcd_band;
    // Reduced-precision 3x3 layers of many tiles run wave-specialised: one resident work-group per CU whose producer
    // waves stage the next chunk while its consumer waves multiply (conv_ws.hip).  Launches too small to give every CU a
    // few tiles keep the one-tile-per-work-group kernel.
    a.ws = 0;
    if ((c->conv_ws < 0 ? a.prec == 1 : c->conv_ws != 0) && a.prec != 0 && dc.ks == 3 && dc.nt == 2 && a.Cin >= 64 && a.Cin % 16 == 0 && a.xcd_group > 0 &&
        (long)a.tiles_x * a.tiles_y * a.B * a.nblk >= c->conv_ws_min_units && a.mw == 32)   // (16 x 16 tiles: measured slower)
        a.ws = 1;
    if (c->no_quarter || fused || a.prec != 0 || dc.ks != 3 || dc.nt != 2) return;
    if ((long)a.tiles_x * a.tiles_y * a.B * a.nblk >= kQuarterBelow) return;
    a.quarter = 1;
    if (4 * a.nblk > kTileCounters) a.persist = 0;  // (two counters per launched cout block, twice as many blocks)
    // pooling pairs the two rows of an 8 x 16 tile's pixel tiles; otherwise the shape with the least padded area
    const long a32 = (long)((H + 3) / 4 * 4) * ((W + 31) / 32 * 32), a16 = (long)((H + 7) / 8 * 8) * ((W + 15) / 16 * 16);
    a.mw = (mode == ppg::CONV_POOL2 || a16 < a32) ? 16 : 32;
    a.tiles_x = (W + a.mw - 1) / a.mw; a.tiles_y = (H + 128 / a.mw - 1) / (128 / a.mw);
}

ppg::ConvGroup conv_group(const DevConv& dc, float* out, int mode, int relu, int nblk0) {
    ppg::ConvGroup g;
    g.out = out; g.bias = dc.bias; g.alpha = dc.alpha; g.beta = dc.beta;
    g.nblk0 = nblk0; g.Cout = dc.cout; g.mode = mode; g.relu = relu;
    return g;
}

int run_conv(ppg_ctx* c, const DevConv& dc, const float* in, float* out, int n, int H, int W, int mode, int relu,
             const uint8_t* fused_img = nullptr) {
    ppg::ConvArgs a;
    if (fused_img) { a.img = fused_img; a.w1a = c->c1a_w; a.b1a = c->c1a_b; }
    a.in = in; a.wpk = dc.w; a.prec = dc.lowp ? c->cfg.precision : 0; a.sat = c->slot[c->cur].sat;
    a.B = n; a.H = H; a.W = W; a.Cin = dc.cin;
    a.nblk = dc.nblk;
    conv_tiling(c, a, dc, mode, fused_img != nullptr);
    a.ngroups = 1;
    a.g[0] = conv_group(dc, out, mode, relu, 0);
    const int cls = fused_img ? PPG_PROF_CONV1A : (dc.ks == 3 && dc.nt == 2 ? PPG_PROF_CONV3X3 : PPG_PROF_CONV_OTHER);
    if (c->prof) c->prof_flops[cls] += 2.0 * n * H * W * (double)dc.cin * dc.cout * dc.ks * dc.ks + (fused_img ? 2.0 * n * H * W * 9.0 * 64 : 0.0);
    ProfScope ps(c, cls, c->stream);
    HIPCHK(c, ppg::launch_conv(a, dc.ks, dc.nt, c->stream));
    return PPG_OK;
}

// convPa, convDa and the first EdgeHeatmap conv all read featureMap (128 -> 256, 3x3): one launch with
// 12 cout blocks instead of three with 4, so the small 1/8-resolution grid fills the chip better.
int run_heads(ppg_ctx* c, int n, int H, int W) {
    ppg::ConvArgs a;
    a.in = c->feat; a.wpk = c->heads_w; a.prec = c->pa.lowp ? c->cfg.precision : 0; a.sat = c->slot[c->cur].sat;
    a.B = n; a.H = H; a.W = W; a.Cin = c->pa.cin;
    a.nblk = c->pa.nblk + c->da.nblk + c->h0.nblk;
    conv_tiling(c, a, c->pa, ppg::CONV_PLAIN, false);
    a.ngroups = 3;
    a.g[0] = conv_group(c->pa, c->pa_o, ppg::CONV_PLAIN, 1, 0);
    a.g[1] = conv_group(c->da, c->da_o, ppg::CONV_PLAIN, 1, c->pa.nblk);
    a.g[2] = conv_group(c->h0, c->hs0, ppg::CONV_PS2, 1, c->pa.nblk + c->da.nblk);
    if (c->prof) c->prof_flops[PPG_PROF_CONV3X3] += 2.0 * n * H * W * 9.0 * c->pa.cin * (c->pa.cout + c->da.cout + c->h0.cout);
    ProfScope ps(c, PPG_PROF_CONV3X3, c->stream);
    HIPCHK(c, ppg::launch_conv(a, 3, 2, c->stream));
    return PPG_OK;
}

ppg::NmsArgs nms_args(ppg_ctx* c, int n) {
    ppg::NmsArgs a;
    a.junc = c->junc; a.score = c->score; a.state = c->state; a.resid = c->resid; a.kept = c->kept;
    a.kept_key = c->kept_key; a.counters = c->counters; a.resid_cap = c->resid_cap; a.kept_cap = c->kept_cap;
    a.B = n; a.H = c->H; a.W = c->W; a.thresh = c->cfg.junction_thresh; a.max_num = c->maxn;
    a.fx = c->cfg.K[0]; a.fy = c->cfg.K[4]; a.cx = c->cfg.K[2]; a.cy = c->cfg.K[5];
    a.d0 = c->cfg.D[0]; a.d1 = c->cfg.D[1]; a.d2 = c->cfg.D[2]; a.d3 = c->cfg.D[3];
    a.fisheye = c->cfg.fisheye;
    a.kp_xy = c->kp_xy; a.kp_score = c->kp_score; a.kp_un = c->kp_un; a.kp_out = c->kp_out;
    return a;
}

ppg::HeatArgs heat_args(ppg_ctx* c, int n) {
    ppg::HeatArgs a;
    a.ps = c->hs2; a.w = c->h3_w; a.bias = c->h3_b; a.raw = c->heat_raw; a.prob = c->prob; a.refined = c->refined;
    a.B = n; a.H = c->H; a.W = c->W; a.refine_sz = c->cfg.heatmap_refine_sz;
    a.valid_thresh = c->cfg.line_valid_thresh; a.valid_ratio = c->cfg.line_valid_ratio;
    return a;
}

size_t tensor_elems(const ppg_ctx* c, int which) {
    const size_t hw = (size_t)c->h8 * c->w8, HW = (size_t)c->H * c->W;
    switch (which) {
        case PPG_T_FEAT: return hw * 128;
        case PPG_T_JUNC: return hw * 65;
        case PPG_T_HEAT_RAW: return HW * 2;
        case PPG_T_DESCMAP: return hw * 256;
        case PPG_T_SCORE: case PPG_T_HEAT: case PPG_T_HEAT_REFINED: return HW;
        case PPG_T_HEAT_REMAPPED: return c->remapped ? HW : 0;
        default: return 0;
    }
}
float* tensor_ptr(ppg_ctx* c, int which) {
    switch (which) {
        case PPG_T_FEAT: return c->feat;
        case PPG_T_JUNC: return c->junc;
        case PPG_T_HEAT_RAW: return c->heat_raw;
        case PPG_T_DESCMAP: return c->descmap;
        case PPG_T_SCORE: return c->score;
        case PPG_T_HEAT: return c->prob;
        case PPG_T_HEAT_REFINED: return c->refined;
        case PPG_T_HEAT_REMAPPED: return c->remapped;
        default: return nullptr;
    }
}

}  // namespace

int ppg_set_error(ppg_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_err = buf;
    return code;
}
int ppg_ctx_device(const ppg_ctx* c) { return c->cfg.device_id; }
int ppg_ctx_max_num(const ppg_ctx* c) { return c->maxn; }

extern "C" {

void ppg_default_config(ppg_config* c, int width, int height) {
    memset(c, 0, sizeof(*c));
    c->width = width; c->height = height;
    c->K[0] = 458.654f; c->K[4] = 457.296f; c->K[2] = width * 0.5f; c->K[5] = height * 0.5f; c->K[8] = 1.f;
    c->max_batch = 1;
    c->desc_dim = 256;                    // feature/src/PPGExtractor.cpp:44-53
    c->junction_thresh = 1.0f / 128.0f;
    c->junction_nms_radius = 4;
    c->junction_max_num = 500;
    c->line_valid_thresh = 1.0e-2f;
    c->line_valid_ratio = 0.3f;
    c->line_dist_thresh = 2.0f;
    c->heatmap_refine_sz = 16;
    c->line_heatmap_thresh = 0.2f;
    c->line_inlier_rate = 0.8f;
}

const char* ppg_last_error(const ppg_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

void ppg_destroy(ppg_ctx* c) {
    if (!c) return;
    delete c->pool;
    c->pool = nullptr;
    (void)hipSetDevice(c->cfg.device_id);
    for (Lane& l : c->lane)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (c->post) (void)hipStreamSynchronize(c->post);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->match_stream) (void)hipStreamSynchronize(c->match_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (void* p : c->dev_allocs) (void)hipFree(p);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    if (c->xPin) (void)hipHostFree(c->xPin);
    void* ms[] = {c->mA.p, c->mB.p, c->mD.p, c->mBest.p, c->mSecond.p, c->mTopD.p, c->mOff.p,
                  c->mIdx.p, c->mMatch.p, c->mTopI.p, c->mSkip.p, c->mClaim.p, c->mNorm.p, c->mListCnt.p, c->mListCol.p, c->mHalf.p, c->mBt.p, c->mExt.p, c->mNearLen.p, c->mListRad.p, c->bWord.p, c->bNode.p, c->bWeight.p, c->mSkipOut.p, c->mRowClaims.p, c->xArena.p,
                  c->voc_desc, c->voc_child_off, c->voc_child_idx, c->voc_word, c->voc_weight};
    for (void* p : ms)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev_all) (void)hipEventDestroy(e);
    for (Slot& sl : c->slot) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.cnn_done) (void)hipEventDestroy(sl.cnn_done);
        if (sl.junc_done) (void)hipEventDestroy(sl.junc_done);
        if (sl.heat_done) (void)hipEventDestroy(sl.heat_done);
        if (sl.match_done) (void)hipEventDestroy(sl.match_done);
    }
    if (c->ev_desc_ready) (void)hipEventDestroy(c->ev_desc_ready);
    for (Lane& l : c->lane)
        if (l.stream) (void)hipStreamDestroy(l.stream);
    if (c->post) (void)hipStreamDestroy(c->post);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->match_stream) (void)hipStreamDestroy(c->match_stream);
    delete c;
}

static int create_impl(ppg_ctx* c) {
    const ppg_config& f = c->cfg;
    if (f.width <= 0 || f.height <= 0 || f.width % 8 || f.height % 8)
        return fail(c, PPG_EINVAL, "image size %dx%d must be positive multiples of 8", f.width, f.height);
    if (f.width < 16 || f.height < 16) return fail(c, PPG_EINVAL, "image too small");
    if (f.junction_nms_radius != 4) return fail(c, PPG_EINVAL, "only JUNCTION_NMS_RADIUS = 4 is built");
    if (f.desc_dim != 256) return fail(c, PPG_EINVAL, "only DESC_DIM_SIZE = 256 is built");
    if (f.junction_max_num < 1 || f.junction_max_num > 4096) return fail(c, PPG_EINVAL, "junction_max_num must be 1..4096");
    if (c->matcher_only) {
        c->cfg.weights_path = nullptr;
        c->cfg.max_batch = 0;
    } else {
        if (f.max_batch < 1 || f.max_batch > 64) return fail(c, PPG_EINVAL, "max_batch must be 1..64");
        if (f.heatmap_refine_sz < 1 || f.heatmap_refine_sz > 16) return fail(c, PPG_EINVAL, "heatmap_refine_sz must be 1..16");
        if (f.precision < 0 || f.precision > 3) return fail(c, PPG_EINVAL, "precision %d is not built", f.precision);
        if (!f.weights_path) return fail(c, PPG_EINVAL, "weights_path is NULL");
        c->weights_path = f.weights_path;
        c->cfg.weights_path = c->weights_path.c_str();
    }
    c->W = f.width; c->H = f.height; c->w8 = c->W / 8; c->h8 = c->H / 8; c->B = c->cfg.max_batch; c->maxn = (int)f.junction_max_num;
    int ndev = 0;
    HIPCHK(c, hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(c, PPG_EHIP, "no HIP device: libppgfront has no CPU path");
    if (f.device_id < 0 || f.device_id >= ndev) return fail(c, PPG_EINVAL, "device_id %d out of range (%d devices)", f.device_id, ndev);
    HIPCHK(c, hipSetDevice(f.device_id));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, f.device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, PPG_EHIP, "device %d is %s; this library is built for gfx950 only", f.device_id, prop.gcnArchName);
    // the fused first convolution keeps one work-group per resident slot (three per CU) walking the tiles
    c->persist_fused = 3 * prop.multiProcessorCount;
    {
        const int rc0 = scratch_setup(c, f.device_id);
        if (rc0) return rc0;
    }
    // the MFMA-bound CNN gets dispatch priority; the latency-bound post-processing fills what is left
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->post, hipStreamNonBlocking, prio_lo));
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // same priority as the CNN: at a lower one the short upload kernel would starve behind every pending convolution work-group
    HIPCHK(c, hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->match_stream, hipStreamNonBlocking, prio_lo));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_desc_ready, hipEventDisableTiming));
    {
        const char* u = getenv("PPG_UNFUSED_CONV1A");
        c->unfused_conv1a = u && atoi(u) != 0;
        const char* nl = getenv("PPG_CNN_LANES");
        if (nl && atoi(nl) == 2) c->nlanes = 2;
        c->dump_tail = getenv("PPG_DUMP_TAIL");
        const char* mm = getenv("PPG_MATCH_MATRIX");
        c->match_matrix = mm && atoi(mm) != 0;
        if (const char* pe = getenv("PPG_CONV_PERSIST")) c->persist_forced = sscanf(pe, "%d,%d", &c->persist_fused, &c->persist_other) == 2;
        if (const char* xm = getenv("PPG_XCD_MAP")) c->xcd_group = std::max(0, atoi(xm));
        if (const char* xb = getenv("PPG_XCD_BAND")) c->xcd_band = atoi(xb) != 0;
        if (const char* w = getenv("PPG_CONV_WS")) c->conv_ws = atoi(w) != 0;
        if (const char* w = getenv("PPG_CONV_WS_MIN")) c->conv_ws_min_units = std::max(8L, atol(w));
        const char* nq = getenv("PPG_NO_QUARTER_TILES");
        c->no_quarter = nq && atoi(nq) != 0;
        const char* eh = getenv("PPG_EXTEND_HOST");
        c->extend_on_host = eh && atoi(eh) != 0;
        const char* lh = getenv("PPG_LINES_HOST");
        c->lines_on_host = (lh && atoi(lh) != 0) || c->dump_tail;
        const char* e = getenv("PPG_HOST_THREADS");
        // Default: the host's hardware threads divided by the ranks sharing it (torchrun exports LOCAL_WORLD_SIZE), at
        // most 8 — the pool only stages image batches into pinned memory and redoes the rare line-graph tails the
        // device hands over, so eight ranks must not start 8 x 16 threads.
        const char* lw = getenv("LOCAL_WORLD_SIZE");
        const int ranks = std::max(1, lw ? atoi(lw) : 1);
        int t = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / (unsigned)ranks));
        c->host_threads = std::max(1, std::min(t, 64));
    }
    int rc = PPG_OK;
#define DA(ptr, n) do { rc = dalloc(c, &c->ptr, (n)); if (rc) return rc; } while (0)
#define HA(ptr, n) do { rc = halloc(c, &c->ptr, (n)); if (rc) return rc; } while (0)
    if (c->matcher_only) {   // the streams above, the match counter, and work arrays that grow with the first calls
        DA(mCount, 16);
        HA(h_mcount, 4);
        return PPG_OK;
    }
    rc = load_weights(c);
    if (rc) return rc;
    const size_t B = c->B, HW = (size_t)c->H * c->W, hw = (size_t)c->h8 * c->w8;
    for (int li = 0; li < c->nlanes; li++) {
        Lane& l = c->lane[li];
#define DL(ptr, n) do { rc = dalloc(c, &l.ptr, (n)); if (rc) return rc; } while (0)
        if (c->unfused_conv1a) DL(a1, B * HW * 64);
        DL(p1, B * HW / 4 * 64);
        DL(a2, B * HW / 4 * 64);
        DL(p2, B * HW / 16 * 64);
        DL(a3, B * HW / 16 * 128);
        DL(p3, B * hw * 128);
        DL(a4, B * hw * 128);
        DL(feat, B * hw * 128);
        DL(pa_o, B * hw * 256);
        DL(da_o, B * hw * 256);
        DL(hs0, B * hw * 4 * 64);
        DL(hs1, B * hw * 16 * 16);
        DL(tile_ctr, kTileCounters);
        HIPCHK(c, hipMemset(l.tile_ctr, 0, kTileCounters * sizeof(int)));
#undef DL
    }
    select_lane(c, 0);
    DA(heat_raw, B * HW * 2);
    DA(prob, B * HW);
    DA(refined, B * HW);
    if (c->cfg.D[0] != 0.0f) {  // feature/src/PPGExtractor.cpp:261: the heat map is remapped only then
        std::vector<float> mx, my;
        undistort_rectify_maps(c->cfg, mx, my);
        rc = upload(c, &c->mapx, mx);
        if (!rc) rc = upload(c, &c->mapy, my);
        if (rc) return rc;
        DA(remapped, B * HW);
    }
    DA(score, B * HW);
    DA(state, B * HW);
    c->resid_cap = (int)HW;
    c->kept_cap = (int)(HW / 16 + 64);
    c->cand_cap = std::max(1, c->maxn * (c->maxn - 1) / 2);
    c->edge_cap = std::max(1, std::min(c->cand_cap, 4 * c->maxn));
    c->tail_cap = std::min(std::max(c->cand_cap, 2 * c->edge_cap + 2), 65536);
    c->pair_cap = std::min(8 * c->tail_cap, 262144);
    DA(resid, B * c->resid_cap);
    DA(kept, B * c->kept_cap);
    DA(kept_key, B * c->kept_cap);
    DA(row_cnt, B * (c->maxn + 1));
    c->mask_words = (c->maxn + 255) / 256 * 4;
    DA(pair_mask, B * (size_t)c->maxn * c->mask_words);
    DA(lt_geo, B * c->tail_cap * 3);
    DA(lt_col_off, B * (c->maxn + 1));
    HIPCHK(c, hipMemset(c->lt_col_off, 0, sizeof(int) * B * (c->maxn + 1)));   // (line_resolve_kernel leaves it zeroed after every use)
    DA(lt_col_cur, B * c->maxn);
    DA(lt_col_ent, B * c->tail_cap * 4);
    DA(lt_pairs, B * (size_t)c->pair_cap * 2);
    DA(lt_kill, B * c->tail_cap);
    DA(lt_pend, B * c->tail_cap);
    DA(lt_cst, B * c->tail_cap);
    DA(mCount, 16);
    HA(h_mcount, 4);
    for (Slot& sl : c->slot) {
#define DS(ptr, n) do { rc = dalloc(c, &sl.ptr, (n)); if (rc) return rc; } while (0)
#define HS(ptr, n) do { rc = halloc(c, &sl.ptr, (n)); if (rc) return rc; } while (0)
        {
            size_t off = 0;
            auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
            const size_t E = c->edge_cap, M1 = c->maxn + 1;
            const size_t o_cnt = take(sizeof(int) * 8 * B), o_xy = take(sizeof(int32_t) * 2 * B * c->maxn),
                         o_sc = take(sizeof(float) * B * c->maxn), o_un = take(sizeof(float) * 2 * B * c->maxn),
                         o_out = take((size_t)B * c->maxn), o_desc = take(sizeof(float) * 256 * B * c->maxn),
                         o_ei = take(sizeof(uint32_t) * 2 * E * B), o_es = take(sizeof(float) * E * B),
                         o_co = take(sizeof(int32_t) * M1 * B), o_ci = take(sizeof(uint32_t) * 2 * E * B),
                         o_lo = take(sizeof(int32_t) * M1 * B), o_lp = take(sizeof(uint32_t) * 2 * E * B);
            sl.res_bytes = off;
            DS(d_res, off);
            HS(h_res, off);
#define RES(dev, host, T, o) do { sl.dev = (T*)(sl.d_res + (o)); sl.host = (T*)(sl.h_res + (o)); } while (0)
            RES(counters, h_counters, int, o_cnt);
            RES(kp_xy, h_kp_xy, int32_t, o_xy);
            RES(kp_score, h_kp_score, float, o_sc);
            RES(kp_un, h_kp_un, float, o_un);
            RES(kp_out, h_kp_out, uint8_t, o_out);
            RES(desc, h_desc, float, o_desc);
            RES(edge_idx, h_edge_idx, uint32_t, o_ei);
            RES(edge_score, h_edge_score, float, o_es);
            RES(conn_off, h_conn_off, int32_t, o_co);
            RES(conn_idx, h_conn_idx, uint32_t, o_ci);
            RES(coline_off, h_coline_off, int32_t, o_lo);
            RES(coline_pairs, h_coline_pairs, uint32_t, o_lp);
#undef RES
        }
        DS(sat, 1);
        DS(cand_ij, B * c->cand_cap);
        DS(cand_sc, B * c->cand_cap * 2);
        DS(d_match, B * c->maxn);
        DS(d_mcounts, B);
        DS(junc, B * hw * 65);
        DS(hs2, B * HW * 4);
        DS(descmap, B * hw * 256);
        HS(h_sat, 1);
        sl.h_sat[0] = 0;
        HS(h_match, B * c->maxn);
        HS(h_mcounts, B);
        HS(h_img, B * HW);
        DS(d_img, B * HW);
#undef DS
#undef HS
        HIPCHK(c, hipMemset(sl.counters, 0, sizeof(int) * 8 * B));
        HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.cnn_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.junc_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.heat_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.match_done, hipEventDisableTiming));
        sl.frames.resize(B);
        sl.nkp.assign(B, 0);
    }
#undef DA
#undef HA
    select_slot(c, 0);
    c->tails.resize(B);
    // warm-up on an all-ones image, as the reference constructor does (PPGExtractor.cpp:100-106)
    // (at the full batch size, so that every kernel has run once at its production shape)
    std::vector<uint8_t> warm(HW, 1);
    std::vector<const uint8_t*> wp(B, warm.data());
    std::vector<ppg_frame_out> tmp(B);
    return ppg_extract_batch(c, wp.data(), (int)B, c->W, tmp.data());
}

static ppg_ctx* create_common(const ppg_config* cfg, bool matcher_only) {
    if (!cfg) { fail(nullptr, PPG_EINVAL, "cfg is NULL"); return nullptr; }
    ppg_ctx* c = new (std::nothrow) ppg_ctx();
    if (!c) { fail(nullptr, PPG_ENOMEM, "out of host memory"); return nullptr; }
    c->cfg = *cfg;
    c->matcher_only = matcher_only;
    const int rc = create_impl(c);
    if (rc != PPG_OK) {
        g_err = c->err;
        ppg_destroy(c);
        return nullptr;
    }
    return c;
}

ppg_ctx* ppg_create(const ppg_config* cfg) { return create_common(cfg, false); }

ppg_ctx* ppg_create_matcher(const ppg_config* cfg) { return create_common(cfg, true); }

int ppg_set_tunables(ppg_ctx* c, const ppg_config* f) {
    if (!c) return PPG_EINVAL;
    if (!f) return fail(c, PPG_EINVAL, "ppg_set_tunables: cfg is NULL");
    NEED_EXTRACTOR(c, "ppg_set_tunables");
    if (any_inflight(c)) return fail(c, PPG_EINVAL, "ppg_set_tunables: a submitted batch is pending: call ppg_wait_batch first");
    if (f->desc_dim != c->cfg.desc_dim || f->junction_nms_radius != c->cfg.junction_nms_radius || f->junction_max_num != c->cfg.junction_max_num)
        return fail(c, PPG_ECAP, "ppg_set_tunables: desc_dim, junction_nms_radius and junction_max_num size buffers and kernels: create a new context");
    if (f->heatmap_refine_sz < 1 || f->heatmap_refine_sz > 16) return fail(c, PPG_EINVAL, "heatmap_refine_sz must be 1..16");
    c->cfg.junction_thresh = f->junction_thresh;
    c->cfg.line_valid_thresh = f->line_valid_thresh;
    c->cfg.line_valid_ratio = f->line_valid_ratio;
    c->cfg.line_dist_thresh = f->line_dist_thresh;
    c->cfg.heatmap_refine_sz = f->heatmap_refine_sz;
    c->cfg.line_heatmap_thresh = f->line_heatmap_thresh;
    c->cfg.line_inlier_rate = f->line_inlier_rate;
    return PPG_OK;
}

void* ppg_stream(ppg_ctx* c) { return c ? (void*)c->stream : nullptr; }

size_t ppg_tensor_elems(const ppg_ctx* c, int which) { return c ? tensor_elems(c, which) : 0; }

int ppg_read_tensor(ppg_ctx* c, int which, int frame, float* dst, size_t n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_read_tensor");
    const size_t e = tensor_elems(c, which);
    if (!e || !dst || n != e || frame < 0 || frame >= c->B) return fail(c, PPG_EINVAL, "ppg_read_tensor: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamSynchronize(l.stream));
    HIPCHK(c, hipStreamSynchronize(c->post));
    HIPCHK(c, hipMemcpy(dst, tensor_ptr(c, which) + (size_t)frame * e, e * sizeof(float), hipMemcpyDeviceToHost));
    return PPG_OK;
}

int ppg_write_tensor(ppg_ctx* c, int which, int frame, const float* src, size_t n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_write_tensor");
    const size_t e = tensor_elems(c, which);
    if (!e || !src || n != e || frame < 0 || frame >= c->B) return fail(c, PPG_EINVAL, "ppg_write_tensor: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamSynchronize(l.stream));
    HIPCHK(c, hipStreamSynchronize(c->post));
    HIPCHK(c, hipMemcpy(tensor_ptr(c, which) + (size_t)frame * e, src, e * sizeof(float), hipMemcpyHostToDevice));
    // an injected map replaces the one the next stage call would compute from the network outputs
    if (which == PPG_T_SCORE) c->skip_score = true;
    if (which == PPG_T_HEAT || which == PPG_T_HEAT_RAW) c->skip_heat = true;
    if (which == PPG_T_HEAT_REFINED) { c->skip_heat = true; c->skip_refine = true; }
    return PPG_OK;
}

bool early_post(const ppg_ctx* c, int n);

int ppg_inference(ppg_ctx* c, const uint8_t* imgs, int n, int on_device) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_inference");
    if (!imgs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_inference: n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const size_t HW = (size_t)c->H * c->W;
    const uint8_t* dimg = imgs;
    c->skip_score = c->skip_heat = c->skip_refine = false;
    {
        const int ns = next_slot(c->cur);
        if (c->slot[ns].inflight || inflight_count(c) >= kMaxInflight)
            return fail(c, PPG_EINVAL, "four batches are in flight: call ppg_wait_batch first");
        select_lane(c, (c->lane_cur + 1) % c->nlanes);
        select_slot(c, ns);
        c->slot[ns].n = n;
        c->slot[ns].matched = false;
        // nkp keeps the counts of the slot's last collected batch: a later ppg_wait_batch of the OTHER slot
        // still needs them to size its match rows
    }
    if (!on_device) {
        // The upload has a stream of its own and the host waits for it here, before the convolutions are enqueued:
        // the copy of batch k+1 then runs beside the CNN of batch k, which is still queued or running, and the CNN
        // stream itself never waits for a copy.  The copy is a kernel reading the pinned buffer (see launch_upload).
        // (An event wait between the two streams instead of the host wait was
        // tried: with two batches in flight the runtime left the second copy unsubmitted until a device-wide
        // synchronisation.)  The slot's buffers are free: a slot is reused only after its batch was waited for.
        Slot& sl = c->slot[c->cur];
        if (imgs != c->h_img) memcpy(c->h_img, imgs, n * HW);  // pinned staging buffer of this slot
        HIPCHK(c, ppg::launch_upload(c->h_img, sl.d_img, n * HW, c->up_stream));
        HIPCHK(c, hipStreamSynchronize(c->up_stream));
        dimg = sl.d_img;
    }
    if (c->cfg.precision == 3) HIPCHK(c, hipMemsetAsync(c->slot[c->cur].sat, 0, sizeof(int), c->stream));
    const int H = c->H, W = c->W;
    // conv1a (u8 -> /255 -> conv 1->64 -> ReLU, backbone.py:28) is evaluated inside conv1b's staging, so the
    // full-resolution 64-channel tensor never goes to memory; PPG_UNFUSED_CONV1A=1 keeps the separate kernel.
    const bool fuse1a = !c->unfused_conv1a;
    if (!fuse1a) {
        ppg::Conv1aArgs a1;
        a1.img = dimg; a1.w = c->c1a_w; a1.bias = c->c1a_b; a1.out = c->a1; a1.B = n; a1.H = H; a1.W = W;
        ProfScope ps(c, PPG_PROF_CONV1A, c->stream);
        HIPCHK(c, ppg::launch_conv1a(a1, c->stream));
    }
    int rc;
#define RC(x) do { rc = (x); if (rc) return rc; } while (0)
    // Backbone (net/Backbone.pt backbone.py:28-37)
    RC(run_conv(c, c->c1b, c->a1, c->p1, n, H, W, ppg::CONV_POOL2, 1, fuse1a ? dimg : nullptr));
    RC(run_conv(c, c->c2a, c->p1, c->a2, n, H / 2, W / 2, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c2b, c->a2, c->p2, n, H / 2, W / 2, ppg::CONV_POOL2, 1));
    RC(run_conv(c, c->c3a, c->p2, c->a3, n, H / 4, W / 4, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c3b, c->a3, c->p3, n, H / 4, W / 4, ppg::CONV_POOL2, 1));
    RC(run_conv(c, c->c4a, c->p3, c->a4, n, H / 8, W / 8, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c4b, c->a4, c->feat, n, H / 8, W / 8, ppg::CONV_PLAIN, 1));
    // PointHeatmap (junction_decoder.py:14-15)
    RC(run_heads(c, n, H / 8, W / 8));  // convPa + convDa + EdgeHeatmap block 0
    RC(run_conv(c, c->pb, c->pa_o, c->junc, n, H / 8, W / 8, ppg::CONV_PLAIN, 0));
    if (early_post(c, n)) HIPCHK(c, hipEventRecord(c->slot[c->cur].junc_done, c->stream));
    // EdgeHeatmap (heatmap_decoder.py:19-22); the last 1x1 conv runs in ppg_detect_lines' first kernel
    RC(run_conv(c, c->h1, c->hs0, c->hs1, n, H / 4, W / 4, ppg::CONV_PS2, 1));
    RC(run_conv(c, c->h2, c->hs1, c->hs2, n, H / 2, W / 2, ppg::CONV_PS2, 1));
    if (early_post(c, n)) HIPCHK(c, hipEventRecord(c->slot[c->cur].heat_done, c->stream));
    // Descriptor (descriptor_decoder.py:14-15)
    RC(run_conv(c, c->db, c->da_o, c->descmap, n, H / 8, W / 8, ppg::CONV_PLAIN, 0));
#undef RC
    HIPCHK(c, hipEventRecord(c->slot[c->cur].cnn_done, c->stream));
    c->last_n = n;
    return PPG_OK;
}

// Small batches start the key-point kernels as soon as the junction logits are written and the line kernels as soon as
// EdgeHeatmap is through, beside the last convolutions: the call returns 5 % sooner.  Large batches wait for the whole CNN:
// started early, the post-processing of a 32-frame batch lands on the next batch's longest convolution instead of
// spreading over the later ones (measured: +0.7 % frames/s, but that launch 5 % longer), and nobody waits for a frame there.
// PPG_EARLY_POST=0 / 1 forces either.
bool early_post(const ppg_ctx* c, int n) {
    static const char* force = getenv("PPG_EARLY_POST");
    if (force) return atoi(force) != 0;
    return (long)n * c->H * c->W <= 16L * 640 * 480;
}

int ppg_detect_keypoints(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    // the slot's counters are cleared before the stream waits for the networks: the clear then runs beside them instead
    // of on the chain behind the junction logits (the slot is free: its last batch was waited for)
    HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(int) * 8 * n, c->post));
    HIPCHK(c, hipStreamWaitEvent(c->post, early_post(c, n) ? c->slot[c->cur].junc_done : c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_POST, c->post);
    ppg::NmsArgs a = nms_args(c, n);
    if (!c->skip_score) HIPCHK(c, ppg::launch_score_map(a, c->post));
    HIPCHK(c, ppg::launch_nms(a, c->post));
    return PPG_OK;
}

int ppg_detect_lines(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamWaitEvent(c->post, early_post(c, n) ? c->slot[c->cur].heat_done : c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_LINES, c->post);
    ppg::HeatArgs ha = heat_args(c, n);
    // last layer of EdgeHeatmap (1x1 conv 4->2, heatmap_decoder.py:22) fused with the 2-way softmax of :242
    if (!c->skip_heat) HIPCHK(c, ppg::launch_heat_tail(ha, c->post));
    if (!c->skip_refine) HIPCHK(c, ppg::launch_refine(ha, c->post));
    if (c->remapped) {  // undistort the heat map (PPGExtractor.cpp:261-262)
        ppg::RemapArgs ra;
        ra.src = c->refined; ra.dst = c->remapped; ra.mapx = c->mapx; ra.mapy = c->mapy; ra.B = n; ra.H = c->H; ra.W = c->W;
        HIPCHK(c, ppg::launch_remap(ra, c->post));
    }
    ppg::LineArgs la;
    la.heat = c->remapped ? c->remapped : c->refined; la.kp_un = c->kp_un; la.kp_out = c->kp_out; la.counters = c->counters;
    la.row_cnt = c->row_cnt; la.pair_mask = c->pair_mask; la.mask_words = c->mask_words; la.cand_ij = c->cand_ij; la.cand_sc = c->cand_sc; la.cand_cap = c->cand_cap;
    la.B = n; la.H = c->H; la.W = c->W; la.max_num = c->maxn;
    la.heat_thresh = c->cfg.line_heatmap_thresh;
    la.inv_scale = 1.0f / std::sqrt((float)(c->H * c->H + c->W * c->W));  // PPGExtractor.cpp:74
    HIPCHK(c, ppg::launch_line_candidates(la, c->post));
    if (!c->lines_on_host) {
        // overlap / score / collinearity logic of :315-441 on the device; frames it flags are redone on the host
        Slot& sl = c->slot[c->cur];
        ppg::LineTailArgs t;
        t.kp_un = c->kp_un; t.counters = c->counters; t.row_off = c->row_cnt; t.cand_ij = c->cand_ij; t.cand_sc = c->cand_sc;
        t.cand_cap = c->cand_cap; t.geo = c->lt_geo; t.col_off = c->lt_col_off; t.col_cur = c->lt_col_cur; t.col_ent = c->lt_col_ent;
        t.pairs = c->lt_pairs; t.pair_cap = c->pair_cap; t.kill_t = c->lt_kill; t.pend_t = c->lt_pend; t.cst = c->lt_cst;
        t.tail_cap = c->tail_cap; t.edge_idx = sl.edge_idx; t.edge_score = sl.edge_score; t.conn_off = sl.conn_off;
        t.conn_idx = sl.conn_idx; t.coline_off = sl.coline_off; t.coline_pairs = sl.coline_pairs; t.edge_cap = c->edge_cap;
        t.B = n; t.max_num = c->maxn; t.dist_thresh = c->cfg.line_dist_thresh; t.inlier_rate = c->cfg.line_inlier_rate;
        t.heat_thresh = c->cfg.line_heatmap_thresh;
        HIPCHK(c, ppg::launch_line_tail(t, c->post));
    }
    return PPG_OK;
}

int ppg_gen_descriptors(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamWaitEvent(c->post, c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_DESC, c->post);
    ppg::DescArgs a;
    a.descmap = c->descmap; a.kp_xy = c->kp_xy; a.counters = c->counters; a.desc = c->desc;
    a.B = n; a.H = c->H; a.W = c->W; a.max_num = c->maxn;
    HIPCHK(c, ppg::launch_desc_sample(a, c->post));
    return PPG_OK;
}

// Device-to-host copies of everything the host needs from slot `si`, then the slot's done event.
static int enqueue_d2h(ppg_ctx* c, int si, int n) {
    Slot& sl = c->slot[si];
    const int M = c->maxn;
    hipStream_t s = c->post;
    if (c->cfg.precision == 3) HIPCHK(c, hipMemcpyAsync(sl.h_sat, sl.sat, sizeof(int), hipMemcpyDeviceToHost, s));
    if (n == c->B && !c->lines_on_host) {
        HIPCHK(c, hipMemcpyAsync(sl.h_res, sl.d_res, sl.res_bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK(c, hipMemcpyAsync(sl.h_counters, sl.counters, sizeof(int) * 8 * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_xy, sl.kp_xy, sizeof(int32_t) * 2 * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_score, sl.kp_score, sizeof(float) * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_un, sl.kp_un, sizeof(float) * 2 * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_out, sl.kp_out, (size_t)M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_desc, sl.desc, sizeof(float) * 256 * M * n, hipMemcpyDeviceToHost, s));
        if (!c->lines_on_host) {
            const size_t E = c->edge_cap;
            HIPCHK(c, hipMemcpyAsync(sl.h_edge_idx, sl.edge_idx, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_edge_score, sl.edge_score, sizeof(float) * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_conn_off, sl.conn_off, sizeof(int32_t) * (M + 1) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_conn_idx, sl.conn_idx, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_coline_off, sl.coline_off, sizeof(int32_t) * (M + 1) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_coline_pairs, sl.coline_pairs, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
        }
    }
    if (sl.matched) {
        hipStream_t ms = c->match_stream;
        HIPCHK(c, hipMemcpyAsync(sl.h_match, sl.d_match, sizeof(int32_t) * M * n, hipMemcpyDeviceToHost, ms));
        HIPCHK(c, hipMemcpyAsync(sl.h_mcounts, sl.d_mcounts, sizeof(int) * n, hipMemcpyDeviceToHost, ms));
        HIPCHK(c, hipEventRecord(sl.match_done, ms));
    }
    HIPCHK(c, hipEventRecord(sl.done, s));
    return PPG_OK;
}

// Waits for slot `si`, runs the host tails of its frames and fills `outs`.
static int finish_slot(ppg_ctx* c, int si, int n, ppg_frame_out* outs) {
    Slot& sl = c->slot[si];
    const int M = c->maxn;
    // the slot's data is complete once `done` fired; late copies use their own stream so that they do
    // not queue behind the next batch already running on the main one
    hipStream_t s = c->copy_stream;
    const double t_sync0 = now_ms();
    HIPCHK(c, hipEventSynchronize(sl.done));
    if (sl.matched) HIPCHK(c, hipEventSynchronize(sl.match_done));
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_WAIT] += now_ms() - t_sync0; c->prof_launches[PPG_PROF_HOST_WAIT] += 1; }
    if (c->cfg.precision == 3 && sl.h_sat[0])
        return fail(c, PPG_ERANGE, "an activation left the range of the fp16 split (|x| > 2047): use precision 0 or 1");
    for (int b = 0; b < n; b++) {
        const int* cnt = sl.h_counters + b * 8;
        if (cnt[0] > c->resid_cap || cnt[1] > c->kept_cap || cnt[2] > c->cand_cap)
            return fail(c, PPG_ECAP, "frame %d: work list overflow (resid %d kept %d cand %d)", b, cnt[0], cnt[1], cnt[2]);
    }
    if (c->prof) prof_resolve(c);
    static const bool dbg_counters = getenv("PPG_DEBUG_COUNTERS") != nullptr;   // measurement aid: work-list sizes of frame 0
    if (dbg_counters) {
        const int* k = sl.h_counters;
        fprintf(stderr, "[ppg] frame 0: %d residual pixels (%d rounds), %d kept, %d line candidates, %d key points, %d edges, %d pairs (%d rounds)\n",
                k[0], k[7] & 255, k[1], k[2], k[3], k[4], k[6], k[7] >> 8);
    }
    for (int b = 0; b < n; b++) sl.nkp[b] = sl.h_counters[b * 8 + 3];
    const double t_tail0 = now_ms();
    const int M1 = M + 1;
    // Frames whose line graph the device tail finished point straight into the pinned mirrors.  The others (a value
    // too close to a rounding boundary, a work array too small, or PPG_LINES_HOST) get the host statement of the
    // tail, lines_tail(), on their candidate list, which is fetched only now.
    std::vector<int> redo;
    for (int b = 0; b < n; b++) {
        const int* cnt = sl.h_counters + b * 8;
        const int N = cnt[3];
        ppg_frame_out& o = outs[b];
        o.n = N;
        o.kp_xy = sl.h_kp_xy + (size_t)b * M * 2;
        o.kp_score = sl.h_kp_score + (size_t)b * M;
        o.kp_un = sl.h_kp_un + (size_t)b * M * 2;
        o.kp_out = sl.h_kp_out + (size_t)b * M;
        o.desc = sl.h_desc + (size_t)b * M * 256;
        if (N > 0 && (c->lines_on_host || cnt[5] != 0)) { redo.push_back(b); continue; }
        FrameHost& fh = sl.frames[b];
        if (N == 0) {
            // PPGExtractor.cpp:239-240 returns before touching mvKeyEdges (stale edges of the previous
            // frame stay in the reference's member); with no key points they index nothing, so this
            // boundary reports an empty graph.
            fh.conn_off.assign(1, 0);
            fh.coline_off.assign(1, 0);
            o.n_edges = 0;
            o.edge_idx = nullptr; o.edge_score = nullptr; o.conn_idx = nullptr; o.coline_pairs = nullptr;
            o.conn_off = fh.conn_off.data(); o.coline_off = fh.coline_off.data();
            continue;
        }
        o.n_edges = cnt[4];
        o.edge_idx = sl.h_edge_idx + (size_t)b * c->edge_cap * 2;
        o.edge_score = sl.h_edge_score + (size_t)b * c->edge_cap;
        o.conn_off = sl.h_conn_off + (size_t)b * M1;
        o.conn_idx = sl.h_conn_idx + (size_t)b * c->edge_cap * 2;
        o.coline_off = sl.h_coline_off + (size_t)b * M1;
        o.coline_pairs = sl.h_coline_pairs + (size_t)b * c->edge_cap * 2;
    }
    c->tail_frames += n;
    c->tail_fallbacks += (long long)redo.size();
    for (int b : redo) {
        const int* cnt = sl.h_counters + b * 8;
        const int N = cnt[3], nc = cnt[2];
        c->fb_ij.resize((size_t)std::max(nc, 1));
        c->fb_sc.resize((size_t)std::max(nc, 1) * 2);
        if (nc > 0) {
            HIPCHK(c, hipMemcpyAsync(c->fb_ij.data(), sl.cand_ij + (size_t)b * c->cand_cap, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->fb_sc.data(), sl.cand_sc + (size_t)b * c->cand_cap * 2, sizeof(float) * 2 * nc, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
        ppg_frame_out& o = outs[b];
        if (c->dump_tail && !c->dump_tail_done && nc > 0) {
            // debugging aid (PPG_DUMP_TAIL=<file>): the first frame's input of lines_tail(), in the format that
            // tools/bench_tail.cpp reads
            c->dump_tail_done = true;
            if (FILE* f = fopen(c->dump_tail, "wb")) {
                const int32_t hdr[2] = {N, nc};
                fwrite(hdr, 4, 2, f);
                fwrite(o.kp_un, 4, (size_t)2 * N, f);
                fwrite(c->fb_ij.data(), 4, (size_t)nc, f);
                fwrite(c->fb_sc.data(), 4, (size_t)2 * nc, f);
                fclose(f);
            }
        }
        FrameHost& fh = sl.frames[b];
        lines_tail(c->cfg, N, o.kp_un, c->fb_ij.data(), c->fb_sc.data(), nc, fh, c->tails[b]);
        o.n_edges = (int)fh.edge_score.size();
        o.edge_idx = fh.edge_idx.data();
        o.edge_score = fh.edge_score.data();
        o.conn_off = fh.conn_off.data();
        o.conn_idx = fh.conn_idx.data();
        o.coline_off = fh.coline_off.data();
        o.coline_pairs = fh.coline_pairs.data();
    }
    const int redone = (int)redo.size();
    if (c->prof) {
        c->prof_ms[PPG_PROF_HOST_TAIL] += now_ms() - t_tail0;
        c->prof_launches[PPG_PROF_HOST_TAIL] += 1;
        c->prof_launches[PPG_PROF_HOST_TAIL_FRAMES] += redone;   // frames the host had to redo
    }
    return PPG_OK;
}

int ppg_collect(ppg_ctx* c, int n, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_collect");
    if (n < 1 || n > c->B || !outs) return fail(c, PPG_EINVAL, "ppg_collect: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    int rc = enqueue_d2h(c, c->cur, n);
    if (!rc) rc = finish_slot(c, c->cur, n, outs);
    return rc;
}

static int match_resident(ppg_ctx* c, const float* dA, const int* nA, const float* dB, const int* nB, int batch,
                          float nnratio, int32_t* d_match, int* d_mcounts);

static int enqueue_extract(ppg_ctx* c, const uint8_t* imgs, int n, int on_device) {
    static const int skip = getenv("PPG_DEBUG_SKIP") ? atoi(getenv("PPG_DEBUG_SKIP")) : 0;  // measurement aid only
    int rc = ppg_inference(c, imgs, n, on_device);
    if (!rc && !(skip & 1)) rc = ppg_detect_keypoints(c, n);
    if (!rc && !(skip & 2)) rc = ppg_detect_lines(c, n);
    if (!rc && !(skip & 4)) rc = ppg_gen_descriptors(c, n);
    return rc;
}

static int extract_common(ppg_ctx* c, const uint8_t* imgs, int n, int on_device, ppg_frame_out* outs) {
    if (any_inflight(c))
        return fail(c, PPG_EINVAL, "a submitted batch is pending: call ppg_wait_batch first");
    int rc = enqueue_extract(c, imgs, n, on_device);
    if (!rc) rc = ppg_collect(c, n, outs);
    return rc;
}

static int stage_host_images(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, const uint8_t** staged);

static int submit_common(ppg_ctx* c, const uint8_t* imgs, int n, int on_device, int match_prev, float nnratio) {
    const double t0 = now_ms();
    if (match_prev && c->slot[c->cur].n < n)
        return fail(c, PPG_EINVAL, "match_prev needs a previous batch of at least %d frames", n);
    int rc = enqueue_extract(c, imgs, n, on_device);  // switches to the other slot or fails if it is in flight
    if (rc) return rc;
    Slot& sl = c->slot[c->cur];
    if (match_prev) {
        rc = match_resident(c, c->desc_prev, c->counters_prev + 3, c->desc, c->counters + 3, n, nnratio, sl.d_match, sl.d_mcounts);
        if (rc) return rc;
        sl.matched = true;
    }
    rc = enqueue_d2h(c, c->cur, n);
    if (rc) return rc;
    sl.inflight = true;
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_SUBMIT] += now_ms() - t0; c->prof_launches[PPG_PROF_HOST_SUBMIT] += 1; }
    return PPG_OK;
}

int ppg_submit_batch_device(ppg_ctx* c, const uint8_t* d_imgs, int n, int match_prev, float nnratio) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_submit_batch_device");
    if (!d_imgs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_submit_batch_device: bad argument");
    return submit_common(c, d_imgs, n, 1, match_prev, nnratio);
}

int ppg_submit_batch(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, int match_prev, float nnratio) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_submit_batch");
    const uint8_t* staged = nullptr;
    int rc = stage_host_images(c, imgs, n, row_stride, &staged);
    if (rc) return rc;
    return submit_common(c, staged, n, 0, match_prev, nnratio);
}

int ppg_wait_batch(ppg_ctx* c, ppg_frame_out* outs, int32_t* m12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_wait_batch");
    if (!outs) return fail(c, PPG_EINVAL, "ppg_wait_batch: outs is NULL");
    // slots alternate, so the older pending batch is the one that is not current (if it is pending)
    int si = c->cur;
    for (int k = 1; k <= kSlots; k++) {
        si = (c->cur + k) % kSlots;
        if (c->slot[si].inflight) break;
    }
    Slot& sl = c->slot[si];
    if (!sl.inflight) return fail(c, PPG_EINVAL, "ppg_wait_batch: nothing was submitted");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int n = sl.n, M = c->maxn;
    const double t0 = now_ms();
    int rc = finish_slot(c, si, n, outs);
    sl.inflight = false;
    if (rc) return rc;
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_FINISH] += now_ms() - t0; c->prof_launches[PPG_PROF_HOST_FINISH] += 1; }
    if (sl.matched) {
        const Slot& prev = c->slot[prev_slot(si)];
        for (int b = 0; b < n; b++) {
            if (m12) {
                memcpy(m12 + (size_t)b * M, sl.h_match + (size_t)b * M, sizeof(int32_t) * prev.nkp[b]);
                for (int k = prev.nkp[b]; k < M; k++) m12[(size_t)b * M + k] = -1;
            }
            if (n_matches) n_matches[b] = sl.h_mcounts[b];
        }
    } else {
        if (m12) for (size_t k = 0; k < (size_t)n * M; k++) m12[k] = -1;
        if (n_matches) for (int b = 0; b < n; b++) n_matches[b] = 0;
    }
    return n;
}

// copies n host images (row_stride >= W) into the pinned staging buffer of the slot the next extract uses
static int stage_host_images(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, const uint8_t** staged) {
    if (!imgs || n < 1 || n > c->B || row_stride < c->W)
        return fail(c, PPG_EINVAL, "bad image batch (n = %d, max_batch = %d, row_stride = %d)", n, c->B, row_stride);
    Slot& nx = c->slot[next_slot(c->cur)];
    if (nx.inflight || inflight_count(c) >= kMaxInflight) return fail(c, PPG_EINVAL, "four batches are in flight: call ppg_wait_batch first");
    const size_t HW = (size_t)c->H * c->W;
    for (int b = 0; b < n; b++)
        if (!imgs[b]) return fail(c, PPG_EINVAL, "image %d is NULL", b);
    auto copy_one = [&](int b) {
        if (row_stride == c->W) memcpy(nx.h_img + b * HW, imgs[b], HW);
        else
            for (int y = 0; y < c->H; y++) memcpy(nx.h_img + b * HW + (size_t)y * c->W, imgs[b] + (size_t)y * row_stride, c->W);
    };
    // a batch of VGA frames is ~10 MB: spread the copy into pinned memory over the host pool
    if (n >= 4 && c->host_threads > 1 && (size_t)n * HW >= (1u << 20)) {
        if (!c->pool) c->pool = new HostPool(c->host_threads - 1);
        c->pool->run(n, copy_one);
    } else {
        for (int b = 0; b < n; b++) copy_one(b);
    }
    *staged = nx.h_img;
    return PPG_OK;
}

int ppg_extract_batch(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_extract_batch");
    if (!outs) return fail(c, PPG_EINVAL, "ppg_extract_batch: outs is NULL");
    if (any_inflight(c))
        return fail(c, PPG_EINVAL, "a submitted batch is pending: call ppg_wait_batch first");
    const uint8_t* staged = nullptr;
    int rc = stage_host_images(c, imgs, n, row_stride, &staged);
    if (rc) return rc;
    return extract_common(c, staged, n, 0, outs);
}

int ppg_extract(ppg_ctx* c, const uint8_t* img, int row_stride, ppg_frame_out* out) {
    return ppg_extract_batch(c, &img, 1, row_stride, out);
}

int ppg_extract_batch_device(ppg_ctx* c, const uint8_t* d_imgs, int n, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_extract_batch_device");
    if (!d_imgs || !outs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_extract_batch_device: bad argument");
    return extract_common(c, d_imgs, n, 1, outs);
}

// ------------------------------------------------------------------------------------------
// matching/
// ------------------------------------------------------------------------------------------

float ppg_descriptor_distance(const float* a, const float* b, int dim) {
    // the chain every kernel of the library evaluates: one fused multiply-add per component, in order, then a
    // correctly rounded square root — no launch, no context
    float acc = 0.f;
    for (int d = 0; d < dim; d++) {
        const float t = a[d] - b[d];
        acc = std::fmaf(t, t, acc);
    }
    return std::sqrt(acc);
}

int ppg_dist_matrix(ppg_ctx* c, const float* A, int nA, const float* B, int nB, int dim, float* D) {
    if (!c) return PPG_EINVAL;
    if (nA < 0 || nB < 0 || dim <= 0 || dim % 16 || (nA && !A) || (nB && !B) || (nA && nB && !D))
        return fail(c, PPG_EINVAL, "ppg_dist_matrix: bad argument (dim must be a multiple of 16)");
    if (nA == 0 || nB == 0) return PPG_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    int rc = grow(c, c->mA, (size_t)nA * dim);
    if (!rc) rc = grow(c, c->mB, (size_t)nB * dim);
    if (!rc) rc = grow(c, c->mD, (size_t)nA * nB);
    if (rc) return rc;
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    HIPCHK(c, hipMemcpyAsync(c->mA.p, A, sizeof(float) * nA * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mB.p, B, sizeof(float) * nB * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, ppg::launch_dist_matrix(c->mA.p, nA, c->mB.p, nB, dim, c->mD.p, nB, s));
    HIPCHK(c, hipMemcpyAsync(D, c->mD.p, sizeof(float) * (size_t)nA * nB, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return PPG_OK;
}

int ppg_match_nn(ppg_ctx* c, const float* A, int nA, const float* B, int nB, int dim,
                 const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* skipB,
                 float th, float ratio, int mode, int32_t* match12, float* best, float* second, int* n_matches) {
    if (!c) return PPG_EINVAL;
    if (nA < 0 || nB < 0 || dim <= 0 || dim % 16 || (nA && !A) || (nB && !B) || (nA && !match12) ||
        (mode != PPG_MATCH_PLAIN && mode != PPG_MATCH_INIT) || ((cand_off == nullptr) != (cand_idx == nullptr)) || nB > 65536)
        return fail(c, PPG_EINVAL, "ppg_match_nn: bad argument");
    if (n_matches) *n_matches = 0;
    if (nA == 0) return PPG_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    const bool csr = cand_off != nullptr;
    const size_t nd = csr ? (size_t)cand_off[nA] : (size_t)nA * nB;
    if (csr) {
        if (cand_off[0] != 0) return fail(c, PPG_EINVAL, "cand_off[0] must be 0");
        for (int i = 0; i < nA; i++)
            if (cand_off[i + 1] < cand_off[i]) return fail(c, PPG_EINVAL, "cand_off must be non-decreasing");
        for (size_t k = 0; k < nd; k++)
            if (cand_idx[k] < 0 || cand_idx[k] >= nB) return fail(c, PPG_EINVAL, "cand_idx[%zu] out of range", k);
    }
    // All-pairs claim matching of rows of norm <= 1 (what genPointDescriptor produces) takes the path without a
    // distance matrix; everything else (candidate lists, skip flags, longer rows, other dimensions) the general one.
    bool fused = !csr && mode == PPG_MATCH_INIT && dim == 256 && !skipB && nA <= 1024 && nB >= 1 && nB <= 1024 && !c->match_matrix;
    auto norms_ok = [dim](const float* X, int n) {
        for (int i = 0; i < n; i++) {
            float s2 = 0.f;
            for (int d = 0; d < dim; d++) s2 += X[(size_t)i * dim + d] * X[(size_t)i * dim + d];
            if (!(s2 <= 1.0001f)) return false;
        }
        return true;
    };
    fused = fused && norms_ok(A, nA) && norms_ok(B, nB);
    int rc = grow(c, c->mA, (size_t)nA * dim);
    if (!rc) rc = grow(c, c->mB, (size_t)std::max(nB, 1) * dim);
    if (!rc && !fused) rc = grow(c, c->mD, std::max<size_t>(nd, 1));
    if (!rc && fused) rc = grow(c, c->mNorm, (size_t)2 * std::max(nA, nB));
    if (!rc && fused) rc = grow(c, c->mListCnt, (size_t)std::max(nA, nB));
    if (!rc && fused) rc = grow(c, c->mListCol, (size_t)std::max(nA, nB) * 64);
    if (!rc && fused) rc = grow(c, c->mHalf, (size_t)2 * std::max(nA, nB) * 256);
    if (!rc && fused) rc = grow(c, c->mBt, (size_t)std::max(nA, nB) * 256);
    if (!rc && fused) rc = grow(c, c->mExt, (size_t)64 * 128);
    if (!rc) rc = grow(c, c->mMatch, nA);
    if (!rc) rc = grow(c, c->mBest, nA);
    if (!rc) rc = grow(c, c->mSecond, nA);
    if (!rc) rc = grow(c, c->mTopD, (size_t)std::max(nA, nB) * 32 + 64);
    if (!rc) rc = grow(c, c->mTopI, (size_t)std::max(nA, nB) * 32 + 64);
    if (!rc && fused) rc = grow(c, c->mNearLen, (size_t)std::max(nA, nB));
    if (!rc && fused) rc = grow(c, c->mListRad, (size_t)std::max(nA, nB));
    if (!rc && csr) rc = grow(c, c->mOff, (size_t)nA + 1);
    if (!rc && csr) rc = grow(c, c->mIdx, std::max<size_t>(nd, 1));
    if (!rc && skipB) rc = grow(c, c->mSkip, std::max(nB, 1));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->mA.p, A, sizeof(float) * nA * dim, hipMemcpyHostToDevice, s));
    if (nB) HIPCHK(c, hipMemcpyAsync(c->mB.p, B, sizeof(float) * nB * dim, hipMemcpyHostToDevice, s));
    if (skipB && nB) HIPCHK(c, hipMemcpyAsync(c->mSkip.p, skipB, nB, hipMemcpyHostToDevice, s));
    if (fused) {
        ppg::FusedMatchArgs f;
        f.A = c->mA.p; f.B = c->mB.p; f.sA = 0; f.sB = 0; f.nA_dev = nullptr; f.nB_dev = nullptr; f.sn = 0; f.nA = nA; f.nB = nB;
        f.maxn = std::max(nA, nB); f.batch = 1; f.th = th; f.ratio = ratio;
        f.norm_work = c->mNorm.p; f.half_work = c->mHalf.p; f.bt_work = c->mBt.p; f.ext_work = c->mExt.p; f.list_cnt = c->mListCnt.p; f.list_col = c->mListCol.p;
        f.list_rad = c->mListRad.p; f.near_len = c->mNearLen.p;
        f.near_idx = c->mTopI.p; f.near_dist = c->mTopD.p; f.match12 = c->mMatch.p; f.best = c->mBest.p; f.second = c->mSecond.p;
        f.n_matches = c->mCount;
        HIPCHK(c, ppg::launch_match_fused(f, s));
    } else if (csr) {
        HIPCHK(c, hipMemcpyAsync(c->mOff.p, cand_off, sizeof(int32_t) * (nA + 1), hipMemcpyHostToDevice, s));
        if (nd) HIPCHK(c, hipMemcpyAsync(c->mIdx.p, cand_idx, sizeof(int32_t) * nd, hipMemcpyHostToDevice, s));
        HIPCHK(c, ppg::launch_dist_csr(c->mA.p, nA, c->mB.p, dim, c->mOff.p, c->mIdx.p, c->mD.p, s));
    } else if (nB) {
        HIPCHK(c, ppg::launch_dist_matrix(c->mA.p, nA, c->mB.p, nB, dim, c->mD.p, nB, s));
    }
    ppg::MatchArgs m;
    m.D = c->mD.p; m.ldd = csr ? 0 : nB; m.cand_off = csr ? c->mOff.p : nullptr; m.cand_idx = csr ? c->mIdx.p : nullptr;
    m.skipB = skipB ? c->mSkip.p : nullptr; m.nA_dev = nullptr; m.nB_dev = nullptr; m.nA = nA; m.nB = nB;
    m.th = th; m.ratio = ratio; m.topk_idx = c->mTopI.p; m.topk_dist = c->mTopD.p; m.claimed = nullptr;
    m.match12 = c->mMatch.p; m.best = c->mBest.p; m.second = c->mSecond.p; m.n_matches = c->mCount;
    if (!csr && nB == 0) {
        // no candidates at all: every row is unmatched
        for (int i = 0; i < nA; i++) { match12[i] = -1; if (best) best[i] = 1e6f; if (second) second[i] = 1e6f; }
        return PPG_OK;
    }
    if (fused) { /* enqueued above */ }
    else if (mode == PPG_MATCH_PLAIN) HIPCHK(c, ppg::launch_match_plain(m, s));
    else HIPCHK(c, ppg::launch_match_init(m, s));
    HIPCHK(c, hipMemcpyAsync(match12, c->mMatch.p, sizeof(int32_t) * nA, hipMemcpyDeviceToHost, s));
    if (best) HIPCHK(c, hipMemcpyAsync(best, c->mBest.p, sizeof(float) * nA, hipMemcpyDeviceToHost, s));
    if (second) HIPCHK(c, hipMemcpyAsync(second, c->mSecond.p, sizeof(float) * nA, hipMemcpyDeviceToHost, s));
    if (mode == PPG_MATCH_INIT) HIPCHK(c, hipMemcpyAsync(c->h_mcount, c->mCount, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (n_matches) {
        if (mode == PPG_MATCH_INIT) *n_matches = c->h_mcount[0];
        else {
            int k = 0;
            for (int i = 0; i < nA; i++) k += match12[i] >= 0;
            *n_matches = k;
        }
    }
    return PPG_OK;
}

int ppg_match_nn_noalloc(ppg_ctx* c, const float* dA, int nA, const float* dB, int nB, int dim, float th, float ratio,
                         int mode, int32_t* d_match12, float* d_best, float* d_second, int* d_n_matches, void* stream) {
    if (!c) return PPG_EINVAL;
    if (nA < 0 || nB < 0 || dim <= 0 || dim % 16 || (mode != PPG_MATCH_PLAIN && mode != PPG_MATCH_INIT) ||
        (nA && (!dA || !d_match12 || !d_best || !d_second)) || (nB && !dB) || (mode == PPG_MATCH_INIT && !d_n_matches))
        return fail(c, PPG_EINVAL, "ppg_match_nn_noalloc: bad argument (dim must be a multiple of 16)");
    if (nA > kNoAllocMaxRows || nB > kNoAllocMaxRows)
        return fail(c, PPG_ECAP, "ppg_match_nn_noalloc: %d x %d rows exceed the fixed work memory (at most %d x %d); "
                    "use ppg_match_nn", nA, nB, kNoAllocMaxRows, kNoAllocMaxRows);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    float* work = (float*)g_scratch_ptr[c->cfg.device_id];  // this GPU's copy, looked up at ppg_create
    if (!work) return fail(c, PPG_EHIP, "ppg_match_nn_noalloc: the work memory of device %d was never set up", c->cfg.device_id);
    hipStream_t s = stream ? (hipStream_t)stream : c->post;
    // One call at a time per GPU on the host, and device-side order between calls on different streams (see ScratchGuard).
    ScratchGuard& guard = g_scratch_guard[c->cfg.device_id];
    std::lock_guard<std::mutex> lk(guard.mu);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) HIPCHK(c, hipStreamIsCapturing(s, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;   // a captured call cannot wait for, or be waited for by, an event outside its graph
    if (!capturing && guard.used && guard.last != s) HIPCHK(c, hipStreamWaitEvent(s, guard.done, 0));
    if (nA > 0) {
        // Nothing in the work memory has to start at zero: the kernels below write every distance and list entry of
        // the nA x nB problem before they read it, and never read beyond it.
        float* D = work;
        float* top_d = work + kScratchDistFloats;
        int32_t* top_i = (int32_t*)(top_d + kScratchListEntries);
        const int ldd = std::max(nB, 1);
        HIPCHK(c, ppg::launch_dist_matrix(dA, nA, dB, nB, dim, D, ldd, s));
        ppg::MatchArgs m;
        m.D = D; m.ldd = ldd; m.cand_off = nullptr; m.cand_idx = nullptr; m.skipB = nullptr; m.nA_dev = nullptr; m.nB_dev = nullptr;
        m.nA = nA; m.nB = nB; m.th = th; m.ratio = ratio; m.topk_idx = top_i; m.topk_dist = top_d; m.claimed = nullptr;
        m.match12 = d_match12; m.best = d_best; m.second = d_second; m.n_matches = d_n_matches;
        if (mode == PPG_MATCH_PLAIN) HIPCHK(c, ppg::launch_match_plain(m, s));
        else HIPCHK(c, ppg::launch_match_init(m, s));
    } else if (mode == PPG_MATCH_INIT) {
        HIPCHK(c, hipMemsetAsync(d_n_matches, 0, sizeof(int), s));
    }
    if (!capturing) {
        HIPCHK(c, hipEventRecord(guard.done, s));
        guard.last = s;
        guard.used = true;
    }
    if (!stream) HIPCHK(c, hipStreamSynchronize(s));   // (still under the lock: the work memory is free when the next caller gets it)
    return PPG_OK;
}

int ppg_match_mutual(ppg_ctx* c, const float* A, int nA, const float* B, int nB, int dim, const int32_t* o12,
                     const int32_t* i12, const int32_t* o21, const int32_t* i21, float th, int32_t* match12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    if (nA < 0 || nB < 0 || (nA && !match12) || ((o12 == nullptr) != (i12 == nullptr)) || ((o21 == nullptr) != (i21 == nullptr)) ||
        ((o12 == nullptr) != (o21 == nullptr)))
        return fail(c, PPG_EINVAL, "ppg_match_mutual: bad argument");
    if (n_matches) *n_matches = 0;
    for (int i = 0; i < nA; i++) match12[i] = -1;
    if (nA == 0 || nB == 0) return PPG_OK;
    std::vector<int32_t> m1(nA, -1), m2(nB, -1);
    std::vector<float> b1(nA), b2(nB);
    int rc = ppg_match_nn(c, A, nA, B, nB, dim, o12, i12, nullptr, th, 0.f, PPG_MATCH_PLAIN, m1.data(), b1.data(), nullptr, nullptr);
    if (!rc) rc = ppg_match_nn(c, B, nB, A, nA, dim, o21, i21, nullptr, th, 0.f, PPG_MATCH_PLAIN, m2.data(), b2.data(), nullptr, nullptr);
    if (rc) return rc;
    int nFound = 0;
    for (int i1 = 0; i1 < nA; i1++) {
        const int idx2 = (m1[i1] >= 0 && b1[i1] <= th) ? m1[i1] : -1;   // :1254-1257
        if (idx2 < 0) continue;
        const int idx1 = (m2[idx2] >= 0 && b2[idx2] <= th) ? m2[idx2] : -1;  // :1309-1312
        if (idx1 == i1) { match12[i1] = idx2; nFound++; }
    }
    if (n_matches) *n_matches = nFound;
    return PPG_OK;
}

}  // extern "C"

// ---- Frame grid (host integer logic) ----
struct ppg_grid {
    int minX, minY, maxX, maxY;
    float invW, invH;
    int n;
    int fisheye;   // camera model of the configuration the grid was built with (projection of ppg_check_in_frustum)
    float D[4];
    std::vector<float> pos;
    std::vector<int32_t> cell[64][48];  // FRAME_GRID_COLS x FRAME_GRID_ROWS, sensors/include/GeometricCamera.h:79-80
};

namespace {
void undistort_pinhole_host(const ppg_config& c, float x, float y, float* ou, float* ov) {
    const double fx = c.K[0], fy = c.K[4], cx = c.K[2], cy = c.K[5];
    const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3];
    double xn = (x - cx) * (1.0 / fx), yn = (y - cy) * (1.0 / fy);
    const double x0 = xn, y0 = yn;
    for (int j = 0; j < 5; j++) {
        const double r2 = xn * xn + yn * yn;
        const double icdist = 1.0 / (1.0 + ((0.0 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) { xn = x0; yn = y0; break; }
        const double dX = 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn);
        const double dY = p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn;
        xn = (x0 - dX) * icdist;
        yn = (y0 - dY) * icdist;
    }
    *ou = (float)(fx * xn + cx);
    *ov = (float)(fy * yn + cy);
}
}  // namespace

extern "C" {

ppg_grid* ppg_grid_build(const ppg_config* cfg, const float* pos, int n) {
    if (!cfg || n < 0 || (n && !pos)) return nullptr;
    ppg_grid* g = new (std::nothrow) ppg_grid();
    if (!g) return nullptr;
    if (!cfg->fisheye) {  // sensors/src/GeometricCamera.cpp:28-47
        const float cs[4][2] = {{0.f, 0.f}, {(float)cfg->width, 0.f}, {0.f, (float)cfg->height}, {(float)cfg->width, (float)cfg->height}};
        float u[4][2];
        for (int k = 0; k < 4; k++) undistort_pinhole_host(*cfg, cs[k][0], cs[k][1], &u[k][0], &u[k][1]);
        g->minX = (int)std::min(u[0][0], u[2][0]);
        g->maxX = (int)std::max(u[1][0], u[3][0]);
        g->minY = (int)std::min(u[0][1], u[1][1]);
        g->maxY = (int)std::max(u[2][1], u[3][1]);
    } else {
        g->minX = 0; g->minY = 0; g->maxX = cfg->width; g->maxY = cfg->height;
    }
    g->fisheye = cfg->fisheye;
    for (int k = 0; k < 4; k++) g->D[k] = cfg->D[k];
    g->invW = 64.0f / (float)(g->maxX - g->minX);
    g->invH = 48.0f / (float)(g->maxY - g->minY);
    g->n = n;
    g->pos.assign(pos, pos + 2 * (size_t)n);
    for (int i = 0; i < n; i++) {  // map/src/Frame.cpp:149-155, :317-327
        const int px = (int)std::round((pos[2 * i] - g->minX) * g->invW);
        const int py = (int)std::round((pos[2 * i + 1] - g->minY) * g->invH);
        if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
        g->cell[px][py].push_back(i);
    }
    return g;
}

void ppg_grid_free(ppg_grid* g) { delete g; }

int ppg_features_in_area(const ppg_grid* g, float x, float y, float r, int32_t* out, int cap) {
    if (!g) return PPG_EINVAL;
    int cnt = 0;  // map/src/Frame.cpp:262-315
    const int minCX = std::max(0, (int)std::floor((x - g->minX - r) * g->invW));
    if (minCX >= 64) return 0;
    const int maxCX = std::min(63, (int)std::ceil((x - g->minX + r) * g->invW));
    if (maxCX < 0) return 0;
    const int minCY = std::max(0, (int)std::floor((y - g->minY - r) * g->invH));
    if (minCY >= 48) return 0;
    const int maxCY = std::min(47, (int)std::ceil((y - g->minY + r) * g->invH));
    if (maxCY < 0) return 0;
    for (int ix = minCX; ix <= maxCX; ix++)
        for (int iy = minCY; iy <= maxCY; iy++)
            for (int32_t id : g->cell[ix][iy]) {
                const float dx = g->pos[2 * id] - x, dy = g->pos[2 * id + 1] - y;
                if (std::fabs(dx) < r && std::fabs(dy) < r) {
                    if (out && cnt < cap) out[cnt] = id;
                    cnt++;
                }
            }
    return cnt;
}

int ppg_dist_lists(ppg_ctx* c, const float* A, int nA, const float* B, int nB, int dim, const int32_t* cand_off,
                   const int32_t* cand_idx, float* D) {
    if (!c) return PPG_EINVAL;
    if (nA < 0 || nB < 0 || dim <= 0 || dim % 16 || (nA && (!A || !cand_off)))
        return fail(c, PPG_EINVAL, "ppg_dist_lists: bad argument");
    if (nA == 0) return PPG_OK;
    if (cand_off[0] != 0) return fail(c, PPG_EINVAL, "cand_off[0] must be 0");
    for (int i = 0; i < nA; i++)
        if (cand_off[i + 1] < cand_off[i]) return fail(c, PPG_EINVAL, "cand_off must be non-decreasing");
    const size_t nd = (size_t)cand_off[nA];
    if (nd == 0) return PPG_OK;
    if (!cand_idx || !D || !B) return fail(c, PPG_EINVAL, "ppg_dist_lists: null list, matrix or output");
    for (size_t k = 0; k < nd; k++)
        if (cand_idx[k] < 0 || cand_idx[k] >= nB) return fail(c, PPG_EINVAL, "cand_idx[%zu] out of range", k);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    int rc = grow(c, c->mA, (size_t)nA * dim);
    if (!rc) rc = grow(c, c->mB, (size_t)nB * dim);
    if (!rc) rc = grow(c, c->mD, nd);
    if (!rc) rc = grow(c, c->mOff, (size_t)nA + 1);
    if (!rc) rc = grow(c, c->mIdx, nd);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->mA.p, A, sizeof(float) * nA * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mB.p, B, sizeof(float) * nB * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mOff.p, cand_off, sizeof(int32_t) * ((size_t)nA + 1), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mIdx.p, cand_idx, sizeof(int32_t) * nd, hipMemcpyHostToDevice, s));
    HIPCHK(c, ppg::launch_dist_csr(c->mA.p, nA, c->mB.p, dim, c->mOff.p, c->mIdx.p, c->mD.p, s));
    HIPCHK(c, hipMemcpyAsync(D, c->mD.p, sizeof(float) * nd, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return PPG_OK;
}

int ppg_distinctive_descriptors(ppg_ctx* c, const float* desc, const int32_t* seg_off, int P, int32_t* best) {
    if (!c) return PPG_EINVAL;
    if (P < 0 || (P && (!seg_off || !best))) return fail(c, PPG_EINVAL, "ppg_distinctive_descriptors: bad argument");
    if (P == 0) return PPG_OK;
    if (seg_off[0] != 0) return fail(c, PPG_EINVAL, "seg_off[0] must be 0");
    for (int p = 0; p < P; p++)
        if (seg_off[p + 1] < seg_off[p]) return fail(c, PPG_EINVAL, "seg_off must be non-decreasing");
    const int rows = seg_off[P], dim = 256;
    if (rows && !desc) return fail(c, PPG_EINVAL, "ppg_distinctive_descriptors: desc is null");
    for (int p = 0; p < P; p++) best[p] = 0;
    if (rows == 0) return PPG_OK;
    // row i of point p is compared with every row of p: a block-diagonal candidate list
    std::vector<int32_t> off((size_t)rows + 1, 0), idx;
    for (int p = 0; p < P; p++)
        for (int i = seg_off[p]; i < seg_off[p + 1]; i++) {
            for (int j = seg_off[p]; j < seg_off[p + 1]; j++) idx.push_back(j);
            off[(size_t)i + 1] = (int32_t)idx.size();
        }
    const size_t nd = idx.size();
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    int rc = grow(c, c->mA, (size_t)rows * dim);
    if (!rc) rc = grow(c, c->mD, nd);
    if (!rc) rc = grow(c, c->mOff, (size_t)rows + 1);
    if (!rc) rc = grow(c, c->mIdx, nd);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->mA.p, desc, sizeof(float) * rows * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mOff.p, off.data(), sizeof(int32_t) * ((size_t)rows + 1), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mIdx.p, idx.data(), sizeof(int32_t) * nd, hipMemcpyHostToDevice, s));
    HIPCHK(c, ppg::launch_dist_csr(c->mA.p, rows, c->mA.p, dim, c->mOff.p, c->mIdx.p, c->mD.p, s));
    std::vector<float> D(nd);
    HIPCHK(c, hipMemcpyAsync(D.data(), c->mD.p, sizeof(float) * nd, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::vector<float> v;
    for (int p = 0; p < P; p++) {
        const int N = seg_off[p + 1] - seg_off[p];
        float BestMedian = 1.0f;  // feature/src/MapPoint.cpp:284
        int BestIdx = 0;
        for (int i = 0; i < N; i++) {
            const float* row = D.data() + off[(size_t)seg_off[p] + i];
            v.assign(row, row + N);
            v[i] = 0.f;  // :274 (the kernel's own value is sqrt(0) = 0 as well)
            std::sort(v.begin(), v.end());
            const float median = v[(size_t)(0.5 * (double)(N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best[p] = BestIdx;
    }
    return PPG_OK;
}

int ppg_grid_is_in_image(const ppg_grid* g, float x, float y) {
    if (!g) return 0;
    return x >= (float)g->minX && x < (float)g->maxX && y >= (float)g->minY && y < (float)g->maxY;
}

int ppg_check_in_frustum(const ppg_grid* g, const float* K, const float* T, int m, const float* pos, const float* normal,
                         const float* min_dist, const float* max_dist, float limit, uint8_t* in_view, float* proj,
                         float* depth, float* view_cos) {
    if (!g || !K || !T || m < 0 || (m && (!pos || !normal || !min_dist || !max_dist || !in_view || !proj))) return PPG_EINVAL;
    float Ow[3];  // camera centre: -R^T t
    for (int r = 0; r < 3; r++) Ow[r] = -(T[r] * T[9] + T[3 + r] * T[10] + T[6 + r] * T[11]);
    for (int i = 0; i < m; i++) {
        in_view[i] = 0;
        proj[2 * i] = -1.f; proj[2 * i + 1] = -1.f;
        if (depth) depth[i] = -1.f;
        const float* P = pos + 3 * (size_t)i;
        float Pc[3];
        for (int r = 0; r < 3; r++) Pc[r] = T[3 * r] * P[0] + T[3 * r + 1] * P[1] + T[3 * r + 2] * P[2] + T[9 + r];
        if (Pc[2] < 0.0f) continue;
        float u, v;   // mpCamera->project (map/src/Frame.cpp:237)
        if (g->fisheye) {
            // KannalaBrandt8::project(const Eigen::Vector3f&) (sensors/src/KannalaBrandt8.cpp:43-59); the unqualified
            // cos / sin of :57-58 are the double functions
            const float x2_plus_y2 = Pc[0] * Pc[0] + Pc[1] * Pc[1];
            const float theta = atan2f(sqrtf(x2_plus_y2), Pc[2]);
            const float psi = atan2f(Pc[1], Pc[0]);
            const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                        theta9 = theta7 * theta2;
            const float r = theta + g->D[0] * theta3 + g->D[1] * theta5 + g->D[2] * theta7 + g->D[3] * theta9;
            u = (float)((double)(K[0] * r) * std::cos((double)psi) + (double)K[2]);
            v = (float)((double)(K[4] * r) * std::sin((double)psi) + (double)K[5]);
        } else {   // Pinhole::project (sensors/src/Pinhole.cpp:32-38)
            u = K[0] * Pc[0] / Pc[2] + K[2];
            v = K[4] * Pc[1] / Pc[2] + K[5];
        }
        if (!ppg_grid_is_in_image(g, u, v)) continue;
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        if (dist < min_dist[i] || dist > max_dist[i]) continue;
        const float* Pn = normal + 3 * (size_t)i;
        const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
        if (viewCos < limit) continue;
        in_view[i] = 1;
        proj[2 * i] = u; proj[2 * i + 1] = v;
        if (depth) depth[i] = dist;
        if (view_cos) view_cos[i] = viewCos;
    }
    return PPG_OK;
}

int ppg_match_projected(ppg_ctx* c, const ppg_proj_args* a, int32_t* best_idx, float* best_dist, int* n_matches) {
    if (!c) return PPG_EINVAL;
    if (!a || !a->grid || a->n1 < 0 || a->n2 < 0 || a->n2 != a->grid->n || (a->n1 && (!a->desc1 || !a->uv || !a->radius || !best_idx)) ||
        (a->n2 && !a->desc2))
        return fail(c, PPG_EINVAL, "ppg_match_projected: bad argument");
    if (n_matches) *n_matches = 0;
    const int n1 = a->n1, n2 = a->n2, dim = 256;
    for (int i = 0; i < n1; i++) { best_idx[i] = -1; if (best_dist) best_dist[i] = 1e6f; }
    if (n1 == 0 || n2 == 0) return PPG_OK;
    // candidate lists in GetFeaturesInArea order, the reprojection gate applied (it does not depend on the loop state)
    std::vector<int32_t> off(n1 + 1, 0), idx, tmp(n2);
    for (int i = 0; i < n1; i++) {
        if (!a->row_ok || a->row_ok[i]) {
            const float u = a->uv[2 * i], v = a->uv[2 * i + 1];
            const int k = ppg_features_in_area(a->grid, u, v, a->radius[i], tmp.data(), n2);
            for (int t = 0; t < k; t++) {
                if (a->e2_max >= 0.f) {
                    const float ex = u - a->grid->pos[2 * tmp[t]], ey = v - a->grid->pos[2 * tmp[t] + 1];
                    const float e2 = ex * ex + ey * ey;
                    if (e2 > a->e2_max) continue;
                }
                idx.push_back(tmp[t]);
            }
        }
        off[i + 1] = (int32_t)idx.size();
    }
    const size_t nd = idx.size();
    if (nd == 0) return PPG_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    int rc = grow(c, c->mA, (size_t)n1 * dim);
    if (!rc) rc = grow(c, c->mB, (size_t)n2 * dim);
    if (!rc) rc = grow(c, c->mD, nd);
    if (!rc) rc = grow(c, c->mOff, (size_t)n1 + 1);
    if (!rc) rc = grow(c, c->mIdx, nd);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->mA.p, a->desc1, sizeof(float) * n1 * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mB.p, a->desc2, sizeof(float) * n2 * dim, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mOff.p, off.data(), sizeof(int32_t) * (n1 + 1), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->mIdx.p, idx.data(), sizeof(int32_t) * nd, hipMemcpyHostToDevice, s));
    HIPCHK(c, ppg::launch_dist_csr(c->mA.p, n1, c->mB.p, dim, c->mOff.p, c->mIdx.p, c->mD.p, s));
    // the order-dependent pass (skip rules, first argmin, accept, slot claiming) on the device as well: only the
    // per-row results come back
    const bool claims = a->slot_blocked && a->row_claims;
    rc = grow(c, c->mMatch, n1);
    if (!rc) rc = grow(c, c->mBest, n1);
    if (!rc) rc = grow(c, c->mTopI, n1);
    if (!rc) rc = grow(c, c->mClaim, n2);
    if (!rc && a->slot_blocked) rc = grow(c, c->mSkip, n2);
    if (!rc && a->slot_blocked) rc = grow(c, c->mSkipOut, n2);
    if (!rc && claims) rc = grow(c, c->mRowClaims, n1);
    if (rc) return rc;
    if (a->slot_blocked) HIPCHK(c, hipMemcpyAsync(c->mSkip.p, a->slot_blocked, n2, hipMemcpyHostToDevice, s));
    if (claims) HIPCHK(c, hipMemcpyAsync(c->mRowClaims.p, a->row_claims, n1, hipMemcpyHostToDevice, s));
    ppg::ProjMatchArgs pm;
    pm.D = c->mD.p; pm.cand_off = c->mOff.p; pm.cand_idx = c->mIdx.p; pm.n1 = n1; pm.n2 = n2;
    pm.blocked_in = a->slot_blocked ? c->mSkip.p : nullptr; pm.row_claims = claims ? c->mRowClaims.p : nullptr;
    pm.th = a->th; pm.ratio = a->ratio; pm.claim = c->mClaim.p; pm.guess = c->mTopI.p; pm.best_idx = c->mMatch.p;
    pm.best_dist = c->mBest.p; pm.blocked_out = claims ? c->mSkipOut.p : nullptr; pm.n_matches = c->mCount;
    HIPCHK(c, ppg::launch_match_projected(pm, s));
    HIPCHK(c, hipMemcpyAsync(best_idx, c->mMatch.p, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
    if (best_dist) HIPCHK(c, hipMemcpyAsync(best_dist, c->mBest.p, sizeof(float) * n1, hipMemcpyDeviceToHost, s));
    if (claims) HIPCHK(c, hipMemcpyAsync(a->slot_blocked, c->mSkipOut.p, n2, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_mcount, c->mCount, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const int nm = c->h_mcount[0];
    if (n_matches) *n_matches = nm;
    return PPG_OK;
}

int ppg_search_for_initialization(ppg_ctx* c, float nnratio, const float* desc1, int n1, const float* desc2, int n2,
                                  const float* pos2, float* prev, int window, int32_t* m12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    if (n1 < 0 || n2 < 0 || (n1 && (!desc1 || !m12 || !prev)) || (n2 && (!desc2 || !pos2)))
        return fail(c, PPG_EINVAL, "ppg_search_for_initialization: bad argument");
    int rc;
    const float TH_LOW = 0.7f;  // matching/src/Matcher.cpp:13
    if (window > 0) {
        ppg_grid* g = ppg_grid_build(&c->cfg, pos2, n2);
        if (!g) return fail(c, PPG_ENOMEM, "grid allocation failed");
        std::vector<int32_t> off(n1 + 1, 0), idx, tmp(std::max(n2, 1));
        for (int i = 0; i < n1; i++) {
            const int k = ppg_features_in_area(g, prev[2 * i], prev[2 * i + 1], (float)window, tmp.data(), n2);
            idx.insert(idx.end(), tmp.begin(), tmp.begin() + k);
            off[i + 1] = (int32_t)idx.size();
        }
        ppg_grid_free(g);
        if (idx.empty()) idx.push_back(0);
        rc = ppg_match_nn(c, desc1, n1, desc2, n2, 256, off.data(), idx.data(), nullptr, TH_LOW, nnratio, PPG_MATCH_INIT,
                          m12, nullptr, nullptr, n_matches);
    } else {
        rc = ppg_match_nn(c, desc1, n1, desc2, n2, 256, nullptr, nullptr, nullptr, TH_LOW, nnratio, PPG_MATCH_INIT, m12,
                          nullptr, nullptr, n_matches);
    }
    if (rc) return rc;
    for (int i = 0; i < n1; i++)  // Matcher.cpp:646-648
        if (m12[i] >= 0) { prev[2 * i] = pos2[2 * m12[i]]; prev[2 * i + 1] = pos2[2 * m12[i] + 1]; }
    return PPG_OK;
}

static int match_resident(ppg_ctx* c, const float* dA, const int* nA, const float* dB, const int* nB, int batch,
                          float nnratio, int32_t* d_match, int* d_mcounts) {
    const int M = c->maxn;
    const bool fused = M <= 1024 && !c->match_matrix;
    int rc = grow(c, c->mBest, (size_t)M * batch);
    if (!rc) rc = grow(c, c->mSecond, (size_t)M * batch);
    if (!rc) rc = grow(c, c->mTopD, (size_t)M * 32 * batch + 64);
    if (!rc) rc = grow(c, c->mTopI, (size_t)M * 32 * batch + 64);
    if (!rc && fused) rc = grow(c, c->mNearLen, (size_t)M * batch);
    if (!rc && fused) rc = grow(c, c->mListRad, (size_t)M * batch);
    if (!rc && !fused) rc = grow(c, c->mD, (size_t)M * M * batch);
    if (!rc && fused) rc = grow(c, c->mNorm, (size_t)2 * M * batch);
    if (!rc && fused) rc = grow(c, c->mListCnt, (size_t)M * batch);
    if (!rc && fused) rc = grow(c, c->mListCol, (size_t)M * batch * 64);
    if (!rc && fused) rc = grow(c, c->mHalf, (size_t)2 * M * batch * 256);
    if (!rc && fused) rc = grow(c, c->mBt, (size_t)M * batch * 256);
    if (!rc && fused) rc = grow(c, c->mExt, (size_t)batch * 64 * 128);
    if (rc) return rc;
    // the descriptors come from the post stream; the matcher's work arrays belong to the match stream
    hipStream_t s = c->match_stream;
    HIPCHK(c, hipEventRecord(c->ev_desc_ready, c->post));
    HIPCHK(c, hipStreamWaitEvent(s, c->ev_desc_ready, 0));
    ProfScope ps(c, PPG_PROF_MATCH, s);
    if (fused) {
        // descriptors out of desc_sample_kernel have norm 1 (or are all zero), which the prefilter's margin assumes
        ppg::FusedMatchArgs f;
        f.A = dA; f.B = dB; f.sA = (size_t)M * 256; f.sB = (size_t)M * 256; f.nA_dev = nA; f.nB_dev = nB; f.sn = 8;
        f.maxn = M; f.batch = batch; f.th = 0.7f /* TH_LOW, Matcher.cpp:13 */; f.ratio = nnratio;
        f.norm_work = c->mNorm.p; f.half_work = c->mHalf.p; f.bt_work = c->mBt.p; f.ext_work = c->mExt.p; f.list_cnt = c->mListCnt.p; f.list_col = c->mListCol.p;
        f.list_rad = c->mListRad.p; f.near_len = c->mNearLen.p;
        f.near_idx = c->mTopI.p; f.near_dist = c->mTopD.p; f.match12 = d_match; f.best = c->mBest.p; f.second = c->mSecond.p;
        f.n_matches = d_mcounts;
        static const int mskip = getenv("PPG_DEBUG_MATCH_SKIP") ? atoi(getenv("PPG_DEBUG_MATCH_SKIP")) : 0;
        f.debug_skip = mskip;
        static const bool dbg = getenv("PPG_DEBUG_MATCH") != nullptr;   // measurement aid: rounds and rescans to stderr
        if (dbg) { HIPCHK(c, hipMemsetAsync(c->mCount + 2, 0, 8 * sizeof(int), s)); f.stats = c->mCount + 2; }
        HIPCHK(c, ppg::launch_match_fused(f, s));
        if (dbg) {
            int st[8];
            HIPCHK(c, hipMemcpyAsync(st, c->mCount + 2, sizeof(st), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            fprintf(stderr, "[ppg] claim pass: %d problems, %d rounds, %d whole-row rescans; slowest problem: %d rounds, %d rescans, "
                            "%.1f us of which %.1f us in rescans (100 MHz clock)\n", batch, st[0], st[1], st[4], st[5], st[2] * 0.01, st[3] * 0.01);
        }
        return PPG_OK;
    }
    HIPCHK(c, ppg::launch_dist_matrix_batched(dA, nA, M, (size_t)M * 256, dB, nB, M, (size_t)M * 256, 256, c->mD.p, M,
                                              (size_t)M * M, 8, batch, s));
    ppg::MatchArgs m;
    m.D = c->mD.p; m.ldd = M; m.cand_off = nullptr; m.cand_idx = nullptr; m.skipB = nullptr;
    m.nA_dev = nA; m.nB_dev = nB; m.nA = M; m.nB = M; m.th = 0.7f /* TH_LOW, Matcher.cpp:13 */; m.ratio = nnratio;
    m.topk_idx = c->mTopI.p; m.topk_dist = c->mTopD.p; m.claimed = nullptr;
    m.match12 = d_match; m.best = c->mBest.p; m.second = c->mSecond.p; m.n_matches = d_mcounts;
    m.batch = batch; m.bs_D = (size_t)M * M; m.bs_rows = M; m.bs_n = 8;
    HIPCHK(c, ppg::launch_match_init(m, s));
    return PPG_OK;
}

int ppg_match_frames(ppg_ctx* c, int fa, int fb, float nnratio, int32_t* m12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_match_frames");
    Slot& sl = c->slot[c->cur];
    if (any_inflight(c)) return fail(c, PPG_EINVAL, "a submitted batch is pending");
    if (fa < 0 || fb < 0 || fa >= sl.n || fb >= sl.n || !m12)
        return fail(c, PPG_EINVAL, "ppg_match_frames: frames must be in the last extracted batch");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int M = c->maxn;
    int rc = match_resident(c, c->desc + (size_t)fa * M * 256, c->counters + fa * 8 + 3, c->desc + (size_t)fb * M * 256,
                            c->counters + fb * 8 + 3, 1, nnratio, sl.d_match, sl.d_mcounts);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(sl.h_match, sl.d_match, sizeof(int32_t) * M, hipMemcpyDeviceToHost, c->match_stream));
    HIPCHK(c, hipMemcpyAsync(sl.h_mcounts, sl.d_mcounts, sizeof(int), hipMemcpyDeviceToHost, c->match_stream));
    HIPCHK(c, hipStreamSynchronize(c->match_stream));
    if (c->prof) prof_resolve(c);
    memcpy(m12, sl.h_match, sizeof(int32_t) * sl.nkp[fa]);
    if (n_matches) *n_matches = sl.h_mcounts[0];
    return PPG_OK;
}

int ppg_match_prev_batch(ppg_ctx* c, int n, float nnratio, int32_t* m12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_match_prev_batch");
    Slot& sl = c->slot[c->cur];
    const Slot& prev = c->slot[prev_slot(c->cur)];
    if (any_inflight(c)) return fail(c, PPG_EINVAL, "a submitted batch is pending");
    if (n < 1 || n > sl.n || n > prev.n || !m12)
        return fail(c, PPG_EINVAL, "ppg_match_prev_batch: needs two consecutive extract calls of >= n frames");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int M = c->maxn;
    int rc = match_resident(c, c->desc_prev, c->counters_prev + 3, c->desc, c->counters + 3, n, nnratio, sl.d_match, sl.d_mcounts);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(sl.h_match, sl.d_match, sizeof(int32_t) * M * n, hipMemcpyDeviceToHost, c->match_stream));
    HIPCHK(c, hipMemcpyAsync(sl.h_mcounts, sl.d_mcounts, sizeof(int) * n, hipMemcpyDeviceToHost, c->match_stream));
    HIPCHK(c, hipStreamSynchronize(c->match_stream));
    if (c->prof) prof_resolve(c);
    for (int b = 0; b < n; b++) {
        memcpy(m12 + (size_t)b * M, sl.h_match + (size_t)b * M, sizeof(int32_t) * prev.nkp[b]);
        for (int k = prev.nkp[b]; k < M; k++) m12[(size_t)b * M + k] = -1;
        if (n_matches) n_matches[b] = sl.h_mcounts[b];
    }
    return PPG_OK;
}

int ppg_profile(ppg_ctx* c, int enable) {
    if (!c) return PPG_EINVAL;
    c->prof = enable == 2 ? 2 : (enable != 0);
    for (int k = 0; k < PPG_PROF_N; k++) { c->prof_ms[k] = 0; c->prof_launches[k] = 0; c->prof_flops[k] = 0; }
    return PPG_OK;
}

int ppg_tail_stats(ppg_ctx* c, long long* frames, long long* host_frames) {
    if (!c || !frames || !host_frames) return PPG_EINVAL;
    *frames = c->tail_frames;
    *host_frames = c->tail_fallbacks;
    return PPG_OK;
}

int ppg_profile_flops(ppg_ctx* c, double* flops) {
    if (!c || !flops) return PPG_EINVAL;
    for (int k = 0; k < PPG_PROF_N; k++) flops[k] = c->prof_flops[k];
    return PPG_OK;
}

int ppg_profile_read(ppg_ctx* c, double* ms, long long* launches) {
    if (!c || !ms || !launches) return PPG_EINVAL;
    for (int k = 0; k < PPG_PROF_N; k++) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_launches[k]; }
    return PPG_OK;
}

// The walk of Matcher::ExtendMapMatches on the host, every distance taken from one m x n GPU distance matrix: the
// statement the device walk (extend_map_matches_kernel) is checked against, and the path for problems beyond that
// kernel's limits (more than 8192 map points, 4096 key points, 64 edges at a point).
static int extend_map_matches_host(ppg_ctx* c, float nnratio, float th, const ppg_mapgraph* g, const float* desc,
                                   const float* pos_un, int n, const uint32_t* edge_idx, int n_edges, const int32_t* conn_off,
                                   const uint32_t* conn_idx, uint8_t* tracked, int32_t* frame_mp, int32_t* frame_me,
                                   int* n_matches) {
    (void)n_edges;
    const int m = g->m;
    const float TH_HIGH = 0.8f;  // matching/src/Matcher.cpp:12
    // every DescriptorDistance the walk can ask for is an entry of the m x n matrix: one GPU pass
    std::vector<float> D((size_t)m * n);
    int rc = ppg_dist_matrix(c, g->desc, m, desc, n, 256, D.data());
    if (rc) return rc;
    int nmatches = 0;
    std::vector<int> cands;  // Matcher.cpp:210-222
    for (int i = 0; i < m; i++)
        if (!g->bad[i] && g->in_view[i]) cands.push_back(i);
    std::sort(cands.begin(), cands.end(), [g](int a, int b) {
        return (g->edge_off[a + 1] - g->edge_off[a]) > (g->edge_off[b + 1] - g->edge_off[b]);
    });
    ppg_grid* grid = ppg_grid_build(&c->cfg, pos_un, n);
    if (!grid) return fail(c, PPG_ENOMEM, "grid allocation failed");
    std::vector<int32_t> idx(n);
    auto key_other = [&](uint32_t e, uint32_t pid) { return edge_idx[2 * e] == pid ? edge_idx[2 * e + 1] : edge_idx[2 * e]; };
    for (int mp : cands) {
        if (tracked[mp] || g->bad[mp]) continue;  // :227
        float best = 1e6f, best2 = 1e6f;
        int bestIdx = -1;
        float r = th;
        if (g->viewcos[mp] > 0.998) r *= 2.5; else r *= 4.0;  // :240-244
        const int cnt = ppg_features_in_area(grid, g->proj[2 * mp], g->proj[2 * mp + 1], r, idx.data(), n);
        if (cnt == 0) continue;
        for (int k = 0; k < cnt; k++) {  // :251-270
            const int id = idx[k];
            if (frame_mp[id] >= 0 && g->nobs[frame_mp[id]] > 0) continue;
            const float d = D[(size_t)mp * n + id];
            if (d < best) { best2 = best; best = d; bestIdx = id; }
            else if (d < best2) best2 = d;
        }
        if (best > TH_HIGH && best > nnratio * best2) continue;  // :276
        if (bestIdx < 0) continue;  // every candidate of the window was skipped (the reference indexes with -1 here, :279)
        frame_mp[bestIdx] = mp;
        tracked[mp] = 1;
        nmatches++;
        std::deque<uint32_t> seed;  // :287-377
        seed.push_back((uint32_t)bestIdx);
        while (!seed.empty()) {
            const uint32_t keyID = seed.front();
            seed.pop_front();
            const int me0 = g->edge_off[mp], nme = g->edge_off[mp + 1] - me0;  // the map side stays the first seed's point (:296)
            const int ke0 = conn_off[keyID], nke = conn_off[keyID + 1] - ke0;
            if (nme == 0 || nke == 0) continue;
            std::vector<float> weight((size_t)nme * nke, 1e6f);
            std::vector<uint32_t> lx, ly;
            for (int i = 0; i < nme; i++)
                if (g->edge_ok[me0 + i] && g->edge_other[me0 + i] >= 0) lx.push_back(i);
            for (int j = 0; j < nke; j++) ly.push_back(j);
            for (uint32_t i : lx)
                for (uint32_t j : ly) {
                    const int mpo = g->edge_other[me0 + i];
                    const uint32_t ko = key_other(conn_idx[ke0 + j], keyID);
                    weight[(size_t)i * nke + j] = mpo == frame_mp[ko] ? -1.f : D[(size_t)mpo * n + ko];
                }
            while (!lx.empty() && !ly.empty()) {
                uint32_t minlx = 0, minly = 0;
                float minW = 1e6f;
                for (uint32_t i = 0; i < lx.size(); i++)
                    for (uint32_t j = 0; j < ly.size(); j++)
                        if (weight[(size_t)lx[i] * nke + ly[j]] < minW) { minW = weight[(size_t)lx[i] * nke + ly[j]]; minlx = i; minly = j; }
                if (minW > TH_HIGH) break;
                const uint32_t mi = lx[minlx], kj = ly[minly];
                lx.erase(lx.begin() + minlx);
                ly.erase(ly.begin() + minly);
                const int mpo = g->edge_other[me0 + mi];
                const uint32_t keyEdge = conn_idx[ke0 + kj];
                const uint32_t ko = key_other(keyEdge, keyID);
                if (mpo < 0 || g->bad[mpo] || tracked[mpo]) continue;
                frame_mp[ko] = mpo;
                if (frame_me) frame_me[keyEdge] = g->edge_id[me0 + mi];
                tracked[mpo] = 1;
                seed.push_back(ko);
            }
        }
        nmatches++;  // counted twice per accepted seed (:281, :378)
    }
    ppg_grid_free(grid);
    if (n_matches) *n_matches = nmatches;
    return PPG_OK;
}

int ppg_extend_map_matches(ppg_ctx* c, float nnratio, float th, const ppg_mapgraph* g, const float* desc,
                           const float* pos_un, int n, const uint32_t* edge_idx, int n_edges, const int32_t* conn_off,
                           const uint32_t* conn_idx, uint8_t* tracked, int32_t* frame_mp, int32_t* frame_me,
                           int* n_matches) {
    if (!c) return PPG_EINVAL;
    if (!g || g->m < 0 || n < 0 || n_edges < 0 || (n && (!desc || !pos_un || !conn_off || !frame_mp)) ||
        (g->m && (!g->desc || !g->proj || !g->viewcos || !g->in_view || !g->bad || !g->nobs || !g->edge_off || !tracked)))
        return fail(c, PPG_EINVAL, "ppg_extend_map_matches: bad argument");
    if (n_matches) *n_matches = 0;
    const int m = g->m;
    if (m == 0 || n == 0) return PPG_OK;
    {   // every id that is used as an index below
        const int ne = g->edge_off[m];
        if (g->edge_off[0] != 0 || ne < 0 || (ne && (!g->edge_other || !g->edge_ok || !g->edge_id)))
            return fail(c, PPG_EINVAL, "ppg_extend_map_matches: bad edge lists");
        for (int i = 0; i < m; i++)
            if (g->edge_off[i + 1] < g->edge_off[i]) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: edge_off must be non-decreasing");
        for (int k = 0; k < ne; k++)
            if (g->edge_other[k] < -1 || g->edge_other[k] >= m) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: edge_other[%d] out of range", k);
        for (int i = 0; i < n; i++)
            if (frame_mp[i] < -1 || frame_mp[i] >= m) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: frame_mp[%d] out of range", i);
        if (conn_off[0] != 0 || conn_off[n] < 0 || (conn_off[n] && (!conn_idx || !edge_idx)))
            return fail(c, PPG_EINVAL, "ppg_extend_map_matches: bad key-edge lists");
        for (int i = 0; i < n; i++)
            if (conn_off[i + 1] < conn_off[i]) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: conn_off must be non-decreasing");
        for (int k = 0; k < conn_off[n]; k++)
            if ((int)conn_idx[k] >= n_edges) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: conn_idx[%d] out of range", k);
        for (int e = 0; e < 2 * n_edges; e++)
            if ((int)edge_idx[e] >= n) return fail(c, PPG_EINVAL, "ppg_extend_map_matches: edge_idx[%d] out of range", e);
    }
    if (c->extend_on_host || m > 8192 || n > 4096)
        return extend_map_matches_host(c, nnratio, th, g, desc, pos_un, n, edge_idx, n_edges, conn_off, conn_idx, tracked, frame_mp, frame_me, n_matches);
    // ---- device walk ----
    std::vector<int> cands;  // Matcher.cpp:210-222
    for (int i = 0; i < m; i++)
        if (!g->bad[i] && g->in_view[i]) cands.push_back(i);
    std::sort(cands.begin(), cands.end(), [g](int a, int b) {
        return (g->edge_off[a + 1] - g->edge_off[a]) > (g->edge_off[b + 1] - g->edge_off[b]);
    });
    const int ncand = (int)cands.size();
    if (ncand == 0) return PPG_OK;
    // the window of every candidate (:240-247): it depends on the projection only, not on the walk
    ppg_grid* grid = ppg_grid_build(&c->cfg, pos_un, n);
    if (!grid) return fail(c, PPG_ENOMEM, "grid allocation failed");
    std::vector<int32_t> woff(ncand + 1, 0), widx, tmp(n);
    for (int k = 0; k < ncand; k++) {
        const int mp = cands[k];
        float r = th;
        if (g->viewcos[mp] > 0.998) r *= 2.5; else r *= 4.0;
        const int cnt = ppg_features_in_area(grid, g->proj[2 * mp], g->proj[2 * mp + 1], r, tmp.data(), n);
        widx.insert(widx.end(), tmp.begin(), tmp.begin() + cnt);
        woff[k + 1] = (int32_t)widx.size();
    }
    ppg_grid_free(grid);
    const size_t nw = widx.size(), ne = (size_t)g->edge_off[m], nconn = (size_t)conn_off[n];
    // what the walk reads of the two graphs, flattened once: per map point its flags, per map edge the point at its
    // usable other end (:298-304), per listed key edge the key point at its other end (KeyEdge::theOtherPid)
    std::vector<uint8_t> flags(m);
    for (int i = 0; i < m; i++) flags[i] = (uint8_t)((tracked[i] ? 1 : 0) | (g->bad[i] ? 2 : 0) | (g->nobs[i] > 0 ? 4 : 0));
    std::vector<int32_t> eusable(ne), cother(nconn);
    for (size_t k = 0; k < ne; k++) eusable[k] = g->edge_ok[k] && g->edge_other[k] >= 0 ? g->edge_other[k] : -1;
    for (int i = 0; i < n; i++)
        for (int k = conn_off[i]; k < conn_off[i + 1]; k++) {
            const uint32_t e = conn_idx[k];
            cother[k] = (int32_t)(edge_idx[2 * e] == (uint32_t)i ? edge_idx[2 * e + 1] : edge_idx[2 * e]);
        }
    // one packed upload from pinned memory: every array at a 256-byte aligned offset of one block; the arrays that come
    // back (F.mvpMapPoints, F.mvpMapEdges, the tracked flags, the counters) lie together at its end
    struct Part { const void* src; size_t bytes; size_t off; };
    std::vector<Part> parts;
    size_t total = 0;
    auto add = [&](const void* src, size_t bytes) { parts.push_back({src, bytes, total}); total += (bytes + 255) / 256 * 256; return parts.size() - 1; };
    const size_t iMdesc = add(g->desc, sizeof(float) * 256 * m), iFdesc = add(desc, sizeof(float) * 256 * n);
    const size_t iCands = add(cands.data(), sizeof(int32_t) * ncand), iWoff = add(woff.data(), sizeof(int32_t) * (ncand + 1));
    const size_t iWidx = add(widx.data(), sizeof(int32_t) * nw), iFlags = add(flags.data(), m);
    const size_t iEoff = add(g->edge_off, sizeof(int32_t) * (m + 1)), iEid = add(g->edge_id, sizeof(int32_t) * ne);
    const size_t iEus = add(eusable.data(), sizeof(int32_t) * ne), iCoff = add(conn_off, sizeof(int32_t) * (n + 1));
    const size_t iCidx = add(conn_idx, sizeof(uint32_t) * nconn), iCoth = add(cother.data(), sizeof(int32_t) * nconn);
    const size_t iFmp = add(frame_mp, sizeof(int32_t) * n);
    const size_t iFme = add(frame_me, frame_me ? sizeof(int32_t) * n_edges : 0);
    const size_t oUp = total;                                  // end of the upload
    const size_t oTr = total; total += ((size_t)m + 255) / 256 * 256;
    const size_t oOut = total; total += 256;
    const size_t oBack = parts[iFmp].off, nBack = total - oBack;   // what comes back
    const size_t oWdist = total; total += (sizeof(float) * nw + 255) / 256 * 256;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    int rc = grow(c, c->xArena, total);
    if (rc) return rc;
    if (c->xPinCap < oWdist) {
        if (c->xPin) HIPCHK(c, hipHostFree(c->xPin));
        c->xPin = nullptr; c->xPinCap = 0;
        void* q = nullptr;
        HIPCHK(c, hipHostMalloc(&q, oWdist + oWdist / 2, hipHostMallocDefault));
        c->xPin = (uint8_t*)q; c->xPinCap = oWdist + oWdist / 2;
    }
    for (const Part& p : parts)
        if (p.bytes) memcpy(c->xPin + p.off, p.src, p.bytes);
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    uint8_t* base = c->xArena.p;
    HIPCHK(c, hipMemcpyAsync(base, c->xPin, oUp, hipMemcpyHostToDevice, s));
    auto dp = [&](size_t i) { return base + parts[i].off; };
    HIPCHK(c, ppg::launch_dist_csr_rows((const float*)dp(iMdesc), (const int32_t*)dp(iCands), ncand, (const float*)dp(iFdesc),
                                        (const int32_t*)dp(iWoff), (const int32_t*)dp(iWidx), (float*)(base + oWdist), s));
    ppg::ExtendArgs x;
    x.m = m; x.n = n; x.n_edges = n_edges; x.ncand = ncand; x.n_map_edges = (int)ne; x.n_conn = (int)nconn;
    x.cands = (const int32_t*)dp(iCands); x.win_off = (const int32_t*)dp(iWoff); x.win_idx = (const int32_t*)dp(iWidx);
    x.win_dist = (const float*)(base + oWdist); x.mdesc = (const float*)dp(iMdesc); x.fdesc = (const float*)dp(iFdesc);
    x.flags = (const uint8_t*)dp(iFlags); x.edge_off = (const int32_t*)dp(iEoff); x.edge_id = (const int32_t*)dp(iEid);
    x.edge_usable = (const int32_t*)dp(iEus); x.conn_off = (const int32_t*)dp(iCoff); x.conn_idx = (const uint32_t*)dp(iCidx);
    x.conn_other = (const int32_t*)dp(iCoth);
    x.tracked = base + oTr; x.frame_mp = (int32_t*)dp(iFmp); x.frame_me = frame_me ? (int32_t*)dp(iFme) : nullptr;
    x.nnratio = nnratio; x.out = (int*)(base + oOut);
    if ((size_t)m * n <= ((size_t)1 << 24)) {
        // every weight a seed can ask for, evaluated in parallel up front (it stays on the device): the walk then looks a
        // weight up instead of running a 256-step chain inside its sequential loop
        rc = grow(c, c->mD, (size_t)m * n);
        if (rc) return rc;
        HIPCHK(c, ppg::launch_dist_matrix(x.mdesc, m, x.fdesc, n, 256, c->mD.p, n, s));
        x.Dfull = c->mD.p;
    }
    HIPCHK(c, ppg::launch_extend_map_matches(x, s));
    // results land in the pinned block first: a walk that exceeded the kernel's limits is redone on the host from the
    // caller's untouched arrays
    HIPCHK(c, hipMemcpyAsync(c->xPin + oBack, base + oBack, nBack, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const int* out = (const int*)(c->xPin + oOut);
    if (out[1])
        return extend_map_matches_host(c, nnratio, th, g, desc, pos_un, n, edge_idx, n_edges, conn_off, conn_idx, tracked, frame_mp, frame_me, n_matches);
    memcpy(tracked, c->xPin + oTr, m);
    memcpy(frame_mp, c->xPin + parts[iFmp].off, sizeof(int32_t) * n);
    if (frame_me && n_edges) memcpy(frame_me, c->xPin + parts[iFme].off, sizeof(int32_t) * n_edges);
    if (n_matches) *n_matches = out[0];
    return PPG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// DBoW3 vocabulary: loader, tree descent on the GPU, BowVector / FeatureVector assembly
// ------------------------------------------------------------------------------------------
namespace {
std::atomic<unsigned long long> g_vocab_serial{1};

int vocab_to_device(ppg_ctx* c, const ppg_vocab* v) {
    if (c->voc_serial == v->serial) return PPG_OK;
    void* old[] = {c->voc_desc, c->voc_child_off, c->voc_child_idx, c->voc_word, c->voc_weight};
    for (void* p : old)
        if (p) HIPCHK(c, hipFree(p));
    c->voc_desc = nullptr; c->voc_child_off = c->voc_child_idx = c->voc_word = nullptr; c->voc_weight = nullptr;
    c->voc_serial = 0;
    auto up = [&](auto** dst, const auto& vec) -> int {
        void* q = nullptr;
        HIPCHK(c, hipMalloc(&q, std::max<size_t>(vec.size(), 1) * sizeof(vec[0])));
        *dst = (std::remove_reference_t<decltype(**dst)>*)q;
        if (!vec.empty()) HIPCHK(c, hipMemcpy(q, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice));
        return PPG_OK;
    };
    int rc = up(&c->voc_desc, v->desc);
    if (!rc) rc = up(&c->voc_child_off, v->child_off);
    if (!rc) rc = up(&c->voc_child_idx, v->child_idx);
    if (!rc) rc = up(&c->voc_word, v->word_id);
    if (!rc) rc = up(&c->voc_weight, v->weight);
    if (rc) return rc;
    c->voc_serial = v->serial;
    return PPG_OK;
}

int bow_run(ppg_ctx* c, const ppg_vocab* v, const float* d_desc, size_t stride_batch, int n, const int* n_dev, int sn, int batch,
            int levelsup, int32_t* word_id, double* weight, int32_t* node_id) {
    int rc = vocab_to_device(c, v);
    if (!rc) rc = grow(c, c->bWord, (size_t)n * batch);
    if (!rc) rc = grow(c, c->bNode, (size_t)n * batch);
    if (!rc) rc = grow(c, c->bWeight, (size_t)n * batch);
    if (rc) return rc;
    hipStream_t s = c->post;
    HIPCHK(c, hipStreamSynchronize(c->match_stream));   // a pipelined match may still be using the work arrays below
    ppg::BowArgs a;
    a.desc = d_desc; a.stride_batch = stride_batch; a.n = n; a.n_dev = n_dev; a.sn = sn;
    a.node_desc = c->voc_desc; a.child_off = c->voc_child_off; a.child_idx = c->voc_child_idx; a.word_id = c->voc_word;
    a.weight = c->voc_weight; a.dim = v->dim; a.nid_level = v->L - levelsup;
    a.out_word = c->bWord.p; a.out_weight = c->bWeight.p; a.out_node = c->bNode.p;
    HIPCHK(c, ppg::launch_bow_transform(a, batch, s));
    HIPCHK(c, hipMemcpyAsync(word_id, c->bWord.p, sizeof(int32_t) * (size_t)n * batch, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(weight, c->bWeight.p, sizeof(double) * (size_t)n * batch, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(node_id, c->bNode.p, sizeof(int32_t) * (size_t)n * batch, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return PPG_OK;
}
}  // namespace

extern "C" {

ppg_vocab* ppg_vocab_load(const char* path) {
    if (!path) { fail(nullptr, PPG_EINVAL, "ppg_vocab_load: path is NULL"); return nullptr; }
    std::string why;
    ppg_vocab* v = vocab_parse(path, why);
    if (!v) { fail(nullptr, PPG_EIO, "%s", why.c_str()); return nullptr; }
    v->serial = g_vocab_serial.fetch_add(1);
    return v;
}

void ppg_vocab_free(ppg_vocab* v) { delete v; }

int ppg_vocab_info(const ppg_vocab* v, int* k, int* L, int* n_nodes, int* n_words, int* dim, int* scoring, int* weighting) {
    if (!v) return PPG_EINVAL;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    if (dim) *dim = v->dim;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    return PPG_OK;
}

int ppg_vocab_node(const ppg_vocab* v, int node, int* parent, double* weight, int* word_id, int* n_children,
                   const int32_t** children, const float** desc) {
    if (!v || node < 0 || node >= v->n_nodes) return PPG_EINVAL;
    if (parent) *parent = v->parent[node];
    if (weight) *weight = v->weight[node];
    if (word_id) *word_id = v->word_id[node];
    if (n_children) *n_children = v->child_off[node + 1] - v->child_off[node];
    if (children) *children = v->child_idx.data() + v->child_off[node];
    if (desc) *desc = v->desc.data() + (size_t)node * v->dim;
    return PPG_OK;
}

int ppg_bow_transform(ppg_ctx* c, const ppg_vocab* v, const float* desc, int n, int levelsup, int32_t* word_id, double* weight,
                      int32_t* node_id) {
    if (!c) return PPG_EINVAL;
    if (!v || n < 0 || (n && (!desc || !word_id || !weight || !node_id)))
        return fail(c, PPG_EINVAL, "ppg_bow_transform: bad argument");
    if (n == 0) return PPG_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    int rc = grow(c, c->mA, (size_t)n * v->dim);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->mA.p, desc, sizeof(float) * (size_t)n * v->dim, hipMemcpyHostToDevice, c->post));
    return bow_run(c, v, c->mA.p, 0, n, nullptr, 0, 1, levelsup, word_id, weight, node_id);
}

int ppg_bow_transform_batch(ppg_ctx* c, const ppg_vocab* v, int n_frames, int levelsup, int32_t* word_id, double* weight,
                            int32_t* node_id) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_bow_transform_batch");
    const Slot& sl = c->slot[c->cur];
    if (any_inflight(c)) return fail(c, PPG_EINVAL, "a submitted batch is pending");
    if (!v || v->dim != 256 || n_frames < 1 || n_frames > sl.n || !word_id || !weight || !node_id)
        return fail(c, PPG_EINVAL, "ppg_bow_transform_batch: needs a 256-component vocabulary and frames of the last extracted batch");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int M = c->maxn;
    int rc = bow_run(c, v, c->desc, (size_t)M * 256, M, c->counters + 3, 8, n_frames, levelsup, word_id, weight, node_id);
    if (rc) return rc;
    for (int b = 0; b < n_frames; b++)   // rows beyond a frame's key points
        for (int i = sl.nkp[b]; i < M; i++) {
            word_id[(size_t)b * M + i] = -1; weight[(size_t)b * M + i] = 0.0; node_id[(size_t)b * M + i] = -1;
        }
    return PPG_OK;
}

int ppg_bow_vectors(const ppg_vocab* v, const int32_t* word_id, const double* weight, const int32_t* node_id, int n,
                    int32_t* bow_word, double* bow_value, int* n_bow, int32_t* fv_node, int32_t* fv_off, int32_t* fv_idx, int* n_fv) {
    if (!v || n < 0 || !n_bow || !n_fv || !fv_off || (n && (!word_id || !weight || !node_id || !bow_word || !bow_value || !fv_node || !fv_idx)))
        return PPG_EINVAL;
    // Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup): both containers are std::map, i.e. ordered
    // by key; a feature whose word has weight 0 enters neither
    std::map<int32_t, double> bow;
    std::map<int32_t, std::vector<int32_t>> fv;
    const bool tf = v->weighting == 0 || v->weighting == 1;   // TF_IDF, TF: weights add up; IDF, BINARY: first one stays
    for (int i = 0; i < n; i++) {
        if (word_id[i] < 0 || !(weight[i] > 0)) continue;
        auto it = bow.find(word_id[i]);
        if (it == bow.end()) bow[word_id[i]] = weight[i];
        else if (tf) it->second += weight[i];
        fv[node_id[i]].push_back(i);
    }
    // ScoringObject::mustNormalize: L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA -> L1; L2_NORM -> L2; DOT_PRODUCT -> none
    const int norm = v->scoring == 1 ? 2 : (v->scoring == 5 ? 0 : 1);
    if (tf && norm == 0 && !bow.empty()) {
        const double nd = (double)bow.size();
        for (auto& kv : bow) kv.second /= nd;
    }
    if (norm != 0) {   // BowVector::normalize
        double s = 0.0;
        for (auto& kv : bow) s += norm == 1 ? std::fabs(kv.second) : kv.second * kv.second;
        if (norm == 2) s = std::sqrt(s);
        if (s > 0.0)
            for (auto& kv : bow) kv.second /= s;
    }
    int k = 0;
    for (auto& kv : bow) { bow_word[k] = kv.first; bow_value[k] = kv.second; k++; }
    *n_bow = k;
    int q = 0, o = 0;
    fv_off[0] = 0;
    for (auto& kv : fv) {
        fv_node[q] = kv.first;
        for (int32_t i : kv.second) fv_idx[o++] = i;
        fv_off[++q] = o;
    }
    *n_fv = q;
    return PPG_OK;
}

}  // extern "C"
