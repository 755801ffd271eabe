// Multi-GPU mode behind the C ABI: the global-shutter barrier of camera streams sharded over the GPUs of one node, and
// the point-to-point descriptor swap of a stereo pair split over two GPUs, alone (ppg_stereo_swap, host arrays) or together
// with the left-right match on descriptors that are still on the GPU (ppg_stereo_match).  RCCL is used directly (no framework in
// between) and is bound at run time: a host that never creates a shutter needs no RCCL installed, and a process that
// already maps an RCCL image (a framework's bundled copy) keeps using that one image.
//
// The reference is one process bound to cuda:0 (feature/src/PPGExtractor.cpp:41, system/src/Tracking.cpp:51) and has no
// counterpart; what these entry points replace is that hard-coded single device.
#include "../../include/ppgfront.h"
#include "ppg_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enumerators only: every function is looked up with dlsym

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

namespace {

static_assert(sizeof(ncclUniqueId) == PPG_SHUTTER_ID_BYTES, "PPG_SHUTTER_ID_BYTES must be RCCL's unique-id size");

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;
char g_rccl_why[256] = "";

// One RCCL image per process: an image that is already mapped wins (RTLD_NOLOAD matches it by its soname), then
// PPG_RCCL_LIB, then the ROCm installation the library was linked against.
const Rccl* rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return &g_rccl;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h)
        if (const char* p = getenv("PPG_RCCL_LIB")) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* e = dlerror();
        snprintf(g_rccl_why, sizeof(g_rccl_why), "cannot load RCCL (librccl.so.1): %s", e ? e : "unknown error");
        return nullptr;
    }
    Rccl r;
    r.handle = h;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) { ok = false; snprintf(g_rccl_why, sizeof(g_rccl_why), "RCCL lacks %s", name); } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) return nullptr;
    g_rccl = r;
    return &g_rccl;
}

constexpr int kDepth = 8;  // barriers in flight before ppg_shutter_release waits for the oldest

}  // namespace

struct ppg_shutter {
    ppg_ctx* ctx = nullptr;
    const Rccl* r = nullptr;
    int device = 0, rank = 0, world = 1, max_num = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int* d_val = nullptr;              // [kDepth] the counters being reduced
    int *h_in = nullptr, *h_out = nullptr;  // [kDepth] pinned: this rank's group number, the reduced one
    hipEvent_t ev[kDepth] = {};
    long long count = 0;               // barriers started
    int pending = 0;                   // started, result not yet read
    int released = -1;                 // newest group known to be finished on every rank
    // stereo swap: [max_num][256] floats + 64 floats whose first carries the row count
    float *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
    size_t swap_floats = 0;
    // resident left-right match (ppg_stereo_match): a matcher context of the shutter's own, so that the match neither
    // queues behind the pipeline's matches nor shares their work arrays
    ppg_ctx* mctx = nullptr;
    hipEvent_t ev_swap = nullptr;
    int32_t *d_match = nullptr, *h_match = nullptr;
    int *d_mcount = nullptr, *h_mcount = nullptr;
    bool match_pending = false;   // a ppg_stereo_match_start whose result has not been fetched
    int match_rows = 0, match_partner = 0;
};

namespace {

#define SH_HIP(sh, expr)                                                                                              \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) return ppg_set_error((sh)->ctx, PPG_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define SH_NCCL(sh, expr)                                                                                             \
    do {                                                                                                              \
        ncclResult_t e_ = (expr);                                                                                     \
        if (e_ != ncclSuccess) return ppg_set_error((sh)->ctx, PPG_ECOMM, "%s: %s (%s:%d)", #expr, (sh)->r->GetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// reads the results of finished barriers, oldest first; with `block` it waits for the oldest one first
int retire(ppg_shutter* sh, bool block) {
    while (sh->pending > 0) {
        const int slot = (int)((sh->count - sh->pending) % kDepth);
        if (block) SH_HIP(sh, hipEventSynchronize(sh->ev[slot]));
        else {
            const hipError_t q = hipEventQuery(sh->ev[slot]);
            if (q == hipErrorNotReady) break;
            SH_HIP(sh, q);
        }
        if (sh->h_out[slot] > sh->released) sh->released = sh->h_out[slot];
        sh->pending--;
        block = false;
    }
    return PPG_OK;
}

int create_impl(ppg_shutter* sh, const uint8_t* id) {
    SH_HIP(sh, hipSetDevice(sh->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    SH_NCCL(sh, sh->r->CommInitRank(&sh->comm, sh->world, uid, sh->rank));
    int lo = 0, hi = 0;
    SH_HIP(sh, hipDeviceGetStreamPriorityRange(&lo, &hi));
    SH_HIP(sh, hipStreamCreateWithPriority(&sh->stream, hipStreamNonBlocking, hi));
    SH_HIP(sh, hipMalloc((void**)&sh->d_val, sizeof(int) * kDepth));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_in, sizeof(int) * kDepth, hipHostMallocDefault));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_out, sizeof(int) * kDepth, hipHostMallocDefault));
    for (int k = 0; k < kDepth; k++) SH_HIP(sh, hipEventCreateWithFlags(&sh->ev[k], hipEventDisableTiming));
    return PPG_OK;
}

int swap_buffers(ppg_shutter* sh) {
    if (sh->d_send) return PPG_OK;
    sh->swap_floats = (size_t)sh->max_num * 256 + 64;
    const size_t bytes = sh->swap_floats * sizeof(float);
    SH_HIP(sh, hipMalloc((void**)&sh->d_send, bytes));
    SH_HIP(sh, hipMalloc((void**)&sh->d_recv, bytes));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_send, bytes, hipHostMallocDefault));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_recv, bytes, hipHostMallocDefault));
    return PPG_OK;
}

int match_buffers(ppg_shutter* sh) {
    if (sh->mctx) return PPG_OK;
    ppg_config cfg = *ppg_ctx_config(sh->ctx);
    ppg_ctx* m = ppg_create_matcher(&cfg);
    if (!m) return ppg_set_error(sh->ctx, PPG_EHIP, "ppg_stereo_match: %s", ppg_last_error(nullptr));
    sh->mctx = m;
    // a caller waits for this match, so its kernels go to a stream of the highest priority (on the matcher contexts' usual
    // low-priority stream a call lasts 1 ms beside a running pipeline instead of 0.2 ms)
    if (ppg_ctx_raise_match_stream(m) != PPG_OK)
        return ppg_set_error(sh->ctx, PPG_EHIP, "ppg_stereo_match: %s", ppg_last_error(m));
    SH_HIP(sh, hipEventCreateWithFlags(&sh->ev_swap, hipEventDisableTiming));
    SH_HIP(sh, hipMalloc((void**)&sh->d_match, sizeof(int32_t) * sh->max_num));
    SH_HIP(sh, hipMalloc((void**)&sh->d_mcount, sizeof(int) * 8));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_match, sizeof(int32_t) * sh->max_num, hipHostMallocDefault));
    SH_HIP(sh, hipHostMalloc((void**)&sh->h_mcount, sizeof(int) * 8, hipHostMallocDefault));
    return PPG_OK;
}

// one fixed-size message each way on the shutter's stream: [max_num][256] floats, then 64 floats whose first holds the row count
int exchange(ppg_shutter* sh, int partner) {
    SH_NCCL(sh, sh->r->GroupStart());
    const ncclResult_t e1 = sh->r->Send(sh->d_send, sh->swap_floats, ncclFloat, partner, sh->comm, sh->stream);
    const ncclResult_t e2 = sh->r->Recv(sh->d_recv, sh->swap_floats, ncclFloat, partner, sh->comm, sh->stream);
    SH_NCCL(sh, sh->r->GroupEnd());
    SH_NCCL(sh, e1);
    SH_NCCL(sh, e2);
    return PPG_OK;
}

}  // namespace

extern "C" {

int ppg_shutter_unique_id(uint8_t* id) {
    if (!id) return ppg_set_error(nullptr, PPG_EINVAL, "ppg_shutter_unique_id: id is NULL");
    const Rccl* r = rccl();
    if (!r) return ppg_set_error(nullptr, PPG_ECOMM, "%s", g_rccl_why);
    ncclUniqueId uid;
    const ncclResult_t e = r->GetUniqueId(&uid);
    if (e != ncclSuccess) return ppg_set_error(nullptr, PPG_ECOMM, "ncclGetUniqueId: %s", r->GetErrorString(e));
    memcpy(id, &uid, sizeof(uid));
    return PPG_OK;
}

ppg_shutter* ppg_shutter_create(ppg_ctx* ctx, int rank, int world, const uint8_t* id) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) {
        ppg_set_error(nullptr, PPG_EINVAL, "ppg_shutter_create: bad argument (rank %d of %d)", rank, world);
        return nullptr;
    }
    const Rccl* r = rccl();
    if (!r) { ppg_set_error(nullptr, PPG_ECOMM, "%s", g_rccl_why); return nullptr; }
    ppg_shutter* sh = new (std::nothrow) ppg_shutter();
    if (!sh) { ppg_set_error(nullptr, PPG_ENOMEM, "out of host memory"); return nullptr; }
    sh->ctx = ctx; sh->r = r; sh->rank = rank; sh->world = world;
    sh->device = ppg_ctx_device(ctx);
    sh->max_num = ppg_ctx_max_num(ctx);
    if (create_impl(sh, id) != PPG_OK) {
        ppg_set_error(nullptr, PPG_ECOMM, "%s", ppg_last_error(ctx));
        ppg_shutter_destroy(sh);
        return nullptr;
    }
    return sh;
}

void ppg_shutter_destroy(ppg_shutter* sh) {
    if (!sh) return;
    (void)hipSetDevice(sh->device);
    if (sh->stream) (void)hipStreamSynchronize(sh->stream);
    if (sh->comm) (void)sh->r->CommDestroy(sh->comm);
    for (hipEvent_t e : sh->ev)
        if (e) (void)hipEventDestroy(e);
    if (sh->d_val) (void)hipFree(sh->d_val);
    if (sh->h_in) (void)hipHostFree(sh->h_in);
    if (sh->h_out) (void)hipHostFree(sh->h_out);
    if (sh->mctx) ppg_destroy(sh->mctx);
    if (sh->ev_swap) (void)hipEventDestroy(sh->ev_swap);
    if (sh->d_match) (void)hipFree(sh->d_match);
    if (sh->d_mcount) (void)hipFree(sh->d_mcount);
    if (sh->h_match) (void)hipHostFree(sh->h_match);
    if (sh->h_mcount) (void)hipHostFree(sh->h_mcount);
    if (sh->d_send) (void)hipFree(sh->d_send);
    if (sh->d_recv) (void)hipFree(sh->d_recv);
    if (sh->h_send) (void)hipHostFree(sh->h_send);
    if (sh->h_recv) (void)hipHostFree(sh->h_recv);
    if (sh->stream) (void)hipStreamDestroy(sh->stream);
    delete sh;
}

int ppg_shutter_release(ppg_shutter* sh, int group, int* released) {
    if (!sh) return PPG_EINVAL;
    SH_HIP(sh, hipSetDevice(sh->device));
    int rc = PPG_OK;
    if (sh->pending >= kDepth) rc = retire(sh, true);  // the ring is full: wait for the oldest barrier only
    if (rc) return rc;
    const int slot = (int)(sh->count % kDepth);
    sh->h_in[slot] = group;
    SH_HIP(sh, hipMemcpyAsync(sh->d_val + slot, sh->h_in + slot, sizeof(int), hipMemcpyHostToDevice, sh->stream));
    SH_NCCL(sh, sh->r->AllReduce(sh->d_val + slot, sh->d_val + slot, 1, ncclInt32, ncclMin, sh->comm, sh->stream));
    SH_HIP(sh, hipMemcpyAsync(sh->h_out + slot, sh->d_val + slot, sizeof(int), hipMemcpyDeviceToHost, sh->stream));
    SH_HIP(sh, hipEventRecord(sh->ev[slot], sh->stream));
    sh->count++;
    sh->pending++;
    rc = retire(sh, false);
    if (released) *released = sh->released;
    return rc;
}

int ppg_shutter_poll(ppg_shutter* sh, int* released) {
    if (!sh) return PPG_EINVAL;
    SH_HIP(sh, hipSetDevice(sh->device));
    const int rc = retire(sh, false);
    if (released) *released = sh->released;
    return rc;
}

int ppg_shutter_drain(ppg_shutter* sh, int* released) {
    if (!sh) return PPG_EINVAL;
    SH_HIP(sh, hipSetDevice(sh->device));
    int rc = PPG_OK;
    while (!rc && sh->pending > 0) rc = retire(sh, true);
    if (released) *released = sh->released;
    return rc;
}

int ppg_shutter_stats(const ppg_shutter* sh, long long* started, int* pending) {
    if (!sh) return PPG_EINVAL;
    if (started) *started = sh->count;
    if (pending) *pending = sh->pending;
    return PPG_OK;
}

int ppg_stereo_swap(ppg_shutter* sh, int partner, const float* desc, int n, float* partner_desc, int* n_partner) {
    if (!sh) return PPG_EINVAL;
    if (partner < 0 || partner >= sh->world || n < 0 || n > sh->max_num || (n && !desc) || !partner_desc || !n_partner)
        return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_swap: bad argument (partner %d of %d ranks, %d rows, at most %d)",
                             partner, sh->world, n, sh->max_num);
    if (sh->match_pending)
        return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_swap: a started match still uses the swap buffers (ppg_stereo_match_finish)");
    SH_HIP(sh, hipSetDevice(sh->device));
    int rc = swap_buffers(sh);
    if (rc) return rc;
    // one fixed-size message each way: the rows, then the count in the tail
    const size_t rows = (size_t)sh->max_num * 256;
    if (n) memcpy(sh->h_send, desc, sizeof(float) * 256 * (size_t)n);
    int cnt = n;
    memcpy(sh->h_send + rows, &cnt, sizeof(int));
    if (n) SH_HIP(sh, hipMemcpyAsync(sh->d_send, sh->h_send, sizeof(float) * 256 * (size_t)n, hipMemcpyHostToDevice, sh->stream));
    SH_HIP(sh, hipMemcpyAsync(sh->d_send + rows, sh->h_send + rows, sizeof(float) * 64, hipMemcpyHostToDevice, sh->stream));
    rc = exchange(sh, partner);
    if (rc) return rc;
    SH_HIP(sh, hipMemcpyAsync(sh->h_recv, sh->d_recv, sizeof(float) * sh->swap_floats, hipMemcpyDeviceToHost, sh->stream));
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    int m = 0;
    memcpy(&m, sh->h_recv + rows, sizeof(int));
    if (m < 0 || m > sh->max_num)
        return ppg_set_error(sh->ctx, PPG_ECOMM, "ppg_stereo_swap: partner %d announced %d rows (its junction_max_num must equal this side's %d)",
                             partner, m, sh->max_num);
    if (m) memcpy(partner_desc, sh->h_recv, sizeof(float) * 256 * (size_t)m);
    *n_partner = m;
    return PPG_OK;
}

int ppg_stereo_match_start(ppg_shutter* sh, int partner, int frame, float nnratio) {
    if (!sh) return PPG_EINVAL;
    if (partner < 0 || partner >= sh->world)
        return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_match: bad argument (partner %d of %d ranks)", partner, sh->world);
    if (sh->match_pending)
        return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_match_start: the previous match was not fetched (ppg_stereo_match_finish)");
    SH_HIP(sh, hipSetDevice(sh->device));
    const float* d_desc = nullptr;
    int n = 0;
    int rc = ppg_ctx_last_desc(sh->ctx, frame, &d_desc, &n);
    if (!rc) rc = swap_buffers(sh);
    if (!rc) rc = match_buffers(sh);
    if (rc) return rc;
    const size_t rows = (size_t)sh->max_num * 256;
    // this side's rows never leave the GPU: slot -> send buffer, the count behind them
    if (n) SH_HIP(sh, hipMemcpyAsync(sh->d_send, d_desc, sizeof(float) * 256 * (size_t)n, hipMemcpyDeviceToDevice, sh->stream));
    memcpy(sh->h_send + rows, &n, sizeof(int));
    SH_HIP(sh, hipMemcpyAsync(sh->d_send + rows, sh->h_send + rows, sizeof(float) * 64, hipMemcpyHostToDevice, sh->stream));
    rc = exchange(sh, partner);
    if (rc) return rc;
    SH_HIP(sh, hipMemcpyAsync(sh->h_recv + rows, sh->d_recv + rows, sizeof(float) * 64, hipMemcpyDeviceToHost, sh->stream));
    SH_HIP(sh, hipEventRecord(sh->ev_swap, sh->stream));
    // Matcher::SearchForInitialization with a window that holds every feature (matching/src/Matcher.cpp:582-651), left rows =
    // this side's, on the matcher context's stream behind the swap; the counts are the ints in the two message tails
    hipStream_t ms = (hipStream_t)ppg_ctx_match_stream(sh->mctx);
    SH_HIP(sh, hipStreamWaitEvent(ms, sh->ev_swap, 0));
    rc = match_resident(sh->mctx, sh->d_send, (const int*)(sh->d_send + rows), sh->d_recv, (const int*)(sh->d_recv + rows), 1,
                        nnratio, sh->d_match, sh->d_mcount);
    if (rc) return ppg_set_error(sh->ctx, rc, "ppg_stereo_match: %s", ppg_last_error(sh->mctx));
    SH_HIP(sh, hipMemcpyAsync(sh->h_match, sh->d_match, sizeof(int32_t) * sh->max_num, hipMemcpyDeviceToHost, ms));
    SH_HIP(sh, hipMemcpyAsync(sh->h_mcount, sh->d_mcount, sizeof(int), hipMemcpyDeviceToHost, ms));
    sh->match_pending = true;
    sh->match_rows = n;
    sh->match_partner = partner;
    return PPG_OK;
}

int ppg_stereo_match_finish(ppg_shutter* sh, int32_t* matches12, int* n_matches, int* n_partner) {
    if (!sh) return PPG_EINVAL;
    if (!matches12) return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_match_finish: matches12 is NULL");
    if (!sh->match_pending) return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_match_finish: no match was started");
    SH_HIP(sh, hipSetDevice(sh->device));
    sh->match_pending = false;
    SH_HIP(sh, hipStreamSynchronize(sh->stream));
    SH_HIP(sh, hipStreamSynchronize((hipStream_t)ppg_ctx_match_stream(sh->mctx)));
    const size_t rows = (size_t)sh->max_num * 256;
    int m = 0;
    memcpy(&m, sh->h_recv + rows, sizeof(int));
    if (m < 0 || m > sh->max_num)
        return ppg_set_error(sh->ctx, PPG_ECOMM, "ppg_stereo_match: partner %d announced %d rows (its junction_max_num must equal this side's %d)",
                             sh->match_partner, m, sh->max_num);
    for (int i = 0; i < sh->max_num; i++) matches12[i] = i < sh->match_rows ? sh->h_match[i] : -1;
    if (n_matches) *n_matches = sh->h_mcount[0];
    if (n_partner) *n_partner = m;
    return PPG_OK;
}

int ppg_stereo_match(ppg_shutter* sh, int partner, int frame, float nnratio, int32_t* matches12, int* n_matches, int* n_partner) {
    if (!sh) return PPG_EINVAL;
    if (!matches12) return ppg_set_error(sh->ctx, PPG_EINVAL, "ppg_stereo_match: matches12 is NULL");
    const int rc = ppg_stereo_match_start(sh, partner, frame, nnratio);
    return rc ? rc : ppg_stereo_match_finish(sh, matches12, n_matches, n_partner);
}

}  // extern "C"
