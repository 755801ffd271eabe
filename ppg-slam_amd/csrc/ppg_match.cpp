// Host side of libppgfront, second half: DescriptorDistance and the distance matrix, the matcher entry points, the frame
// grid and the frustum check, the point-pair-graph matcher, the DBoW3 vocabulary.  C ABI in include/ppgfront.h.
#include "ppg_ctx.h"

namespace {

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

}  // namespace

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

extern "C" {

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

int match_resident(ppg_ctx* c, const float* dA, const int* nA, const float* dB, const int* nB, int batch,
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
