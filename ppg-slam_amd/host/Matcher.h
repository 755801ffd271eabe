// Host-side mirror of the reference's Matcher (matching/include/Matcher.h:20-64) and of
// DescriptorDistance (feature/include/MapPoint.h:25) over the C ABI of libppgfront.
// Header-only; link with -lppgfront.
//
// The reference's Frame / MapPoint / MapEdge are pointer graphs guarded by per-object mutexes.  The
// mirror takes the same information as flat views: the caller gathers it under the reference's locks
// (MapPoint::GetDescriptor / isBad / getEdges, feature/src/MapPoint.cpp:203-210, 304-308, 421-425),
// calls in, and scatters the returned ids back into F.mvpMapPoints / F.mvpMapEdges.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ppgfront.h"
#include "ppg_types.h"

namespace ppg_host {

// The part of Frame (map/include/Frame.h:42) that the matchers read.
struct FrameView {
    std::vector<KeyPointEx> mvKeysUn;
    std::vector<KeyEdge> mvKeyEdges;
    DescMat mDescriptors;
    std::vector<int> mvpMapPoints;  // map-point id per key point, -1 = nullptr
    std::vector<int> mvpMapEdges;   // map-edge id per key edge, -1 = nullptr
};

// Flattened local map for ExtendMapMatches: one entry per MapPoint* of vpMapPoints, in order.
struct MapView {
    DescMat desc;                       // GetDescriptor()
    std::vector<float> proj;            // mTrackProjX, mTrackProjY
    std::vector<float> viewcos;         // mTrackViewCos
    std::vector<uint8_t> in_view, bad;  // mbTrackInView, isBad()
    std::vector<int32_t> nobs;          // Observations()
    std::vector<uint8_t> tracked;       // mnTrackedbyFrame == F.mnId (in/out)
    std::vector<int32_t> edge_off;      // CSR over getEdges()
    std::vector<int32_t> edge_id, edge_other;
    std::vector<uint8_t> edge_ok;       // !isBad() && mbValid
};

// Sophus::SE3f as the matchers use it: camera-from-world rotation (row-major) and translation.  "T * p" is the
// row-major product with left-to-right sums (the reference's Eigen code is vectorised; the order is a definition).
struct SE3f {
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float t[3] = {0, 0, 0};
    void apply(const float* p, float* o) const {
        for (int r = 0; r < 3; r++) o[r] = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2] + t[r];
    }
    void center(float* ow) const {  // inverse().translation()
        for (int r = 0; r < 3; r++) ow[r] = -(R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
    }
};

// What the projection matchers read from the MapPoint objects (feature/include/MapPoint.h), one entry per point.
struct MapPointsView {
    std::vector<float> pos, normal;           // GetWorldPos(), GetNormal()  (3 per point)
    std::vector<float> min_dist, max_dist;    // GetMin/MaxDistanceInvariance()
    DescMat desc;                             // GetDescriptor()
    std::vector<uint8_t> bad;                 // isBad()
    std::vector<int32_t> nobs;                // Observations()
    int size() const { return (int)bad.size(); }
};

// feature/src/MapPoint.cpp:22-29.  One pair per call launches a 1x1 distance matrix on the GPU; batch
// pairs with ppg_dist_matrix wherever the caller has more than one.
inline float DescriptorDistance(ppg_ctx* ctx, const DescMat& a, const DescMat& b) {
    if (a.cols != b.cols) throw std::invalid_argument("DescriptorDistance: column mismatch");
    float d = 0.f;
    if (ppg_dist_matrix(ctx, a.data.data(), 1, b.data.data(), 1, a.cols, &d) != PPG_OK)
        throw std::runtime_error(std::string("libppgfront: ") + ppg_last_error(ctx));
    return d;
}

class Matcher {
public:
    // matching/include/Matcher.h:24; the context supplies the GPU (one per calling thread)
    Matcher(ppg_ctx* ctx, const CameraModel* pCam, float nnratio = 0.6f) : mCtx(ctx), mpCamera(pCam), mfNNratio(nnratio) {}

    // matching/src/Matcher.cpp:582-651
    int SearchForInitialization(FrameView& F1, FrameView& F2, std::vector<Point2f>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10) {
        const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
        vnMatches12.assign(n1, -1);
        if (n1 == 0) return 0;
        std::vector<float> pos2(2 * (size_t)n2), prev(2 * (size_t)n1);
        for (int i = 0; i < n2; i++) { pos2[2 * i] = F2.mvKeysUn[i].mPos[0]; pos2[2 * i + 1] = F2.mvKeysUn[i].mPos[1]; }
        for (int i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        int nmatches = 0;
        check(ppg_search_for_initialization(mCtx, mfNNratio, F1.mDescriptors.data.data(), n1, F2.mDescriptors.data.data(), n2,
                                            pos2.data(), prev.data(), windowSize, vnMatches12.data(), &nmatches));
        for (int i = 0; i < n1; i++) vbPrevMatched[i] = Point2f(prev[2 * i], prev[2 * i + 1]);
        return nmatches;
    }

    // matching/src/Matcher.cpp:203-381
    int ExtendMapMatches(FrameView& F, MapView& map, const float th) {
        const int n = (int)F.mvKeysUn.size(), ne = (int)F.mvKeyEdges.size(), m = map.desc.rows;
        std::vector<float> pos(2 * (size_t)n);
        std::vector<uint32_t> eidx(2 * (size_t)ne), cidx;
        std::vector<int32_t> coff(n + 1, 0);
        for (int i = 0; i < n; i++) {
            pos[2 * i] = F.mvKeysUn[i].mPos[0]; pos[2 * i + 1] = F.mvKeysUn[i].mPos[1];
            for (unsigned e : F.mvKeysUn[i].mvConnected) cidx.push_back(e);
            coff[i + 1] = (int32_t)cidx.size();
        }
        for (int e = 0; e < ne; e++) { eidx[2 * e] = F.mvKeyEdges[e].startIdx; eidx[2 * e + 1] = F.mvKeyEdges[e].endIdx; }
        if (cidx.empty()) cidx.push_back(0);
        F.mvpMapPoints.resize(n, -1);
        F.mvpMapEdges.resize(ne, -1);
        ppg_mapgraph g;
        g.m = m; g.desc = map.desc.data.data(); g.proj = map.proj.data(); g.viewcos = map.viewcos.data();
        g.in_view = map.in_view.data(); g.bad = map.bad.data(); g.nobs = map.nobs.data(); g.edge_off = map.edge_off.data();
        g.edge_id = map.edge_id.data(); g.edge_other = map.edge_other.data(); g.edge_ok = map.edge_ok.data();
        int nmatches = 0;
        check(ppg_extend_map_matches(mCtx, mfNNratio, th, &g, F.mDescriptors.data.data(), pos.data(), n, eidx.data(), ne,
                                     coff.data(), cidx.data(), map.tracked.data(), F.mvpMapPoints.data(),
                                     F.mvpMapEdges.data(), &nmatches));
        return nmatches;
    }

    // matching/src/Matcher.cpp:31-87.  Last: mvpMapPoints + mvbOutlier; Cur.mvpMapPoints is updated.
    int SearchByProjection(FrameView& Cur, const FrameView& Last, const std::vector<uint8_t>& lastOutlier,
                           const SE3f& Tcw, const MapPointsView& map, const float th) {
        const int nl = (int)Last.mvKeysUn.size(), nc = (int)Cur.mvKeysUn.size();
        Cur.mvpMapPoints.resize(nc, -1);
        Grid grid(mpCamera, Cur);
        std::vector<float> uv(2 * (size_t)nl, 0.f), radius(nl, th);
        std::vector<uint8_t> ok(nl, 0), claims(nl, 0), blocked(nc, 0);
        DescMat d1(nl, 256);
        for (int i = 0; i < nl; i++) {
            const int id = i < (int)Last.mvpMapPoints.size() ? Last.mvpMapPoints[i] : -1;
            if (id < 0 || lastOutlier[i]) continue;
            float pc[3];
            Tcw.apply(&map.pos[3 * (size_t)id], pc);
            const float invzc = (float)(1.0 / (double)pc[2]);
            if (invzc < 0) continue;
            project(pc, &uv[2 * i]);
            if (!ppg_grid_is_in_image(grid.g, uv[2 * i], uv[2 * i + 1])) continue;
            ok[i] = 1;
            claims[i] = map.nobs[id] > 0;  // :67-69 sees the slots this loop has filled
            std::memcpy(d1.ptr(i), map.desc.ptr(id), sizeof(float) * 256);
        }
        for (int j = 0; j < nc; j++) blocked[j] = Cur.mvpMapPoints[j] >= 0 && map.nobs[Cur.mvpMapPoints[j]] > 0;
        std::vector<int32_t> best(nl, -1);
        int nmatches = 0;
        run(grid, Cur.mDescriptors, d1, uv, radius, ok, &claims, &blocked, TH_HIGH, 0.f, -1.f, best, nmatches);
        for (int i = 0; i < nl; i++)
            if (best[i] >= 0) Cur.mvpMapPoints[best[i]] = Last.mvpMapPoints[i];  // in loop order: later rows overwrite (:80)
        return nmatches;
    }

    // matching/src/Matcher.cpp:479-568.  Tcw = SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale()) (:483);
    // vpMatched holds map-point ids or -1.
    int SearchByProjection(const FrameView& KF, const SE3f& Tcw, const MapPointsView& map, const std::vector<int>& vpPoints,
                           std::vector<int>& vpMatched, int th, float ratioHamming) {
        const int np = (int)vpPoints.size(), nk = (int)KF.mvKeysUn.size();
        vpMatched.resize(nk, -1);
        Grid grid(mpCamera, KF);
        float Ow[3];
        Tcw.center(Ow);
        std::vector<uint8_t> found(map.size(), 0);  // spAlreadyFound (:487-488), a snapshot
        for (int id : vpMatched)
            if (id >= 0) found[id] = 1;
        std::vector<float> uv(2 * (size_t)np, 0.f), radius(np, (float)th);
        std::vector<uint8_t> ok(np, 0), claims(np, 1), blocked(nk, 0);
        DescMat d1(np, 256);
        for (int i = 0; i < np; i++) {
            const int id = vpPoints[i];
            if (map.bad[id] || found[id]) continue;
            if (!gates(grid, Tcw, Ow, map, id, &uv[2 * i])) continue;
            ok[i] = 1;
            std::memcpy(d1.ptr(i), map.desc.ptr(id), sizeof(float) * 256);
        }
        for (int j = 0; j < nk; j++) blocked[j] = vpMatched[j] >= 0;
        std::vector<int32_t> best(np, -1);
        int nmatches = 0;
        run(grid, KF.mDescriptors, d1, uv, radius, ok, &claims, &blocked, TH_LOW * ratioHamming, 0.f, -1.f, best, nmatches);
        for (int i = 0; i < np; i++)
            if (best[i] >= 0) vpMatched[best[i]] = vpPoints[i];
        return nmatches;
    }

    // The matching part of Fuse(KeyFrame*, vpMapPoints, th) (matching/src/Matcher.cpp:897-1036, withReprojGate = true)
    // and of Fuse(KeyFrame*, Sim3f&, vpPoints, th, vpReplacePoint) (:1038-1135, false).  inKF[i]: IsInKeyFrame(pKF) /
    // spAlreadyFound.count(pMP); vpPoints may hold -1 (nullptr).  bestIdx[i] = the key point to fuse with or -1; the
    // caller performs Replace / AddObservation / vpReplacePoint on its map (:1013-1028, :1119-1129).
    int Fuse(const FrameView& KF, const SE3f& Tcw, const MapPointsView& map, const std::vector<int>& vpPoints,
             const std::vector<uint8_t>& inKF, const float th, bool withReprojGate, std::vector<int>& bestIdx) {
        const int np = (int)vpPoints.size();
        Grid grid(mpCamera, KF);
        float Ow[3];
        Tcw.center(Ow);
        std::vector<float> uv(2 * (size_t)np, 0.f), radius(np, th);
        std::vector<uint8_t> ok(np, 0);
        DescMat d1(np, 256);
        for (int i = 0; i < np; i++) {
            const int id = vpPoints[i];
            if (id < 0 || map.bad[id] || inKF[i]) continue;
            if (!gates(grid, Tcw, Ow, map, id, &uv[2 * i])) continue;
            ok[i] = 1;
            std::memcpy(d1.ptr(i), map.desc.ptr(id), sizeof(float) * 256);
        }
        std::vector<int32_t> best(np, -1);
        int nFused = 0;
        run(grid, KF.mDescriptors, d1, uv, radius, ok, nullptr, nullptr, TH_LOW, 0.f, withReprojGate ? 5.99f : -1.f, best, nFused);
        bestIdx.assign(best.begin(), best.end());
        return nFused;
    }

public:
    static constexpr float TH_LOW = 0.7f;   // matching/src/Matcher.cpp:13
    static constexpr float TH_HIGH = 0.8f;  // matching/src/Matcher.cpp:12
    ppg_ctx* mCtx;
    const CameraModel* mpCamera;
    float mfNNratio;

private:
    // Frame::AssignFeaturesToGrid over mvKeysUn (map/src/Frame.cpp:138-156), freed with the scope
    struct Grid {
        ppg_grid* g = nullptr;
        Grid(const CameraModel* cam, const FrameView& F) {
            ppg_config c;
            ppg_default_config(&c, cam->width, cam->height);
            for (int k = 0; k < 9; k++) c.K[k] = cam->K[k];
            for (int k = 0; k < 4; k++) c.D[k] = cam->D[k];
            c.fisheye = cam->mnType == CameraModel::CAM_FISHEYE;
            std::vector<float> pos(2 * F.mvKeysUn.size());
            for (size_t i = 0; i < F.mvKeysUn.size(); i++) { pos[2 * i] = F.mvKeysUn[i].mPos[0]; pos[2 * i + 1] = F.mvKeysUn[i].mPos[1]; }
            g = ppg_grid_build(&c, pos.data(), (int)F.mvKeysUn.size());
            if (!g) throw std::runtime_error("libppgfront: grid allocation failed");
        }
        ~Grid() { ppg_grid_free(g); }
        Grid(const Grid&) = delete;
        Grid& operator=(const Grid&) = delete;
    };
    // Pinhole::project (sensors/src/Pinhole.cpp:32-38)
    void project(const float* p, float* uv) const {
        uv[0] = mpCamera->K[0] * p[0] / p[2] + mpCamera->K[2];
        uv[1] = mpCamera->K[4] * p[1] / p[2] + mpCamera->K[5];
    }
    // depth, image bounds, scale range, viewing angle: matching/src/Matcher.cpp:501-531 = :931-969 = :1060-1090
    bool gates(const Grid& grid, const SE3f& Tcw, const float* Ow, const MapPointsView& map, int id, float* uv) const {
        const float* pw = &map.pos[3 * (size_t)id];
        float pc[3];
        Tcw.apply(pw, pc);
        if (pc[2] < 0.0f) return false;
        project(pc, uv);
        if (!ppg_grid_is_in_image(grid.g, uv[0], uv[1])) return false;
        const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
        const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        if (dist < map.min_dist[id] || dist > map.max_dist[id]) return false;
        const float* Pn = &map.normal[3 * (size_t)id];
        const float dot = PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2];
        return !((double)dot < 0.5 * (double)dist);
    }
    void run(const Grid& grid, const DescMat& desc2, const DescMat& d1, const std::vector<float>& uv, const std::vector<float>& radius,
             const std::vector<uint8_t>& ok, const std::vector<uint8_t>* claims, std::vector<uint8_t>* blocked, float th,
             float ratio, float e2max, std::vector<int32_t>& best, int& nmatches) {
        ppg_proj_args a;
        a.grid = grid.g; a.desc2 = desc2.data.data(); a.n2 = desc2.rows; a.desc1 = d1.data.data(); a.n1 = d1.rows;
        a.uv = uv.data(); a.radius = radius.data(); a.row_ok = ok.data();
        a.row_claims = claims ? claims->data() : nullptr; a.slot_blocked = blocked ? blocked->data() : nullptr;
        a.th = th; a.ratio = ratio; a.e2_max = e2max;
        check(ppg_match_projected(mCtx, &a, best.data(), nullptr, &nmatches));
    }
    void check(int rc) {
        if (rc != PPG_OK) throw std::runtime_error(std::string("libppgfront: ") + ppg_last_error(mCtx));
    }
};

}  // namespace ppg_host
