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

// DBoW3::FeatureVector (a std::map<NodeId, std::vector<unsigned>>) as the matchers walk it: entries sorted by node id.
// With the shipped vocabularies (k = 9, L = 3) and the reference's levelsup = 4 (map/src/Frame.cpp:131) every feature
// is filed under the root, i.e. one entry {0, [0 .. N-1]}: all_in_root().
struct FeatureVectorView {
    std::vector<std::pair<unsigned, std::vector<unsigned>>> nodes;
    static FeatureVectorView all_in_root(int n) {
        FeatureVectorView v;
        v.nodes.emplace_back(0u, std::vector<unsigned>());
        for (int i = 0; i < n; i++) v.nodes[0].second.push_back((unsigned)i);
        return v;
    }
};

// feature/src/MapPoint.cpp:22-29 (also called in a loop from MapPoint.cpp:277).  Host arithmetic only: the same
// fused-multiply-add chain as every kernel of the library, bit for bit, with no launch, no context and no effect on
// batches in flight.  Callers with many pairs batch them with ppg_dist_matrix / ppg_dist_lists.
inline float DescriptorDistance(const DescMat& a, const DescMat& b) {
    if (a.cols != b.cols) throw std::invalid_argument("DescriptorDistance: column mismatch");
    return ppg_descriptor_distance(a.data.data(), b.data.data(), a.cols);
}
// (the context is not needed; kept so that code written against the earlier signature still compiles)
inline float DescriptorDistance(ppg_ctx*, const DescMat& a, const DescMat& b) { return DescriptorDistance(a, b); }

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

    // matching/src/Matcher.cpp:89-189 (no call site in the reference).  map: the candidate points with their
    // mTrackProjX/Y, mTrackViewCos, mbTrackInView; F.mvpMapPoints is updated (map.tracked is read, never written).
    int SearchByProjection(FrameView& F, const MapView& map, const float th) {
        const int m = (int)map.bad.size(), nc = (int)F.mvKeysUn.size();
        F.mvpMapPoints.resize(nc, -1);
        Grid grid(mpCamera, F);
        std::vector<int> rows;   // candidateMapPoints (:94-101), then the per-row skips (:111-117)
        for (int i = 0; i < m; i++)
            if (!map.bad[i] && map.in_view[i] && !map.tracked[i]) rows.push_back(i);
        const int nr = (int)rows.size();
        const bool bFactor = th != 1.0;
        std::vector<float> uv(2 * (size_t)nr, 0.f), radius(nr, 4.0f);
        std::vector<uint8_t> ok(nr, 1), claims(nr, 0), blocked(nc, 0);
        DescMat d1(nr, 256);
        for (int k = 0; k < nr; k++) {
            const int id = rows[k];
            float r = 4.0f;                                   // :127-133
            if (map.viewcos[id] > 0.998) r = 2.5;
            if (bFactor) r *= th;
            radius[k] = r;
            uv[2 * k] = map.proj[2 * id]; uv[2 * k + 1] = map.proj[2 * id + 1];
            claims[k] = map.nobs[id] > 0;                     // :154-156 sees the slots this loop has filled
            std::memcpy(d1.ptr(k), map.desc.ptr(id), sizeof(float) * 256);
        }
        for (int j = 0; j < nc; j++) blocked[j] = F.mvpMapPoints[j] >= 0 && map.nobs[F.mvpMapPoints[j]] > 0;
        std::vector<int32_t> best(nr, -1);
        int nmatches = 0;
        run(grid, F.mDescriptors, d1, uv, radius, ok, &claims, &blocked, TH_HIGH, mfNNratio, -1.f, best, nmatches);
        for (int k = 0; k < nr; k++)
            if (best[k] >= 0) F.mvpMapPoints[best[k]] = rows[k];   // in loop order: later rows overwrite (:180)
        return nmatches;
    }

    // matching/src/Matcher.cpp:1337-1411 (reached only from the uncalled Relocalization).  KF.mvpMapPoints:
    // GetMapPointMatches() as ids; alreadyFound[id] = sAlreadyFound.count(pMP); Tcw = CurrentFrame.GetPose().
    int SearchByProjection(FrameView& Cur, const FrameView& KF, const std::vector<uint8_t>& alreadyFound, const SE3f& Tcw,
                           const MapPointsView& map, const float th, const float descDist) {
        const int nk = (int)KF.mvpMapPoints.size(), nc = (int)Cur.mvKeysUn.size();
        Cur.mvpMapPoints.resize(nc, -1);
        Grid grid(mpCamera, Cur);
        float Ow[3];
        Tcw.center(Ow);
        std::vector<float> uv(2 * (size_t)nk, 0.f), radius(nk, th);
        std::vector<uint8_t> ok(nk, 0), claims(nk, 1), blocked(nc, 0);
        DescMat d1(nk, 256);
        for (int i = 0; i < nk; i++) {
            const int id = KF.mvpMapPoints[i];
            if (id < 0 || map.bad[id] || alreadyFound[id]) continue;
            const float* pw = &map.pos[3 * (size_t)id];
            float pc[3];
            Tcw.apply(pw, pc);
            project(pc, &uv[2 * i]);                          // no depth gate here (:1358-1362)
            if (!ppg_grid_is_in_image(grid.g, uv[2 * i], uv[2 * i + 1])) continue;
            const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
            const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
            if (dist3D < map.min_dist[id] || dist3D > map.max_dist[id]) continue;
            ok[i] = 1;
            std::memcpy(d1.ptr(i), map.desc.ptr(id), sizeof(float) * 256);
        }
        for (int j = 0; j < nc; j++) blocked[j] = Cur.mvpMapPoints[j] >= 0;   // :1388-1389, and every accepted match fills its slot
        std::vector<int32_t> best(nk, -1);
        int nmatches = 0;
        run(grid, Cur.mDescriptors, d1, uv, radius, ok, &claims, &blocked, descDist, 0.f, -1.f, best, nmatches);
        for (int i = 0; i < nk; i++)
            if (best[i] >= 0) Cur.mvpMapPoints[best[i]] = KF.mvpMapPoints[i];
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

    // matching/src/Matcher.cpp:393-477.  KF.mvpMapPoints: GetMapPointMatches() as ids; bad[id] = isBad().
    // vpMapPointMatches (size F.N) receives map-point ids.
    int SearchByBoW(const FrameView& KF, const FeatureVectorView& fvKF, const FrameView& F, const FeatureVectorView& fvF,
                    const std::vector<uint8_t>& bad, std::vector<int>& vpMapPointMatches) {
        vpMapPointMatches.assign(F.mvKeysUn.size(), -1);
        Lists L(*this, KF, fvKF, F, fvF, [&](unsigned idx) { const int id = mp(KF, idx); return id >= 0 && !bad[id]; });
        int nmatches = 0;
        for (size_t r = 0; r < L.rows.size(); r++) {
            float bestDist1 = 1e6f, bestDist2 = 1e6f;
            int bestIdxF = -1;
            for (int k = L.off[r]; k < L.off[r + 1]; k++) {
                const int realIdxF = L.idx[k];
                if (vpMapPointMatches[realIdxF] >= 0) continue;
                const float dist = L.D[k];
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= TH_LOW && bestDist1 < mfNNratio * bestDist2) {
                vpMapPointMatches[bestIdxF] = mp(KF, L.rows[r]);
                nmatches++;
            }
        }
        return nmatches;
    }

    // matching/src/Matcher.cpp:663-754.  vpMatches12 (size KF1.N) receives the ids of KF2's map points.  (Frame and
    // KeyFrame are one view type here, so the key-frame/key-frame overload carries a tag.)
    struct BetweenKeyFrames {};
    int SearchByBoW(const FrameView& KF1, const FeatureVectorView& fv1, const FrameView& KF2, const FeatureVectorView& fv2,
                    const std::vector<uint8_t>& bad, std::vector<int>& vpMatches12, BetweenKeyFrames) {
        vpMatches12.assign(KF1.mvKeysUn.size(), -1);
        std::vector<uint8_t> vbMatched2(KF2.mvKeysUn.size(), 0);
        Lists L(*this, KF1, fv1, KF2, fv2, [&](unsigned idx) { const int id = mp(KF1, idx); return id >= 0 && !bad[id]; });
        int nmatches = 0;
        for (size_t r = 0; r < L.rows.size(); r++) {
            float bestDist1 = 1e6f, bestDist2 = 1e6f;
            int bestIdx2 = -1;
            for (int k = L.off[r]; k < L.off[r + 1]; k++) {
                const int idx2 = L.idx[k];
                const int id2 = mp(KF2, idx2);
                if (vbMatched2[idx2] || id2 < 0) continue;
                if (bad[id2]) continue;
                const float dist = L.D[k];
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < TH_LOW && bestDist1 < mfNNratio * bestDist2) {
                vpMatches12[L.rows[r]] = mp(KF2, bestIdx2);
                vbMatched2[bestIdx2] = 1;
                nmatches++;
            }
        }
        return nmatches;
    }

    // matching/src/Matcher.cpp:767-885 (pinhole epipolar test, sensors/src/Pinhole.cpp:98-115).  T1w, T2w: the key
    // frames' poses.  Features that already hold a map point are skipped on both sides; vbMatched2 is never set by the
    // reference, so one KF2 feature may pair with several KF1 features.
    int SearchForTriangulation(const FrameView& KF1, const FeatureVectorView& fv1, const SE3f& T1w, const FrameView& KF2,
                               const FeatureVectorView& fv2, const SE3f& T2w, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
        float Cw[3], C2[3], ep[2];
        T1w.center(Cw);
        T2w.apply(Cw, C2);
        project(C2, ep);
        // T12 = T1w * Tw2: R12 = R1 R2^T, t12 = t1 - R12 t2
        float R12[9], t12[3];
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++)
                R12[3 * r + q] = T1w.R[3 * r] * T2w.R[3 * q] + T1w.R[3 * r + 1] * T2w.R[3 * q + 1] + T1w.R[3 * r + 2] * T2w.R[3 * q + 2];
        for (int r = 0; r < 3; r++) t12[r] = T1w.t[r] - (R12[3 * r] * T2w.t[0] + R12[3 * r + 1] * T2w.t[1] + R12[3 * r + 2] * T2w.t[2]);
        float F12[9];
        fundamental(R12, t12, F12);
        std::vector<int> vMatches12(KF1.mvKeysUn.size(), -1);
        Lists L(*this, KF1, fv1, KF2, fv2, [&](unsigned idx) { return mp(KF1, idx) < 0; });
        int nmatches = 0;
        for (size_t r = 0; r < L.rows.size(); r++) {
            const KeyPointEx& kp1 = KF1.mvKeysUn[L.rows[r]];
            float bestDist = TH_LOW;
            int bestIdx2 = -1;
            for (int k = L.off[r]; k < L.off[r + 1]; k++) {
                const int idx2 = L.idx[k];
                if (mp(KF2, idx2) >= 0) continue;
                const float dist = L.D[k];
                if (dist > TH_LOW || dist > bestDist) continue;
                const KeyPointEx& kp2 = KF2.mvKeysUn[idx2];
                const float ex = ep[0] - kp2.mPos[0], ey = ep[1] - kp2.mPos[1];
                if (std::sqrt(ex * ex + ey * ey) < 10.0f) continue;
                if (epipolarConstrain(kp1, kp2, F12)) { bestIdx2 = idx2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) { vMatches12[L.rows[r]] = bestIdx2; nmatches++; }
        }
        vMatchedPairs.clear();
        for (size_t i = 0; i < vMatches12.size(); i++)
            if (vMatches12[i] >= 0) vMatchedPairs.emplace_back(i, (size_t)vMatches12[i]);
        return nmatches;
    }

public:
    static constexpr float TH_LOW = 0.7f;   // matching/src/Matcher.cpp:13
    static constexpr float TH_HIGH = 0.8f;  // matching/src/Matcher.cpp:12
    ppg_ctx* mCtx;
    const CameraModel* mpCamera;
    float mfNNratio;

private:
    static int mp(const FrameView& F, unsigned idx) { return idx < F.mvpMapPoints.size() ? F.mvpMapPoints[idx] : -1; }
    // The lock-step walk over two FeatureVectors (matching/src/Matcher.cpp:405-475): rows = the features of frame 1 that
    // pass `take`, in walk order, each with the features of frame 2 filed under the same node; every listed
    // DescriptorDistance comes from one GPU launch.
    struct Lists {
        std::vector<unsigned> rows;
        std::vector<int32_t> off, idx;
        std::vector<float> D;
        template <typename Take>
        Lists(Matcher& m, const FrameView& F1, const FeatureVectorView& fv1, const FrameView& F2, const FeatureVectorView& fv2, Take take) {
            off.push_back(0);
            size_t a = 0, b = 0;
            while (a < fv1.nodes.size() && b < fv2.nodes.size()) {
                if (fv1.nodes[a].first == fv2.nodes[b].first) {
                    for (unsigned i1 : fv1.nodes[a].second) {
                        if (!take(i1)) continue;
                        rows.push_back(i1);
                        for (unsigned i2 : fv2.nodes[b].second) idx.push_back((int32_t)i2);
                        off.push_back((int32_t)idx.size());
                    }
                    a++; b++;
                } else if (fv1.nodes[a].first < fv2.nodes[b].first) {
                    while (a < fv1.nodes.size() && fv1.nodes[a].first < fv2.nodes[b].first) a++;  // lower_bound
                } else {
                    while (b < fv2.nodes.size() && fv2.nodes[b].first < fv1.nodes[a].first) b++;
                }
            }
            D.assign(idx.size(), 0.f);
            if (rows.empty() || idx.empty()) return;
            DescMat d1((int)rows.size(), 256);
            for (size_t r = 0; r < rows.size(); r++) std::memcpy(d1.ptr((int)r), F1.mDescriptors.ptr((int)rows[r]), sizeof(float) * 256);
            m.check(ppg_dist_lists(m.mCtx, d1.data.data(), d1.rows, F2.mDescriptors.data.data(), F2.mDescriptors.rows, 256,
                                   off.data(), idx.data(), D.data()));
        }
    };
    // F12 = K^-T [t12]x R12 K^-1 (sensors/src/Pinhole.cpp:101-104), with the closed-form inverse of the upper-triangular K
    void fundamental(const float* R12, const float* t12, float* F12) const {
        const float fx = mpCamera->K[0], fy = mpCamera->K[4], cx = mpCamera->K[2], cy = mpCamera->K[5];
        const float Ki[9] = {1.f / fx, 0.f, -cx / fx, 0.f, 1.f / fy, -cy / fy, 0.f, 0.f, 1.f};
        const float tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
        float A[9], B[9];
        mul3(tx, R12, A);
        mul3(A, Ki, B);
        // K^-T * B: (K^-T)[r][q] = Ki[q][r]
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) F12[3 * r + q] = Ki[r] * B[q] + Ki[3 + r] * B[3 + q] + Ki[6 + r] * B[6 + q];
    }
    static void mul3(const float* X, const float* Y, float* Z) {
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) Z[3 * r + q] = X[3 * r] * Y[q] + X[3 * r + 1] * Y[3 + q] + X[3 * r + 2] * Y[6 + q];
    }
    static bool epipolarConstrain(const KeyPointEx& kp1, const KeyPointEx& kp2, const float* F12) {
        const float a = kp1.mPos[0] * F12[0] + kp1.mPos[1] * F12[3] + F12[6];
        const float b = kp1.mPos[0] * F12[1] + kp1.mPos[1] * F12[4] + F12[7];
        const float c = kp1.mPos[0] * F12[2] + kp1.mPos[1] * F12[5] + F12[8];
        const float num = a * kp2.mPos[0] + b * kp2.mPos[1] + c;
        const float den = a * a + b * b;
        if (den == 0) return false;
        const float dsqr = num * num / den;
        return (double)dsqr < 3.84;
    }
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
    // mpCamera->project: Pinhole::project (sensors/src/Pinhole.cpp:32-38) or KannalaBrandt8::project
    // (sensors/src/KannalaBrandt8.cpp:43-59; its unqualified cos / sin are the double functions)
    void project(const float* p, float* uv) const {
        if (mpCamera->mnType == CameraModel::CAM_FISHEYE) {
            const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
            const float theta = atan2f(sqrtf(x2_plus_y2), p[2]);
            const float psi = atan2f(p[1], p[0]);
            const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                        theta9 = theta7 * theta2;
            const float r = theta + mpCamera->D[0] * theta3 + mpCamera->D[1] * theta5 + mpCamera->D[2] * theta7 + mpCamera->D[3] * theta9;
            uv[0] = (float)((double)(mpCamera->K[0] * r) * std::cos((double)psi) + (double)mpCamera->K[2]);
            uv[1] = (float)((double)(mpCamera->K[4] * r) * std::sin((double)psi) + (double)mpCamera->K[5]);
            return;
        }
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
