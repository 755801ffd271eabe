// Host-side mirror of the reference's Matcher (matching/include/Matcher.h:20-64) and of
// DescriptorDistance (feature/include/MapPoint.h:25) over the C ABI of libppgfront.
// Header-only; link with -lppgfront.
//
// The reference's Frame / MapPoint / MapEdge are pointer graphs guarded by per-object mutexes.  The
// mirror takes the same information as flat views: the caller gathers it under the reference's locks
// (MapPoint::GetDescriptor / isBad / getEdges, feature/src/MapPoint.cpp:203-210, 304-308, 421-425),
// calls in, and scatters the returned ids back into F.mvpMapPoints / F.mvpMapEdges.
#pragma once
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

public:
    static constexpr float TH_LOW = 0.7f;   // matching/src/Matcher.cpp:13
    static constexpr float TH_HIGH = 0.8f;  // matching/src/Matcher.cpp:12
    ppg_ctx* mCtx;
    const CameraModel* mpCamera;
    float mfNNratio;

private:
    void check(int rc) {
        if (rc != PPG_OK) throw std::runtime_error(std::string("libppgfront: ") + ppg_last_error(mCtx));
    }
};

}  // namespace ppg_host
