// Exercises the host-side mirror (PPGExtractor::run, Matcher::SearchForInitialization) the way
// map/src/Frame.cpp:62 and system/src/Tracking.cpp:524-525 use the reference classes.
// Usage: shim_demo <weights dir> <raw u8 image A> <raw u8 image B> <width> <height> [projection dump file]
// Prints one line of integers that tests/test_gpu_parity.py compares with the C-ABI results.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Matcher.h"
#include "PPGExtractor.h"

using namespace ppg_host;

static std::vector<uint8_t> read_raw(const char* path, size_t n) {
    std::vector<uint8_t> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

template <typename T>
static void dump(FILE* f, const std::vector<T>& v) {
    const int32_t n = (int32_t)v.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(v.data(), sizeof(T), v.size(), f);
}

// The projection matchers on a synthetic map built from frame A's key points (tests/test_gpu_parity.py rebuilds
// nothing: it reads the dumped inputs, gives them to the CPU oracle and compares the dumped results).
static int projection_demo(Matcher& matcher, const CameraModel& cam, FrameView F[2], const char* out_path) {
    const int na = (int)F[0].mvKeysUn.size(), nb = (int)F[1].mvKeysUn.size();
    MapPointsView map;
    map.desc = DescMat(na, 256);
    for (int i = 0; i < na; i++) {
        const float z = i % 2 == 0 ? 3.5f : 2.0f + 0.5f * (float)(i % 7);  // half of the points reproject within a pixel
        const float x = (F[0].mvKeysUn[i].mPos[0] - cam.K[2]) / cam.K[0] * z, y = (F[0].mvKeysUn[i].mPos[1] - cam.K[5]) / cam.K[4] * z;
        const float n = std::sqrt(x * x + y * y + z * z);
        map.pos.insert(map.pos.end(), {x, y, z});
        map.normal.insert(map.normal.end(), {x / n, y / n, z / n});
        map.min_dist.push_back(n * (i % 11 == 3 ? 1.2f : 0.6f));   // some points fail the scale gate
        map.max_dist.push_back(n * 1.8f);
        map.bad.push_back(i % 13 == 5);
        map.nobs.push_back(i % 4);                                  // a quarter of the points is unobserved
        std::memcpy(map.desc.ptr(i), F[0].mDescriptors.ptr(i), sizeof(float) * 256);
    }
    // frame B shows frame A's content displaced by (-3, -1) px: at depth z that is a translation of (-3z/fx, -z/fy); use the
    // mean depth, so projections land within a few pixels of the true key points
    SE3f Tcw;
    Tcw.t[0] = -3.0f * 3.5f / cam.K[0]; Tcw.t[1] = -1.0f * 3.5f / cam.K[4]; Tcw.t[2] = 0.01f;
    std::vector<uint8_t> outlier(na, 0);
    F[0].mvpMapPoints.assign(na, -1);
    for (int i = 0; i < na; i++) {
        if (i % 5 != 4) F[0].mvpMapPoints[i] = i;
        outlier[i] = i % 17 == 9;
    }
    F[1].mvpMapPoints.assign(nb, -1);
    for (int j = 0; j < nb; j += 9) F[1].mvpMapPoints[j] = (j * 7) % std::max(na, 1);  // some slots already taken
    std::vector<int> cur_before(F[1].mvpMapPoints.begin(), F[1].mvpMapPoints.end());
    FrameView cur = F[1];
    const int n1 = matcher.SearchByProjection(cur, F[0], outlier, Tcw, map, 15.f);
    std::vector<int> points, matched(nb, -1), matched_before;
    for (int i = na - 1; i >= 0; i--) points.push_back(i);
    for (int j = 0; j < nb; j += 6) matched[j] = (j * 5) % std::max(na, 1);
    matched_before = matched;
    const int n2 = matcher.SearchByProjection(F[1], Tcw, map, points, matched, 10, 1.0f);
    std::vector<uint8_t> in_kf(points.size());
    for (size_t i = 0; i < points.size(); i++) in_kf[i] = i % 10 == 7;
    std::vector<int> fuse_pts = points, best_a, best_b;
    for (size_t i = 0; i < fuse_pts.size(); i += 23) fuse_pts[i] = -1;
    const int n3 = matcher.Fuse(F[1], Tcw, map, fuse_pts, in_kf, 3.f, true, best_a);
    const int n4 = matcher.Fuse(F[1], Tcw, map, fuse_pts, in_kf, 4.f, false, best_b);
    {   // the two overloads without a call site in the reference (Matcher.cpp:1337-1411, :89-189): they must run and
        // respect their own rules (only empty slots are filled; no slot twice)
        FrameView cur2 = F[1];
        std::vector<uint8_t> found(map.size(), 0);
        for (int i = 0; i < na; i += 7) found[i] = 1;
        const int nd1 = matcher.SearchByProjection(cur2, F[0], found, Tcw, map, 15.f, 0.8f);
        int filled = 0;
        for (int j = 0; j < nb; j++) {
            if (cur_before[j] >= 0 && cur2.mvpMapPoints[j] != cur_before[j]) { fprintf(stderr, "dead overload 1337 touched a taken slot\n"); return 3; }
            if (cur_before[j] < 0 && cur2.mvpMapPoints[j] >= 0) { filled++; if (found[cur2.mvpMapPoints[j]]) return 3; }
        }
        if (nd1 != filled || nd1 < 10) { fprintf(stderr, "dead overload 1337: %d matches, %d slots filled\n", nd1, filled); return 3; }
        MapView mv;
        mv.desc = map.desc; mv.bad = map.bad; mv.nobs = map.nobs;
        mv.in_view.assign(na, 1); mv.tracked.assign(na, 0); mv.viewcos.assign(na, 0.9f); mv.proj.assign(2 * (size_t)na, 0.f);
        for (int i = 0; i < na; i++) {
            float pc[3];
            Tcw.apply(&map.pos[3 * (size_t)i], pc);
            mv.proj[2 * i] = cam.K[0] * pc[0] / pc[2] + cam.K[2]; mv.proj[2 * i + 1] = cam.K[4] * pc[1] / pc[2] + cam.K[5];
            mv.in_view[i] = i % 6 != 1; mv.tracked[i] = i % 10 == 2; if (i % 3 == 0) mv.viewcos[i] = 0.999f;
        }
        mv.edge_off.assign(na + 1, 0);
        FrameView cur3 = F[1];
        Matcher loose2(matcher.mCtx, &cam, 0.9f);
        const int nd2 = loose2.SearchByProjection(cur3, mv, 3.f);
        int changed = 0;
        for (int j = 0; j < nb; j++) {
            const int before = cur_before[j], after = cur3.mvpMapPoints[j];
            if (before >= 0 && map.nobs[before] > 0 && after != before) { fprintf(stderr, "dead overload 89 replaced an observed point\n"); return 3; }
            if (after != before) { changed++; if (map.bad[after] || !mv.in_view[after] || mv.tracked[after]) return 3; }
        }
        if (nd2 < changed || nd2 < 10) { fprintf(stderr, "dead overload 89: %d matches, %d slots changed\n", nd2, changed); return 3; }
    }
    // ---- FeatureVector matchers: features filed by a 160-px position bucket, some nodes missing on either side ----
    auto file_features = [&](const FrameView& F, int drop_mod, int drop_rem) {
        std::vector<std::vector<unsigned>> bucket(64);
        for (size_t i = 0; i < F.mvKeysUn.size(); i++) {
            const int node = 1 + (int)(F.mvKeysUn[i].mPos[1] / 160.f) * 4 + (int)(F.mvKeysUn[i].mPos[0] / 160.f);
            if (node >= 0 && node < 64) bucket[node].push_back((unsigned)i);
        }
        FeatureVectorView fv;
        for (unsigned n = 0; n < 64; n++)
            if (!bucket[n].empty() && (int)(n % drop_mod) != drop_rem) fv.nodes.emplace_back(n, bucket[n]);
        return fv;
    };
    auto flatten = [](const FeatureVectorView& fv, std::vector<int>& node, std::vector<int>& off, std::vector<int>& idx) {
        off.push_back(0);
        for (auto& e : fv.nodes) {
            node.push_back((int)e.first);
            for (unsigned i : e.second) idx.push_back((int)i);
            off.push_back((int)idx.size());
        }
    };
    const FeatureVectorView fvA = file_features(F[0], 5, 2), fvB = file_features(F[1], 7, 3);
    FrameView A = F[0], B = F[1];
    for (int j = 0; j < nb; j++) B.mvpMapPoints[j] = j % 3 ? (j * 3) % std::max(na, 1) : -1;
    std::vector<int> bow_f, bow_kk;
    Matcher loose(matcher.mCtx, &cam, 0.9f);
    const int n5 = loose.SearchByBoW(A, fvA, B, fvB, map.bad, bow_f);
    const int n6 = loose.SearchByBoW(A, fvA, B, fvB, map.bad, bow_kk, Matcher::BetweenKeyFrames());
    FrameView A2 = F[0], B2 = F[1];
    for (int i = 0; i < na; i++) A2.mvpMapPoints[i] = i % 2 ? i : -1;
    for (int j = 0; j < nb; j++) B2.mvpMapPoints[j] = j % 4 == 1 ? 0 : -1;
    std::vector<std::pair<size_t, size_t>> tri;
    SE3f T1w;
    const int n7 = loose.SearchForTriangulation(A2, fvA, T1w, B2, fvB, Tcw, tri);
    std::vector<int> tri_flat;
    for (auto& pr : tri) { tri_flat.push_back((int)pr.first); tri_flat.push_back((int)pr.second); }
    // the root-only FeatureVector of the shipped vocabularies
    std::vector<int> bow_root;
    const int n8 = loose.SearchByBoW(A, FeatureVectorView::all_in_root(na), B, FeatureVectorView::all_in_root(nb), map.bad, bow_root);

    FILE* f = fopen(out_path, "wb");
    if (!f) return 2;
    std::vector<float> pose(Tcw.R, Tcw.R + 9), posb, camk(cam.K, cam.K + 9);
    pose.insert(pose.end(), Tcw.t, Tcw.t + 3);
    for (auto& kp : F[1].mvKeysUn) { posb.push_back(kp.mPos[0]); posb.push_back(kp.mPos[1]); }
    std::vector<int> counts = {n1, n2, n3, n4, cam.width, cam.height};
    dump(f, counts); dump(f, camk); dump(f, pose);
    dump(f, map.pos); dump(f, map.normal); dump(f, map.min_dist); dump(f, map.max_dist); dump(f, map.desc.data);
    dump(f, map.bad); dump(f, map.nobs);
    dump(f, F[0].mvpMapPoints); dump(f, outlier); dump(f, posb); dump(f, F[1].mDescriptors.data);
    dump(f, cur_before); dump(f, cur.mvpMapPoints);
    dump(f, points); dump(f, matched_before); dump(f, matched);
    dump(f, fuse_pts); dump(f, in_kf); dump(f, best_a); dump(f, best_b);
    std::vector<int> nodeA, offA, idxA, nodeB, offB, idxB, counts2 = {n5, n6, n7, n8};
    flatten(fvA, nodeA, offA, idxA);
    flatten(fvB, nodeB, offB, idxB);
    std::vector<float> posa;
    for (auto& kp : F[0].mvKeysUn) { posa.push_back(kp.mPos[0]); posa.push_back(kp.mPos[1]); }
    dump(f, counts2); dump(f, nodeA); dump(f, offA); dump(f, idxA); dump(f, nodeB); dump(f, offB); dump(f, idxB);
    dump(f, posa); dump(f, A.mvpMapPoints); dump(f, B.mvpMapPoints); dump(f, bow_f); dump(f, bow_kk);
    dump(f, A2.mvpMapPoints); dump(f, B2.mvpMapPoints); dump(f, tri_flat); dump(f, bow_root);
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) { fprintf(stderr, "usage: %s weights_dir imgA imgB W H [projection dump]\n", argv[0]); return 2; }
    const int W = atoi(argv[4]), H = atoi(argv[5]);
    CameraModel cam;
    cam.width = W; cam.height = H;
    cam.K[0] = 458.654f; cam.K[4] = 457.296f; cam.K[2] = W * 0.5f; cam.K[5] = H * 0.5f;
    try {
        PPGExtractor ext(&cam, argv[1]);
        FrameView F[2];
        for (int k = 0; k < 2; k++) {
            std::vector<uint8_t> img = read_raw(argv[2 + k], (size_t)W * H);
            GrayImage g;
            g.data = img.data(); g.rows = H; g.cols = W; g.step = W;
            std::vector<KeyPointEx> keys;
            ext.run(g, keys, F[k].mvKeysUn, F[k].mvKeyEdges, F[k].mDescriptors);
        }
        Matcher matcher(ext.context(), &cam, 0.9f);
        std::vector<Point2f> prev;
        for (auto& kp : F[0].mvKeysUn) prev.emplace_back(kp.mPos[0], kp.mPos[1]);
        std::vector<int> m12;
        const int nm = matcher.SearchForInitialization(F[0], F[1], prev, m12, 50);
        {   // the free function DescriptorDistance (host arithmetic, no context) against the GPU's distance matrix, bit for bit
            const int na = std::min<int>(16, (int)F[0].mvKeysUn.size()), nb = std::min<int>(16, (int)F[1].mvKeysUn.size());
            std::vector<float> D((size_t)na * nb);
            if (na && nb && ppg_dist_matrix(ext.context(), F[0].mDescriptors.ptr(0), na, F[1].mDescriptors.ptr(0), nb, 256, D.data()) != PPG_OK) {
                fprintf(stderr, "ppg_dist_matrix: %s\n", ppg_last_error(ext.context()));
                return 3;
            }
            for (int i = 0; i < na; i++)
                for (int j = 0; j < nb; j++) {
                    DescMat a(1, 256), b(1, 256);
                    std::memcpy(a.ptr(0), F[0].mDescriptors.ptr(i), sizeof(float) * 256);
                    std::memcpy(b.ptr(0), F[1].mDescriptors.ptr(j), sizeof(float) * 256);
                    const float d = DescriptorDistance(a, b);
                    if (std::memcmp(&d, &D[(size_t)i * nb + j], 4) != 0) {
                        fprintf(stderr, "DescriptorDistance(%d, %d) = %.9g on the host, %.9g on the GPU\n", i, j, d, D[(size_t)i * nb + j]);
                        return 3;
                    }
                }
        }
        long long chk = 0;
        for (size_t i = 0; i < m12.size(); i++) chk = chk * 31 + (m12[i] + 2);
        size_t conn = 0;
        for (auto& kp : F[0].mvKeysUn) conn += kp.mvConnected.size();
        printf("%zu %zu %zu %zu %d %lld\n", F[0].mvKeysUn.size(), F[0].mvKeyEdges.size(), F[1].mvKeysUn.size(), conn, nm, chk);
        if (argc == 7 && projection_demo(matcher, cam, F, argv[6]) != 0) { fprintf(stderr, "cannot write %s\n", argv[6]); return 2; }
        {   // what another thread of the reference does (LocalMapping.cpp:214, LoopClosing.cpp:408): a Matcher of its own, on a
            // context without the extractor's weights and buffers; the same match must come out
            ppg_config mc;
            ppg_default_config(&mc, W, H);
            ppg_ctx* mctx = ppg_create_matcher(&mc);
            if (!mctx) { fprintf(stderr, "ppg_create_matcher: %s\n", ppg_last_error(nullptr)); return 4; }
            Matcher other(mctx, &cam, 0.9f);
            std::vector<int> m12b;
            const int nmb = other.SearchForInitialization(F[0], F[1], prev, m12b, 50);
            ppg_frame_out none;
            const bool refused = ppg_extract(mctx, nullptr, 0, &none) == PPG_EINVAL;     // an extractor call on it is an error
            ppg_destroy(mctx);
            if (nmb != nm || m12b != m12 || !refused) { fprintf(stderr, "matcher context: %d matches against %d\n", nmb, nm); return 4; }
        }
        {   // one process per camera: the barrier between the ranks and the swap of a stereo pair's descriptors, here in a world
            // of one rank (the partner of the swap is the rank itself).  Skipped when no RCCL can be loaded.
            uint8_t id[PPG_SHUTTER_ID_BYTES];
            if (ppg_shutter_unique_id(id) != PPG_OK) {
                fprintf(stderr, "note: no RCCL (%s): barrier and stereo swap not exercised\n", ppg_last_error(nullptr));
            } else {
                ppg_shutter* sh = ppg_shutter_create(ext.context(), 0, 1, id);
                if (!sh) { fprintf(stderr, "ppg_shutter_create: %s\n", ppg_last_error(ext.context())); return 5; }
                int released = -2;
                for (int group = 0; group < 3; group++)
                    if (ppg_shutter_release(sh, group, nullptr) != PPG_OK) { fprintf(stderr, "ppg_shutter_release: %s\n", ppg_last_error(ext.context())); return 5; }
                if (ppg_shutter_drain(sh, &released) != PPG_OK || released != 2) { fprintf(stderr, "ppg_shutter_drain: released %d\n", released); return 5; }
                const int n0 = (int)F[0].mvKeysUn.size();
                std::vector<float> back((size_t)std::max(n0, 1) * 256);
                int nback = -1;
                if (ppg_stereo_swap(sh, 0, n0 ? F[0].mDescriptors.ptr(0) : back.data(), n0, back.data(), &nback) != PPG_OK || nback != n0 ||
                    (n0 && std::memcmp(back.data(), F[0].mDescriptors.ptr(0), sizeof(float) * 256 * n0) != 0)) {
                    fprintf(stderr, "ppg_stereo_swap: %d rows back of %d (%s)\n", nback, n0, ppg_last_error(ext.context()));
                    return 5;
                }
                // swap + left-right match in one call on the descriptors still resident from the last run() (frame F[1]); with
                // the rank as its own partner left = right, so (nearly) every row matches itself
                const int n1 = (int)F[1].mvKeysUn.size();
                std::vector<int32_t> lr(PPGExtractor::JUNCTION_MAX_NUM);
                int nlr = -1, npart = -1;
                if (ppg_stereo_match(sh, 0, 0, 0.9f, lr.data(), &nlr, &npart) != PPG_OK || npart != n1) {
                    fprintf(stderr, "ppg_stereo_match: partner rows %d of %d (%s)\n", npart, n1, ppg_last_error(ext.context()));
                    return 5;
                }
                int self = 0;
                for (int i = 0; i < n1; i++) self += lr[i] == i;
                if (self * 10 < n1 * 9) { fprintf(stderr, "ppg_stereo_match: %d of %d rows match themselves\n", self, n1); return 5; }
                ppg_shutter_destroy(sh);
            }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
