// Host-side, order-dependent tail of detectLines.  Pure C++ (no HIP): included by ppg_front.cpp and by
// the host micro-benchmark tools/bench_tail.cpp.  Must be compiled with -ffp-contract=off.
#pragma once
#include "../../include/ppgfront.h"

#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct FrameHost {
    std::vector<uint32_t> edge_idx;
    std::vector<float> edge_score;
    std::vector<int32_t> conn_off, coline_off;
    std::vector<uint32_t> conn_idx, coline_pairs;
};

struct Cand {
    uint32_t i, j;
    float inlier, lscore_heat;
    float dist;        // |Pj - Pi|                                   (:278-282)
    float dir_ij;      // direction seen from i                        (:283)
    float dir_ji;      // direction seen from j = dir_ij - pi, wrapped (:284-286)
    float lscore;
    float dir_from(uint32_t p) const { return p == i ? dir_ij : dir_ji; }
    uint32_t other(uint32_t p) const { return p == i ? j : i; }
};

// One entry of a point's incidence list: what the overlap test needs about a line, stored inline so
// that the walk over a list touches the point's own small array only.
struct Inc {
    float dir;     // direction of the line seen from this point
    float dist;
    uint32_t lid;  // index into TailScratch::lines
    int32_t next;  // next entry of the same direction sector, -1 = end
};

constexpr int kBuckets = 32;

// The lines incident to one point: entries in arrival order in ONE contiguous array, chained per direction
// sector (head[s] = newest entry of sector s), so that listing a line is an append and a walk over
// neighbouring sectors stays inside a few cache lines.
struct PointLists {
    int32_t head[kBuckets];
    float dmin;  // lower bound of the length of every line ever listed here
    std::vector<Inc> e;
};

// Per-frame scratch that survives between calls so the tail allocates nothing in steady state.
struct TailScratch {
    std::vector<Cand> lines;
    std::vector<uint8_t> bad;  // isBad per line
    std::vector<PointLists> pts;
    std::vector<std::vector<uint32_t>> conn2;
};

// ---------------------------------------------------------------------------------------------
// Order-dependent tail of detectLines (feature/src/PPGExtractor.cpp:290-441) on the candidate list
// the GPU produced.  Candidates arrive sorted by (i, j) — the reference's loop order — each with its
// inlier rate and mean heat.  Everything here is O(candidates x live degree).
// ---------------------------------------------------------------------------------------------
const double kPi = 3.1415926535897932384626433832795, k2Pi = 6.283185307179586476925286766559;

void lines_tail(const ppg_config& cfg, int n, const float* un, const uint32_t* ij, const float* sc, int nc,
                FrameHost& fh, TailScratch& ts) {
    std::vector<Cand>& lines = ts.lines;  // candidateLines of the reference (survivors of the overlap test)
    std::vector<uint8_t>& bad = ts.bad;
    lines.clear();
    bad.clear();
    if (ts.pts.size() < (size_t)n) { ts.pts.resize(n); ts.conn2.resize(n); }
    std::vector<PointLists>& pts = ts.pts;
    for (int p = 0; p < n; p++) {
        pts[p].e.clear();
        pts[p].dmin = 3.0e38f;
        for (int t = 0; t < kBuckets; t++) pts[p].head[t] = -1;
    }
    const float distTh = cfg.line_dist_thresh;
    // Incidence lists are split into kBuckets direction sectors per point.  Only lines within 0.2*pi of
    // the new direction can pass the angle gate of :326/:348, so the walk visits the sectors that can
    // hold them (with slack for rounding) and skips the rest — the skipped entries would `continue`.
    // The loop body has no early exit and its effects commute, so the visiting order is immaterial.
    const float bscale = (float)(kBuckets / k2Pi);
    auto sector = [&](float dir) -> int {
        int b = (int)((dir + (float)kPi) * bscale);
        return b < 0 ? 0 : (b >= kBuckets ? kBuckets - 1 : b);
    };
    auto insert = [&](uint32_t p, float dir, float dist, uint32_t lid) {
        PointLists& pl = pts[p];
        const int sct = sector(dir);
        pl.e.push_back(Inc{dir, dist, lid, pl.head[sct]});
        pl.head[sct] = (int32_t)pl.e.size() - 1;
        if (dist < pl.dmin) pl.dmin = dist;
    };
    auto overlap = [&](uint32_t a, float dirAB, float dNew) -> bool {  // :316-335 / :338-357
        PointLists& pl = pts[a];
        if (pl.e.empty()) return false;
        bool isOverlap = false;
        // A listed line matters only if it passes the angle gate AND (shorter length) * sin(angle) < distTh.  The
        // shorter length is at least dLow = min(dNew, shortest line ever listed here), and sin(x) >= x * 2/pi on
        // [0, pi/2], so lines further than distTh / dLow * pi/2 from the new direction cannot matter: the walk is
        // narrowed to that angle (plus slack for rounding) when it is smaller than the gate.
        float slack = 0.2f * (float)kPi + 1e-3f;
        const float dLow = dNew < pl.dmin ? dNew : pl.dmin;
        const float narrow = distTh / dLow * 1.5708f + 2e-3f;
        if (narrow < slack) slack = narrow;
        // floor() of values in (-2, kBuckets + 2) by truncation of the value shifted to be positive
        const int b0 = (int)((dirAB - slack + (float)kPi) * bscale + 8.f) - 8;
        int b1 = (int)((dirAB + slack + (float)kPi) * bscale + 8.f) - 8;
        if (b1 - b0 >= kBuckets) b1 = b0 + kBuckets - 1;
        auto walk = [&](int s0, int s1) {  // sectors s0..s1, both inside [0, kBuckets)
          for (int sct = s0; sct <= s1; sct++) {
            int32_t* link = &pl.head[sct];
            while (*link >= 0) {
                Inc& ent = pl.e[*link];
                // Dead lines are skipped by the reference and never come back to life: unlink them while walking
                // (keeps the walks proportional to the live degree).
                if (bad[ent.lid]) { *link = ent.next; continue; }
                link = &ent.next;
                const Inc e = ent;
                float ad = dirAB - e.dir;
                if (ad < -kPi) ad = (float)(ad + k2Pi);
                if (ad > kPi) ad = (float)(ad - k2Pi);
                ad = std::fabs(ad);
                if (ad > 0.2 * kPi) continue;
                const float dOld = e.dist;
                // Both tests of the reference ask whether (shorter length) * sin(ad) < distTh.  For 0 <= x <= 0.2*pi
                // the sine lies between x - x^3/6 (>= 0.934 x) and x; when the product with a bound is on the same
                // side of the threshold by more than every rounding involved (the float sine, the float product:
                // < 3e-7 relative) the answer is known without evaluating the sine.
                const float dMin = dNew <= dOld ? dNew : dOld;
                const float pa = dMin * ad;
                bool close;
                if (pa * 0.93f > distTh) close = false;
                else if (pa * 1.00001f < distTh) close = true;
                else {
                    const double x = ad, x2 = x * x, lo = x - x * x2 * (1.0 / 6.0), hi = lo + x * x2 * x2 * (1.0 / 120.0);
                    if ((double)dMin * hi * (1.0 + 1e-6) < (double)distTh) close = true;
                    else if ((double)dMin * lo * (1.0 - 1e-6) > (double)distTh) close = false;
                    else close = dMin * (float)std::sin((double)ad) < distTh;
                }
                if (!close) continue;
                if (dNew <= dOld) bad[e.lid] = 1;
                else isOverlap = true;
            }
          }
        };
        // dirAB lies in [-pi, pi], so b1 >= 0 and b0 < kBuckets: at most one end of the range wraps
        if (b0 < 0) { walk(b0 + kBuckets, kBuckets - 1); walk(0, b1); }
        else if (b1 >= kBuckets) { walk(b0, kBuckets - 1); walk(0, b1 - kBuckets); }
        else walk(b0, b1);
        return isOverlap;
    };
    for (int k = 0; k < nc; k++) {
        // The second end point of a candidate is a random point: its lists are fetched ahead of their use.
        if (k + 8 < nc) {
            const char* ph = (const char*)&pts[ij[k + 8] & 0xFFFFu];
            __builtin_prefetch(ph);
            __builtin_prefetch(ph + 64);
            __builtin_prefetch(ph + 128);
        }
        if (k + 4 < nc) {
            const PointLists& pn = pts[ij[k + 4] & 0xFFFFu];
            const char* pe = (const char*)pn.e.data();
            const size_t bytes = pn.e.size() * sizeof(Inc);
            for (size_t o = 0; o < bytes && o < 512; o += 64) __builtin_prefetch(pe + o);
        }
        Cand cd;
        cd.i = ij[k] >> 16; cd.j = ij[k] & 0xFFFFu;
        float vx = un[2 * cd.j] - un[2 * cd.i], vy = un[2 * cd.j + 1] - un[2 * cd.i + 1];
        cd.dist = std::sqrt(vx * vx + vy * vy);
        vx = vx / cd.dist; vy = vy / cd.dist;
        // libm-independent statement of std::atan2(float, float): the double function rounded to float (the GPU
        // statement of this tail, line_resolve_kernel in post_kernels.hip, evaluates the same expression)
        cd.dir_ij = (float)std::atan2((double)vy, (double)vx);
        float r = (float)(cd.dir_ij - kPi);
        if (r < -kPi) r = (float)(r + k2Pi);
        cd.dir_ji = r;
        if (overlap(cd.i, cd.dir_ij, cd.dist)) continue;
        if (overlap(cd.j, cd.dir_ji, cd.dist)) continue;
        cd.inlier = sc[2 * k]; cd.lscore_heat = sc[2 * k + 1];
        cd.lscore = 0.f;
        const uint32_t lid = (uint32_t)lines.size();
        lines.push_back(cd);
        bad.push_back(0);
        insert(cd.i, cd.dir_ij, cd.dist, lid);
        insert(cd.j, cd.dir_ji, cd.dist, lid);
    }
    std::vector<std::vector<uint32_t>>& c2 = ts.conn2;
    for (int p = 0; p < n; p++) c2[p].clear();
    for (uint32_t k = 0; k < lines.size(); k++) {  // :367-389
        Cand& kl = lines[k];
        if (bad[k]) continue;
        if (kl.inlier < cfg.line_inlier_rate) { bad[k] = 1; continue; }
        if (kl.lscore_heat < cfg.line_heatmap_thresh) { bad[k] = 1; continue; }
        kl.lscore = kl.inlier * kl.lscore_heat;
        c2[kl.i].push_back(k);
        c2[kl.j].push_back(k);
    }
    fh.coline_off.assign(n + 1, 0);
    fh.coline_pairs.clear();
    std::vector<uint32_t> idx;
    for (int p = 0; p < n; p++) {  // :392-432
        fh.coline_off[p] = (int32_t)(fh.coline_pairs.size() / 2);
        if (c2[p].size() < 2) continue;
        idx = c2[p];
        while (idx.size() > 1) {
            double minDiff = 1e9;
            int b1 = -1, b2 = -1, bestId = -1;
            const Cand& kl1 = lines[idx.back()];
            if (bad[idx.back()]) { idx.pop_back(); continue; }
            for (uint32_t q = 0; q + 1 < idx.size(); q++) {
                if (bad[idx[q]]) continue;
                const Cand& kl2 = lines[idx[q]];
                const float ad = kl1.dir_from(p) - kl2.dir_from(p);
                const double pd = 0.5 * (kl1.dist + kl2.dist) * std::fabs((float)std::sin((double)ad));
                if (minDiff > pd) { minDiff = pd; bestId = (int)q; b1 = (int)kl1.other(p); b2 = (int)kl2.other(p); }
            }
            if (minDiff > cfg.line_dist_thresh) { idx.pop_back(); continue; }
            fh.coline_pairs.push_back((uint32_t)b1);
            fh.coline_pairs.push_back((uint32_t)b2);
            idx.pop_back();
            idx[bestId] = idx.back();
            idx.pop_back();
        }
    }
    fh.coline_off[n] = (int32_t)(fh.coline_pairs.size() / 2);
    fh.edge_idx.clear();
    fh.edge_score.clear();
    for (int p = 0; p < n; p++) c2[p].clear();
    for (uint32_t k = 0; k < lines.size(); k++) {  // :433-441
        if (bad[k]) continue;
        const Cand& cl = lines[k];
        const uint32_t e = (uint32_t)fh.edge_score.size();
        fh.edge_idx.push_back(cl.i);
        fh.edge_idx.push_back(cl.j);
        fh.edge_score.push_back(cl.lscore);
        c2[cl.i].push_back(e);
        c2[cl.j].push_back(e);
    }
    fh.conn_off.assign(n + 1, 0);
    fh.conn_idx.clear();
    for (int p = 0; p < n; p++) {
        fh.conn_off[p] = (int32_t)fh.conn_idx.size();
        for (uint32_t e : c2[p]) fh.conn_idx.push_back(e);
    }
    fh.conn_off[n] = (int32_t)fh.conn_idx.size();
}

}  // namespace
