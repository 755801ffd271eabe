// CPU oracle, post-processing and matcher half (TEST INFRASTRUCTURE ONLY — see ppg_oracle.h).
//
// Line-by-line restatement of
//   feature/src/PPGExtractor.cpp:158-234 (detectKeyPoint), :236-442 (detectLines),
//   :461-513 (heatMapInlierRate / heatMapLineScore), :515-538 (genPointDescriptor),
//   :540-589 (refineHeatMap / bilinearInterpolation);
//   feature/src/MapPoint.cpp:22-29 (DescriptorDistance);
//   map/src/Frame.cpp:138-156, 262-327 and sensors/src/GeometricCamera.cpp:26-61 (grid);
//   matching/src/Matcher.cpp:203-381 (ExtendMapMatches), :582-651 (SearchForInitialization).
// Compiled with -ffp-contract=off and without FMA code generation so that every float
// expression rounds as the reference's (-O3, no -march) build does.  The reference's C++ cannot be
// built here (OpenCV, Eigen, DBoW3, LibTorch absent), so this half is "parity unpinned" against a
// running reference; the third-party pieces (cv::undistortPoints) follow OpenCV's published
// algorithm and reduce to the identity for D = 0, the only case the parity tests use.
#include "ppg_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <numeric>
#include <tuple>
#include <utility>
#include <vector>

namespace {
const double kPi = 3.1415926535897932384626433832795;  // CV_PI
const double k2Pi = 6.283185307179586476925286766559;  // CV_2PI
const float kInvSampleGap[4] = {0.3333f, 0.200f, 0.1427f, 0.1111f};  // PPGExtractor.cpp:19
const float TH_HIGH = 0.8f, TH_LOW = 0.7f;                            // Matcher.cpp:12-13

// cv::undistortPoints(src, dst, K, D(4), noArray(), P = K): 5 fixed-point iterations in double.
void undistort_pinhole(const ppgo_cfg* c, float x, float y, float* ou, float* ov) {
    const double fx = c->K[0], fy = c->K[4], cx = c->K[2], cy = c->K[5];
    const double k1 = c->D[0], k2 = c->D[1], p1 = c->D[2], p2 = c->D[3];
    double xn = (x - cx) * (1.0 / fx), yn = (y - cy) * (1.0 / fy);
    const double x0 = xn, y0 = yn;
    for (int j = 0; j < 5; j++) {
        double r2 = xn * xn + yn * yn;
        double icdist = 1.0 / (1.0 + ((0.0 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) { xn = x0; yn = y0; break; }
        double dX = 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn);
        double dY = p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn;
        xn = (x0 - dX) * icdist;
        yn = (y0 - dY) * icdist;
    }
    *ou = (float)(fx * xn + cx);
    *ov = (float)(fy * yn + cy);
}

// cv::fisheye::undistortPoints(src, dst, K, D, noArray(), P = K): Newton on theta, 10 iterations.
void undistort_fisheye(const ppgo_cfg* c, float x, float y, float* ou, float* ov) {
    const double fx = c->K[0], fy = c->K[4], cx = c->K[2], cy = c->K[5];
    const double k0 = c->D[0], k1 = c->D[1], k2 = c->D[2], k3 = c->D[3];
    double px = (x - cx) / fx, py = (y - cy) / fy;
    double theta_d = std::sqrt(px * px + py * py);
    theta_d = std::min(std::max(-kPi / 2., theta_d), kPi / 2.);
    double scale = 0.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; j++) {
            double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            double k0t2 = k0 * t2, k1t4 = k1 * t4, k2t6 = k2 * t6, k3t8 = k3 * t8;
            double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) /
                         (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
            theta = theta - fix;
            if (std::fabs(fix) < 1e-8) break;
        }
        scale = std::tan(theta) / theta_d;
    } else {
        scale = 1.0;
    }
    double ux = px * scale, uy = py * scale;
    *ou = (float)(fx * ux + cx);
    *ov = (float)(fy * uy + cy);
}

struct Edge {
    unsigned s, e;
    bool bad;
    float lscore;
    unsigned other(unsigned pid) const { return pid == s ? e : s; }
};

struct LineCtx {
    const ppgo_cfg* c;
    const float* heat;
    int W;
    float invScale;
    float at(int y, int x) const { return heat[(size_t)y * W + x]; }
};

// PPGExtractor.cpp:580-589
float bilinear(const LineCtx& L, float ptX, float ptY) {
    int x1 = (int)ptX, x2 = x1 + 1, y1 = (int)ptY, y2 = y1 + 1;
    float d1 = (x2 - ptX) * L.at(y1, x1) + (ptX - x1) * L.at(y1, x2);
    float d2 = (x2 - ptX) * L.at(y2, x1) + (ptX - x1) * L.at(y2, x2);
    return (y2 - ptY) * d1 + (ptY - y1) * d2;
}

inline float norm2(float dx, float dy) { return std::sqrt(dx * dx + dy * dy); }

// PPGExtractor.cpp:484-497
float inlier_rate(const LineCtx& L, const float* ps, const float* pe) {
    float dist = norm2(ps[0] - pe[0], ps[1] - pe[1]);
    int lenLevel = (int)(dist * L.invScale * 4.0);
    int segNum = (int)(dist * kInvSampleGap[lenLevel]);
    float step = (float)(1.0 / (float)segNum);
    int inl = 0;
    for (int i = 1; i < segNum; i++) {
        float sx = ps[0] * step * (float)i + pe[0] * step * (float)(segNum - i);
        float sy = ps[1] * step * (float)i + pe[1] * step * (float)(segNum - i);
        int px = (int)(sx + 0.5), py = (int)(sy + 0.5);
        if (L.at(py, px) > L.c->line_heatmap_thresh) inl++;
    }
    return (float)inl / (float)(segNum - 1);
}

// PPGExtractor.cpp:500-513
float line_score(const LineCtx& L, const float* ps, const float* pe) {
    float dist = norm2(ps[0] - pe[0], ps[1] - pe[1]);
    int lenLevel = (int)(dist * L.invScale * 4.0);
    int segNum = (int)(dist * kInvSampleGap[lenLevel]);
    float step = (float)(1.0 / (float)segNum);
    float sum = 0.f;
    for (int i = 1; i < segNum; i++) {
        float sx = ps[0] * step * (float)i + pe[0] * step * (float)(segNum - i);
        float sy = ps[1] * step * (float)i + pe[1] * step * (float)(segNum - i);
        sum += bilinear(L, sx, sy);
    }
    return sum / (float)(segNum - 1);
}

// PPGExtractor.cpp:540-578 on the sub-rectangle [y0,y1) x [x0,x1)
void refine_tile(const ppgo_cfg* c, float* heat, int W, int y0, int y1, int x0, int x1) {
    std::vector<float> vals;
    const int segH = y1 - y0, segW = x1 - x0;
    for (int i = y0; i < y1; i++)
        for (int j = x0; j < x1; j++)
            if (heat[(size_t)i * W + j] > c->line_valid_thresh) vals.push_back(heat[(size_t)i * W + j]);
    int valCount = (int)(c->line_valid_ratio * vals.size());
    if (valCount < 1) return;
    if (vals.size() >= segH * segW * 0.9 && vals[(size_t)(vals.size() * 0.9)] > 0.1) {
        for (int i = y0; i < y1; i++)
            for (int j = x0; j < x1; j++) heat[(size_t)i * W + j] = 0.f;
        return;
    }
    std::sort(vals.begin(), vals.end(), [](float a, float b) { return a > b; });
    float ave = std::accumulate(vals.begin(), vals.begin() + valCount, 0.0) / (float)valCount;
    for (int i = y0; i < y1; i++)
        for (int j = x0; j < x1; j++) {
            float cur = heat[(size_t)i * W + j];
            if (cur > c->line_valid_thresh) {
                float ns = cur / ave;
                heat[(size_t)i * W + j] = ns > 1.0 ? 1.0 : ns;
            } else {
                heat[(size_t)i * W + j] = 0;
            }
        }
}
}  // namespace

struct ppgo_grid {
    int minX, minY, maxX, maxY;
    float invW, invH;
    int n;
    std::vector<float> pos;
    std::vector<int32_t> cell[64][48];  // FRAME_GRID_COLS x FRAME_GRID_ROWS (GeometricCamera.h:79-80)
};

extern "C" {

void ppgo_default_cfg(ppgo_cfg* c, int width, int height) {
    memset(c, 0, sizeof(*c));
    c->width = width; c->height = height;
    c->K[0] = 458.654f; c->K[4] = 457.296f; c->K[2] = width * 0.5f; c->K[5] = height * 0.5f; c->K[8] = 1.f;
    c->desc_dim = 256;
    c->junction_thresh = 1.0f / 128.0f;
    c->junction_nms_radius = 4;
    c->junction_max_num = 500;
    c->line_valid_thresh = 1.0e-2f;
    c->line_valid_ratio = 0.3f;
    c->line_dist_thresh = 2.0f;
    c->heatmap_refine_sz = 16;
    c->line_heatmap_thresh = 0.2f;
    c->line_inlier_rate = 0.8f;
    c->stable_sort = 1;
    c->math_mode = 1;
}

int ppgo_detect_keypoints(const ppgo_cfg* c, const float* score, int32_t* kp_xy, float* kp_score,
                          float* kp_un, uint8_t* kp_out) {
    const int W = c->width, H = c->height, R = c->junction_nms_radius;
    std::vector<std::tuple<unsigned, unsigned, float>> pts;
    pts.reserve((size_t)(0.2 * H * W));
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            float s = score[(size_t)i * W + j];
            if (s < c->junction_thresh) continue;
            pts.emplace_back(j, i, s);
        }
    if (c->stable_sort) {
        std::sort(pts.begin(), pts.end(), [W](const auto& a, const auto& b) {
            if (std::get<2>(a) != std::get<2>(b)) return std::get<2>(a) > std::get<2>(b);
            return std::get<1>(a) * (unsigned)W + std::get<0>(a) < std::get<1>(b) * (unsigned)W + std::get<0>(b);
        });
    } else {
        std::sort(pts.begin(), pts.end(), [](auto a, auto b) { return std::get<2>(a) > std::get<2>(b); });
    }
    std::vector<unsigned char> flag((size_t)W * H, 0);
    int n = 0;
    for (auto& kp : pts) {
        int px = std::get<0>(kp), py = std::get<1>(kp);
        if (px < R || px > (W - R - 1) || py < R || py > (H - R - 1) || flag[(size_t)py * W + px] != 0) continue;
        flag[(size_t)py * W + px] = 1;
        kp_xy[2 * n] = px; kp_xy[2 * n + 1] = py; kp_score[n] = std::get<2>(kp);
        n++;
        if ((unsigned)n + 1 > c->junction_max_num) break;
        for (int i = py - R; i <= py + R; i++)
            for (int j = px - R; j <= px + R; j++) {
                if (i < 0 || i > H || j < 0 || j > W) continue;
                flag[(size_t)i * W + j] = (unsigned char)-1;
            }
    }
    for (int i = 0; i < n; i++) {
        float u, v;
        if (c->fisheye) undistort_fisheye(c, (float)kp_xy[2 * i], (float)kp_xy[2 * i + 1], &u, &v);
        else undistort_pinhole(c, (float)kp_xy[2 * i], (float)kp_xy[2 * i + 1], &u, &v);
        kp_out[i] = !(u >= 1 && u < W - 1 && v >= 1 && v < H - 1);
        kp_un[2 * i] = u; kp_un[2 * i + 1] = v;
    }
    return n;
}

void ppgo_undistort_maps(const ppgo_cfg* c, float* mapx, float* mapy) {
    const int W = c->width, H = c->height;
    const double fx = c->K[0], fy = c->K[4], u0 = c->K[2], v0 = c->K[5];
    // inverse of the upper-triangular camera matrix
    const double ir[9] = {1.0 / fx, 0.0, -u0 / fx, 0.0, 1.0 / fy, -v0 / fy, 0.0, 0.0, 1.0};
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double u, v;
            if (!c->fisheye) {
                const double k1 = c->D[0], k2 = c->D[1], p1 = c->D[2], p2 = c->D[3], k3 = 0.0;
                const double w = 1. / _w, x = _x * w, y = _y * w;
                const double x2 = x * x, y2 = y * y;
                const double r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
                const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
                const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
                u = fx * xd + u0;
                v = fy * yd + v0;
            } else {
                const double x = _x / _w, y = _y / _w;
                const double r = std::sqrt(x * x + y * y);
                const double theta = std::atan(r);
                const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
                const double theta_d = theta * (1 + c->D[0] * theta2 + c->D[1] * theta4 + c->D[2] * theta6 + c->D[3] * theta8);
                const double scale = (r == 0) ? 1.0 : theta_d / r;
                u = fx * x * scale + u0;
                v = fy * y * scale + v0;
            }
            mapx[(size_t)i * W + j] = (float)u;
            mapy[(size_t)i * W + j] = (float)v;
        }
    }
}

void ppgo_remap_heat(const ppgo_cfg* c, const float* src, float* dst) {
    const int W = c->width, H = c->height;
    std::vector<float> mx((size_t)W * H), my((size_t)W * H);
    ppgo_undistort_maps(c, mx.data(), my.data());
    const int INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS;
    // bilinear weight table, as initInterTab2D builds it for float data
    std::vector<float> tab((size_t)INTER_TAB_SIZE * INTER_TAB_SIZE * 4);
    const float scale = 1.f / INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++) {
            const float vy[2] = {1.f - i * scale, i * scale}, vx[2] = {1.f - j * scale, j * scale};
            float* w = &tab[((size_t)i * INTER_TAB_SIZE + j) * 4];
            for (int k1 = 0; k1 < 2; k1++)
                for (int k2 = 0; k2 < 2; k2++) w[k1 * 2 + k2] = vy[k1] * vx[k2];
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const int sx = (int)std::lrint((double)(mx[p] * (float)INTER_TAB_SIZE));  // cvRound
            const int sy = (int)std::lrint((double)(my[p] * (float)INTER_TAB_SIZE));
            const float* w = &tab[((size_t)(sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1))) * 4];
            const int x0 = sx >> INTER_BITS, y0 = sy >> INTER_BITS;
            float v[4];
            for (int k = 0; k < 4; k++) {
                const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                v[k] = (xx >= 0 && yy >= 0 && xx < W && yy < H) ? src[(size_t)yy * W + xx] : 0.f;
            }
            dst[p] = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
        }
}

void ppgo_refine_heat(const ppgo_cfg* c, float* heat) {
    const int W = c->width, H = c->height, S = c->heatmap_refine_sz;
    const int gx = W / S, gy = H / S;
    for (int i = 0; i < gy; i++)
        for (int j = 0; j < gx; j++) {
            int h0 = i * S, w0 = j * S;
            int h1 = i == (gy - 1) ? H : (i + 1) * S;
            int w1 = j == (gx - 1) ? W : (j + 1) * S;
            refine_tile(c, heat, W, h0, h1, w0, w1);
        }
}

int ppgo_detect_lines(const ppgo_cfg* c, const float* heat, int n, const float* kp_un, const uint8_t* kp_out,
                      uint32_t* edge_idx, float* edge_score, int edge_cap,
                      int32_t* conn_off, uint32_t* conn_idx,
                      int32_t* coline_off, uint32_t* coline_pairs, int coline_cap) {
    const int W = c->width, H = c->height;
    LineCtx L{c, heat, W, 1.0f / std::sqrt((float)(H * H + W * W))};
    const unsigned N = (unsigned)n;
    const int mm = c->math_mode;
    auto sinm = [mm](float x) -> float { return mm ? (float)std::sin((double)x) : std::sin(x); };
    std::vector<float> distMat((size_t)N * N, 0.f), dirMat((size_t)N * N, 0.f);
    auto P = [&](unsigned i, int k) { return kp_un[2 * i + k]; };
    for (unsigned i = 0; i < N; i++) {
        if (kp_out[i]) continue;
        for (unsigned j = i + 1; j < N; j++) {
            if (kp_out[j]) continue;
            float vx = P(j, 0) - P(i, 0), vy = P(j, 1) - P(i, 1);
            float dist = norm2(vx, vy);
            vx = vx / dist; vy = vy / dist;
            distMat[(size_t)i * N + j] = dist;
            distMat[(size_t)j * N + i] = dist;
            float d = mm ? (float)std::atan2((double)vy, (double)vx) : std::atan2(vy, vx);
            dirMat[(size_t)i * N + j] = d;
            float r = (float)(d - kPi);
            if (r < -kPi) r = (float)(r + k2Pi);
            dirMat[(size_t)j * N + i] = r;
        }
    }
    std::vector<Edge> cand;
    std::vector<std::vector<unsigned>> conn(N);
    const float T = c->line_heatmap_thresh;
    auto overlap_pass = [&](unsigned a, unsigned b) -> bool {
        // PPGExtractor.cpp:316-335 with (i,j) = (a,b); :338-357 with (a,b) = (j,i)
        bool isOverlap = false;
        for (unsigned lid : conn[a]) {
            Edge& old = cand[lid];
            if (old.bad) continue;
            unsigned pid_old = old.other(a);
            float ad = dirMat[(size_t)a * N + b] - dirMat[(size_t)a * N + pid_old];
            if (ad < -kPi) ad = (float)(ad + k2Pi);
            if (ad > kPi) ad = (float)(ad - k2Pi);
            ad = mm == 2 ? (float)std::abs((int)ad) : std::fabs(ad);
            if (ad > 0.2 * kPi) continue;
            float dNew = distMat[(size_t)a * N + b], dOld = distMat[(size_t)a * N + pid_old];
            float s = sinm(ad);
            if (dNew <= dOld && dNew * s < c->line_dist_thresh) old.bad = true;
            if (dOld < dNew && dOld * s < c->line_dist_thresh) isOverlap = true;
        }
        return isOverlap;
    };
    for (unsigned i = 0; i < N; i++) {
        if (kp_out[i]) continue;
        for (unsigned j = i + 1; j < N; j++) {
            if (kp_out[j]) continue;
            float c1x = P(j, 0) * 0.2f + P(i, 0) * 0.8f, c1y = P(j, 1) * 0.2f + P(i, 1) * 0.8f;
            float c2x = P(j, 0) * 0.8f + P(i, 0) * 0.2f, c2y = P(j, 1) * 0.8f + P(i, 1) * 0.2f;
            float c3x = P(j, 0) * 0.5f + P(i, 0) * 0.5f, c3y = P(j, 1) * 0.5f + P(i, 1) * 0.5f;
            if (L.at((int)(c1y + 0.5), (int)(c1x + 0.5)) < T) continue;
            if (L.at((int)(c2y + 0.5), (int)(c2x + 0.5)) < T) continue;
            if (L.at((int)(c3y + 0.5), (int)(c3x + 0.5)) < T) continue;
            if (overlap_pass(i, j)) continue;
            if (overlap_pass(j, i)) continue;
            cand.push_back(Edge{i, j, false, 0.f});
            unsigned lid = (unsigned)cand.size() - 1;
            conn[i].push_back(lid);
            conn[j].push_back(lid);
        }
    }
    for (auto& v : conn) v.clear();
    for (unsigned i = 0; i < cand.size(); i++) {
        Edge& kl = cand[i];
        if (kl.bad) continue;
        const float* ps = &kp_un[2 * kl.s];
        const float* pe = &kp_un[2 * kl.e];
        float si = inlier_rate(L, ps, pe);
        if (si < c->line_inlier_rate) { kl.bad = true; continue; }
        float sh = line_score(L, ps, pe);
        if (sh < c->line_heatmap_thresh) { kl.bad = true; continue; }
        kl.lscore = si * sh;
        conn[kl.s].push_back(i);
        conn[kl.e].push_back(i);
    }
    // collinear pairs, PPGExtractor.cpp:392-432
    int ncol = 0;
    for (unsigned p = 0; p < N; p++) {
        coline_off[p] = ncol;
        std::vector<unsigned> idx = conn[p];
        while (idx.size() > 1) {
            double minDiff = 1e9;
            int b1 = -1, b2 = -1, bestId = -1;
            Edge& kl1 = cand[idx.back()];
            if (kl1.bad) { idx.pop_back(); continue; }
            for (unsigned i = 0; i < idx.size() - 1; i++) {
                Edge& kl2 = cand[idx[i]];
                if (kl2.bad) continue;
                unsigned p1 = kl1.other(p), p2 = kl2.other(p);
                float ad = dirMat[(size_t)p * N + p1] - dirMat[(size_t)p * N + p2];
                double pd = 0.5 * (distMat[(size_t)p * N + p1] + distMat[(size_t)p * N + p2]) * std::abs(sinm(ad));
                if (minDiff > pd) { minDiff = pd; bestId = (int)i; b1 = (int)p1; b2 = (int)p2; }
            }
            if (minDiff > c->line_dist_thresh) { idx.pop_back(); continue; }
            if (ncol >= coline_cap) return -1;
            coline_pairs[2 * ncol] = (uint32_t)b1; coline_pairs[2 * ncol + 1] = (uint32_t)b2;
            ncol++;
            idx.pop_back();
            idx[bestId] = idx.back();
            idx.pop_back();
        }
    }
    coline_off[N] = ncol;
    // compact, PPGExtractor.cpp:433-441
    int E = 0;
    std::vector<std::vector<unsigned>> out_conn(N);
    for (Edge& cl : cand) {
        if (cl.bad) continue;
        if (E >= edge_cap) return -1;
        edge_idx[2 * E] = cl.s; edge_idx[2 * E + 1] = cl.e; edge_score[E] = cl.lscore;
        out_conn[cl.s].push_back(E);
        out_conn[cl.e].push_back(E);
        E++;
    }
    int k = 0;
    for (unsigned p = 0; p < N; p++) {
        conn_off[p] = k;
        for (unsigned e : out_conn[p]) conn_idx[k++] = e;
    }
    conn_off[N] = k;
    return E;
}

void ppgo_gen_descriptors(const ppgo_cfg* c, const float* dm, int n, const int32_t* kp_xy, float* out) {
    const int D = c->desc_dim;
    if (n < 10) { memset(out, 0, sizeof(float) * (size_t)n * D); return; }
    const int W = c->width, H = c->height, w = W / 8, h = H / 8;
    for (int i = 0; i < n; i++) {
        // PPGExtractor.cpp:530-531: f32 / f32, then * 2. - 1. in double, stored as f32
        float gx = (float)((double)((float)kp_xy[2 * i] / (float)W) * 2. - 1.);
        float gy = (float)((double)((float)kp_xy[2 * i + 1] / (float)H) * 2. - 1.);
        // ATen grid_sampler_2d, bilinear, zeros padding, align_corners = false
        float ix = ((gx + 1.f) * w - 1.f) / 2.f, iy = ((gy + 1.f) * h - 1.f) / 2.f;
        float fx = std::floor(ix), fy = std::floor(iy);
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
        float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
        auto in = [&](int y, int x) { return x >= 0 && x < w && y >= 0 && y < h; };
        float* o = out + (size_t)i * D;
        double ss = 0.0;
        for (int k = 0; k < D; k++) {
            const float* p = dm + (size_t)k * h * w;
            float v = 0.f;
            if (in(y0, x0)) v += p[y0 * w + x0] * nw;
            if (in(y0, x1)) v += p[y0 * w + x1] * ne;
            if (in(y1, x0)) v += p[y1 * w + x0] * sw;
            if (in(y1, x1)) v += p[y1 * w + x1] * se;
            o[k] = v;
            ss += (double)v * v;
        }
        float nrm = (float)std::sqrt(ss);
        float den = nrm > 1e-12f ? nrm : 1e-12f;
        for (int k = 0; k < D; k++) o[k] = o[k] / den;
    }
}

float ppgo_descriptor_distance(const float* a, const float* b, int dim) {
    float acc = 0.f;
    for (int k = 0; k < dim; k++) {
        float d = a[k] - b[k];
        acc = std::fmaf(d, d, acc);
    }
    return std::sqrt(acc);
}

ppgo_grid* ppgo_grid_build(const ppgo_cfg* c, int n, const float* kp_pos) {
    ppgo_grid* g = new ppgo_grid();
    if (!c->fisheye) {
        float cs[4][2] = {{0.f, 0.f}, {(float)c->width, 0.f}, {0.f, (float)c->height}, {(float)c->width, (float)c->height}};
        float u[4][2];
        for (int k = 0; k < 4; k++) undistort_pinhole(c, cs[k][0], cs[k][1], &u[k][0], &u[k][1]);
        g->minX = (int)std::min(u[0][0], u[2][0]);
        g->maxX = (int)std::max(u[1][0], u[3][0]);
        g->minY = (int)std::min(u[0][1], u[1][1]);
        g->maxY = (int)std::max(u[2][1], u[3][1]);
    } else {
        g->minX = 0; g->minY = 0; g->maxX = c->width; g->maxY = c->height;
    }
    g->invW = 64.0f / (float)(g->maxX - g->minX);
    g->invH = 48.0f / (float)(g->maxY - g->minY);
    g->n = n;
    g->pos.assign(kp_pos, kp_pos + 2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        int px = (int)std::round((kp_pos[2 * i] - g->minX) * g->invW);
        int py = (int)std::round((kp_pos[2 * i + 1] - g->minY) * g->invH);
        if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
        g->cell[px][py].push_back(i);
    }
    return g;
}

void ppgo_grid_free(ppgo_grid* g) { delete g; }

int ppgo_features_in_area(const ppgo_grid* g, float x, float y, float r, int32_t* out, int cap) {
    int cnt = 0;
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
                float dx = g->pos[2 * id] - x, dy = g->pos[2 * id + 1] - y;
                if (std::fabs(dx) < r && std::fabs(dy) < r) {
                    if (cnt < cap) out[cnt] = id;
                    cnt++;
                }
            }
    return cnt;
}

int ppgo_search_for_initialization(const ppgo_cfg* c, float nnratio, int n1, const float* desc1,
                                   int n2, const float* desc2, const float* pos2,
                                   float* prev, int window, int32_t* m12) {
    const int D = c->desc_dim;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) m12[i] = -1;
    std::vector<int> matchedDist(n2, INT_MAX);  // vector<int> on purpose (Matcher.cpp:588)
    std::vector<int> m21(n2, -1);
    ppgo_grid* g = window > 0 ? ppgo_grid_build(c, n2, pos2) : nullptr;
    std::vector<int32_t> idx(n2);
    for (int i1 = 0; i1 < n1; i1++) {
        int cnt;
        if (g) cnt = ppgo_features_in_area(g, prev[2 * i1], prev[2 * i1 + 1], (float)window, idx.data(), n2);
        else { cnt = n2; for (int k = 0; k < n2; k++) idx[k] = k; }
        if (cnt == 0) continue;
        const float* d1 = desc1 + (size_t)i1 * D;
        float best = 1e6f, best2 = 1e6f;
        int bestIdx = -1;
        for (int k = 0; k < cnt; k++) {
            int i2 = idx[k];
            float dist = ppgo_descriptor_distance(d1, desc2 + (size_t)i2 * D, D);
            if (matchedDist[i2] <= dist) continue;
            if (dist < best) { best2 = best; best = dist; bestIdx = i2; }
            else if (dist < best2) best2 = dist;
        }
        if (best <= TH_LOW) {
            if (best < (float)best2 * nnratio) {
                if (m21[bestIdx] >= 0) { m12[m21[bestIdx]] = -1; nmatches--; }
                m12[i1] = bestIdx;
                m21[bestIdx] = i1;
                matchedDist[bestIdx] = (int)best;
                nmatches++;
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (m12[i1] >= 0) { prev[2 * i1] = pos2[2 * m12[i1]]; prev[2 * i1 + 1] = pos2[2 * m12[i1] + 1]; }
    if (g) ppgo_grid_free(g);
    return nmatches;
}

int ppgo_window_nn(const float* q, int dim, const float* desc, const int32_t* cand, int ncand,
                   const uint8_t* skip, float* best, float* second) {
    float b = 1e6f, b2 = 1e6f;
    int bi = -1;
    for (int k = 0; k < ncand; k++) {
        int id = cand[k];
        if (skip && skip[id]) continue;
        float d = ppgo_descriptor_distance(q, desc + (size_t)id * dim, dim);
        if (d < b) { b2 = b; b = d; bi = id; }
        else if (d < b2) b2 = d;
    }
    *best = b; *second = b2;
    return bi;
}

namespace {
// Tcw * p, Pinhole::project (sensors/src/Pinhole.cpp:32-38), GeometricCamera::IsInImage (GeometricCamera.cpp:21-24)
inline void se3_apply(const ppgo_pose* T, const float* p, float* o) {
    for (int r = 0; r < 3; r++) o[r] = T->R[3 * r] * p[0] + T->R[3 * r + 1] * p[1] + T->R[3 * r + 2] * p[2] + T->t[r];
}
inline void cam_center(const ppgo_pose* T, float* ow) {  // Tcw.inverse().translation() = -R^T t
    for (int r = 0; r < 3; r++) ow[r] = -(T->R[r] * T->t[0] + T->R[3 + r] * T->t[1] + T->R[6 + r] * T->t[2]);
}
// mpCamera->project: Pinhole::project (sensors/src/Pinhole.cpp:32-38) or, for a fisheye rig, KannalaBrandt8::project
// (sensors/src/KannalaBrandt8.cpp:43-59; the unqualified cos / sin of :57-58 are the double functions)
inline void pinhole_project(const ppgo_cfg* c, const float* p, float* uv) {
    if (c->fisheye) {
        const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
        const float theta = atan2f(sqrtf(x2_plus_y2), p[2]);
        const float psi = atan2f(p[1], p[0]);
        const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                    theta9 = theta7 * theta2;
        const float r = theta + c->D[0] * theta3 + c->D[1] * theta5 + c->D[2] * theta7 + c->D[3] * theta9;
        uv[0] = (float)((double)(c->K[0] * r) * std::cos((double)psi) + (double)c->K[2]);
        uv[1] = (float)((double)(c->K[4] * r) * std::sin((double)psi) + (double)c->K[5]);
        return;
    }
    uv[0] = c->K[0] * p[0] / p[2] + c->K[2];
    uv[1] = c->K[4] * p[1] / p[2] + c->K[5];
}
inline bool in_image(const ppgo_grid* g, float x, float y) {
    return x >= (float)g->minX && x < (float)g->maxX && y >= (float)g->minY && y < (float)g->maxY;
}
// the gates shared by :501-531, :931-969, :1060-1090
inline bool project_gates(const ppgo_cfg* c, const ppgo_grid* g, const ppgo_pose* Tcw, const float* Ow,
                          const ppgo_mappoints* mp, int id, float* uv) {
    const float* pw = mp->pos + 3 * (size_t)id;
    float pc[3];
    se3_apply(Tcw, pw, pc);
    if (pc[2] < 0.0f) return false;
    pinhole_project(c, pc, uv);
    if (!in_image(g, uv[0], uv[1])) return false;
    const float PO[3] = {pw[0] - Ow[0], pw[1] - Ow[1], pw[2] - Ow[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < mp->min_dist[id] || dist > mp->max_dist[id]) return false;
    const float* Pn = mp->normal + 3 * (size_t)id;
    const float dot = PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2];
    if ((double)dot < 0.5 * (double)dist) return false;
    return true;
}
}  // namespace

void ppgo_check_in_frustum(const ppgo_cfg* c, const ppgo_pose* Tcw, const ppgo_mappoints* mp, float viewingCosLimit,
                           uint8_t* in_view, float* proj, float* depth, float* view_cos) {
    ppgo_grid* g = ppgo_grid_build(c, 0, nullptr);  // image bounds only
    float mOw[3];
    cam_center(Tcw, mOw);
    for (int i = 0; i < mp->m; i++) {
        in_view[i] = 0;          // mbTrackInView = false
        proj[2 * i] = -1;        // mTrackProjX
        proj[2 * i + 1] = -1;    // mTrackProjY
        depth[i] = -1;           // mTrackDepth
        const float* P = mp->pos + 3 * (size_t)i;
        float Pc[3], uv[2];
        se3_apply(Tcw, P, Pc);
        if (Pc[2] < 0.0f) continue;
        pinhole_project(c, Pc, uv);
        if (!in_image(g, uv[0], uv[1])) continue;
        const float PO[3] = {P[0] - mOw[0], P[1] - mOw[1], P[2] - mOw[2]};
        const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        if (dist < mp->min_dist[i] || dist > mp->max_dist[i]) continue;
        const float* Pn = mp->normal + 3 * (size_t)i;
        const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
        if (viewCos < viewingCosLimit) continue;
        in_view[i] = 1;
        proj[2 * i] = uv[0];
        proj[2 * i + 1] = uv[1];
        depth[i] = dist;
        view_cos[i] = viewCos;
    }
    ppgo_grid_free(g);
}

int ppgo_search_by_projection_frames(const ppgo_cfg* c, const ppgo_pose* Tcw, const ppgo_mappoints* mp,
                                     int nl, const int32_t* last_mp, const uint8_t* last_outlier,
                                     int nc, const float* cur_pos, const float* cur_desc, int32_t* cur_mp, float th) {
    const int D = c->desc_dim;
    const float TH_HIGH = 0.8f;
    ppgo_grid* g = ppgo_grid_build(c, nc, cur_pos);
    std::vector<int32_t> idx(std::max(nc, 1));
    int nmatches = 0;
    for (int i = 0; i < nl; i++) {
        const int id = last_mp[i];
        if (id < 0 || last_outlier[i]) continue;
        float pc[3], uv[2];
        se3_apply(Tcw, mp->pos + 3 * (size_t)id, pc);
        const float invzc = (float)(1.0 / (double)pc[2]);  // :46: float <- double division
        if (invzc < 0) continue;
        pinhole_project(c, pc, uv);
        if (!in_image(g, uv[0], uv[1])) continue;
        const int cnt = std::min(ppgo_features_in_area(g, uv[0], uv[1], th, idx.data(), nc), nc);
        if (cnt == 0) continue;
        float bestDist = 1e6f;
        int bestIdx2 = -1;
        for (int k = 0; k < cnt; k++) {
            const int i2 = idx[k];
            if (cur_mp[i2] >= 0 && mp->nobs[cur_mp[i2]] > 0) continue;
            const float dist = ppgo_descriptor_distance(mp->desc + (size_t)id * D, cur_desc + (size_t)i2 * D, D);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) { cur_mp[bestIdx2] = id; nmatches++; }
    }
    ppgo_grid_free(g);
    return nmatches;
}

int ppgo_search_by_projection_sim3(const ppgo_cfg* c, const ppgo_pose* Tcw, const ppgo_mappoints* mp,
                                   int np, const int32_t* points,
                                   int nk, const float* kf_pos, const float* kf_desc, int32_t* matched,
                                   int th, float ratio_hamming) {
    const int D = c->desc_dim;
    const float TH_LOW = 0.7f;
    float Ow[3];
    cam_center(Tcw, Ow);
    std::vector<uint8_t> found(std::max(mp->m, 1), 0);  // spAlreadyFound, :487-488 (a snapshot)
    for (int k = 0; k < nk; k++)
        if (matched[k] >= 0) found[matched[k]] = 1;
    ppgo_grid* g = ppgo_grid_build(c, nk, kf_pos);
    std::vector<int32_t> idx(std::max(nk, 1));
    int nmatches = 0;
    for (int iMP = 0; iMP < np; iMP++) {
        const int id = points[iMP];
        if (mp->bad[id] || found[id]) continue;
        float uv[2];
        if (!project_gates(c, g, Tcw, Ow, mp, id, uv)) continue;
        const int cnt = std::min(ppgo_features_in_area(g, uv[0], uv[1], (float)th, idx.data(), nk), nk);
        if (cnt == 0) continue;
        float bestDist = 1e6f;
        int bestIdx = -1;
        for (int k = 0; k < cnt; k++) {
            const int i2 = idx[k];
            if (matched[i2] >= 0) continue;
            const float dist = ppgo_descriptor_distance(mp->desc + (size_t)id * D, kf_desc + (size_t)i2 * D, D);
            if (dist < bestDist) { bestDist = dist; bestIdx = i2; }
        }
        if (bestDist <= TH_LOW * ratio_hamming) { matched[bestIdx] = id; nmatches++; }
    }
    ppgo_grid_free(g);
    return nmatches;
}

int ppgo_fuse_match(const ppgo_cfg* c, const ppgo_pose* Tcw, const ppgo_mappoints* mp,
                    int np, const int32_t* points, const uint8_t* in_kf,
                    int nk, const float* kf_pos, const float* kf_desc, float th, int with_e2, int32_t* best_idx) {
    const int D = c->desc_dim;
    const float TH_LOW = 0.7f;
    float Ow[3];
    cam_center(Tcw, Ow);
    ppgo_grid* g = ppgo_grid_build(c, nk, kf_pos);
    std::vector<int32_t> idx(std::max(nk, 1));
    int nFused = 0;
    for (int i = 0; i < np; i++) {
        best_idx[i] = -1;
        const int id = points[i];
        if (id < 0) continue;
        if (mp->bad[id]) continue;
        if (in_kf[i]) continue;
        float uv[2];
        if (!project_gates(c, g, Tcw, Ow, mp, id, uv)) continue;
        const int cnt = std::min(ppgo_features_in_area(g, uv[0], uv[1], th, idx.data(), nk), nk);
        if (cnt == 0) continue;
        float bestDist = 1e6f;
        int bestIdx = -1;
        for (int k = 0; k < cnt; k++) {
            const int i2 = idx[k];
            if (with_e2) {
                const float ex = uv[0] - kf_pos[2 * i2], ey = uv[1] - kf_pos[2 * i2 + 1];
                const float e2 = ex * ex + ey * ey;
                if ((double)e2 > 5.99) continue;
            }
            const float dist = ppgo_descriptor_distance(mp->desc + (size_t)id * D, kf_desc + (size_t)i2 * D, D);
            if (dist < bestDist) { bestDist = dist; bestIdx = i2; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; nFused++; }
    }
    ppgo_grid_free(g);
    return nFused;
}

namespace {
// std::map<NodeId, vector<unsigned>>::lower_bound on the flattened form
inline int fv_lower_bound(const ppgo_featvec* fv, uint32_t key) {
    int k = 0;
    while (k < fv->nn && fv->node[k] < key) k++;
    return k;
}
}  // namespace

int ppgo_search_by_bow_frame(const ppgo_cfg* c, float nnratio, const ppgo_featvec* fvKF, const ppgo_featvec* fvF,
                             int nKF, const float* descKF, const int32_t* mpKF, const uint8_t* bad,
                             int nF, const float* descF, int32_t* out) {
    (void)nKF;
    const int D = c->desc_dim;
    const float TH_LOW = 0.7f;
    for (int i = 0; i < nF; i++) out[i] = -1;
    int nmatches = 0;
    int KFit = 0, Fit = 0;
    while (KFit != fvKF->nn && Fit != fvF->nn) {
        if (fvKF->node[KFit] == fvF->node[Fit]) {
            for (int iKF = fvKF->off[KFit]; iKF < fvKF->off[KFit + 1]; iKF++) {
                const unsigned realIdxKF = fvKF->idx[iKF];
                const int pMP = mpKF[realIdxKF];
                if (pMP < 0) continue;
                if (bad[pMP]) continue;
                const float* dKF = descKF + (size_t)realIdxKF * D;
                float bestDist1 = 1e6f, bestDist2 = 1e6f;
                int bestIdxF = -1;
                for (int iF = fvF->off[Fit]; iF < fvF->off[Fit + 1]; iF++) {
                    const unsigned realIdxF = fvF->idx[iF];
                    if (out[realIdxF] >= 0) continue;
                    const float dist = ppgo_descriptor_distance(dKF, descF + (size_t)realIdxF * D, D);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (bestDist1 < nnratio * bestDist2) { out[bestIdxF] = pMP; nmatches++; }
                }
            }
            KFit++; Fit++;
        } else if (fvKF->node[KFit] < fvF->node[Fit]) {
            KFit = fv_lower_bound(fvKF, fvF->node[Fit]);
        } else {
            Fit = fv_lower_bound(fvF, fvKF->node[KFit]);
        }
    }
    return nmatches;
}

int ppgo_search_by_bow_keyframes(const ppgo_cfg* c, float nnratio, const ppgo_featvec* fv1, const ppgo_featvec* fv2,
                                 int n1, const float* desc1, const int32_t* mp1,
                                 int n2, const float* desc2, const int32_t* mp2, const uint8_t* bad, int32_t* out) {
    const int D = c->desc_dim;
    const float TH_LOW = 0.7f;
    for (int i = 0; i < n1; i++) out[i] = -1;
    std::vector<bool> vbMatched2(std::max(n2, 1), false);
    int nmatches = 0;
    int f1it = 0, f2it = 0;
    while (f1it != fv1->nn && f2it != fv2->nn) {
        if (fv1->node[f1it] == fv2->node[f2it]) {
            for (int i1 = fv1->off[f1it]; i1 < fv1->off[f1it + 1]; i1++) {
                const size_t idx1 = fv1->idx[i1];
                const int pMP1 = mp1[idx1];
                if (pMP1 < 0) continue;
                if (bad[pMP1]) continue;
                const float* d1 = desc1 + idx1 * D;
                float bestDist1 = 1e6f, bestDist2 = 1e6f;
                int bestIdx2 = -1;
                for (int i2 = fv2->off[f2it]; i2 < fv2->off[f2it + 1]; i2++) {
                    const size_t idx2 = fv2->idx[i2];
                    const int pMP2 = mp2[idx2];
                    if (vbMatched2[idx2] || pMP2 < 0) continue;
                    if (bad[pMP2]) continue;
                    float dist = ppgo_descriptor_distance(d1, desc2 + idx2 * D, D);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if (bestDist1 < nnratio * bestDist2) {
                        out[idx1] = mp2[bestIdx2];
                        vbMatched2[bestIdx2] = true;
                        nmatches++;
                    }
                }
            }
            f1it++; f2it++;
        } else if (fv1->node[f1it] < fv2->node[f2it]) {
            f1it = fv_lower_bound(fv1, fv2->node[f2it]);
        } else {
            f2it = fv_lower_bound(fv2, fv1->node[f1it]);
        }
    }
    return nmatches;
}

int ppgo_search_for_triangulation(const ppgo_cfg* c, const ppgo_pose* T1w, const ppgo_pose* T2w,
                                  const ppgo_featvec* fv1, const ppgo_featvec* fv2,
                                  int n1, const float* desc1, const float* pos1, const int32_t* mp1,
                                  int n2, const float* desc2, const float* pos2, const int32_t* mp2, int32_t* pairs) {
    (void)n2;
    const int D = c->desc_dim;
    const float TH_LOW = 0.7f;
    // epipole in KF2 (:779-782) and the relative pose T12 = T1w * Tw2 (:785-787)
    float Cw[3], C2[3], ep[2];
    cam_center(T1w, Cw);
    se3_apply(T2w, Cw, C2);
    pinhole_project(c, C2, ep);
    float R12[9], t12[3];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++)
            R12[3 * r + q] = T1w->R[3 * r] * T2w->R[3 * q] + T1w->R[3 * r + 1] * T2w->R[3 * q + 1] + T1w->R[3 * r + 2] * T2w->R[3 * q + 2];
    for (int r = 0; r < 3; r++) t12[r] = T1w->t[r] - (R12[3 * r] * T2w->t[0] + R12[3 * r + 1] * T2w->t[1] + R12[3 * r + 2] * T2w->t[2]);
    // F12 = K1^-T * t12x * R12 * K2^-1 (sensors/src/Pinhole.cpp:101-104); Eigen's inverse() is replaced by the closed
    // form of the upper-triangular K, products left to right
    const float fx = c->K[0], fy = c->K[4], cx = c->K[2], cy = c->K[5];
    const float Ki[9] = {1.f / fx, 0.f, -cx / fx, 0.f, 1.f / fy, -cy / fy, 0.f, 0.f, 1.f};
    const float tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
    auto mul = [](const float* X, const float* Y, float* Z) {
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) Z[3 * r + q] = X[3 * r] * Y[q] + X[3 * r + 1] * Y[3 + q] + X[3 * r + 2] * Y[6 + q];
    };
    float A[9], B[9], F12[9];
    mul(tx, R12, A);
    mul(A, Ki, B);
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) F12[3 * r + q] = Ki[r] * B[q] + Ki[3 + r] * B[3 + q] + Ki[6 + r] * B[6 + q];

    int nmatches = 0;
    std::vector<int> vMatches12(std::max(n1, 1), -1);
    int f1it = 0, f2it = 0;
    while (f1it != fv1->nn && f2it != fv2->nn) {
        if (fv1->node[f1it] == fv2->node[f2it]) {
            for (int i1 = fv1->off[f1it]; i1 < fv1->off[f1it + 1]; i1++) {
                const size_t idx1 = fv1->idx[i1];
                if (mp1[idx1] >= 0) continue;
                const float* kp1 = pos1 + 2 * idx1;
                const float* d1 = desc1 + idx1 * D;
                float bestDist = TH_LOW;
                int bestIdx2 = -1;
                for (int i2 = fv2->off[f2it]; i2 < fv2->off[f2it + 1]; i2++) {
                    size_t idx2 = fv2->idx[i2];
                    if (mp2[idx2] >= 0) continue;  // vbMatched2 is never set (:794)
                    const float dist = ppgo_descriptor_distance(d1, desc2 + idx2 * D, D);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const float* kp2 = pos2 + 2 * idx2;
                    const float ex = ep[0] - kp2[0], ey = ep[1] - kp2[1];
                    if (std::sqrt(ex * ex + ey * ey) < 10.0f) continue;
                    // Pinhole::epipolarConstrain
                    const float a = kp1[0] * F12[0] + kp1[1] * F12[3] + F12[6];
                    const float b = kp1[0] * F12[1] + kp1[1] * F12[4] + F12[7];
                    const float cc = kp1[0] * F12[2] + kp1[1] * F12[5] + F12[8];
                    const float num = a * kp2[0] + b * kp2[1] + cc;
                    const float den = a * a + b * b;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if ((double)dsqr < 3.84) { bestIdx2 = (int)idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) { vMatches12[idx1] = bestIdx2; nmatches++; }
            }
            f1it++; f2it++;
        } else if (fv1->node[f1it] < fv2->node[f2it]) {
            f1it = fv_lower_bound(fv1, fv2->node[f2it]);
        } else {
            f2it = fv_lower_bound(fv2, fv1->node[f1it]);
        }
    }
    int k = 0;
    for (int i = 0; i < n1; i++) {
        if (vMatches12[i] < 0) continue;
        pairs[2 * k] = i; pairs[2 * k + 1] = vMatches12[i];
        k++;
    }
    return nmatches;
}

int ppgo_distinctive_descriptor(const float* desc, int n, int dim) {
    if (n <= 0) return 0;
    const size_t N = (size_t)n;
    std::vector<float> Distances(N * N);
    for (size_t i = 0; i < N; i++) {
        Distances[i * N + i] = 0;
        for (size_t j = i + 1; j < N; j++) {
            float distij = ppgo_descriptor_distance(desc + i * dim, desc + j * dim, dim);
            Distances[i * N + j] = distij;
            Distances[j * N + i] = distij;
        }
    }
    float BestMedian = 1.0f;
    int BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
        std::vector<float> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
        std::sort(vDists.begin(), vDists.end());
        float median = vDists[(size_t)(0.5 * (N - 1))];
        if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
    }
    return BestIdx;
}

int ppgo_extend_map_matches(const ppgo_cfg* c, float nnratio, float th, const ppgo_mapgraph* g,
                            int n, const float* desc, const float* pos_un,
                            int n_edges, const uint32_t* edge_idx,
                            const int32_t* conn_off, const uint32_t* conn_idx,
                            uint8_t* tracked, int32_t* frame_mp, int32_t* frame_me) {
    (void)n_edges;
    const int D = c->desc_dim;
    int nmatches = 0;
    std::vector<int> cands;
    for (int i = 0; i < g->m; i++) {
        if (g->bad[i] || !g->in_view[i]) continue;
        cands.push_back(i);
    }
    std::sort(cands.begin(), cands.end(), [g](int a, int b) {
        return (g->edge_off[a + 1] - g->edge_off[a]) > (g->edge_off[b + 1] - g->edge_off[b]);
    });
    ppgo_grid* grid = ppgo_grid_build(c, n, pos_un);
    std::vector<int32_t> idx(n);
    auto key_other = [&](unsigned e, unsigned pid) { return edge_idx[2 * e] == pid ? edge_idx[2 * e + 1] : edge_idx[2 * e]; };
    for (int mp : cands) {
        if (tracked[mp] || g->bad[mp]) continue;
        const float* dmp = g->desc + (size_t)mp * D;
        float best = 1e6f, best2 = 1e6f;
        int bestIdx = -1;
        float r = th;
        if (g->viewcos[mp] > 0.998) r *= 2.5; else r *= 4.0;
        int cnt = ppgo_features_in_area(grid, g->proj[2 * mp], g->proj[2 * mp + 1], r, idx.data(), n);
        if (cnt == 0) continue;
        for (int k = 0; k < cnt; k++) {
            int id = idx[k];
            if (frame_mp[id] >= 0 && g->nobs[frame_mp[id]] > 0) continue;
            float d = ppgo_descriptor_distance(dmp, desc + (size_t)id * D, D);
            if (d < best) { best2 = best; best = d; bestIdx = id; }
            else if (d < best2) best2 = d;
        }
        if (best > TH_HIGH && best > nnratio * best2) continue;
        // every candidate of the window was skipped and nnratio >= 1 lets 1e6 > nnratio * 1e6 fail: the reference indexes
        // with -1 here (Matcher.cpp:279, undefined behaviour); the library and this statement skip the map point
        if (bestIdx < 0) continue;
        frame_mp[bestIdx] = mp;
        tracked[mp] = 1;
        nmatches++;
        std::deque<unsigned> seed;
        seed.push_back((unsigned)bestIdx);
        while (!seed.empty()) {
            unsigned keyID = seed.front();
            seed.pop_front();
            const int me0 = g->edge_off[mp], nme = g->edge_off[mp + 1] - me0;  // pMP stays the first seed's point
            const int ke0 = conn_off[keyID], nke = conn_off[keyID + 1] - ke0;
            if (nme == 0 || nke == 0) continue;
            std::vector<float> weight((size_t)nme * nke, 1e6f);
            std::vector<unsigned> lx, ly;
            for (int i = 0; i < nme; i++) {
                if (!g->edge_ok[me0 + i] || g->edge_other[me0 + i] < 0) continue;
                lx.push_back(i);
            }
            for (int j = 0; j < nke; j++) ly.push_back(j);
            for (unsigned i : lx)
                for (unsigned j : ly) {
                    int mpo = g->edge_other[me0 + i];
                    unsigned ko = key_other(conn_idx[ke0 + j], keyID);
                    if (mpo == frame_mp[ko]) weight[(size_t)i * nke + j] = -1;
                    else weight[(size_t)i * nke + j] =
                             ppgo_descriptor_distance(g->desc + (size_t)mpo * D, desc + (size_t)ko * D, D);
                }
            while (!lx.empty() && !ly.empty()) {
                unsigned minlx = 0, minly = 0;
                float minW = 1e6f;
                for (unsigned i = 0; i < lx.size(); i++)
                    for (unsigned j = 0; j < ly.size(); j++)
                        if (weight[(size_t)lx[i] * nke + ly[j]] < minW) {
                            minW = weight[(size_t)lx[i] * nke + ly[j]];
                            minlx = i; minly = j;
                        }
                if (minW > TH_HIGH) break;
                unsigned mi = lx[minlx], kj = ly[minly];
                lx.erase(lx.begin() + minlx);
                ly.erase(ly.begin() + minly);
                int mpo = g->edge_other[me0 + mi];
                unsigned keyEdge = conn_idx[ke0 + kj];
                unsigned ko = key_other(keyEdge, keyID);
                if (mpo < 0 || g->bad[mpo] || tracked[mpo]) continue;
                frame_mp[ko] = mpo;
                frame_me[keyEdge] = g->edge_id[me0 + mi];
                tracked[mpo] = 1;
                seed.push_back(ko);
            }
        }
        nmatches++;
    }
    ppgo_grid_free(grid);
    return nmatches;
}

void ppgo_bow_transform(const float* desc, int n, int dim, const float* node_desc, const int32_t* child_off,
                        const int32_t* child_idx, const int32_t* word_id, const double* weight, int L, int levelsup,
                        int32_t* out_word, double* out_weight, int32_t* out_node) {
    const int nid_level = L - levelsup;
    for (int f = 0; f < n; f++) {
        const float* x = desc + (size_t)f * dim;
        int final_id = 0, current_level = 0, nid = 0;   // nid_level <= 0: the root
        do {
            ++current_level;
            double best_d = std::numeric_limits<double>::max();
            const int start = final_id;
            for (int k = child_off[start]; k < child_off[start + 1]; k++) {
                const int id = child_idx[k];
                const float* y = node_desc + (size_t)id * dim;
                double sqd = 0.;
                for (int i = 0; i < dim; i++) sqd += (x[i] - y[i]) * (x[i] - y[i]);
                if (sqd < best_d) { best_d = sqd; final_id = id; }
            }
            // (a descriptor with NaN components compares nearer to no child and DBoW3 would loop here for ever: the
            // library and this statement take the first child instead)
            if (final_id == start) final_id = child_idx[child_off[start]];
            if (current_level == nid_level) nid = final_id;
        } while (child_off[final_id + 1] != child_off[final_id]);
        out_word[f] = word_id[final_id];
        out_weight[f] = weight[final_id];
        out_node[f] = nid;
    }
}

}  // extern "C"
