// gfx950 (MI355X, CDNA4) kernels of the PPG front-end after the CNN: score map, NMS, heat-map tail and refinement, remap, the
// line graph (candidates, scores, the order-dependent tail), descriptor sampling.  Wave = 64 lanes everywhere.
//
// Built with -ffp-contract=off: every float expression below rounds exactly as written, and fused
// multiply-adds appear only where fmaf() is spelled out.  That is what lets the integer outputs
// (keypoints, line candidates) agree bit for bit with a strict-IEEE CPU evaluation of the reference's formulas.
#include "ppg_kernels.h"
#include "conv_split.h"
#include "conv_store.h"
#include "wave_util.h"

#include <math.h>
#include <algorithm>
#include <type_traits>

namespace ppg {

__global__ __launch_bounds__(256) void score_map_kernel(NmsArgs a) {
    const int h = a.H / 8, w = a.W / 8;
    const long cell = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (cell >= (long)a.B * h * w) return;
    const int cx = (int)(cell % w), cy = (int)((cell / w) % h), b = (int)(cell / ((long)w * h));
    const float* p = a.junc + (size_t)cell * 65;
    const float v = p[lane], dust = p[64];
    const float m = wave_max(fmaxf(v, dust));
    const float e = expf(v - m), ed = expf(dust - m);
    const float s = wave_sum(e) + ed;
    a.score[((size_t)b * a.H + cy * 8 + (lane >> 3)) * a.W + cx * 8 + (lane & 7)] = e / s;
}

hipError_t launch_score_map(const NmsArgs& a, hipStream_t s) {
    const long cells = (long)a.B * (a.H / 8) * (a.W / 8);
    hipLaunchKernelGGL(score_map_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// Greedy radius-4 NMS of feature/src/PPGExtractor.cpp:166-206, reformulated so that it runs in
// parallel yet returns exactly the sequential result.
//
// The reference visits candidates (score >= thresh, >= 4 px from the border) in descending score and
// keeps one iff no already-kept candidate lies within +-4 px.  With the total order
//     p beats q  <=>  score_p > score_q  or  (score_p == score_q and index_p < index_q)
// (index = y*W + x; the reference's std::sort leaves ties unspecified) that is the unique fixed
// point of:  kept(p) <=> no kept q in the window beats p.  Passes:
//   A  (tiles) local maxima of the order are kept; every other candidate is undecided
//   B  (tiles) undecided next to a kept pixel -> suppressed; the rest go to a residual list
//   C  (one work-group per frame) iterate on the residual list until nothing is undecided
//   D  (one work-group per frame) rank the kept pixels by the same order, emit the first max_num
//      (the reference's early break at JUNCTION_MAX_NUM, :195-196), undistort them (:212-233).
// =====================================================================================
constexpr int NT_TILE = 32, NR = 4, NT_LD = NT_TILE + 2 * NR;  // 40

__device__ inline bool beats(float sq, int dy, int dx, float sp) {
    return sq > sp || (sq == sp && (dy < 0 || (dy == 0 && dx < 0)));
}

// 64-bit priority of a candidate pixel: larger = better.  Scores are positive floats, so their bit
// patterns order like the values; ties go to the smaller row-major index.  0 = not a candidate.
__device__ inline unsigned long long prio_key(float s, uint32_t idx) {
    return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// Pass A: a candidate is kept at once iff it is the best pixel of its own 9x9 window.  The window
// maximum of a total order is separable: row maxima over 9 columns, then column maxima over 9 rows.
__global__ __launch_bounds__(256) void nms_local_max_kernel(NmsArgs a) {
    __shared__ unsigned long long key[NT_LD][NT_LD + 1];
    __shared__ unsigned long long rowmax[NT_LD][NT_TILE + 1];
    const int b = blockIdx.z, ox = blockIdx.x * NT_TILE, oy = blockIdx.y * NT_TILE;
    const float* sc = a.score + (size_t)b * a.H * a.W;
    for (int i = threadIdx.x; i < NT_LD * NT_LD; i += 256) {
        const int ly = i / NT_LD, lx = i % NT_LD;
        const int gy = oy + ly - NR, gx = ox + lx - NR;
        unsigned long long v = 0;
        if (gx >= NR && gx <= a.W - NR - 1 && gy >= NR && gy <= a.H - NR - 1) {
            const float s = sc[(size_t)gy * a.W + gx];
            if (!(s < a.thresh)) v = prio_key(s, (uint32_t)(gy * a.W + gx));
        }
        key[ly][lx] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT_LD * NT_TILE; i += 256) {
        const int ly = i / NT_TILE, lx = i % NT_TILE;
        unsigned long long m = 0;
#pragma unroll
        for (int d = 0; d <= 2 * NR; d++) { const unsigned long long v = key[ly][lx + d]; m = v > m ? v : m; }
        rowmax[ly][lx] = m;
    }
    __syncthreads();
    uint8_t* st = a.state + (size_t)b * a.H * a.W;
    for (int i = threadIdx.x; i < NT_TILE * NT_TILE; i += 256) {
        const int ly = i / NT_TILE, lx = i % NT_TILE;
        const int gy = oy + ly, gx = ox + lx;
        if (gy >= a.H || gx >= a.W) continue;
        unsigned long long m = 0;
#pragma unroll
        for (int d = 0; d <= 2 * NR; d++) { const unsigned long long v = rowmax[ly + d][lx]; m = v > m ? v : m; }
        const unsigned long long me = key[ly + NR][lx + NR];
        st[(size_t)gy * a.W + gx] = me == 0 ? 0 : (me == m ? 2 : 1);
    }
}

// Pass B: an undecided candidate with a kept pixel in its window is suppressed (that pixel beats it,
// being the best of a window that contains this one); the rest go to the residual list.  "Any kept
// pixel in the window" is separable in the same way.
__global__ __launch_bounds__(256) void nms_suppress_kernel(NmsArgs a) {
    __shared__ uint8_t sh[NT_LD][NT_LD + 8];
    __shared__ uint8_t rowany[NT_LD][NT_TILE + 8];
    const int b = blockIdx.z, ox = blockIdx.x * NT_TILE, oy = blockIdx.y * NT_TILE;
    uint8_t* st = a.state + (size_t)b * a.H * a.W;
    for (int i = threadIdx.x; i < NT_LD * NT_LD; i += 256) {
        const int ly = i / NT_LD, lx = i % NT_LD;
        const int gy = oy + ly - NR, gx = ox + lx - NR;
        uint8_t v = 0;
        if (gx >= 0 && gx < a.W && gy >= 0 && gy < a.H) v = st[(size_t)gy * a.W + gx];
        sh[ly][lx] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT_LD * NT_TILE; i += 256) {
        const int ly = i / NT_TILE, lx = i % NT_TILE;
        uint8_t any = 0;
#pragma unroll
        for (int d = 0; d <= 2 * NR; d++) any |= (uint8_t)(sh[ly][lx + d] == 2);
        rowany[ly][lx] = any;
    }
    __syncthreads();
    // the tile's kept and residual pixels are gathered in LDS and appended to the frame's lists with one
    // global atomic each (the lists are unordered; a per-pixel atomic on the frame counter serialises)
    __shared__ uint32_t l_kept[NT_TILE * NT_TILE], l_res[NT_TILE * NT_TILE];
    __shared__ int n_kept, n_res, g_kept, g_res;
    if (threadIdx.x == 0) { n_kept = 0; n_res = 0; }
    __syncthreads();
    int* cnt = a.counters + b * 8;
    for (int i = threadIdx.x; i < NT_TILE * NT_TILE; i += 256) {
        const int ly = i / NT_TILE, lx = i % NT_TILE;
        const int gy = oy + ly, gx = ox + lx;
        if (gy >= a.H || gx >= a.W) continue;
        const uint8_t s0 = sh[ly + NR][lx + NR];
        const uint32_t idx = (uint32_t)(gy * a.W + gx);
        if (s0 == 2) {
            l_kept[atomicAdd(&n_kept, 1)] = idx;
        } else if (s0 == 1) {
            uint8_t sup = 0;
#pragma unroll
            for (int d = 0; d <= 2 * NR; d++) sup |= rowany[ly + d][lx];
            if (sup) {
                // only this thread writes this pixel; neighbours test for == 2 only, which never changes here
                st[idx] = 3;
            } else {
                l_res[atomicAdd(&n_res, 1)] = idx;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g_kept = n_kept ? atomicAdd(&cnt[1], n_kept) : 0;
        g_res = n_res ? atomicAdd(&cnt[0], n_res) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_kept; i += 256)
        if (g_kept + i < a.kept_cap) a.kept[(size_t)b * a.kept_cap + g_kept + i] = l_kept[i];
    for (int i = threadIdx.x; i < n_res; i += 256)
        if (g_res + i < a.resid_cap) a.resid[(size_t)b * a.resid_cap + g_res + i] = l_res[i];
}

__global__ __launch_bounds__(1024) void nms_resolve_kernel(NmsArgs a) {
    const int b = blockIdx.x;
    uint8_t* st = a.state + (size_t)b * a.H * a.W;
    const float* sc = a.score + (size_t)b * a.H * a.W;
    int* cnt = a.counters + b * 8;
    const int nres = min(cnt[0], a.resid_cap);
    const uint32_t* resid = a.resid + (size_t)b * a.resid_cap;
    // Every round decides at least the best undecided pixel, so nres + 1 rounds always suffice.
    for (int round = 0; round <= nres; round++) {
        int remaining = 0;
        for (int k = threadIdx.x; k < nres; k += 1024) {
            const int p = (int)resid[k];
            if (st[p] != 1) continue;
            const int py = p / a.W, px = p % a.W;
            const float sp = sc[p];
            // The 80 state bytes of the window are fetched without a branch between them, so they are all in flight
            // together (a scan that stops at the first kept pixel pays one memory latency per pixel looked at, and
            // this kernel is one work-group per frame); only the undecided neighbours then need their score.
            // (Candidates keep NR pixels from the border, so the whole window lies inside the image.)
            bool sup = false, blocked = false;
            unsigned long long und0 = 0ull;   // undecided neighbours, window positions 0..63
            unsigned und1 = 0u;               // positions 64..80
            constexpr int NW = 2 * NR + 1;
            const uint8_t* w0 = st + (py - NR) * a.W + (px - NR);
            uint8_t sq[NW * NW];
#pragma unroll
            for (int t = 0; t < NW * NW; t++) sq[t] = w0[(t / NW) * a.W + (t % NW)];
#pragma unroll
            for (int t = 0; t < NW * NW; t++) {
                if (t == NW * NR + NR) continue;
                sup |= sq[t] == 2;
                if (t < 64) und0 |= (unsigned long long)(sq[t] == 1) << t;
                else und1 |= (unsigned)(sq[t] == 1) << (t - 64);
            }
            if (!sup) {
                while (und0 | und1) {
                    int t;
                    if (und0) { t = __ffsll((long long)und0) - 1; und0 &= und0 - 1; }
                    else { t = 64 + __ffs((int)und1) - 1; und1 &= und1 - 1; }
                    const int dy = t / (2 * NR + 1) - NR, dx = t % (2 * NR + 1) - NR;
                    if (beats(sc[(py + dy) * a.W + px + dx], dy, dx, sp)) { blocked = true; break; }
                }
            }
            if (sup) st[p] = 3;
            else if (!blocked) {
                st[p] = 2;
                const int kk = atomicAdd(&cnt[1], 1);
                if (kk < a.kept_cap) a.kept[(size_t)b * a.kept_cap + kk] = (uint32_t)p;
            } else remaining++;
        }
        // barrier + work-group-wide visibility of the state bytes written above
        if (__syncthreads_count(remaining) == 0) { if (threadIdx.x == 0) cnt[7] = round + 1; break; }
    }
}

// cv::undistortPoints(src, dst, K, D, noArray(), P = K) for a 4-coefficient pinhole model: five
// fixed-point iterations in double (OpenCV's published algorithm; identity for D = 0).
__device__ inline void undistort_pinhole_dev(const NmsArgs& a, float x, float y, float* ou, float* ov) {
    const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
    const double k1 = a.d0, k2 = a.d1, p1 = a.d2, p2 = a.d3;
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

// cv::fisheye::undistortPoints(src, dst, K, D, noArray(), P = K): Newton iteration on theta.
__device__ inline void undistort_fisheye_dev(const NmsArgs& a, float x, float y, float* ou, float* ov) {
    const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
    const double k0 = a.d0, k1 = a.d1, k2 = a.d2, k3 = a.d3;
    const double kPi = 3.1415926535897932384626433832795;
    const double px = (x - cx) / fx, py = (y - cy) / fy;
    double theta_d = sqrt(px * px + py * py);
    theta_d = fmin(fmax(-kPi / 2., theta_d), kPi / 2.);
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double k0t2 = k0 * t2, k1t4 = k1 * t4, k2t6 = k2 * t6, k3t8 = k3 * t8;
            const double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) /
                               (1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
            theta = theta - fix;
            if (fabs(fix) < 1e-8) break;
        }
        scale = tan(theta) / theta_d;
    }
    *ou = (float)(fx * (px * scale) + cx);
    *ov = (float)(fy * (py * scale) + cy);
}

constexpr int EMIT_SORT_CAP = 8192;  // keys the LDS sort buffer holds (64 KiB)
constexpr int EMIT_SORT_ALL = 2048;  // up to this many kept pixels everything is sorted; of larger sets only the best max_num

__device__ inline void emit_point(const NmsArgs& a, int b, const float* sc, uint32_t p, int rank) {
    const int x = (int)(p % (uint32_t)a.W), y = (int)(p / (uint32_t)a.W);
    const size_t o = (size_t)b * a.max_num + rank;
    a.kp_xy[2 * o] = x;
    a.kp_xy[2 * o + 1] = y;
    a.kp_score[o] = sc[p];
    float u, v;
    if (a.fisheye) undistort_fisheye_dev(a, (float)x, (float)y, &u, &v);
    else undistort_pinhole_dev(a, (float)x, (float)y, &u, &v);
    a.kp_un[2 * o] = u;
    a.kp_un[2 * o + 1] = v;
    a.kp_out[o] = !(u >= 1 && u < a.W - 1 && v >= 1 && v < a.H - 1);
}

// Pass D: order the kept pixels by priority and emit the best max_num (the reference's sorted visit
// order with its early break at JUNCTION_MAX_NUM, PPGExtractor.cpp:180, :195-196).
__global__ __launch_bounds__(1024) void nms_emit_kernel(NmsArgs a) {
    __shared__ unsigned long long skey[EMIT_SORT_CAP];
    __shared__ int hist[256];
    __shared__ int wtot[4];
    __shared__ int s_bin, s_k, s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* cnt = a.counters + b * 8;
    const int S = min(cnt[1], a.kept_cap);
    const float* sc = a.score + (size_t)b * a.H * a.W;
    const uint32_t* kept = a.kept + (size_t)b * a.kept_cap;
    const int n = min(S, a.max_num);
    int m = S;  // keys to sort
    if (S <= EMIT_SORT_ALL || S <= n) {
        for (int k = tid; k < S; k += 1024) skey[k] = prio_key(sc[kept[k]], kept[k]);
    } else {
        // Many more kept pixels than are emitted: only the best n = max_num matter, so find the n-th largest key by
        // radix select (8 bits per pass over the keys in memory; keys are distinct) and sort just those n — a
        // 91-stage bitonic sort of 8192 keys costs far more than eight counting passes and a sort of n.
        unsigned long long* key = a.kept_key + (size_t)b * a.kept_cap;
        for (int k = tid; k < S; k += 1024) key[k] = prio_key(sc[kept[k]], kept[k]);
        __syncthreads();
        unsigned long long prefix = 0, mask = 0;
        int kk = n;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int k = tid; k < S; k += 1024) {
                const unsigned long long u = key[k];
                if ((u & mask) == prefix) atomicAdd(&hist[(int)((u >> shift) & 255ull)], 1);
            }
            __syncthreads();
            int c = 0, inc = 0;
            if (tid < 256) {
                c = hist[255 - tid];  // bins in descending order
                inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                if (lane == 63) wtot[wave] = inc;
            }
            __syncthreads();
            if (tid < 256) {
                for (int w = 0; w < wave; w++) inc += wtot[w];
                const int exc = inc - c;
                if (exc < kk && kk <= inc) { s_bin = 255 - tid; s_k = kk - exc; }
            }
            __syncthreads();
            prefix |= (unsigned long long)s_bin << shift;
            mask |= 255ull << shift;
            kk = s_k;
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int k = tid; k < S; k += 1024) {
            const unsigned long long u = key[k];
            if (u >= prefix) {
                const int o = atomicAdd(&s_cnt, 1);
                if (o < EMIT_SORT_CAP) skey[o] = u;
            }
        }
        __syncthreads();
        m = min(s_cnt, EMIT_SORT_CAP);  // == n
    }
    int P = 1;
    while (P < m) P <<= 1;
    __syncthreads();
    for (int k = m + tid; k < P; k += 1024) skey[k] = 0ull;
    __syncthreads();
    if (P <= 1024) {
        // one key per thread: the exchanges over fewer than 64 positions are lane shuffles, only the ten stages with
        // a stride of 64 or more go through LDS (a sort of 1024 keys in LDS alone is 55 stages with a barrier each)
        unsigned long long k = tid < P ? skey[tid] : 0ull;
        for (int size = 2; size <= P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                unsigned long long other;
                if (stride < 64) {
                    const unsigned lo32 = __shfl_xor((unsigned)(k & 0xFFFFFFFFull), stride);
                    const unsigned hi32 = __shfl_xor((unsigned)(k >> 32), stride);
                    other = ((unsigned long long)hi32 << 32) | lo32;
                } else {
                    __syncthreads();
                    if (tid < P) skey[tid] = k;
                    __syncthreads();
                    other = tid < P ? skey[tid ^ stride] : 0ull;
                }
                const bool take_max = ((tid & stride) == 0) == ((tid & size) == 0);
                k = take_max ? (k > other ? k : other) : (k < other ? k : other);
            }
        __syncthreads();
        if (tid < P) skey[tid] = k;
        __syncthreads();
    } else
    // bitonic sort, descending
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long x = skey[lo], y = skey[hi];
                if ((x < y) == desc) { skey[lo] = y; skey[hi] = x; }
            }
            __syncthreads();
        }
    for (int r = tid; r < n; r += 1024) emit_point(a, b, sc, 0xFFFFFFFFu - (uint32_t)(skey[r] & 0xFFFFFFFFull), r);
    if (tid == 0) cnt[3] = n;
}

hipError_t launch_nms(const NmsArgs& a, hipStream_t s) {
    dim3 grid((a.W + NT_TILE - 1) / NT_TILE, (a.H + NT_TILE - 1) / NT_TILE, a.B);
    hipLaunchKernelGGL(nms_local_max_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms_suppress_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(nms_resolve_kernel, dim3(a.B), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(a.B), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// Line heat map: 1x1 conv 4->2 (net/EdgeHeatmap.pt .../heatmap_decoder.py:22), softmax over the two
// channels, channel 1 (feature/src/PPGExtractor.cpp:242).
// =====================================================================================
__device__ inline float softmax2_ch1(float h0, float h1) {
    const float m = fmaxf(h0, h1);
    const float e0 = expf(h0 - m), e1 = expf(h1 - m);
    return e1 / (e0 + e1);
}

__global__ __launch_bounds__(256) void heat_tail_kernel(HeatArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.H * a.W;
    if (i >= total) return;
    const f32x4 v = *(const f32x4*)(a.ps + (size_t)i * 4);
    float h0 = a.bias[0], h1 = a.bias[1];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        h0 = fmaf(v[c], a.w[c], h0);
        h1 = fmaf(v[c], a.w[4 + c], h1);
    }
    if (a.raw) { a.raw[2 * i] = h0; a.raw[2 * i + 1] = h1; }
    a.prob[i] = softmax2_ch1(h0, h1);
}

__global__ __launch_bounds__(256) void heat_prob_kernel(HeatArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.H * a.W;
    if (i >= total) return;
    a.prob[i] = softmax2_ch1(a.raw[2 * i], a.raw[2 * i + 1]);
}

hipError_t launch_heat_tail(const HeatArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.H * a.W;
    hipLaunchKernelGGL(heat_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_heat_prob_from_raw(const HeatArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.H * a.W;
    hipLaunchKernelGGL(heat_prob_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) of a float image through float maps, as OpenCV's generic path
// evaluates it: the map coordinate is rounded to 1/32 pixel (cvRound(m * INTER_TAB_SIZE), ties to even), the
// four weights are float products of (1 - a, a) with a = k/32, and the sum runs tap by tap from the left.
// =====================================================================================
__global__ __launch_bounds__(256) void remap_kernel(RemapArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)a.H * a.W;
    if (i >= hw * a.B) return;
    const long p = i % hw;
    const float* src = a.src + (i - p);
    const int sx = (int)rintf(a.mapx[p] * 32.0f), sy = (int)rintf(a.mapy[p] * 32.0f);
    const int x0 = sx >> 5, y0 = sy >> 5;
    const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
    const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
    auto tap = [&](int y, int x) { return (x >= 0 && y >= 0 && x < a.W && y < a.H) ? src[(size_t)y * a.W + x] : 0.0f; };
    a.dst[i] = tap(y0, x0) * w0 + tap(y0, x0 + 1) * w1 + tap(y0 + 1, x0) * w2 + tap(y0 + 1, x0 + 1) * w3;
}

hipError_t launch_remap(const RemapArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.H * a.W;
    hipLaunchKernelGGL(remap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// refineHeatMap (feature/src/PPGExtractor.cpp:540-578) on the tile grid of :243-256.
// One work-group per tile (tiles are refine_sz^2, the last row/column of tiles absorbs the
// remainder, so at most (2*sz-1)^2 pixels).  The row-major list of values > valid_thresh is rebuilt
// in LDS in the reference's order (needed for the percentile probe on the UNSORTED list, :557),
// ranked by counting, and the top-30 % mean is accumulated in double in sorted order by one lane —
// the same additions in the same order as std::accumulate(…, 0.0).
// =====================================================================================
constexpr int RF_MAXPIX = 1024;  // (2*16-1)^2 = 961

__global__ __launch_bounds__(256) void refine_kernel(HeatArgs a) {
    __shared__ float vals[RF_MAXPIX];
    __shared__ float sorted[RF_MAXPIX];
    __shared__ int wave_cnt[4];
    __shared__ int s_n;
    __shared__ float s_ave;
    __shared__ int s_action;  // 0 leave, 1 zero, 2 rescale
    __shared__ int s_big;     // some value exceeds 1 (cannot happen for softmax output)
    const int S = a.refine_sz;
    const int gxn = a.W / S, gyn = a.H / S;
    const int tj = blockIdx.x % gxn, ti = (blockIdx.x / gxn) % gyn, b = blockIdx.x / (gxn * gyn);
    const int h0 = ti * S, w0 = tj * S;
    const int h1 = ti == gyn - 1 ? a.H : (ti + 1) * S, w1 = tj == gxn - 1 ? a.W : (tj + 1) * S;
    const int segH = h1 - h0, segW = w1 - w0, area = segH * segW;
    const float* src = a.prob + (size_t)b * a.H * a.W;
    float* dst = a.refined + (size_t)b * a.H * a.W;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    if (tid == 0) s_big = 0;
    __syncthreads();
    // ordered compaction, 256 pixels per pass
    int base = 0;
    for (int p0 = 0; p0 < area; p0 += 256) {
        const int p = p0 + tid;
        float v = 0.f;
        bool ok = false;
        if (p < area) {
            v = src[(size_t)(h0 + p / segW) * a.W + w0 + p % segW];
            ok = v > a.valid_thresh;
            if (v > 1.0f) s_big = 1;
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int wv = 0; wv < wave; wv++) off += wave_cnt[wv];
        if (ok) vals[off + __popcll(m & ((1ull << lane) - 1ull))] = v;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int n = base;
    if (tid == 0) {
        int action = 2;
        const int valCount = (int)(a.valid_ratio * (float)n);
        if (valCount < 1) action = 0;
        else if ((double)n >= (double)(segH * segW) * 0.9 && (double)vals[(size_t)((double)n * 0.9)] > 0.1) action = 1;
        s_action = action;
        s_n = valCount;
    }
    __syncthreads();
    const int action = s_action, valCount = s_n;
    // The mean of the valCount largest values is accumulated in double by the reference.  Every value lies in
    // (valid_thresh, 1] (checked: s_big); with valid_thresh >= 2^-20 all of them are multiples of 2^-43, so any sum of up to
    // 1024 of them is exact in double and the order of the additions cannot change the result.  Then it is
    // enough to find the valCount-th largest value T (radix select on the bit patterns, 8 bits per pass):
    // sum = sum of the values above T + (valCount - their number) * T.
    if (action == 2 && a.valid_thresh >= 0x1p-20f && !s_big) {
        __shared__ int hist[256];
        __shared__ int wtot[4];
        __shared__ int s_bin, s_k;
        __shared__ double wsum[4];
        __shared__ int wcnt2[4];
        unsigned prefix = 0, mask = 0;
        int kk = valCount;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            for (int k = tid; k < n; k += 256) {
                const unsigned u = __float_as_uint(vals[k]);
                if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
            }
            __syncthreads();
            const int c = hist[255 - tid];  // bins in descending order
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) wtot[wave] = inc;
            __syncthreads();
            for (int w = 0; w < wave; w++) inc += wtot[w];
            const int exc = inc - c;
            if (exc < kk && kk <= inc) { s_bin = 255 - tid; s_k = kk - exc; }
            __syncthreads();
            prefix |= (unsigned)s_bin << shift;
            mask |= 255u << shift;
            kk = s_k;
        }
        double part = 0.0;
        int above = 0;
        for (int k = tid; k < n; k += 256) {
            const float v = vals[k];
            if (__float_as_uint(v) > prefix) { part += (double)v; above++; }
        }
        part = wave_sum_d(part);
        for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o);
        if (lane == 0) { wsum[wave] = part; wcnt2[wave] = above; }
        __syncthreads();
        if (tid == 0) {
            const double acc = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) +
                               (double)(valCount - (wcnt2[0] + wcnt2[1] + wcnt2[2] + wcnt2[3])) * (double)__uint_as_float(prefix);
            s_ave = (float)(acc / (double)(float)valCount);
        }
        __syncthreads();
    } else if (action == 2) {
        for (int k = tid; k < n; k += 256) {
            const float v = vals[k];
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const float u = vals[j];
                rank += (u > v) || (u == v && j < k);
            }
            sorted[rank] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double acc = 0.0;
            for (int k = 0; k < valCount; k++) acc += (double)sorted[k];
            s_ave = (float)(acc / (double)(float)valCount);
        }
        __syncthreads();
    }
    const float ave = s_ave;
    for (int p = tid; p < area; p += 256) {
        const size_t o = (size_t)(h0 + p / segW) * a.W + w0 + p % segW;
        const float v = src[o];
        float r = v;
        if (action == 1) r = 0.f;
        else if (action == 2) {
            if (v > a.valid_thresh) {
                const float ns = v / ave;
                r = (double)ns > 1.0 ? 1.0f : ns;
            } else r = 0.f;
        }
        dst[o] = r;
    }
}

// The same for tiles of at most 256 pixels that all have the full size (image sides multiples of refine_sz): one wave
// per tile, four pixels per lane, no work-group barrier.  The ordered list of valid values exists only as (lane, slot)
// positions — a lane's pixels are consecutive in row-major order, so its slice of the list starts at the wave prefix sum
// of the valid counts —, the probe of the unsorted list is one lane lookup, and the k-th largest value comes from a
// bit-by-bit bisection of the bit patterns with wave ballots.
__global__ __launch_bounds__(256) void refine_wave_kernel(HeatArgs a) {
    __shared__ float s_vals[4][256], s_sorted[4][256];   // only for the sequential fallback
    const int S = a.refine_sz, area = S * S;
    const int gxn = a.W / S, gyn = a.H / S;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long tile = (long)blockIdx.x * 4 + wave;
    if (tile >= (long)gxn * gyn * a.B) return;
    const int tj = (int)(tile % gxn), ti = (int)((tile / gxn) % gyn), b = (int)(tile / ((long)gxn * gyn));
    const int h0 = ti * S, w0 = tj * S;
    const float* src = a.prob + (size_t)b * a.H * a.W;
    float* dst = a.refined + (size_t)b * a.H * a.W;
    float v[4];
    size_t o[4];
    bool ok[4], in[4];
    int cnt = 0;
    bool big = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int p = 4 * lane + e;
        in[e] = p < area;
        o[e] = (size_t)(h0 + (in[e] ? p / S : 0)) * a.W + w0 + (in[e] ? p % S : 0);
        v[e] = in[e] ? src[o[e]] : 0.f;
        ok[e] = in[e] && v[e] > a.valid_thresh;
        big |= in[e] && v[e] > 1.0f;
        cnt += ok[e];
    }
    int inc = cnt;   // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    const int off = inc - cnt;
    const int n = __shfl(inc, 63);
    const bool any_big = __ballot(big) != 0ull;
    const int valCount = (int)(a.valid_ratio * (float)n);
    int action = 2;
    if (valCount < 1) action = 0;
    else if ((double)n >= (double)area * 0.9) {
        // element (size_t)(n * 0.9) of the unsorted list
        const int idx = (int)(size_t)((double)n * 0.9);
        float mine = 0.f;
        int r = idx - off;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (ok[e]) { if (r == 0) mine = v[e]; r--; }
        const unsigned long long holder = __ballot(idx >= off && idx < off + cnt);
        const float probe = __shfl(mine, __ffsll((long long)holder) - 1);
        if ((double)probe > 0.1) action = 1;
    }
    float ave = 0.f;
    if (action == 2 && a.valid_thresh >= 0x1p-20f && !any_big) {
        // every value is a multiple of 2^-43 in (2^-20, 1]: sums of up to 256 of them are exact in double in any order
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; e++) u[e] = ok[e] ? __float_as_uint(v[e]) : 0u;
        uint32_t T = 0u;   // largest T with at least valCount keys >= T: the valCount-th largest key
        for (int bit = 30; bit >= 0; bit--) {
            const uint32_t cand = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) c += __popcll(__ballot(u[e] >= cand));
            if (c >= valCount) T = cand;
        }
        double part = 0.0;
        int above = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (u[e] > T) { part += (double)v[e]; above++; }
        part = wave_sum_d(part);
        for (int d = 32; d > 0; d >>= 1) above += __shfl_xor(above, d);
        const double acc = part + (double)(valCount - above) * (double)__uint_as_float(T);
        ave = (float)(acc / (double)(float)valCount);
    } else if (action == 2) {
        // sequential statement: rank by counting, then the additions in sorted order by one lane
        float* vals = s_vals[wave];
        float* sorted = s_sorted[wave];
        int r = off;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (ok[e]) vals[r++] = v[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = lane; k < n; k += 64) {
            const float x = vals[k];
            int rank = 0;
            for (int j = 0; j < n; j++) {
                const float y = vals[j];
                rank += (y > x) || (y == x && j < k);
            }
            sorted[rank] = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double acc = 0.0;
        if (lane == 0)
            for (int k = 0; k < valCount; k++) acc += (double)sorted[k];
        acc = __shfl(acc, 0);
        ave = (float)(acc / (double)(float)valCount);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (!in[e]) continue;
        float r = v[e];
        if (action == 1) r = 0.f;
        else if (action == 2) {
            if (v[e] > a.valid_thresh) {
                const float ns = v[e] / ave;
                r = (double)ns > 1.0 ? 1.0f : ns;
            } else r = 0.f;
        }
        dst[o[e]] = r;
    }
}

hipError_t launch_refine(const HeatArgs& a, hipStream_t s) {
    const int S = a.refine_sz;
    if (S < 1 || (2 * S - 1) * (2 * S - 1) > RF_MAXPIX) return hipErrorInvalidValue;
    const int tiles = (a.W / S) * (a.H / S) * a.B;
    if (S * S <= 256 && a.W % S == 0 && a.H % S == 0) {
        hipLaunchKernelGGL(refine_wave_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(refine_kernel, dim3(tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// detectLines, state-independent part (feature/src/PPGExtractor.cpp:290-313 and :367-389).
// For every in-image pair i < j: the three-point heat probe (:304-313).  Survivors are listed in
// (i, j) order with the two scores the reference computes later for lines that are still alive
// (heatMapInlierRate :484-497, heatMapLineScore :502-512); both depend only on the end points, so
// they can be evaluated up front.  The order-dependent overlap / collinearity logic consumes this
// list on the host.
// =====================================================================================
__constant__ float kInvSampleGap[4] = {0.3333f, 0.200f, 0.1427f, 0.1111f};  // PPGExtractor.cpp:19

__device__ inline float heat_at(const float* heat, int W, int y, int x) { return heat[(size_t)y * W + x]; }

// three-point heat probe of PPGExtractor.cpp:304-313 (most pairs fail the first lookup: the early exits save
// more memory traffic than issuing the three gathers together saves latency)
__device__ inline bool line_probe(const LineArgs& a, const float* un, const uint8_t* out, const float* heat, int i, int j) {
    if (out[j]) return false;
    const float ix = un[2 * i], iy = un[2 * i + 1], jx = un[2 * j], jy = un[2 * j + 1];
    const float c1x = jx * 0.2f + ix * 0.8f, c1y = jy * 0.2f + iy * 0.8f;
    if (heat_at(heat, a.W, (int)((double)c1y + 0.5), (int)((double)c1x + 0.5)) < a.heat_thresh) return false;
    const float c2x = jx * 0.8f + ix * 0.2f, c2y = jy * 0.8f + iy * 0.2f;
    if (heat_at(heat, a.W, (int)((double)c2y + 0.5), (int)((double)c2x + 0.5)) < a.heat_thresh) return false;
    const float c3x = jx * 0.5f + ix * 0.5f, c3y = jy * 0.5f + iy * 0.5f;
    if (heat_at(heat, a.W, (int)((double)c3y + 0.5), (int)((double)c3x + 0.5)) < a.heat_thresh) return false;
    return true;
}

// Candidates leave the GPU already in the reference's loop order (i ascending, then j ascending):
// pass 1 probes every pair, keeps the wave ballots and counts the survivors of every row i, a scan turns the
// counts into offsets, pass 2 reads the ballots back and writes each survivor at its ordered position (ballot
// prefix inside the work-group).
template <bool EMIT>
__global__ __launch_bounds__(256) void line_pairs_kernel(LineArgs a) {
    __shared__ int wcnt[4];
    const int b = blockIdx.y, i = blockIdx.x;
    int* cnt = a.counters + b * 8;
    const int n = cnt[3];
    if (i >= n) return;
    const uint8_t* out = a.kp_out + (size_t)b * a.max_num;
    const float* un = a.kp_un + (size_t)b * a.max_num * 2;
    const float* heat = a.heat + (size_t)b * a.H * a.W;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int* rowcnt = a.row_cnt + (size_t)b * (a.max_num + 1);
    const bool row_ok = !out[i];
    // ballots of the row: word 4*t + w = wave w of iteration t
    unsigned long long* mrow = a.pair_mask + ((size_t)b * a.max_num + i) * a.mask_words;
    if (!EMIT) {
        int mine = 0;  // survivors seen by this wave
        for (int j0 = i + 1; j0 < n; j0 += 256, mrow += 4) {
            const int j = j0 + tid;
            const bool ok = row_ok && j < n && line_probe(a, un, out, heat, i, j);
            const unsigned long long m = __ballot(ok);
            if (lane == 0) mrow[wave] = m;
            mine += __popcll(m);
        }
        if (lane == 0) wcnt[wave] = mine;
        __syncthreads();
        if (tid == 0) rowcnt[i] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    } else {
        int base = rowcnt[i];  // after the scan rowcnt[i] holds the exclusive offset
        for (int j0 = i + 1; j0 < n; j0 += 256, mrow += 4) {
            const unsigned long long m0 = mrow[0], m1 = mrow[1], m2 = mrow[2], m3 = mrow[3];
            const unsigned long long m = wave == 0 ? m0 : wave == 1 ? m1 : wave == 2 ? m2 : m3;
            int off = base + __popcll(m & ((1ull << lane) - 1ull));
            if (wave > 0) off += __popcll(m0);
            if (wave > 1) off += __popcll(m1);
            if (wave > 2) off += __popcll(m2);
            if (((m >> lane) & 1ull) && off < a.cand_cap)
                a.cand_ij[(size_t)b * a.cand_cap + off] = ((uint32_t)i << 16) | (uint32_t)(j0 + tid);
            base += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    }
}

// Inclusive prefix sum of one value per thread over a 1024-thread work-group: shuffles inside each wave, the sixteen wave
// totals through LDS (part: 16 ints), two barriers in all — a work-group barrier of sixteen waves costs about a
// microsecond, and a doubling scan over the whole work-group has twenty of them.  *total = the sum over all threads.
__device__ inline int block_incl_scan_1024(int v, int* part, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    __syncthreads();   // (an earlier use of part is over)
    if (lane == 63) part[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int p = part[w];
        tot += p;
        if (w < wave) base += p;
    }
    *total = tot;
    return base + x;
}

// exclusive scan of the per-row counts of one frame (n <= 4096 rows), total -> counters[2]
__global__ __launch_bounds__(1024) void line_scan_kernel(LineArgs a) {
    __shared__ int part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    int* cnt = a.counters + b * 8;
    const int n = cnt[3];
    int* rowcnt = a.row_cnt + (size_t)b * (a.max_num + 1);
    const int per = (n + 1023) / 1024;
    int local = 0;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) local += rowcnt[r];
    }
    int total;
    int run = block_incl_scan_1024(local, part, &total) - local;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) { const int c = rowcnt[r]; rowcnt[r] = run; run += c; }
    }
    if (tid == 1023) cnt[2] = total;
}

// Sixteen lanes per candidate: the samples of a line are evaluated sixteen at a time (the loads and the bilinear
// terms, which is where the time goes), then added to the running sum one by one in sample order — the order of
// the reference's loop, which fixes the rounding of the score.
__global__ __launch_bounds__(256) void line_score_kernel(LineArgs a) {
    const int b = blockIdx.y;
    const int* cnt = a.counters + b * 8;
    const int nc = min(cnt[2], a.cand_cap);
    const float* un = a.kp_un + (size_t)b * a.max_num * 2;
    const float* heat = a.heat + (size_t)b * a.H * a.W;
    const int lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
    const int groups = gridDim.x * 16;
    // every lane of a wave runs the same number of rounds (wave-wide shuffles inside)
    for (int kw = (blockIdx.x * 256 + (threadIdx.x & ~63)) >> 4; kw < nc; kw += groups) {  // kw: the wave's first candidate
        const int k0 = kw + g;
        const bool live = k0 < nc;
        const int k = live ? k0 : 0;
        const uint32_t ij = a.cand_ij[(size_t)b * a.cand_cap + k];
        const int i = ij >> 16, j = ij & 0xFFFF;
        // ps = start (i), pe = end (j)
        const float psx = un[2 * i], psy = un[2 * i + 1], pex = un[2 * j], pey = un[2 * j + 1];
        const float dx = psx - pex, dy = psy - pey;
        const float dist = sqrtf(dx * dx + dy * dy);
        const int lenLevel = (int)((double)(dist * a.inv_scale) * 4.0);
        const int segNum = live ? (int)(dist * kInvSampleGap[lenLevel < 0 ? 0 : (lenLevel > 3 ? 3 : lenLevel)]) : 0;
        const float step = (float)(1.0 / (double)(float)segNum);
        int inl = 0;
        float sum = 0.f;
        int segMax = max(segNum, __shfl_xor(segNum, 16));
        segMax = max(segMax, __shfl_xor(segMax, 32));
        for (int base = 1; base < segMax; base += 16) {
            const int t = base + gl;
            float val = 0.f;
            bool hit = false;
            if (t < segNum) {
                const float sx = psx * step * (float)t + pex * step * (float)(segNum - t);
                const float sy = psy * step * (float)t + pey * step * (float)(segNum - t);
                const int px = (int)((double)sx + 0.5), py = (int)((double)sy + 0.5);
                // bilinearInterpolation, PPGExtractor.cpp:580-589.  Its four taps are two 8-byte loads, and the
                // rounded sample position of the inlier test is one of them (sx, sy >= 0 for in-image end points).
                const int x1 = (int)sx, x2 = x1 + 1, y1 = (int)sy, y2 = y1 + 1;
                const f32x2u r1 = *(const f32x2u*)(heat + (size_t)y1 * a.W + x1);
                const f32x2u r2 = *(const f32x2u*)(heat + (size_t)y2 * a.W + x1);
                float hp;
                if ((px == x1 || px == x2) && (py == y1 || py == y2)) {
                    const f32x2u r = py == y1 ? r1 : r2;
                    hp = px == x1 ? r[0] : r[1];
                } else {
                    hp = heat_at(heat, a.W, py, px);
                }
                hit = hp > a.heat_thresh;
                const float d1 = ((float)x2 - sx) * r1[0] + (sx - (float)x1) * r1[1];
                const float d2 = ((float)x2 - sx) * r2[0] + (sx - (float)x1) * r2[1];
                val = ((float)y2 - sy) * d1 + (sy - (float)y1) * d2;
            }
            const unsigned long long hits = __ballot(hit);
            inl += __popcll((hits >> (16 * g)) & 0xFFFFull);
            const int left = segNum - base;  // samples of this line in the round
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const float v = __shfl(val, 16 * g + u);
                if (u < left) sum += v;
            }
        }
        if (live && gl == 0) {
            float* o = a.cand_sc + ((size_t)b * a.cand_cap + k) * 2;
            o[0] = (float)inl / (float)(segNum - 1);
            o[1] = sum / (float)(segNum - 1);
        }
    }
}

hipError_t launch_line_candidates(const LineArgs& a, hipStream_t s) {
    dim3 g1(a.max_num, a.B);
    hipLaunchKernelGGL(line_pairs_kernel<false>, g1, dim3(256), 0, s, a);
    hipLaunchKernelGGL(line_scan_kernel, dim3(a.B), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(line_pairs_kernel<true>, g1, dim3(256), 0, s, a);
    dim3 g2(1024, a.B);
    hipLaunchKernelGGL(line_score_kernel, g2, dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// detectLines, order-dependent part (feature/src/PPGExtractor.cpp:315-441), on the device.
//
// The reference walks the probe survivors in (i, j) order; each new line is tested against the live lines at its
// two end points (angle within 0.2*pi and lateral offset below LINE_DISTTHRESH): the shorter of the two wins, a
// killed line stays dead, a rejected line is never listed.  Whether two candidates interact, and which one would
// win, depends on geometry alone; only "is the partner alive at that moment" depends on the order.  So:
//   line_geo_kernel     per candidate: length and the two directions (the reference's distMat / dirMat entries)
//   line_col_*          candidates grouped by their second end point j (lists at end point j)
//   line_inter_kernel   every (candidate, earlier candidate at a shared end point) pair is classified once, in
//                       parallel; the few that interact (about 1.5 %) are listed
//   line_resolve_kernel one work-group per frame: the sequential pass is the unique solution of
//                         rejected_at_i(c) = some listed partner at i that beats c is alive when c arrives
//                         accepted(c)      = not rejected at i and not rejected at j
//                         alive(y, t)      = accepted(y) and no earlier-than-t candidate attempted to kill y
//                         z attempts its kills at i always, its kills at j only if it was not rejected at i
//                       whose dependencies all point back in time; it is found by relaxation with three-valued
//                       states (a handful of rounds: every round settles at least the earliest open candidate).
//                       The same work-group then applies the score thresholds (:367-389), compacts the edges in
//                       candidate order, builds the per-point lists and pairs collinear lines (:392-432).
// Arithmetic: every float expression is the one of the host statement (lines_tail.h); atan2 and sin are the double
// functions rounded to float, and a result within 64 double ulps of a float rounding boundary raises LT_GUARD, upon
// which the host statement (whose libm may round that value the other way) decides the frame instead.
// =====================================================================================
constexpr double kPiD = 3.1415926535897932384626433832795, k2PiD = 6.283185307179586476925286766559;
constexpr uint32_t LT_INF = 0xFFFFFFFFu;
// candidate state bits
constexpr uint32_t CS_HAS_RI = 1, CS_HAS_RJ = 2, CS_RI_KNOWN = 4, CS_RI = 8, CS_ACC_KNOWN = 16, CS_ACC = 32,
                   CS_AI_T = 64, CS_AI_U = 128, CS_AJ_T = 256, CS_AJ_U = 512;

// (float)v, and whether v is so close to the midpoint of two floats that another libm's last-bit error could round it
// the other way
__device__ inline float round_guarded(double v, bool& risky) {
    const float f = (float)v;
    if (f != 0.f) {
        int ex;
        (void)frexp((double)f, &ex);                    // |f| in [2^(ex-1), 2^ex): float spacing 2^(ex-24) above f
        double half = ldexp(1.0, ex - 25);
        if ((__float_as_uint(f) & 0x7FFFFFu) == 0u && fabs(v) < fabs((double)f)) half *= 0.5;  // below a power of two
        const double e = fabs(v - (double)f);
        if (half - e < ldexp(1.0, ex - 47)) risky = true;
    }
    return f;
}

__global__ __launch_bounds__(256) void line_geo_kernel(LineTailArgs a) {
    const int b = blockIdx.y;
    int* cnt = a.counters + b * 8;
    const int nc = cnt[2];
    if (nc > a.tail_cap) return;
    const float* un = a.kp_un + (size_t)b * a.max_num * 2;
    const uint32_t* cij = a.cand_ij + (size_t)b * a.cand_cap;
    float* geo = a.geo + (size_t)b * a.tail_cap * 3;
    int* colc = a.col_off + (size_t)b * (a.max_num + 1);
    bool risky = false;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < nc; k += gridDim.x * 256) {
        const uint32_t ij = cij[k];
        const int i = ij >> 16, j = ij & 0xFFFFu;
        float vx = un[2 * j] - un[2 * i], vy = un[2 * j + 1] - un[2 * i + 1];   // :278-283
        const float dist = sqrtf(vx * vx + vy * vy);
        vx = vx / dist; vy = vy / dist;
        const float dij = round_guarded(atan2((double)vy, (double)vx), risky);
        float r = (float)((double)dij - kPiD);                                   // :284-286
        if ((double)r < -kPiD) r = (float)((double)r + k2PiD);
        geo[3 * k] = dist; geo[3 * k + 1] = dij; geo[3 * k + 2] = r;
        a.cst[(size_t)b * a.tail_cap + k] = 0u;
        a.kill_t[(size_t)b * a.tail_cap + k] = LT_INF;
        atomicAdd(&colc[j], 1);
    }
    if (risky) atomicOr(&cnt[5], LT_GUARD);
}

// exclusive scan of arr[0..n) in place by a 1024-thread work-group; returns the total (part: 1024 ints of LDS)
__device__ inline int block_scan_1024(int* arr, int n, int* part) {
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    int local = 0;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) local += arr[r];
    }
    int total;
    int run = block_incl_scan_1024(local, part, &total) - local;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) { const int c = arr[r]; arr[r] = run; run += c; }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void line_col_scan_kernel(LineTailArgs a) {
    __shared__ int part[1024];
    const int b = blockIdx.x;
    int* cnt = a.counters + b * 8;
    if (cnt[2] > a.tail_cap) {
        if (threadIdx.x == 0) atomicOr(&cnt[5], LT_CAP);
        return;
    }
    const int n = cnt[3];
    int* col = a.col_off + (size_t)b * (a.max_num + 1);
    const int total = block_scan_1024(col, n, part);
    if (threadIdx.x == 0) col[n] = total;
    for (int p = threadIdx.x; p < n; p += 1024) a.col_cur[(size_t)b * a.max_num + p] = 0;
}

__global__ __launch_bounds__(256) void line_col_fill_kernel(LineTailArgs a) {
    const int b = blockIdx.y;
    const int* cnt = a.counters + b * 8;
    const int nc = cnt[2];
    if (nc > a.tail_cap) return;
    const uint32_t* cij = a.cand_ij + (size_t)b * a.cand_cap;
    const int* col = a.col_off + (size_t)b * (a.max_num + 1);
    const float* geo = a.geo + (size_t)b * a.tail_cap * 3;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < nc; k += gridDim.x * 256) {
        const int j = cij[k] & 0xFFFFu;
        const int pos = col[j] + atomicAdd(&a.col_cur[(size_t)b * a.max_num + j], 1);   // order inside a column is immaterial
        // the list entry carries what a partner test needs (length, direction seen from j), so that walking a list is
        // one contiguous read instead of a gather per entry
        f32x4 e = {geo[3 * k], geo[3 * k + 2], __uint_as_float((uint32_t)k), 0.f};
        *(f32x4*)(a.col_ent + ((size_t)b * a.tail_cap + pos) * 4) = e;
    }
}

// Does the new line (direction dn, length ln at the shared end point) interact with the listed one (dy, ly)?
// 0 no, 1 the new line kills the listed one (:331-332), 2 the listed one rejects the new line (:333-334).
// The body is the one of lines_tail.h's walk.
__device__ inline int line_interaction(float dn, float ln, float dy, float ly, float distTh, bool& risky) {
    float ad = dn - dy;
    if ((double)ad < -kPiD) ad = (float)((double)ad + k2PiD);
    if ((double)ad > kPiD) ad = (float)((double)ad - k2PiD);
    ad = fabsf(ad);
    if ((double)ad > 0.2 * kPiD) return 0;
    const float dMin = ln <= ly ? ln : ly;
    const float pa = dMin * ad;
    bool close;
    if (pa * 0.93f > distTh) close = false;
    else if (pa * 1.00001f < distTh) close = true;
    else {
        const double x = ad, x2 = x * x, lo = x - x * x2 * (1.0 / 6.0), hi = lo + x * x2 * x2 * (1.0 / 120.0);
        if ((double)dMin * hi * (1.0 + 1e-6) < (double)distTh) close = true;
        else if ((double)dMin * lo * (1.0 - 1e-6) > (double)distTh) close = false;
        else close = dMin * round_guarded(sin((double)ad), risky) < distTh;
    }
    if (!close) return 0;
    return ln <= ly ? 1 : 2;
}

// Two lines interact only at a shared end point, so the pairs are enumerated point by point: one wave per point p
// holds the lines incident at p (those listed at p as second end point, then p's own row) sixty-four at a time in
// registers and tests every line against every earlier one (candidate order) by lane broadcast.  For a candidate
// c = (i, j) that is exactly its partners at i (the lines listed at i and the earlier ones of its row) and at j
// (the earlier lines listed at j).  Interacting pairs are gathered in LDS and written with one slot reservation
// per flush: a reservation per pass would put a returning atomic into the innermost loop.
constexpr int LI_BUF = 256;
__global__ __launch_bounds__(256) void line_inter_kernel(LineTailArgs a) {
    __shared__ uint32_t buf[4][LI_BUF][2];
    const int b = blockIdx.y;
    int* cnt = a.counters + b * 8;
    const int nc = cnt[2];
    if (nc > a.tail_cap) return;
    const int n = cnt[3];
    const float* geo = a.geo + (size_t)b * a.tail_cap * 3;
    const int* col = a.col_off + (size_t)b * (a.max_num + 1);
    const float* cole = a.col_ent + (size_t)b * a.tail_cap * 4;
    const int* rowoff = a.row_off + (size_t)b * (a.max_num + 1);
    uint32_t* cst = a.cst + (size_t)b * a.tail_cap;
    uint32_t* pairs = a.pairs + (size_t)b * a.pair_cap * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t (*wb)[2] = buf[wave];
    bool risky = false, full = false;
    int nh = 0;   // pairs waiting in the wave's buffer (wave-uniform)
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int base = 0;
        if (lane == 0) base = atomicAdd(&cnt[6], nh);
        base = __shfl(base, 0);
        for (int q = lane; q < nh; q += 64) {
            const int slot = base + q;
            if (slot >= a.pair_cap) { full = true; continue; }
            pairs[2 * slot] = wb[q][0];
            pairs[2 * slot + 1] = wb[q][1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        nh = 0;
    };
    for (int p = blockIdx.x * 4 + wave; p < n; p += gridDim.x * 4) {
        const int a0 = col[p], nA = col[p + 1] - a0;
        const int r0 = rowoff[p], nR = (p + 1 < n ? rowoff[p + 1] : nc) - r0;
        const int tot = nA + nR;
        if (tot < 2) continue;
        auto load = [&](int e, float& len, float& dir, int& k) {   // entry e of the point's list, its direction seen from p
            if (e < nA) {
                const f32x4 v = *(const f32x4*)(cole + (size_t)(a0 + e) * 4);
                len = v[0]; dir = v[1]; k = (int)__float_as_uint(v[2]);
            } else if (e < tot) {
                k = r0 + (e - nA); len = geo[3 * k]; dir = geo[3 * k + 1];
            } else { k = -1; len = 0.f; dir = 0.f; }
        };
        for (int cb = 0; cb < tot; cb += 64) {
            float cl, cd; int ck;
            load(cb + lane, cl, cd, ck);
            const int ep = (cb + lane) < nA ? 1 : 0;
            // earlier lines sit in the chunks up to the own one, except that the listed lines of a chunk are in no
            // particular order: their chunks are all visited
            const int y_end = cb + 64 < nA ? nA : min(cb + 64, tot);
            for (int yb = 0; yb < y_end; yb += 64) {
                float yl0, yd0; int yk0;
                load(yb + lane, yl0, yd0, yk0);
                const int m = min(64, tot - yb);
                for (int sidx = 0; sidx < m; sidx++) {
                    const float yl = __shfl(yl0, sidx), yd = __shfl(yd0, sidx);
                    const int yk = __shfl(yk0, sidx);
                    int code = 0;
                    if (yk < ck) code = line_interaction(cd, cl, yd, yl, a.dist_thresh, risky);
                    const unsigned long long hit = __ballot(code != 0);
                    if (hit == 0ull) continue;
                    if (code != 0) {
                        const int q = nh + __popcll(hit & ((1ull << lane) - 1ull));
                        wb[q][0] = (uint32_t)ck;
                        wb[q][1] = (uint32_t)yk | ((uint32_t)ep << 30) | ((uint32_t)(code == 1) << 31);
                        if (code == 2) atomicOr(&cst[ck], ep ? CS_HAS_RJ : CS_HAS_RI);
                    }
                    nh += __popcll(hit);
                    if (nh > LI_BUF - 64) flush();
                }
            }
        }
    }
    if (nh) flush();
    if (risky) atomicOr(&cnt[5], LT_GUARD);
    if (full) atomicOr(&cnt[5], LT_CAP);
}

constexpr int LT_MAXPTS = 4096;   // junction_max_num limit of the library
constexpr int LT_MAXDEG = 32;     // incident surviving lines per point the collinearity search holds in registers

// Where the relaxation keeps its three per-candidate arrays: in global memory (any frame), or as 16-bit entries in LDS
// (frames of at most LR_NC_LDS candidates, i.e. nearly all): with one work-group per frame every pass of a round is a
// round trip to wherever the arrays live, and the L2 is ten times farther than the LDS.
constexpr int LR_NC_LDS = 20480;
struct LrGlobal {
    uint32_t *cst, *kill, *pend;
    __device__ uint32_t get_cst(uint32_t k) const { return cst[k]; }
    __device__ void set_cst(uint32_t k, uint32_t v) const { cst[k] = v; }
    __device__ void or_cst(uint32_t k, uint32_t bits) const { atomicOr(&cst[k], bits); }
    __device__ uint32_t get_kill(uint32_t k) const { return kill[k]; }
    __device__ void min_kill(uint32_t k, uint32_t c) const { atomicMin(&kill[k], c); }
    __device__ uint32_t get_pend(uint32_t k) const { return pend[k]; }
    __device__ void min_pend(uint32_t k, uint32_t c) const { atomicMin(&pend[k], c); }
    __device__ void clear_pend(uint32_t k) const { pend[k] = LT_INF; }
};
struct LrLds {   // 16-bit entries, 0xFFFF = LT_INF (candidate indices stay below LR_NC_LDS)
    uint16_t *cst, *kill, *pend;
    __device__ static uint32_t widen(uint16_t v) { return v == 0xFFFFu ? LT_INF : (uint32_t)v; }
    __device__ static void min16(uint16_t* arr, uint32_t k, uint32_t c) {
        uint32_t* w = (uint32_t*)arr + (k >> 1);
        const int sh = 16 * (int)(k & 1u);
        uint32_t old = *w;
        while (true) {
            if (((old >> sh) & 0xFFFFu) <= c) break;
            const uint32_t nw = (old & ~(0xFFFFu << sh)) | (c << sh);
            const uint32_t prev = atomicCAS(w, old, nw);
            if (prev == old) break;
            old = prev;
        }
    }
    __device__ uint32_t get_cst(uint32_t k) const { return cst[k]; }
    __device__ void set_cst(uint32_t k, uint32_t v) const { cst[k] = (uint16_t)v; }
    __device__ void or_cst(uint32_t k, uint32_t bits) const { atomicOr((uint32_t*)cst + (k >> 1), bits << (16 * (k & 1u))); }
    __device__ uint32_t get_kill(uint32_t k) const { return widen(kill[k]); }
    __device__ void min_kill(uint32_t k, uint32_t c) const { min16(kill, k, c); }
    __device__ uint32_t get_pend(uint32_t k) const { return widen(pend[k]); }
    __device__ void min_pend(uint32_t k, uint32_t c) const { min16(pend, k, c); }
    __device__ void clear_pend(uint32_t k) const { pend[k] = 0xFFFFu; }
};

// The sequential overlap pass (:315-337) as a relaxation over the classified pairs; on return st holds the final
// "accepted" bits and kill times.  Returns the number of rounds.
template <class S>
__device__ int line_relax(const S st, const uint32_t* __restrict__ pairs, int np, int nc) {
    const int tid = threadIdx.x;
    for (int k = tid; k < nc; k += 1024) {
        uint32_t s = st.get_cst(k) & (CS_HAS_RI | CS_HAS_RJ);
        if (!(s & CS_HAS_RI)) s |= CS_RI_KNOWN;                                    // nothing at i can reject it
        if (!(s & (CS_HAS_RI | CS_HAS_RJ))) s |= CS_ACC_KNOWN | CS_ACC;
        st.set_cst(k, s);
    }
    // A thread's first LR_PR pairs stay in registers for all rounds, and every pass first issues all its loads and then
    // works on the results.
    constexpr int LR_PR = 8;
    uint32_t pw[LR_PR], pc[LR_PR];
#pragma unroll
    for (int q = 0; q < LR_PR; q++) {
        const int p = tid + 1024 * q;
        pw[q] = p < np ? pairs[2 * p + 1] : 0u;
        pc[q] = p < np ? pairs[2 * p] : 0xFFFFFFFFu;   // 0xFFFFFFFF: no pair
    }
    auto is_kill_i = [](uint32_t w) { return (w >> 31) && !((w >> 30) & 1u); };
    auto is_kill_j = [](uint32_t w) { return (w >> 31) && ((w >> 30) & 1u); };
    __syncthreads();   // (the LDS variant was filled by other threads)
#pragma unroll
    for (int q = 0; q < LR_PR; q++)   // kills at the first end point are attempted whatever happens to c (:316-335)
        if (pc[q] != 0xFFFFFFFFu && is_kill_i(pw[q])) st.min_kill(pw[q] & 0xFFFFu, pc[q]);
    for (int p = tid + 1024 * LR_PR; p < np; p += 1024) {
        const uint32_t w = pairs[2 * p + 1];
        if (is_kill_i(w)) st.min_kill(w & 0xFFFFu, pairs[2 * p]);
    }
    __syncthreads();
    int rounds = 0;
    for (int round = 0; round <= nc; round++) {
        for (int k = tid; k < nc; k += 1024) st.clear_pend(k);
        __syncthreads();
        // kills at the second end point happen only if c got past the first (:336-337)
        {
            uint32_t sq[LR_PR];
#pragma unroll
            for (int q = 0; q < LR_PR; q++) sq[q] = (pc[q] != 0xFFFFFFFFu && is_kill_j(pw[q])) ? st.get_cst(pc[q]) : 0u;
#pragma unroll
            for (int q = 0; q < LR_PR; q++) {
                if (pc[q] == 0xFFFFFFFFu || !is_kill_j(pw[q])) continue;
                if (sq[q] & CS_RI_KNOWN) { if (!(sq[q] & CS_RI)) st.min_kill(pw[q] & 0xFFFFu, pc[q]); }
                else st.min_pend(pw[q] & 0xFFFFu, pc[q]);
            }
        }
        for (int p = tid + 1024 * LR_PR; p < np; p += 1024) {
            const uint32_t w = pairs[2 * p + 1];
            if (!is_kill_j(w)) continue;
            const uint32_t c = pairs[2 * p], s = st.get_cst(c);
            if (s & CS_RI_KNOWN) { if (!(s & CS_RI)) st.min_kill(w & 0xFFFFu, c); }
            else st.min_pend(w & 0xFFFFu, c);
        }
        __syncthreads();
        // is the partner that would reject c alive when c arrives?
        auto reject_bit = [&](uint32_t c, uint32_t w, uint32_t s, uint32_t sy, uint32_t kt, uint32_t pt) {
            const uint32_t ep = (w >> 30) & 1u;
            if (s & CS_ACC_KNOWN) return;
            if (!ep && (s & CS_RI_KNOWN)) return;
            if (((sy & CS_ACC_KNOWN) && !(sy & CS_ACC)) || kt < c) return;                      // dead or never listed
            uint32_t bit;
            if ((sy & CS_ACC_KNOWN) && pt > c) bit = ep ? CS_AJ_T : CS_AI_T;                   // alive
            else bit = ep ? CS_AJ_U : CS_AI_U;                                                  // not settled yet
            st.or_cst(c, bit);
        };
        {
            uint32_t s_[LR_PR], sy_[LR_PR], kt_[LR_PR], pt_[LR_PR];
#pragma unroll
            for (int q = 0; q < LR_PR; q++) {
                const bool on = pc[q] != 0xFFFFFFFFu && !(pw[q] >> 31);
                const uint32_t y = pw[q] & 0xFFFFu;
                s_[q] = on ? st.get_cst(pc[q]) : CS_ACC_KNOWN;
                sy_[q] = on ? st.get_cst(y) : 0u;
                kt_[q] = on ? st.get_kill(y) : 0u;
                pt_[q] = on ? st.get_pend(y) : 0u;
            }
#pragma unroll
            for (int q = 0; q < LR_PR; q++)
                if (pc[q] != 0xFFFFFFFFu && !(pw[q] >> 31)) reject_bit(pc[q], pw[q], s_[q], sy_[q], kt_[q], pt_[q]);
        }
        for (int p = tid + 1024 * LR_PR; p < np; p += 1024) {
            const uint32_t w = pairs[2 * p + 1];
            if (w >> 31) continue;
            const uint32_t c = pairs[2 * p], y = w & 0xFFFFu;
            reject_bit(c, w, st.get_cst(c), st.get_cst(y), st.get_kill(y), st.get_pend(y));
        }
        __syncthreads();
        int open = 0;
        for (int k = tid; k < nc; k += 1024) {
            uint32_t s = st.get_cst(k);
            if (s & CS_ACC_KNOWN) continue;
            if (!(s & CS_RI_KNOWN)) {
                if (s & CS_AI_T) s |= CS_RI_KNOWN | CS_RI | CS_ACC_KNOWN;        // rejected at i: never listed
                else if (!(s & CS_AI_U)) s |= CS_RI_KNOWN;
            }
            if ((s & CS_RI_KNOWN) && !(s & CS_RI) && !(s & CS_ACC_KNOWN)) {
                if (s & CS_AJ_T) s |= CS_ACC_KNOWN;
                else if (!(s & CS_AJ_U)) s |= CS_ACC_KNOWN | CS_ACC;
            }
            s &= ~(CS_AI_T | CS_AI_U | CS_AJ_T | CS_AJ_U);
            st.set_cst(k, s);
            open += !(s & CS_ACC_KNOWN);
        }
        rounds = round + 1;
        if (__syncthreads_count(open) == 0) break;
    }
    // the last round's second-end-point kills of candidates settled in that round
    for (int p = tid; p < np; p += 1024) {
        const uint32_t w = pairs[2 * p + 1];
        if (!is_kill_j(w)) continue;
        const uint32_t c = pairs[2 * p], s = st.get_cst(c);
        if ((s & CS_RI_KNOWN) && !(s & CS_RI)) st.min_kill(w & 0xFFFFu, c);
    }
    __syncthreads();
    return rounds;
}

__global__ __launch_bounds__(1024) void line_resolve_kernel(LineTailArgs a) {
    // one buffer: the relaxation's 16-bit arrays first, the per-point counters of the later phases afterwards
    __shared__ __attribute__((aligned(16))) uint32_t lds_words[3 * LR_NC_LDS / 2];
    static_assert(3 * LR_NC_LDS / 2 >= 2 * (LT_MAXPTS + 1), "deg and cur live in the same buffer");
    int* deg = (int*)lds_words;
    int* cur = deg + LT_MAXPTS + 1;
    __shared__ int part[1024];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x;
    int* cnt = a.counters + b * 8;
    const int n = cnt[3], nc = cnt[2];
    int32_t* conn_off = a.conn_off + (size_t)b * (a.max_num + 1);
    int32_t* coline_off = a.coline_off + (size_t)b * (a.max_num + 1);
    // The per-point counts that line_geo_kernel accumulates (col_off) were last read by line_inter_kernel: this frame's row
    // goes back to zero here for the next batch, which spares the chain a clearing launch (the array starts zeroed).
    for (int p = tid; p <= a.max_num; p += 1024) a.col_off[(size_t)b * (a.max_num + 1) + p] = 0;
    if (n == 0 || nc == 0) {
        // PPGExtractor.cpp:239-240 returns early without key points; without candidates every list is empty
        for (int p = tid; p <= n; p += 1024) { conn_off[p] = 0; coline_off[p] = 0; }
        if (tid == 0) cnt[4] = 0;
        return;
    }
    if (nc > a.tail_cap || cnt[5] != 0) return;   // the host decides this frame (flags are complete: earlier kernels)
    const int np = min(cnt[6], a.pair_cap);
    const uint32_t* cij = a.cand_ij + (size_t)b * a.cand_cap;
    const float* csc = a.cand_sc + (size_t)b * a.cand_cap * 2;
    const float* geo = a.geo + (size_t)b * a.tail_cap * 3;
    const uint32_t* pairs = a.pairs + (size_t)b * a.pair_cap * 2;
    uint32_t* cst = a.cst + (size_t)b * a.tail_cap;
    uint32_t* kill_t = a.kill_t + (size_t)b * a.tail_cap;
    uint32_t* pend_t = a.pend_t + (size_t)b * a.tail_cap;
    if (tid == 0) s_flag = 0;

    // ---- the sequential overlap pass as a relaxation ----
    int rounds;
    if (nc <= LR_NC_LDS) {
        LrLds st;
        st.cst = (uint16_t*)lds_words;
        st.kill = st.cst + LR_NC_LDS;
        st.pend = st.kill + LR_NC_LDS;
        // the interaction kernel left the "something may reject it" bits in cst; every kill time starts at "never"
        for (int k = tid; k < nc; k += 1024) { st.cst[k] = (uint16_t)cst[k]; st.kill[k] = 0xFFFFu; }
        __syncthreads();
        rounds = line_relax(st, pairs, np, nc);
        for (int k = tid; k < nc; k += 1024) { cst[k] = st.cst[k]; kill_t[k] = LrLds::widen(st.kill[k]); }
        __syncthreads();
    } else {
        LrGlobal st{cst, kill_t, pend_t};
        rounds = line_relax(st, pairs, np, nc);
    }
    if (tid == 0) cnt[7] |= rounds << 8;
    for (int p = tid; p <= n; p += 1024) deg[p] = 0;
    __syncthreads();

    // ---- score thresholds (:367-389) and ordered compaction (:433-441) ----
    // A thread takes a run of consecutive candidates: it decides them from loads issued together, the runs' survivor counts
    // go through one scan, and the survivors are written at their ordered positions.
    uint32_t* e2k = pend_t;   // edge -> candidate (pend_t is free now)
    const int per = (nc + 1023) / 1024;   // <= 64: tail_cap <= 65536
    const int kbeg = tid * per, kend = min(nc, kbeg + per);
    unsigned long long keepm = 0ull;
    for (int i0 = 0; i0 < per; i0 += 8) {
        uint32_t s8[8], kt8[8];
        f32x2 sc8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = kbeg + i0 + u;
            const bool in = i0 + u < per && k < kend;
            s8[u] = in ? cst[k] : 0u;
            kt8[u] = in ? kill_t[k] : 0u;
            sc8[u] = in ? *(const f32x2*)(csc + 2 * (size_t)k) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            bool keep = (s8[u] & CS_ACC) && kt8[u] == LT_INF;
            if (keep) {
                if (sc8[u][0] < a.inlier_rate) keep = false;
                else if (sc8[u][1] < a.heat_thresh) keep = false;
            }
            if (keep) keepm |= 1ull << (i0 + u);
        }
    }
    const int mine_cnt = __popcll(keepm);
    int ebase;
    const int mine_end = block_incl_scan_1024(mine_cnt, part, &ebase);
    {
        int e = mine_end - mine_cnt;
        unsigned long long m = keepm;
        while (m) {
            const int i = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int k = kbeg + i;
            if (e < a.edge_cap) {
                const uint32_t ij = cij[k];
                const f32x2 sc = *(const f32x2*)(csc + 2 * (size_t)k);
                a.edge_idx[((size_t)b * a.edge_cap + e) * 2] = ij >> 16;
                a.edge_idx[((size_t)b * a.edge_cap + e) * 2 + 1] = ij & 0xFFFFu;
                a.edge_score[(size_t)b * a.edge_cap + e] = sc[0] * sc[1];
                e2k[e] = (uint32_t)k;
                atomicAdd(&deg[ij >> 16], 1);
                atomicAdd(&deg[ij & 0xFFFFu], 1);
            }
            e++;
        }
    }
    __syncthreads();
    const int E = ebase;
    if (E > a.edge_cap) {
        if (tid == 0) atomicOr(&cnt[5], LT_CAP);
        return;
    }
    // ---- per-point lists of edge ids, ascending (KeyPointEx::mvConnected) ----
    for (int p = tid; p < n; p += 1024) cur[p] = 0;
    (void)block_scan_1024(deg, n, part);
    if (tid == 0) deg[n] = 2 * E;
    __syncthreads();
    for (int p = tid; p <= n; p += 1024) conn_off[p] = deg[p];
    uint32_t* conn = a.conn_idx + (size_t)b * a.edge_cap * 2;
    const uint32_t* eidx = a.edge_idx + (size_t)b * a.edge_cap * 2;
    for (int e = tid; e < E; e += 1024) {
        const int i = eidx[2 * e], j = eidx[2 * e + 1];
        conn[deg[i] + atomicAdd(&cur[i], 1)] = (uint32_t)e;
        conn[deg[j] + atomicAdd(&cur[j], 1)] = (uint32_t)e;
    }
    __syncthreads();
    // ---- per point: sort its list, then the collinearity search (:392-432) ----
    uint32_t* ctmp = kill_t;   // pairs of point p at (deg[p] / 2 ...): at most half its degree
    bool risky = false;
    for (int p = tid; p < n; p += 1024) {
        const int o = deg[p], d = deg[p + 1] - o;
        for (int x = 1; x < d; x++) {   // insertion sort (lists are short)
            const uint32_t v = conn[o + x];
            int y = x - 1;
            while (y >= 0 && conn[o + y] > v) { conn[o + y + 1] = conn[o + y]; y--; }
            conn[o + y + 1] = v;
        }
        int npair = 0;
        if (d > LT_MAXDEG) { atomicOr(&s_flag, 1); }
        else if (d >= 2) {
            uint32_t idx[LT_MAXDEG];
            for (int x = 0; x < d; x++) idx[x] = conn[o + x];
            int sz = d;
            auto dir_from = [&](uint32_t e) { const uint32_t k = e2k[e]; return (int)eidx[2 * e] == p ? geo[3 * k + 1] : geo[3 * k + 2]; };
            auto other = [&](uint32_t e) { return (int)eidx[2 * e] == p ? eidx[2 * e + 1] : eidx[2 * e]; };
            while (sz > 1) {
                double minDiff = 1e9;
                int bestId = -1;
                uint32_t b1 = 0, b2 = 0;
                const uint32_t e1 = idx[sz - 1];
                const float d1 = geo[3 * e2k[e1]], dir1 = dir_from(e1);
                for (int q = 0; q + 1 < sz; q++) {
                    const uint32_t e2 = idx[q];
                    const float ad = dir1 - dir_from(e2);
                    const float sn = round_guarded(sin((double)ad), risky);
                    const double pd = 0.5 * (double)(d1 + geo[3 * e2k[e2]]) * (double)fabsf(sn);
                    if (minDiff > pd) { minDiff = pd; bestId = q; b1 = other(e1); b2 = other(e2); }
                }
                if (minDiff > (double)a.dist_thresh) { sz--; continue; }
                ctmp[2 * (o / 2 + npair)] = b1;
                ctmp[2 * (o / 2 + npair) + 1] = b2;
                npair++;
                sz--;
                idx[bestId] = idx[sz - 1];
                sz--;
            }
        }
        cur[p] = npair;
    }
    __syncthreads();
    if (risky) atomicOr(&s_flag, 2);
    __syncthreads();
    if (s_flag) {
        if (tid == 0) atomicOr(&cnt[5], (s_flag & 1 ? LT_CAP : 0) | (s_flag & 2 ? LT_GUARD : 0));
        return;
    }
    const int npairs = block_scan_1024(cur, n, part);
    if (tid == 0) cur[n] = npairs;
    __syncthreads();
    uint32_t* cpairs = a.coline_pairs + (size_t)b * a.edge_cap * 2;
    for (int p = tid; p <= n; p += 1024) coline_off[p] = cur[p];
    for (int p = tid; p < n; p += 1024) {
        const int cnum = cur[p + 1] - cur[p], src = deg[p] / 2;
        for (int x = 0; x < cnum; x++) {
            cpairs[2 * (cur[p] + x)] = ctmp[2 * (src + x)];
            cpairs[2 * (cur[p] + x) + 1] = ctmp[2 * (src + x) + 1];
        }
    }
    if (tid == 0) cnt[4] = E;
}

hipError_t launch_line_tail(const LineTailArgs& a, hipStream_t s) {
    if (a.max_num > LT_MAXPTS || a.tail_cap > 65536 || a.tail_cap < 2 * a.edge_cap + 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(line_geo_kernel, dim3(32, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(line_col_scan_kernel, dim3(a.B), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(line_col_fill_kernel, dim3(32, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(line_inter_kernel, dim3(256, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(line_resolve_kernel, dim3(a.B), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// genPointDescriptor (feature/src/PPGExtractor.cpp:515-538): grid_sampler(bilinear, zeros,
// align_corners=false) of the raw descriptor map at the integer key-point positions, then
// F::normalize(dim=1).  One wave per key point, four channels per lane (1 KiB coalesced per tap).
// =====================================================================================
__global__ __launch_bounds__(256) void desc_sample_kernel(DescArgs a) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = a.counters[b * 8 + 3];
    if (k >= n) return;
    float* o = a.desc + ((size_t)b * a.max_num + k) * 256 + lane * 4;
    if (n < 10) {  // PPGExtractor.cpp:520-524
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)o = z;
        return;
    }
    const int w = a.W / 8, h = a.H / 8;
    const int32_t* xy = a.kp_xy + ((size_t)b * a.max_num + k) * 2;
    // :530-531: f32 / f32, then "* 2. - 1." in double, stored as f32
    const float gx = (float)((double)((float)xy[0] / (float)a.W) * 2. - 1.);
    const float gy = (float)((double)((float)xy[1] / (float)a.H) * 2. - 1.);
    const float ix = ((gx + 1.f) * (float)w - 1.f) / 2.f, iy = ((gy + 1.f) * (float)h - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
    const float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
    const float* dm = a.descmap + (size_t)b * h * w * 256 + lane * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    auto tap = [&](int y, int x, float wt) {
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const f32x4 t = *(const f32x4*)(dm + ((size_t)y * w + x) * 256);
            v[0] = v[0] + t[0] * wt; v[1] = v[1] + t[1] * wt; v[2] = v[2] + t[2] * wt; v[3] = v[3] + t[3] * wt;
        }
    };
    tap(y0, x0, nw); tap(y0, x1, ne); tap(y1, x0, sw); tap(y1, x1, se);
    double ss = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    ss = wave_sum_d(ss);
    const float nrm = (float)sqrt(ss);
    const float den = nrm > 1e-12f ? nrm : 1e-12f;
    v[0] = v[0] / den; v[1] = v[1] / den; v[2] = v[2] / den; v[3] = v[3] / den;
    *(f32x4*)o = v;
}

hipError_t launch_desc_sample(const DescArgs& a, hipStream_t s) {
    dim3 grid((a.max_num + 3) / 4, a.B);
    hipLaunchKernelGGL(desc_sample_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ppg
