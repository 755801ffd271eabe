// gfx950 (MI355X, CDNA4) kernels of the PPG front-end's matchers: descriptor distances, the nearest-neighbour searches and
// their order-dependent claim passes, the BoW tree descent, the projection matchers' pass, the point-pair-graph walk.
// Wave = 64 lanes everywhere.
//
// Built with -ffp-contract=off: every float expression below rounds exactly as written, and fused
// multiply-adds appear only where fmaf()/MFMA is spelled out.  That is what lets the match indices agree bit for bit
// with a strict-IEEE CPU evaluation of the reference's formulas.
#include "ppg_kernels.h"
#include "conv_split.h"
#include "conv_store.h"
#include "wave_util.h"

#include <math.h>
#include <algorithm>
#include <type_traits>

namespace ppg {

// =====================================================================================
// DescriptorDistance (feature/src/MapPoint.cpp:22-29) for all pairs of two descriptor sets.
// 64x64 output tile per work-group, 4x4 outputs per thread, 16-deep slices through LDS.  Each
// output is one fmaf chain over d = 0..dim-1 in order, then a correctly rounded square root, so the
// value does not depend on the tiling.
// =====================================================================================
constexpr int DT = 64, DK = 16;

__global__ __launch_bounds__(256) void dist_matrix_kernel(const float* __restrict__ A, int nA, const float* __restrict__ B,
                                                          int nB, int dim, float* __restrict__ D, int ldd,
                                                          const int* nA_dev, const int* nB_dev, size_t sA, size_t sB,
                                                          size_t sD, int sn) {
    const int bz = blockIdx.z;
    A += bz * sA; B += bz * sB; D += bz * sD;
    if (nA_dev) nA = min(nA, nA_dev[bz * sn]);
    if (nB_dev) nB = min(nB, nB_dev[bz * sn]);
    const int r0 = blockIdx.y * DT, c0 = blockIdx.x * DT;
    if (r0 >= nA || c0 >= nB) return;
    __shared__ float As[DK][DT + 4];
    __shared__ float Bs[DK][DT + 4];
    const int tid = threadIdx.x, tr = tid / 16, tc = tid % 16;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    // staging: thread loads one float4 of A and one of B per slice: row = tid/4, k4 = tid%4
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    for (int k0 = 0; k0 < dim; k0 += DK) {
        f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (r0 + lr < nA) va = *(const f32x4*)(A + (size_t)(r0 + lr) * dim + k0 + lk);
        if (c0 + lr < nB) vb = *(const f32x4*)(B + (size_t)(c0 + lr) * dim + k0 + lk);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) { As[lk + q][lr] = va[q]; Bs[lk + q][lr] = vb[q]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DK; k++) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) av[i] = As[k][tr * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = Bs[k][tc * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float d = av[i] - bv[j];
                    acc[i][j] = fmaf(d, d, acc[i][j]);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + tr * 4 + i, c = c0 + tc * 4 + j;
            if (r < nA && c < nB) D[(size_t)r * ldd + c] = sqrtf(acc[i][j]);
        }
}

hipError_t launch_dist_matrix(const float* A, int nA, const float* B, int nB, int dim, float* D, int ldd, hipStream_t s) {
    if (nA <= 0 || nB <= 0) return hipSuccess;
    dim3 grid((nB + DT - 1) / DT, (nA + DT - 1) / DT);
    hipLaunchKernelGGL(dist_matrix_kernel, grid, dim3(256), 0, s, A, nA, B, nB, dim, D, ldd, (const int*)nullptr,
                       (const int*)nullptr, (size_t)0, (size_t)0, (size_t)0, 0);
    return hipGetLastError();
}

hipError_t launch_dist_matrix_batched(const float* A, const int* nA_dev, int maxA, size_t sA, const float* B,
                                      const int* nB_dev, int maxB, size_t sB, int dim, float* D, int ldd, size_t sD,
                                      int sn, int batch, hipStream_t s) {
    dim3 grid((maxB + DT - 1) / DT, (maxA + DT - 1) / DT, batch);
    hipLaunchKernelGGL(dist_matrix_kernel, grid, dim3(256), 0, s, A, maxA, B, maxB, dim, D, ldd, nA_dev, nB_dev, sA, sB, sD, sn);
    return hipGetLastError();
}

// one lane per (row, candidate) pair of a CSR candidate list; same fmaf chain
__global__ __launch_bounds__(256) void dist_csr_kernel(const float* __restrict__ A, int nA, const float* __restrict__ B, int dim,
                                                       const int32_t* __restrict__ cand_off,
                                                       const int32_t* __restrict__ cand_idx, float* __restrict__ Dc) {
    const int row = blockIdx.x;
    if (row >= nA) return;
    const int o0 = cand_off[row], o1 = cand_off[row + 1];
    const float* a = A + (size_t)row * dim;
    for (int k = o0 + threadIdx.x; k < o1; k += 256) {
        const float* bb = B + (size_t)cand_idx[k] * dim;
        float acc = 0.f;
        for (int d = 0; d < dim; d += 4) {
            const f32x4 x = *(const f32x4*)(a + d), y = *(const f32x4*)(bb + d);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float t = x[q] - y[q];
                acc = fmaf(t, t, acc);
            }
        }
        Dc[k] = sqrtf(acc);
    }
}

hipError_t launch_dist_csr(const float* A, int nA, const float* B, int dim, const int32_t* cand_off,
                           const int32_t* cand_idx, float* Dc, hipStream_t s) {
    if (nA <= 0) return hipSuccess;
    hipLaunchKernelGGL(dist_csr_kernel, dim3(nA), dim3(256), 0, s, A, nA, B, dim, cand_off, cand_idx, Dc);
    return hipGetLastError();
}

// =====================================================================================
// Nearest-neighbour reductions with wavefront shuffles.
//
// Sequential semantics being reproduced (e.g. matching/src/Matcher.cpp:607-627): walk the candidates
// in order; `dist < best` shifts best into second; `else if dist < second` updates second.  The
// result is (min, first position of the min, second smallest of the multiset), which merges
// associatively — so lanes scan strided sub-lists and combine with shuffles.
// =====================================================================================
struct Best2 {
    float b, b2;
    int pos;  // position in candidate order (for ties), -1 when empty
    int idx;  // B row
};

__device__ inline Best2 best2_merge(Best2 x, Best2 y) {
    const bool xwins = x.b < y.b || (x.b == y.b && (unsigned)x.pos < (unsigned)y.pos);
    Best2 r;
    if (xwins) { r.b = x.b; r.pos = x.pos; r.idx = x.idx; r.b2 = fminf(x.b2, y.b); }
    else { r.b = y.b; r.pos = y.pos; r.idx = y.idx; r.b2 = fminf(y.b2, x.b); }
    return r;
}

__device__ inline Best2 best2_wave(Best2 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best2 t;
        t.b = __shfl_xor(v.b, o); t.b2 = __shfl_xor(v.b2, o);
        t.pos = __shfl_xor(v.pos, o); t.idx = __shfl_xor(v.idx, o);
        v = best2_merge(v, t);
    }
    return v;
}

__device__ inline void apply_batch(MatchArgs& a, int b) {
    a.D += b * a.bs_D;
    a.topk_idx += b * a.bs_rows * 8; a.topk_dist += b * a.bs_rows * 8;
    a.match12 += b * a.bs_rows; a.best += b * a.bs_rows; a.second += b * a.bs_rows;
    if (a.nA_dev) a.nA_dev += b * a.bs_n;
    if (a.nB_dev) a.nB_dev += b * a.bs_n;
    a.n_matches += b;
}

__device__ inline void row_candidates(const MatchArgs& a, int row, int nB, const float** drow, const int32_t** irow, int* cnt) {
    if (a.ldd > 0) { *drow = a.D + (size_t)row * a.ldd; *irow = nullptr; *cnt = nB; }
    else { const int o0 = a.cand_off[row]; *drow = a.D + o0; *irow = a.cand_idx + o0; *cnt = a.cand_off[row + 1] - o0; }
}

// PLAIN: one wave per A row
__global__ __launch_bounds__(256) void match_plain_kernel(MatchArgs a) {
    apply_batch(a, blockIdx.y);
    const int nA = a.nA_dev ? min(a.nA, *a.nA_dev) : a.nA, nB = a.nB_dev ? min(a.nB, *a.nB_dev) : a.nB;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float* drow; const int32_t* irow; int cnt;
    row_candidates(a, row, nB, &drow, &irow, &cnt);
    Best2 v = {1e6f, 1e6f, -1, -1};
    for (int k = lane; k < cnt; k += 64) {
        const int idx = irow ? irow[k] : k;
        if (a.skipB && a.skipB[idx]) continue;
        const float d = drow[k];
        if (d < v.b) { v.b2 = v.b; v.b = d; v.pos = k; v.idx = idx; }
        else if (d < v.b2) v.b2 = d;
    }
    v = best2_wave(v);
    if (lane == 0) { a.match12[row] = v.idx; a.best[row] = v.b; a.second[row] = v.b2; }
}

hipError_t launch_match_plain(const MatchArgs& a, hipStream_t s) {
    if (a.nA <= 0) return hipSuccess;
    hipLaunchKernelGGL(match_plain_kernel, dim3((a.nA + 3) / 4, a.batch), dim3(256), 0, s, a);
    return hipGetLastError();
}

// INIT step 1: per A row, the 8 nearest candidates ordered by (distance, candidate position),
// ignoring claims.  One wave per row: every lane keeps a sorted list of its own 8 best, then the
// wave pops the global minimum 8 times.
constexpr int TOPK = 8;

__global__ __launch_bounds__(256) void match_topk_kernel(MatchArgs a) {
    apply_batch(a, blockIdx.y);
    const int nA = a.nA_dev ? min(a.nA, *a.nA_dev) : a.nA, nB = a.nB_dev ? min(a.nB, *a.nB_dev) : a.nB;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nA) return;
    const float* drow; const int32_t* irow; int cnt;
    row_candidates(a, row, nB, &drow, &irow, &cnt);
    float ld[TOPK];
    int lp[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; t++) { ld[t] = 3e38f; lp[t] = 0x7FFFFFFF; }
    for (int k = lane; k < cnt; k += 64) {
        const int idx = irow ? irow[k] : k;
        if (a.skipB && a.skipB[idx]) continue;
        float d = drow[k];
        int p = k;
        // insertion into the sorted list; positions ascend within a lane, so ties keep the earlier one first
#pragma unroll
        for (int t = 0; t < TOPK; t++) {
            if (d < ld[t]) {
                const float td = ld[t]; const int tp = lp[t];
                ld[t] = d; lp[t] = p; d = td; p = tp;
            }
        }
    }
    for (int t = 0; t < TOPK; t++) {
        // wave argmin over list heads
        float d = ld[0];
        int p = lp[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(d, o);
            const int op = __shfl_xor(p, o);
            if (od < d || (od == d && op < p)) { d = od; p = op; }
        }
        if (lp[0] == p && p != 0x7FFFFFFF) {  // this lane owned the winner: pop
#pragma unroll
            for (int q = 0; q < TOPK - 1; q++) { ld[q] = ld[q + 1]; lp[q] = lp[q + 1]; }
            ld[TOPK - 1] = 3e38f; lp[TOPK - 1] = 0x7FFFFFFF;
        }
        if (lane == 0) {
            const bool valid = p != 0x7FFFFFFF;
            a.topk_dist[(size_t)row * TOPK + t] = valid ? d : 3e38f;
            a.topk_idx[(size_t)row * TOPK + t] = valid ? (irow ? irow[p] : p) : -1;
        }
    }
}

// INIT step 2: the reference's sequential pass over i1 (Matcher.cpp:591-644) by ONE wave.
// A claimed F2 feature stores (int)bestDist == 0 in vMatchedDistance (:588, :640), so the test
// `vMatchedDistance[i2] <= dist` (:614) skips it for every later i1; the un-claim branch (:633-637)
// can therefore never run.  Rows whose 8-entry list cannot supply two unclaimed candidates while the
// row has more candidates fall back to a full scan of the row by the wave.  The claim bitmap lives
// in LDS and the lists are fetched eight rows at a time, so a row costs LDS latency only.
constexpr int CLAIM_WORDS = 2048;  // up to 65536 B rows

__global__ __launch_bounds__(64) void match_init_resolve_kernel(MatchArgs a) {
    __shared__ uint32_t claimed[CLAIM_WORDS];
    apply_batch(a, blockIdx.x);
    const int nA = a.nA_dev ? min(a.nA, *a.nA_dev) : a.nA, nB = a.nB_dev ? min(a.nB, *a.nB_dev) : a.nB;
    const int lane = threadIdx.x;
    for (int k = lane; k < CLAIM_WORDS; k += 64) claimed[k] = 0;
    __syncthreads();
    int nmatches = 0;
    // lists of the first group; each later group is fetched while the previous one is being resolved
    float dv = 3e38f;
    int iv = -1;
    if ((lane >> 3) < nA) {
        dv = a.topk_dist[lane];
        iv = a.topk_idx[lane];
    }
    for (int r0 = 0; r0 < nA; r0 += 8) {
        const float dcur = dv;
        const int icur = iv;
        dv = 3e38f;
        iv = -1;
        if (r0 + 8 + (lane >> 3) < nA) {  // lane -> (row r0 + 8 + lane/8, entry lane%8): one coalesced 256-B read per array
            dv = a.topk_dist[(size_t)(r0 + 8) * TOPK + lane];
            iv = a.topk_idx[(size_t)(r0 + 8) * TOPK + lane];
        }
        const int rows = min(8, nA - r0);
        int om = -1;
        float ob = 1e6f, os = 1e6f;  // results of row r0 + lane (lane < rows), stored once per group
        for (int rr = 0; rr < rows; rr++) {
            const int row = r0 + rr;
            const int srcl = rr * 8 + (lane & 7);
            float d = __shfl(dcur, srcl);
            int idx = __shfl(icur, srcl);
            if (lane >= TOPK) { d = 3e38f; idx = -1; }
            const bool listed = idx >= 0;
            bool free_ = false;
            if (listed) free_ = !((claimed[idx >> 5] >> (idx & 31)) & 1u);
            const unsigned long long mlist = __ballot(listed), mfree = __ballot(free_);
            const int nlisted = __popcll(mlist), nfree = __popcll(mfree);
            float best = 1e6f, second = 1e6f;
            int bidx = -1;
            if (nfree >= 2 || nlisted < TOPK) {
                // the list suffices: first two unclaimed entries in (distance, position) order
                if (nfree >= 1) {
                    const int l0 = __ffsll((long long)mfree) - 1;
                    best = __shfl(d, l0);
                    bidx = __shfl(idx, l0);
                    const unsigned long long rest = mfree & ~(1ull << l0);
                    if (rest) second = __shfl(d, __ffsll((long long)rest) - 1);
                }
            } else {
                const float* drow; const int32_t* irow; int cnt;
                row_candidates(a, row, nB, &drow, &irow, &cnt);
                Best2 v = {1e6f, 1e6f, -1, -1};
                for (int k = lane; k < cnt; k += 64) {
                    const int id = irow ? irow[k] : k;
                    if (a.skipB && a.skipB[id]) continue;
                    if ((claimed[id >> 5] >> (id & 31)) & 1u) continue;
                    const float dd = drow[k];
                    if (dd < v.b) { v.b2 = v.b; v.b = dd; v.pos = k; v.idx = id; }
                    else if (dd < v.b2) v.b2 = dd;
                }
                v = best2_wave(v);
                best = v.b; second = v.b2; bidx = v.idx;
            }
            int m = -1;
            if (bidx >= 0 && best <= a.th && best < second * a.ratio) {
                m = bidx;
                nmatches++;
                if (lane == 0) claimed[bidx >> 5] |= 1u << (bidx & 31);
            }
            if (lane == rr) { om = m; ob = best; os = second; }
            // One wave: its LDS operations execute in program order, so the claim above is in LDS before
            // the next row's lanes read the bitmap; the fence only stops the compiler from reordering them.
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < rows) { a.match12[r0 + lane] = om; a.best[r0 + lane] = ob; a.second[r0 + lane] = os; }
    }
    if (lane == 0) *a.n_matches = nmatches;
}

// INIT step 2 for problems of at most MJ_MAX rows on either side: the same pass as a fixed point, all rows at once.
// Row i's match depends only on the claims of the rows before it, so a guess of every row's match that reproduces
// itself when each row recomputes its own from the claims the guesses of the earlier rows imply IS the sequential result
// (rows 0..t-1 are final after round t; a handful of rounds in practice, at most nA + 1).  A thread per row works
// from the row's eight-nearest list; the rows whose list cannot supply two free candidates (while the row has more than
// eight) are queued and scanned whole, a wave per row, reading the row's distances side by side.  Claims (the
// smallest row whose guess is the column), guesses and the queue live in LDS.
constexpr int MJ_MAX = 8192;

__global__ __launch_bounds__(1024) void match_init_jacobi_kernel(MatchArgs a) {
    __shared__ uint32_t claim[MJ_MAX];
    __shared__ int32_t guess[MJ_MAX];
    __shared__ uint16_t fb_rows[MJ_MAX];
    __shared__ int nfb, tot;
    apply_batch(a, blockIdx.x);
    const int nA = a.nA_dev ? min(a.nA, *a.nA_dev) : a.nA, nB = a.nB_dev ? min(a.nB, *a.nB_dev) : a.nB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nA; i += 1024) guess[i] = -1;
    if (tid == 0) tot = 0;
    __syncthreads();
    for (int round = 0; round <= nA; round++) {
        for (int j = tid; j < nB; j += 1024) claim[j] = 0xFFFFFFFFu;
        if (tid == 0) nfb = 0;
        __syncthreads();
        for (int i = tid; i < nA; i += 1024) {
            const int g = guess[i];
            if (g >= 0) atomicMin(&claim[g], (uint32_t)i);
        }
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < nA; i += 1024) {
            const f32x4 d0 = *(const f32x4*)(a.topk_dist + (size_t)i * TOPK), d1 = *(const f32x4*)(a.topk_dist + (size_t)i * TOPK + 4);
            const int4 i0 = *(const int4*)(a.topk_idx + (size_t)i * TOPK), i1 = *(const int4*)(a.topk_idx + (size_t)i * TOPK + 4);
            const float d[TOPK] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            const int id[TOPK] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            float best = 1e6f, second = 1e6f;
            int bidx = -1, nfree = 0, nlisted = 0;
#pragma unroll
            for (int t = 0; t < TOPK; t++) {
                if (id[t] < 0) continue;
                nlisted++;
                if (claim[id[t]] < (uint32_t)i) continue;   // taken by an earlier row
                if (nfree == 0) { best = d[t]; bidx = id[t]; }
                else if (nfree == 1) second = d[t];
                nfree++;
            }
            if (nfree < 2 && nlisted == TOPK) {   // the list ran dry and the row has more candidates
                fb_rows[atomicAdd(&nfb, 1)] = (uint16_t)i;
                continue;
            }
            const int m = (bidx >= 0 && best <= a.th && best < second * a.ratio) ? bidx : -1;
            changed |= m != guess[i];
            guess[i] = m;   // (read again only by this thread, when the claims of the next round are built)
            a.best[i] = best;
            a.second[i] = second;
        }
        __syncthreads();
        const int nq = nfb;
        for (int q = wave; q < nq; q += 16) {
            const int i = fb_rows[q];
            const float* drow; const int32_t* irow; int cnt;
            row_candidates(a, i, nB, &drow, &irow, &cnt);
            Best2 v = {1e6f, 1e6f, -1, -1};
            for (int k = lane; k < cnt; k += 64) {
                const int idx = irow ? irow[k] : k;
                if (a.skipB && a.skipB[idx]) continue;
                if (claim[idx] < (uint32_t)i) continue;
                const float dd = drow[k];
                if (dd < v.b) { v.b2 = v.b; v.b = dd; v.pos = k; v.idx = idx; }
                else if (dd < v.b2) v.b2 = dd;
            }
            v = best2_wave(v);
            if (lane == 0) {
                const int m = (v.idx >= 0 && v.b <= a.th && v.b < v.b2 * a.ratio) ? v.idx : -1;
                changed |= m != guess[i];
                guess[i] = m;
                a.best[i] = v.b;
                a.second[i] = v.b2;
            }
        }
        if (!__syncthreads_count(changed)) break;
    }
    int cnt = 0;
    for (int i = tid; i < nA; i += 1024) {
        const int m = guess[i];
        a.match12[i] = m;
        cnt += m >= 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && cnt) atomicAdd(&tot, cnt);
    __syncthreads();
    if (tid == 0) *a.n_matches = tot;
}

hipError_t launch_match_init(const MatchArgs& a, hipStream_t s) {
    const int rows = a.nA;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(match_topk_kernel, dim3((rows + 3) / 4, a.batch), dim3(256), 0, s, a);
    // the one-wave replay only where the fixed point's tables would not fit LDS
    if (a.nA <= MJ_MAX && a.nB <= MJ_MAX) hipLaunchKernelGGL(match_init_jacobi_kernel, dim3(a.batch), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(match_init_resolve_kernel, dim3(a.batch), dim3(64), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// All-pairs SearchForInitialization (matching/src/Matcher.cpp:591-644) without the distance matrix.
//
// Every distance the reference's loop can ask for is DescriptorDistance = one fused-multiply-add chain over the 256
// components; evaluating it for all pairs is 2 x 256 VALU operations per pair on the issue port the convolutions'
// MFMAs share.  Only the few nearest columns of a row can ever become its best or second-best match, so:
//   row_norm_kernel        |row|^2 of every descriptor and a copy of the row rounded to fp16
//   match_prefilter_kernel |a|^2 + |b|^2 - 2 a.b for all pairs on the fp16 MFMA (fp32 accumulate), kept in registers,
//                          never written: per row a bound T on its 16th smallest value, and the columns within
//                          T + margin are listed.  The margin covers the rounding of the operands to fp16 and of
//                          both evaluations (|approx - chain| <= 2.2e-3 for rows of norm <= 1; the values lie in
//                          [0, 4], so that costs about one more listed column per row), which makes the list a
//                          superset of the 16 columns that are nearest by the exact chain.  In fp16 the pass needs
//                          1/16 of the matrix-core time of an fp32 MFMA; that time is what it takes from the
//                          convolutions running beside it.
//   match_exact_kernel     the exact chain for the listed columns only (about 23 per row), sorted by (distance,
//                          column): up to the radius the list is complete for, the row's nearest-neighbour order
//   desc_transpose_kernel  a column-major copy of B for the rows that have to be scanned whole
//   match_jacobi_kernel    the sequential claim pass as a fixed point: given a guess of every row's match, each row
//                          recomputes its own from the claims of the rows before it; the system is triangular, so
//                          a guess that reproduces itself is the sequential result (rows 0..t-1 are final after
//                          round t; in practice a handful of rounds).  A row whose list runs dry is scanned whole
//                          and keeps the columns up to its second-best free one as an extended list.
// =====================================================================================
constexpr int PF_ROWS = 32, PF_MAXCOLS = 1024, PF_LIST = 64;
constexpr int PF_WAVES = 8, PF_THREADS = 64 * PF_WAVES, PF_Q = PF_MAXCOLS / 32 / PF_WAVES;   // column tiles per wave
constexpr int PF_K = 16;        // T bounds the PF_K-th smallest value of a row
constexpr int PF_NEAR = 32;     // length of the sorted nearest lists handed to the claim pass
// |approx - chain| <= PF_EPS for rows of norm <= 1.  Rounding a and b to fp16 (relative 2^-11 per component, absolute
// 2^-25 below the normal range) moves a.b by at most (2^-10 + 2^-22) |a||b| + 2^-20 <= 9.8e-4, twice that in the
// squared distance; the fp16 products are exact in fp32, the fp32 accumulation, the two norms and the chain's own
// rounding add less than 2e-4.  The listing threshold is T + PF_MARGIN with PF_MARGIN >= 2 * PF_EPS, so every column
// whose exact squared distance is <= T + PF_MARGIN - PF_EPS is listed: up to that radius the sorted list is the
// true nearest-neighbour order with nothing missing.
constexpr float PF_MARGIN = 4.5e-3f, PF_EPS = 2.2e-3f;

__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ X, const int* n_dev, int sn, int maxn, size_t sX,
                                                       float* __restrict__ out, uint16_t* __restrict__ half_out, int n_host) {
    const int b = blockIdx.y;
    const int n = n_dev ? min(maxn, n_dev[b * sn]) : n_host;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const f32x4 v = *(const f32x4*)(X + b * sX + (size_t)row * 256 + lane * 4);
    float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)b * maxn + row] = s;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};   // round to nearest even
    *(f16x4*)(half_out + ((size_t)b * maxn + row) * 256 + lane * 4) = h;
}

// order-preserving map of float bit patterns to unsigned
__device__ inline uint32_t fkey(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct PrefilterArgs {
    const float* A; const float* B;        // [batch][maxn][256]
    size_t sA, sB;
    const int* nA_dev; const int* nB_dev; int sn;   // row counts on the device, or nullptr: nA / nB below
    int nA, nB;
    const float* normA; const float* normB;  // [batch][maxn]
    const uint16_t* halfA; const uint16_t* halfB;   // [batch][maxn][256] the rows in fp16
    int maxn;
    int* list_cnt;                         // [batch][maxn]
    uint16_t* list_col;                    // [batch][maxn][PF_LIST]
    float* list_rad;                       // [batch][maxn] squared radius up to which the list is complete
};

__global__ __launch_bounds__(PF_THREADS) void match_prefilter_kernel(PrefilterArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t As[PF_ROWS][264];   // fp16 rows, padded against bank conflicts
    __shared__ float nas[PF_ROWS];
    __shared__ uint32_t gmin[PF_ROWS][32];
    __shared__ float thr[PF_ROWS];
    __shared__ int lcount[PF_ROWS];
    __shared__ uint16_t llist[PF_ROWS][PF_LIST];
    const int b = blockIdx.y, r0 = blockIdx.x * PF_ROWS;
    const int nA = a.nA_dev ? min(a.maxn, a.nA_dev[b * a.sn]) : a.nA, nB = a.nB_dev ? min(a.maxn, a.nB_dev[b * a.sn]) : a.nB;
    if (r0 >= nA) return;
    const uint16_t* A = a.halfA + (size_t)b * a.maxn * 256;
    const uint16_t* B = a.halfB + (size_t)b * a.maxn * 256;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int idx = tid; idx < PF_ROWS * 32; idx += PF_THREADS) {
        const int r = idx >> 5, c8 = idx & 31;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < nA) v = *(const u32x4*)(A + (size_t)(r0 + r) * 256 + c8 * 8);
        *(u32x4*)&As[r][c8 * 8] = v;
    }
    if (tid < PF_ROWS) {
        nas[tid] = r0 + tid < nA ? a.normA[(size_t)b * a.maxn + r0 + tid] : 0.f;
        lcount[tid] = 0;
    }
    for (int idx = tid; idx < PF_ROWS * 32; idx += PF_THREADS) gmin[idx >> 5][idx & 31] = 0xFFFFFFFFu;
    __syncthreads();
    const int ntiles = (nB + 31) / 32;
    f32x16 acc[PF_Q];
#pragma unroll
    for (int q = 0; q < PF_Q; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    // The order of the 256 products is free here, so lane (li, lh) walks its own half of the row / column: step u
    // (0..15) of v_mfma_f32_32x32x16_f16 takes components 128*lh + 8*u .. +7 from it, sixteen bytes, the half row
    // being two whole 128-byte lines.  One column tile's sixteen loads are in flight while the previous tile is
    // multiplied.
    const uint16_t* brow[PF_Q];
    bool bval[PF_Q];
#pragma unroll
    for (int q = 0; q < PF_Q; q++) {
        const int col = (wave + PF_WAVES * q) * 32 + li;
        bval[q] = col < nB;
        brow[q] = B + (size_t)(bval[q] ? col : 0) * 256 + 128 * lh;   // a column past the end reads row 0; its sums are discarded
    }
    const uint16_t* arow = &As[li][128 * lh];
    u32x4 bb[2][16];
    auto load_tile = [&](int q, u32x4 (&dst)[16]) {
        if (wave + PF_WAVES * q >= ntiles) return;   // uniform per wave
#pragma unroll
        for (int u = 0; u < 16; u++) dst[u] = *(const u32x4*)(brow[q] + 8 * u);
    };
    load_tile(0, bb[0]);
#pragma unroll
    for (int q = 0; q < PF_Q; q++) {
        if (q + 1 < PF_Q) load_tile(q + 1, bb[(q + 1) & 1]);
        if (wave + PF_WAVES * q >= ntiles) continue;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const u32x4 av = *(const u32x4*)(arow + 8 * u);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bb[q & 1][u]), acc[q], 0, 0, 0);
        }
    }
    // approx = |a|^2 + |b|^2 - 2 a.b; lane holds column li of each of its tiles and rows (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
    for (int q = 0; q < PF_Q; q++) {
        if (wave + PF_WAVES * q >= ntiles) continue;
        const int col = (wave + PF_WAVES * q) * 32 + li;
        const float nb = bval[q] ? a.normB[(size_t)b * a.maxn + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            acc[q][r] = bval[q] ? (nas[row] + nb) - 2.f * acc[q][r] : 3.0e38f;
        }
    }
    // T(row) = 8th smallest of the 32 per-lane-column minima: at least 8 different columns lie at or below it
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float m = 3.0e38f;
#pragma unroll
        for (int q = 0; q < PF_Q; q++)
            if (wave + PF_WAVES * q < ntiles) m = fminf(m, acc[q][r]);
        atomicMin(&gmin[(r & 3) + 8 * (r >> 2) + 4 * lh][li], fkey(m));
    }
    __syncthreads();
    for (int rr = 0; rr < PF_ROWS / PF_WAVES; rr++) {
        const int row = wave * (PF_ROWS / PF_WAVES) + rr;
        const uint32_t mine = gmin[row][li];
        int rank = 0;
#pragma unroll
        for (int o = 0; o < 32; o++) {
            const uint32_t other = __shfl(mine, o);
            rank += (other < mine) || (other == mine && o < li);
        }
        if (rank == PF_K - 1 && lh == 0) {
            const float T = fkey_inv(mine);
            thr[row] = T < 1.0e38f ? T + PF_MARGIN : 3.0e38f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PF_Q; q++) {
        if (wave + PF_WAVES * q >= ntiles || !bval[q]) continue;
        const int col = (wave + PF_WAVES * q) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (acc[q][r] <= thr[row]) {
                const int slot = atomicAdd(&lcount[row], 1);
                if (slot < PF_LIST) llist[row][slot] = (uint16_t)col;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < PF_ROWS * PF_LIST; idx += PF_THREADS) {
        const int r = idx / PF_LIST, sidx = idx % PF_LIST;
        if (r0 + r < nA && sidx < min(lcount[r], PF_LIST))
            a.list_col[((size_t)b * a.maxn + r0 + r) * PF_LIST + sidx] = llist[r][sidx];
    }
    if (tid < PF_ROWS && r0 + tid < nA) {
        a.list_cnt[(size_t)b * a.maxn + r0 + tid] = lcount[tid];
        a.list_rad[(size_t)b * a.maxn + r0 + tid] = thr[tid] < 1.0e38f ? thr[tid] - PF_EPS : 3.0e38f;
    }
}

// The chain takes a whole 128-byte line of each operand at a time: a lane that walks its row sixteen bytes per load
// comes back to every line eight times, and with a row per lane the L1 has dropped it by then.
__device__ inline float chain_distance2(const float* __restrict__ x, const float* __restrict__ y) {
    float acc = 0.f;
    for (int d = 0; d < 256; d += 32) {
        f32x4 u[8], v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { u[q] = *(const f32x4*)(x + d + 4 * q); v[q] = *(const f32x4*)(y + d + 4 * q); }
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float t = u[q][e] - v[q][e];
                acc = fmaf(t, t, acc);
            }
    }
    return acc;
}
__device__ inline float chain_distance(const float* __restrict__ x, const float* __restrict__ y) { return sqrtf(chain_distance2(x, y)); }

// The listed columns of one row by W lanes (W = 64: a wave per row; W = 32: two rows per wave, for the usual lists of at
// most 32 columns, so that half of the lanes do not idle through the 256-step chains): gl = lane inside the group.
template <int W>
__device__ inline void exact_listed(const PrefilterArgs& a, int b, int row, int cnt, int nB, int gl, int gshift,
                                    int32_t* near_idx, float* near_dist, int* near_len) {
    const float* x = a.A + b * a.sA + (size_t)row * 256;
    const float* B = a.B + b * a.sB;
    int32_t* oi = near_idx + ((size_t)b * a.maxn + row) * PF_NEAR;
    float* od = near_dist + ((size_t)b * a.maxn + row) * PF_NEAR;
    const float rad = a.list_rad[(size_t)b * a.maxn + row];
    float d = 3e38f;
    int p = 0x7FFFFFFF;
    bool certain = false;
    if (gl < cnt) {
        p = a.list_col[((size_t)b * a.maxn + row) * PF_LIST + gl];
        const float d2 = chain_distance2(x, B + (size_t)p * 256);
        certain = d2 <= rad;
        d = sqrtf(d2);
    }
    // rank of this lane's entry among the listed ones by (distance, column)
    int rank = 0;
    for (int o = 0; o < cnt; o++) {
        const float od2 = __shfl(d, o, W);
        const int op = __shfl(p, o, W);
        rank += od2 < d || (od2 == d && op < p);
    }
    // entries are certain up to the first uncertain one in rank order
    const unsigned long long unc = (__ballot(gl < cnt && !certain) >> gshift) & (W == 64 ? ~0ull : 0xFFFFFFFFull);
    int first_unc = cnt;
    for (int o = 0; o < cnt; o++) {
        const int r = __shfl(rank, o, W);
        if ((unc >> o) & 1ull) first_unc = min(first_unc, r);
    }
    // every column of the row is listed: the list is the whole row, certain to its end
    const int len = cnt >= nB ? cnt : first_unc;
    if (gl < cnt && rank < PF_NEAR) { oi[rank] = p; od[rank] = d; }
    if (gl == 0) near_len[(size_t)b * a.maxn + row] = min(len, PF_NEAR);
}

// Exact distances of the listed columns, sorted by (distance, column): near_idx/near_dist hold the row's nearest
// columns in the order the reference's scan would rank them, near_len how many of them are certain (nothing nearer is
// missing).  A wave takes two rows; a row whose list overflowed is scanned completely (eight certain entries).
__global__ __launch_bounds__(256) void match_exact_kernel(PrefilterArgs a, int32_t* near_idx, float* near_dist, int* near_len) {
    const int b = blockIdx.y;
    const int nA = a.nA_dev ? min(a.maxn, a.nA_dev[b * a.sn]) : a.nA, nB = a.nB_dev ? min(a.maxn, a.nB_dev[b * a.sn]) : a.nB;
    const int lane = threadIdx.x & 63;
    const int row0 = 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row0 >= nA) return;
    const int cnt0 = a.list_cnt[(size_t)b * a.maxn + row0];
    const int cnt1 = row0 + 1 < nA ? a.list_cnt[(size_t)b * a.maxn + row0 + 1] : 0;
    if (cnt0 <= 32 && cnt1 <= 32) {
        const int h = lane >> 5;
        // (both halves run the same code; the shuffles and the ballot slice stay inside a half)
        const int row = row0 + h, cnt = h ? cnt1 : cnt0;
        if (row < nA) exact_listed<32>(a, b, row, cnt, nB, lane & 31, 32 * h, near_idx, near_dist, near_len);
        return;
    }
    for (int rr = 0; rr < 2; rr++) {
        const int row = row0 + rr, cnt = rr ? cnt1 : cnt0;
        if (row >= nA) break;
        if (cnt <= PF_LIST) {
            exact_listed<64>(a, b, row, cnt, nB, lane, 0, near_idx, near_dist, near_len);
            continue;
        }
        const float* x = a.A + b * a.sA + (size_t)row * 256;
        const float* B = a.B + b * a.sB;
        int32_t* oi = near_idx + ((size_t)b * a.maxn + row) * PF_NEAR;
        float* od = near_dist + ((size_t)b * a.maxn + row) * PF_NEAR;
        float ld[TOPK];
        int lp[TOPK];
#pragma unroll
        for (int t = 0; t < TOPK; t++) { ld[t] = 3e38f; lp[t] = 0x7FFFFFFF; }
        for (int col = lane; col < nB; col += 64) {
            float d = chain_distance(x, B + (size_t)col * 256);
            int p = col;
#pragma unroll
            for (int t = 0; t < TOPK; t++)
                if (d < ld[t] || (d == ld[t] && p < lp[t])) {
                    const float td = ld[t]; const int tp = lp[t];
                    ld[t] = d; lp[t] = p; d = td; p = tp;
                }
        }
        int len = 0;
        for (int t = 0; t < TOPK; t++) {
            float d = ld[0];
            int p = lp[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od2 = __shfl_xor(d, o);
                const int op = __shfl_xor(p, o);
                if (od2 < d || (od2 == d && op < p)) { d = od2; p = op; }
            }
            if (lp[0] == p && p != 0x7FFFFFFF) {
#pragma unroll
                for (int q = 0; q < TOPK - 1; q++) { ld[q] = ld[q + 1]; lp[q] = lp[q + 1]; }
                ld[TOPK - 1] = 3e38f; lp[TOPK - 1] = 0x7FFFFFFF;
            }
            if (p != 0x7FFFFFFF) {
                if (lane == 0) { od[t] = d; oi[t] = p; }
                len = t + 1;
            }
        }
        if (lane == 0) near_len[(size_t)b * a.maxn + row] = len;
    }
}

// Bt[b][d][col] = B[b][col][d]: the column-major copy the claim pass scans when a row needs all its distances (a thread
// per column then reads consecutive addresses; from the row-major descriptors every lane walks its own row, 64 cache
// lines per load instruction, and a scan of 1024 columns by one CU takes 35 us instead of 8).  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void desc_transpose_kernel(const float* __restrict__ B, const int* n_dev, int sn, int n_host,
                                                             int maxn, size_t sB, float* __restrict__ Bt) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int n = n_dev ? min(maxn, n_dev[b * sn]) : n_host;
    const int c0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    if (c0 >= n) return;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 16; idx += 256) {
        const int r = idx >> 4, q = idx & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c0 + r < n) v = *(const f32x4*)(B + b * sB + (size_t)(c0 + r) * 256 + d0 + 4 * q);
        tile[r][4 * q] = v[0]; tile[r][4 * q + 1] = v[1]; tile[r][4 * q + 2] = v[2]; tile[r][4 * q + 3] = v[3];
    }
    __syncthreads();
    float* out = Bt + (size_t)b * 256 * maxn;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int d = idx >> 6, r = idx & 63;
        if (c0 + r < maxn) out[(size_t)(d0 + d) * maxn + c0 + r] = tile[r][d];
    }
}

struct JacobiArgs {
    const float* A; const float* B; size_t sA, sB;
    const float* Bt;    // [batch][256][maxn] column-major copy of B
    uint2* ext;         // [batch][JB_EXT_ROWS][JB_EXT_LEN] work: (distance bits, column) lists written by whole-row scans
    const int* nA_dev; const int* nB_dev; int sn;
    int nA, nB;
    int maxn;
    const int32_t* near_idx; const float* near_dist;   // [batch][maxn][PF_NEAR]
    const int* near_len;                               // [batch][maxn]
    float th, ratio;
    int32_t* match12;   // [batch][maxn]
    float* best; float* second;   // [batch][maxn] or nullptr
    int* n_matches;     // [batch]
    int* stats;         // optional: [0] += rounds, [1] += rows that needed a whole-row scan (summed over rounds)
};

constexpr int JB_FB = 8;   // rows of one whole-row scan pass
constexpr int JB_EXT_ROWS = 64, JB_EXT_LEN = 128;   // rows that can keep an extended list after a whole-row scan, entries of one

__global__ __launch_bounds__(1024) void match_jacobi_kernel(JacobiArgs a) {
    __shared__ uint32_t claim[PF_MAXCOLS];     // smallest row whose current guess is this column
    __shared__ int32_t guess[PF_MAXCOLS];
    __shared__ int fb_rows[PF_MAXCOLS];
    __shared__ int fb_n, fb_changed;
    __shared__ float rb[JB_FB][16], rs[JB_FB][16];
    __shared__ int rp[JB_FB][16];
    __shared__ __attribute__((aligned(16))) float xs[JB_FB][256];
    // After a whole-row scan the row keeps every column up to its second-best free one (claimed ones included) as an
    // unsorted list: all columns within that radius, so the two nearest free entries of the list are the row's two nearest
    // free columns whatever the claims become, and the row is scanned again only if the list runs dry in its turn.
    __shared__ int ext_n;                       // rows that own a list
    __shared__ int ext_cnt[JB_EXT_ROWS];        // entries appended to each (may exceed JB_EXT_LEN: then the list is void)
    __shared__ float fb_radius[JB_FB];
    __shared__ int fb_slot[JB_FB];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nA = a.nA_dev ? min(a.maxn, a.nA_dev[b * a.sn]) : a.nA, nB = a.nB_dev ? min(a.maxn, a.nB_dev[b * a.sn]) : a.nB;
    const float* A = a.A + b * a.sA;
    uint2* ext = a.ext + (size_t)b * JB_EXT_ROWS * JB_EXT_LEN;
    int my_ext = -1;                            // this row's list, or -1
    if (tid == 0) ext_n = 0;
    // thread i owns row i (nA <= 1024); its sorted nearest list stays in registers for every round
    const int i = tid;
    float nd[PF_NEAR];
    int ni[PF_NEAR];
    int len = 0;
    if (i < nA) {
        len = a.near_len[(size_t)b * a.maxn + i];
        const f32x4* pd = (const f32x4*)(a.near_dist + ((size_t)b * a.maxn + i) * PF_NEAR);
        const int4* pi = (const int4*)(a.near_idx + ((size_t)b * a.maxn + i) * PF_NEAR);
#pragma unroll
        for (int q = 0; q < PF_NEAR / 4; q++) {
            const f32x4 dv = pd[q];
            const int4 iv = pi[q];
            nd[4 * q] = dv[0]; nd[4 * q + 1] = dv[1]; nd[4 * q + 2] = dv[2]; nd[4 * q + 3] = dv[3];
            ni[4 * q] = iv.x; ni[4 * q + 1] = iv.y; ni[4 * q + 2] = iv.z; ni[4 * q + 3] = iv.w;
        }
        guess[i] = -1;
    }
    int mine = -1;
    float mbest = 1e6f, msecond = 1e6f;
    __syncthreads();
    const long long t_begin = a.stats ? (long long)wall_clock64() : 0;
    long long t_fb = 0;
    int n_round = 0, n_fb = 0;
    for (int round = 0; round <= nA; round++) {
        for (int j = tid; j < nB; j += 1024) claim[j] = 0xFFFFFFFFu;
        if (tid == 0) { fb_n = 0; fb_changed = 0; }
        __syncthreads();
        if (i < nA && guess[i] >= 0) atomicMin(&claim[guess[i]], (uint32_t)i);
        __syncthreads();
        int changed = 0;
        bool fallback = false;
        if (i < nA) {
            float best = 1e6f, second = 1e6f;
            int bidx = -1, nfree = 0;
            bool whole = len >= nB;   // the list at hand covers the whole row
            if (my_ext >= 0) {
                const int cnt = ext_cnt[my_ext];
                const uint2* e = ext + (size_t)my_ext * JB_EXT_LEN;
                whole = cnt >= nB;
                for (int t = 0; t < cnt; t++) {   // unsorted: keep the two smallest free entries by (distance, column)
                    const uint2 v = e[t];
                    const int col = (int)v.y;
                    if (claim[col] < (uint32_t)i) continue;
                    const float d = __uint_as_float(v.x);
                    if (d < best || (d == best && col < bidx)) { second = best; best = d; bidx = col; }
                    else if (d < second) second = d;
                    nfree++;
                }
            } else {
#pragma unroll
                for (int t = 0; t < PF_NEAR; t++) {
                    if (t >= len || nfree >= 2) continue;
                    if (claim[ni[t]] < (uint32_t)i) continue;   // claimed by an earlier row
                    if (nfree == 0) { best = nd[t]; bidx = ni[t]; }
                    else second = nd[t];
                    nfree++;
                }
            }
            if (nfree < 2 && !whole) {
                // the certain part of the list ran dry while the row has more columns: scan the whole row
                fallback = true;
                fb_rows[atomicAdd(&fb_n, 1)] = i;
            } else {
                const int m = (bidx >= 0 && best <= a.th && best < second * a.ratio) ? bidx : -1;
                changed = m != mine;
                mine = m; mbest = best; msecond = second;
            }
        }
        __syncthreads();
        const int nfb = fb_n;
        if (a.stats && tid == 0) { atomicAdd(&a.stats[0], 1); atomicAdd(&a.stats[1], nfb); n_round++; n_fb += nfb; }
        const long long t_f0 = a.stats ? (long long)wall_clock64() : 0;
        // rare; these rows all read the same claims, so their order is immaterial.  Up to JB_FB of them share one
        // pass over the columns: a thread reads its column once and runs one chain per row against the rows in LDS.
        for (int f0 = 0; f0 < nfb; f0 += JB_FB) {
            const int nb = min(JB_FB, nfb - f0);
            for (int idx = tid; idx < nb * 64; idx += 1024) {
                const int rr = idx >> 6, c4 = (idx & 63) * 4;
                *(f32x4*)&xs[rr][c4] = *(const f32x4*)(A + (size_t)fb_rows[f0 + rr] * 256 + c4);
            }
            __syncthreads();
            float acc[JB_FB];
#pragma unroll
            for (int rr = 0; rr < JB_FB; rr++) acc[rr] = 0.f;
            const int col = tid;
            if (col < nB) {
                const float* y = a.Bt + (size_t)b * 256 * a.maxn + col;   // component d of this column at y[d * maxn]
                for (int d = 0; d < 256; d += 16) {
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = y[(size_t)(d + q) * a.maxn];
#pragma unroll
                    for (int rr = 0; rr < JB_FB; rr++) {
                        if (rr >= nb) continue;   // uniform
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const f32x4 u = *(const f32x4*)&xs[rr][d + 4 * q];
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float t = u[e] - v[4 * q + e];
                                acc[rr] = fmaf(t, t, acc[rr]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < JB_FB; rr++) {
                if (rr >= nb) continue;
                const int r = fb_rows[f0 + rr];
                Best2 v = {1e6f, 1e6f, -1, -1};
                if (col < nB && !(claim[col] < (uint32_t)r)) { v.b = sqrtf(acc[rr]); v.pos = col; v.idx = col; }
                v = best2_wave(v);
                if (lane == 0) { rb[rr][wave] = v.b; rs[rr][wave] = v.b2; rp[rr][wave] = v.pos; }
            }
            __syncthreads();
            if (fallback) {
                for (int rr = 0; rr < nb; rr++) {
                    if (fb_rows[f0 + rr] != i) continue;
                    Best2 q = {rb[rr][0], rs[rr][0], rp[rr][0], rp[rr][0]};
                    for (int w = 1; w < 16; w++) q = best2_merge(q, Best2{rb[rr][w], rs[rr][w], rp[rr][w], rp[rr][w]});
                    const int m = (q.pos >= 0 && q.b <= a.th && q.b < q.b2 * a.ratio) ? q.pos : -1;
                    if (m != mine) fb_changed = 1;
                    mine = m; mbest = q.b; msecond = q.b2;
                    // the row's list from now on: every column up to the second-best free one (none if there is no such
                    // column, if no list is left to hand out, or if the list would not fit)
                    if (my_ext < 0 && q.b2 < 1e6f) {
                        const int sl = atomicAdd(&ext_n, 1);
                        if (sl < JB_EXT_ROWS) my_ext = sl;
                    }
                    if (my_ext >= 0) ext_cnt[my_ext] = 0;
                    fb_slot[rr] = q.b2 < 1e6f ? my_ext : -1;
                    fb_radius[rr] = q.b2;
                }
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < JB_FB; rr++) {
                if (rr >= nb) continue;
                const int sl = fb_slot[rr];
                if (sl < 0 || col >= nB) continue;
                const float d = sqrtf(acc[rr]);
                if (d <= fb_radius[rr]) {
                    const int o = atomicAdd(&ext_cnt[sl], 1);
                    if (o < JB_EXT_LEN) ext[(size_t)sl * JB_EXT_LEN + o] = make_uint2(__float_as_uint(d), (uint32_t)col);
                }
            }
            __syncthreads();
            if (fallback && my_ext >= 0 && ext_cnt[my_ext] > JB_EXT_LEN) {
                // does not fit: the row goes back to whole-row scans (its slot stays taken; this does not happen with
                // real descriptors, where the second-best free column is a few dozen columns out)
                my_ext = -1;
            }
            __syncthreads();
        }
        (void)fallback;
        if (a.stats) t_fb += (long long)wall_clock64() - t_f0;
        if (i < nA) guess[i] = mine;
        const int fbc = fb_changed;   // read before the barrier: thread 0 clears it at the top of the next round
        if (__syncthreads_count(changed | fbc) == 0) break;
    }
    if (a.stats && tid == 0) {   // the slowest problem of the batch: [2] its time, [3] of which rescans, [4] rounds, [5] rescans
        const int t_all = (int)((long long)wall_clock64() - t_begin);
        if (atomicMax(&a.stats[2], t_all) < t_all) { a.stats[3] = (int)t_fb; a.stats[4] = n_round; a.stats[5] = n_fb; }
    }
    if (i < nA) {
        a.match12[(size_t)b * a.maxn + i] = mine;
        if (a.best) { a.best[(size_t)b * a.maxn + i] = mbest; a.second[(size_t)b * a.maxn + i] = msecond; }
    }
    int cnt = (i < nA && mine >= 0) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    if (lane == 0) rp[0][wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) tot += rp[0][w];
        a.n_matches[b] = tot;
    }
}

hipError_t launch_match_fused(const FusedMatchArgs& f, hipStream_t s) {
    if (f.maxn > PF_MAXCOLS || f.batch < 1) return hipErrorInvalidValue;
    PrefilterArgs p;
    p.A = f.A; p.B = f.B; p.sA = f.sA; p.sB = f.sB; p.nA_dev = f.nA_dev; p.nB_dev = f.nB_dev; p.sn = f.sn;
    p.nA = f.nA; p.nB = f.nB;
    if (!f.nA_dev && (f.nA > f.maxn || f.nB > f.maxn || f.nA < 0 || f.nB < 0)) return hipErrorInvalidValue;
    p.normA = f.norm_work; p.normB = f.norm_work + (size_t)f.batch * f.maxn; p.maxn = f.maxn;
    p.halfA = f.half_work; p.halfB = f.half_work + (size_t)f.batch * f.maxn * 256;
    p.list_cnt = f.list_cnt; p.list_col = f.list_col; p.list_rad = f.list_rad;
    dim3 gn((f.maxn + 3) / 4, f.batch);
    hipLaunchKernelGGL(row_norm_kernel, gn, dim3(256), 0, s, f.A, f.nA_dev, f.sn, f.maxn, f.sA, f.norm_work, f.half_work, f.nA);
    hipLaunchKernelGGL(row_norm_kernel, gn, dim3(256), 0, s, f.B, f.nB_dev, f.sn, f.maxn, f.sB, f.norm_work + (size_t)f.batch * f.maxn,
                       f.half_work + (size_t)f.batch * f.maxn * 256, f.nB);
    if (!(f.debug_skip & 4)) hipLaunchKernelGGL(match_prefilter_kernel, dim3((f.maxn + PF_ROWS - 1) / PF_ROWS, f.batch), dim3(PF_THREADS), 0, s, p);
    if (!(f.debug_skip & 2)) hipLaunchKernelGGL(match_exact_kernel, dim3((f.maxn + 7) / 8, f.batch), dim3(256), 0, s, p, f.near_idx, f.near_dist, f.near_len);
    if (f.debug_skip & 1) return hipGetLastError();
    hipLaunchKernelGGL(desc_transpose_kernel, dim3((f.maxn + 63) / 64, 4, f.batch), dim3(256), 0, s, f.B, f.nB_dev, f.sn, f.nB, f.maxn, f.sB, f.bt_work);
    JacobiArgs j;
    j.A = f.A; j.B = f.B; j.sA = f.sA; j.sB = f.sB; j.Bt = f.bt_work; j.ext = (uint2*)f.ext_work; j.nA_dev = f.nA_dev; j.nB_dev = f.nB_dev; j.sn = f.sn; j.maxn = f.maxn;
    j.nA = f.nA; j.nB = f.nB;
    j.near_idx = f.near_idx; j.near_dist = f.near_dist; j.near_len = f.near_len; j.th = f.th; j.ratio = f.ratio; j.match12 = f.match12;
    j.best = f.best; j.second = f.second; j.n_matches = f.n_matches; j.stats = f.stats;
    hipLaunchKernelGGL(match_jacobi_kernel, dim3(f.batch), dim3(1024), 0, s, j);
    return hipGetLastError();
}

// =====================================================================================
// DBoW3::Vocabulary::transform(feature, word_id, weight, nid, levelsup) (called for every descriptor of a frame at
// map/src/Frame.cpp:131 and :338): descend the k-ary tree from the root, at every level to the child whose node
// descriptor is nearest (squared L2: float differences, float squares, summed in double in component order; the first
// minimum in child order wins), until a leaf; report the leaf's word and weight and the node passed at level
// L - levelsup (the root when that is <= 0).  Sixteen lanes per descriptor, a lane per child.
// =====================================================================================
__global__ __launch_bounds__(256) void bow_transform_kernel(BowArgs a) {
    const int b = blockIdx.y;
    const int n = a.n_dev ? min(a.n, a.n_dev[b * a.sn]) : a.n;
    const int f = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
    if (f >= n) return;   // whole 16-lane groups leave together
    const float* x = a.desc + b * a.stride_batch + (size_t)f * a.dim;
    int cur = 0, level = 0, nid = 0;
    for (;;) {
        level++;
        const int c0 = a.child_off[cur], nc = a.child_off[cur + 1] - c0;
        double best = 1.7976931348623157e308;
        int bpos = 0x7FFFFFFF;
        for (int t = gl; t < nc; t += 16) {
            const float* y = a.node_desc + (size_t)a.child_idx[c0 + t] * a.dim;
            double sqd = 0.0;
            for (int d = 0; d < a.dim; d++) {
                const float df = x[d] - y[d];
                sqd += (double)(df * df);
            }
            if (sqd < best) { best = sqd; bpos = t; }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int op = __shfl_xor(bpos, o);
            if (ob < best || (ob == best && op < bpos)) { best = ob; bpos = op; }
        }
        // No child compares nearer only for a descriptor with NaN components: DBoW3's loop would then stay at this node
        // for ever; the first child is taken instead, so that the descent ends and no index is out of range.
        if (bpos == 0x7FFFFFFF) bpos = 0;
        cur = a.child_idx[c0 + bpos];
        if (level == a.nid_level) nid = cur;
        if (a.child_off[cur + 1] == a.child_off[cur]) break;   // leaf
    }
    if (gl == 0) {
        const size_t o = (size_t)b * a.n + f;
        a.out_word[o] = a.word_id[cur];
        a.out_weight[o] = a.weight[cur];
        a.out_node[o] = nid;
    }
}

hipError_t launch_bow_transform(const BowArgs& a, int batch, hipStream_t s) {
    if (a.n <= 0 || batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(bow_transform_kernel, dim3((a.n * 16 + 255) / 256, batch), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// The inner loop shared by Matcher::SearchByProjection (matching/src/Matcher.cpp:31-87, :89-189, :479-568) and
// Matcher::Fuse (:897-1036, :1038-1135) with its order dependence: rows are visited in order, a row skips blocked
// slots, takes the first argmin of its candidates, accepts by threshold (and ratio), and an accepted row that claims
// blocks its slot for the rows after it.  As for the initialisation matcher, the sequential result is the fixed point
// of "every row recomputes its match from the claims of the rows before it" (triangular system): a few parallel
// rounds instead of one sequential pass.  One work-group; rows and candidates are strided over its threads.
// =====================================================================================
__global__ __launch_bounds__(1024) void match_projected_kernel(ProjMatchArgs a) {
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n1; i += 1024) a.guess[i] = -1;
    __syncthreads();
    for (int round = 0; round <= a.n1; round++) {
        for (int j = tid; j < a.n2; j += 1024) a.claim[j] = 0xFFFFFFFFu;
        __syncthreads();
        for (int i = tid; i < a.n1; i += 1024)
            if (a.guess[i] >= 0 && a.row_claims && a.row_claims[i]) atomicMin(&a.claim[a.guess[i]], (uint32_t)i);
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < a.n1; i += 1024) {
            float best = 1e6f, second = 1e6f;
            int bi = -1;
            for (int k = a.cand_off[i]; k < a.cand_off[i + 1]; k++) {
                const int j = a.cand_idx[k];
                if ((a.blocked_in && a.blocked_in[j]) || a.claim[j] < (uint32_t)i) continue;
                const float d = a.D[k];
                if (d < best) { second = best; best = d; bi = j; }
                else if (d < second) second = d;
            }
            int m = -1;
            if (best <= a.th && !(a.ratio > 0.f && best > a.ratio * second)) m = bi;
            a.best_dist[i] = best;
            changed |= m != a.guess[i];
            a.best_idx[i] = m;
        }
        const int any = __syncthreads_count(changed);
        for (int i = tid; i < a.n1; i += 1024) a.guess[i] = a.best_idx[i];
        __syncthreads();
        if (!any) break;
    }
    // slots blocked afterwards: the given ones and every claimed one (claim[] holds the final claims: it was rebuilt from
    // the final guesses at the top of the last round)
    int cnt = 0;
    for (int i = tid; i < a.n1; i += 1024) cnt += a.guess[i] >= 0;
    if (a.blocked_out)
        for (int j = tid; j < a.n2; j += 1024) a.blocked_out[j] = (uint8_t)((a.blocked_in && a.blocked_in[j]) || a.claim[j] != 0xFFFFFFFFu);
    __shared__ int tot;
    if (tid == 0) tot = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) atomicAdd(&tot, cnt);
    __syncthreads();
    if (tid == 0) *a.n_matches = tot;
}

hipError_t launch_match_projected(const ProjMatchArgs& a, hipStream_t s) {
    if (a.n1 <= 0) return hipSuccess;
    hipLaunchKernelGGL(match_projected_kernel, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// Matcher::ExtendMapMatches (matching/src/Matcher.cpp:203-381), the point-pair-graph matcher, by ONE wave.
//
// The loop is sequential by nature: every accepted map point changes F.mvpMapPoints, which the skip rule of the next
// window search reads (:253-255), and the seed growing (:287-377) walks the two graphs breadth first, assigning as it
// goes.  What bounds such a walk is the chain of dependent memory accesses, so nothing the walk decides on is fetched
// from global memory inside it: the flags of the map points (tracked / bad / observed), F.mvpMapPoints, both adjacency
// structures (for every listed edge the point at its other end) and the seed queue live in LDS; the per-candidate
// scalars (map point, window range, edge range) are fetched 64 candidates at a time, a lane each, and read back by
// lane index; a candidate's window (indices and precomputed distances) is fetched while the candidate before it is
// processed.  What remains is one global access per popped seed: the weights of its (map edges) x (key edges)
// matrix, a lane per pair, looked up in the distance matrix.  Minima over the wave go through DPP row shifts.
// =====================================================================================
constexpr int EX_MAXM = 8192, EX_MAXN = 4096, EX_MAXDEG = 64, EX_MAXW = 1024, EX_MAXCONN = 16384, EX_MAXME = 16384;

// minimum over the 64 lanes (result uniform): inclusive scan inside each row of 16 lanes by row shifts, then the row
// totals carried over by the two row broadcasts; lane 63 ends up with the minimum of the wave
__device__ inline uint32_t wave_min_u32(uint32_t v) {
#define PPG_DPP_MIN(ctrl, rowmask) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rowmask, 0xf, false))
    PPG_DPP_MIN(0x111, 0xf);   // row_shr:1
    PPG_DPP_MIN(0x112, 0xf);   // row_shr:2
    PPG_DPP_MIN(0x114, 0xf);   // row_shr:4
    PPG_DPP_MIN(0x118, 0xf);   // row_shr:8
    PPG_DPP_MIN(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    PPG_DPP_MIN(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef PPG_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// sixteen bytes per lane from a padded global array into a narrower LDS array
template <class D>
__device__ inline void ex_fill_i32(D* dst, const int32_t* __restrict__ src, int count, int lane) {
#pragma unroll 4
    for (int k = lane * 4; k < count; k += 256) {
        const int4 v = *reinterpret_cast<const int4*>(src + k);
        dst[k] = (D)v.x;
        if (k + 1 < count) dst[k + 1] = (D)v.y;
        if (k + 2 < count) dst[k + 2] = (D)v.z;
        if (k + 3 < count) dst[k + 3] = (D)v.w;
    }
}

__global__ __launch_bounds__(64) void extend_map_matches_kernel(ExtendArgs a) {
    __shared__ uint8_t flags[EX_MAXM];            // bit 0 tracked, bit 1 bad, bit 2 observed
    __shared__ int16_t fmp[EX_MAXN];              // F.mvpMapPoints as map point ids (m <= 8192), -1 = none
    __shared__ uint16_t queue[EX_MAXM + 1];       // every push marks another map point tracked: at most m pushes per walk
    __shared__ float W[EX_MAXW];
    __shared__ int32_t coff[EX_MAXN + 1];
    __shared__ uint16_t cother[EX_MAXCONN];
    __shared__ int16_t eother[EX_MAXME];
    __shared__ int16_t rowmp[EX_MAXDEG];
    __shared__ uint8_t lxs[EX_MAXDEG];
    const int lane = threadIdx.x;
    const float TH_HIGH = 0.8f;   // matching/src/Matcher.cpp:12
    if (a.n_conn > EX_MAXCONN || a.n_map_edges > EX_MAXME) {   // (uniform)
        if (lane == 0) { a.out[0] = 0; a.out[1] = 1; }
        return;
    }
#pragma unroll 2
    for (int k = lane * 4; k < a.m; k += 256) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(a.flags + k);
        flags[k] = v.x;
        if (k + 1 < a.m) flags[k + 1] = v.y;
        if (k + 2 < a.m) flags[k + 2] = v.z;
        if (k + 3 < a.m) flags[k + 3] = v.w;
    }
    ex_fill_i32(fmp, a.frame_mp, a.n, lane);
    ex_fill_i32(coff, a.conn_off, a.n + 1, lane);
    ex_fill_i32(cother, a.conn_other, a.n_conn, lane);
    ex_fill_i32(eother, a.edge_usable, a.n_map_edges, lane);
    __syncthreads();
    int nmatches = 0;
    bool overflow = false;
    for (int c0 = 0; c0 < a.ncand && !overflow; c0 += 64) {
        // the scalars of 64 candidates, a lane each
        const int cl = min(c0 + lane, a.ncand - 1);
        const int my_mp = a.cands[cl];
        const int my_w0 = a.win_off[cl], my_w1 = a.win_off[cl + 1];
        const int my_e0 = a.edge_off[my_mp], my_e1 = a.edge_off[my_mp + 1];
        const int nchunk = min(64, a.ncand - c0);
        int nx_id = 0;
        float nx_d = 0.f;
        {
            const int w0 = __builtin_amdgcn_readlane(my_w0, 0), w1 = __builtin_amdgcn_readlane(my_w1, 0);
            if (lane < w1 - w0) { nx_id = a.win_idx[w0 + lane]; nx_d = a.win_dist[w0 + lane]; }
        }
        for (int ci = 0; ci < nchunk && !overflow; ci++) {
            const int mp = __builtin_amdgcn_readlane(my_mp, ci);
            const int w0 = __builtin_amdgcn_readlane(my_w0, ci), cnt = __builtin_amdgcn_readlane(my_w1, ci) - w0;
            const int cur_id = nx_id;
            const float cur_d = nx_d;
            if (ci + 1 < nchunk) {   // the next candidate's window is on its way while this one is processed
                const int nw0 = __builtin_amdgcn_readlane(my_w0, ci + 1), nw1 = __builtin_amdgcn_readlane(my_w1, ci + 1);
                if (lane < nw1 - nw0) { nx_id = a.win_idx[nw0 + lane]; nx_d = a.win_dist[nw0 + lane]; }
            }
            if (flags[mp] & 3) continue;                                              // :227
            if (cnt == 0) continue;
            // :251-270 — best and second best of the window; the first of equal minima wins (window order)
            float b = 1e6f, b2 = 1e6f;
            int pos = -1, idx = -1;
            auto consider = [&](int id, float d, int k) {
                const int cur = fmp[id];
                if (cur >= 0 && (flags[cur] & 4)) return;
                if (d < b) { b2 = b; b = d; pos = k; idx = id; }
                else if (d < b2) b2 = d;
            };
            if (lane < cnt) consider(cur_id, cur_d, lane);
            for (int k = lane + 64; k < cnt; k += 64) consider(a.win_idx[w0 + k], a.win_dist[w0 + k], k);
            // distances are non-negative: their bit patterns order like the values
            const uint32_t gb = wave_min_u32(__float_as_uint(b));
            const uint32_t gpos = wave_min_u32(__float_as_uint(b) == gb ? (uint32_t)pos : 0xFFFFFFFFu);
            const bool winner = __float_as_uint(b) == gb && (uint32_t)pos == gpos && pos >= 0;
            const float best = __uint_as_float(gb);
            const float best2 = __uint_as_float(wave_min_u32(__float_as_uint(winner ? b2 : b)));
            if (best > TH_HIGH && best > a.nnratio * best2) continue;                 // :276
            const unsigned long long wm = __ballot(winner);
            if (wm == 0) continue;   // every candidate of the window was skipped (the reference would index with -1, :279)
            const int bestIdx = __builtin_amdgcn_readlane(idx, __builtin_ctzll(wm));
            const int me0 = __builtin_amdgcn_readlane(my_e0, ci), nme = __builtin_amdgcn_readlane(my_e1, ci) - me0;
            if (lane == 0) { fmp[bestIdx] = (int16_t)mp; flags[mp] |= 1; queue[0] = (uint16_t)bestIdx; }
            nmatches++;
            // lx: the usable map edges of this point, in order (the map side stays the first seed's point, :296)
            int nlx = 0;
            if (nme > 0 && nme <= EX_MAXDEG) {
                const int er = lane < nme ? (int)eother[me0 + lane] : -1;
                const unsigned long long mrow = __ballot(er >= 0);
                nlx = __popcll(mrow);
                if (er >= 0) {
                    const int p = __popcll(mrow & ((1ull << lane) - 1ull));
                    lxs[p] = (uint8_t)lane;
                    rowmp[p] = (int16_t)er;
                }
            }
            __syncthreads();
            int qh = 0, qt = 1;
            while (qh < qt) {
                const int keyID = queue[qh];
                qh++;
                const int ke0 = coff[keyID], nke = coff[keyID + 1] - ke0;
                if (nme == 0 || nke == 0) continue;
                if (nme > EX_MAXDEG || nke > EX_MAXDEG) { overflow = true; break; }
                const int P = nlx * nke;
                if (P > EX_MAXW) { overflow = true; break; }
                // :323-339 — the weight of every (map edge, key edge) pair, a lane per pair; the lane's first pair stays
                // in registers (most matrices have at most 64 entries)
                int ip0 = 0, jp0 = 0;
                float w0v = 1e6f;
                for (int p = lane; p < P; p += 64) {
                    const int ip = p / nke, jp = p - ip * nke;
                    const int mpo = rowmp[ip];
                    const int ko = cother[ke0 + jp];
                    const float w = mpo == fmp[ko] ? -1.f
                                                   : (a.Dfull ? a.Dfull[(size_t)mpo * a.n + ko]
                                                              : chain_distance(a.mdesc + (size_t)mpo * 256, a.fdesc + (size_t)ko * 256));
                    if (p == lane) { ip0 = ip; jp0 = jp; w0v = w; }
                    else W[p] = w;
                }
                __syncthreads();
                unsigned long long rused = 0, cused = 0;   // (uniform)
                for (;;) {                                                             // :341-376
                    // smallest weight among the unused rows and columns, the first in (row, column) order of equal ones;
                    // weights are -1 or non-negative: key 0 for -1, bit pattern + 1 otherwise
                    uint32_t bk = 0xFFFFFFFFu;
                    int bp = 0x7FFFFFFF;
                    if (lane < P && !((rused >> ip0) & 1) && !((cused >> jp0) & 1) && w0v < 1e6f) {
                        bk = w0v < 0.f ? 0u : __float_as_uint(w0v) + 1u;
                        bp = lane;
                    }
                    for (int p = lane + 64; p < P; p += 64) {
                        const int ip = p / nke, jp = p - ip * nke;
                        if (((rused >> ip) & 1) || ((cused >> jp) & 1)) continue;
                        const float w = W[p];
                        if (!(w < 1e6f)) continue;
                        const uint32_t k = w < 0.f ? 0u : __float_as_uint(w) + 1u;
                        if (k < bk) { bk = k; bp = p; }
                    }
                    const uint32_t gk = wave_min_u32(bk);
                    if (gk == 0xFFFFFFFFu) break;
                    const float bw = gk == 0u ? -1.f : __uint_as_float(gk - 1u);
                    if (bw > TH_HIGH) break;
                    const int gp = (int)wave_min_u32(bk == gk ? (uint32_t)bp : 0xFFFFFFFFu);
                    const int ip = gp / nke, jp = gp - ip * nke;
                    const int mpo = rowmp[ip];
                    const int ko = cother[ke0 + jp];
                    const bool take = !(flags[mpo] & 3);   // (the other end exists: only usable edges are rows)
                    rused |= 1ull << ip;
                    cused |= 1ull << jp;
                    __syncthreads();
                    if (take) {
                        if (lane == 0) {
                            fmp[ko] = (int16_t)mpo;
                            flags[mpo] |= 1;
                            queue[qt] = (uint16_t)ko;
                            if (a.frame_me) a.frame_me[a.conn_idx[ke0 + jp]] = a.edge_id[me0 + lxs[ip]];
                        }
                        qt++;
                    }
                    __syncthreads();
                }
            }
            nmatches++;   // counted twice per accepted seed (:281, :378)
        }
    }
    __syncthreads();
    for (int k = lane; k < a.m; k += 64) a.tracked[k] = flags[k] & 1;
    for (int k = lane; k < a.n; k += 64) a.frame_mp[k] = fmp[k];
    if (lane == 0) { a.out[0] = nmatches; a.out[1] = overflow ? 1 : 0; }
}

hipError_t launch_extend_map_matches(const ExtendArgs& a, hipStream_t s) {
    if (a.m > EX_MAXM || a.n > EX_MAXN) return hipErrorInvalidValue;
    hipLaunchKernelGGL(extend_map_matches_kernel, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

// distances of CSR candidate lists whose rows are picked from A by an index: Dc[off[r] + k] = |A[rows[r]] - B[idx[off[r]+k]]|
__global__ __launch_bounds__(256) void dist_csr_rows_kernel(const float* __restrict__ A, const int32_t* __restrict__ rows, int nrows,
                                                            const float* __restrict__ B, const int32_t* __restrict__ cand_off,
                                                            const int32_t* __restrict__ cand_idx, float* __restrict__ Dc) {
    const int r = blockIdx.x;
    if (r >= nrows) return;
    const float* x = A + (size_t)rows[r] * 256;
    for (int k = cand_off[r] + threadIdx.x; k < cand_off[r + 1]; k += 256) Dc[k] = chain_distance(x, B + (size_t)cand_idx[k] * 256);
}

hipError_t launch_dist_csr_rows(const float* A, const int32_t* rows, int nrows, const float* B, const int32_t* cand_off,
                                const int32_t* cand_idx, float* Dc, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(dist_csr_rows_kernel, dim3(nrows), dim3(256), 0, s, A, rows, nrows, B, cand_off, cand_idx, Dc);
    return hipGetLastError();
}

}  // namespace ppg
