// Wave-specialised form of the reduced-precision 3x3 convolution (the arithmetic of conv_bf16_kernel in ppg_kernels.hip:
// operands split into bf16 / fp16 hi + lo, three products per pair, fp32 accumulate; same tile, same operand layouts,
// same order of the additions, hence the same bits) for launches of many tiles.
//
// Reference op: the `torch._convolution` + ReLU / max_pool2d / batch_norm + pixel_shuffle groups of net/*.pt
// (net/Backbone.pt code/__torch__/model/nets/backbone.py:28-37, net/EdgeHeatmap.pt .../heatmap_decoder.py:19-22).
//
// One work-group of 512 threads per CU, resident for the whole launch:
//   waves 0-3 ("consumers", one per SIMD) read operand fragments from LDS and issue MFMAs, nothing else inside a tile;
//   waves 4-7 ("producers", one per SIMD) bring the NEXT 16-channel chunk in: the packed weights by LDS-DMA (they lie in
//             memory in LDS image order), the input tile through registers (it has to be split into hi/lo on the way;
//             FUSE1A: it is conv1a of the image patch, evaluated here), and they draw the tiles from per-XCD counters.
// Two LDS stages; ONE barrier per chunk, which every wave reaches: consumers compute on stage g & 1 while producers fill
// stage (g + 1) & 1.  The chunk stream runs on across tile boundaries, so a tile's prologue (index set-up, first fetch,
// image patch) is hidden behind the MFMAs of the tile before it; only the epilogue (bias / batch norm / ReLU / pooling
// and the stores, with the vectors already in LDS) interrupts the consumers.
#include "ppg_kernels.h"
#include "conv_split.h"
#include "conv_store.h"

#include <algorithm>

namespace ppg {

typedef __attribute__((address_space(3))) void* ws_lds_ptr;

// Every wave of the work-group, once per chunk.  A raw s_barrier (no vmcnt(0): the producers' register loads for the
// chunk after next stay in flight across it); the asm's memory clobber keeps LDS accesses on their side of it.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef PPG_WS_TIMING
// microbenchmark builds only (tools/ubench/conv_bench.hip): shader cycles summed over work-groups:
// [0] consumer wave 0 in total, [1] of that waiting at the chunk barriers, [2] in epilogues, [3] producer wave 4 in total,
// [4] of that waiting at the barriers, [5] work-groups, [6] tiles
__device__ unsigned long long* g_ws_timing = nullptr;
#define WS_T0(var) const long long var = clock64()
#define WS_ADD(acc, t0) acc += clock64() - (t0)
#else
#define WS_T0(var)
#define WS_ADD(acc, t0)
#endif

#ifndef WS_FAST_EPILOGUE
#define WS_FAST_EPILOGUE 1   // 0: every tile through conv_store (measurement builds)
#endif

template <bool FUSE1A, int MW, int TERMS, bool F16>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(ConvArgs a) {
    constexpr int KC = 16, TAPS = 9, NT = 2, NB = 64;
    constexpr int TW = MW, TH = 256 / MW;
    constexpr int IH = TH + 2, IW = TW + 2, IN_PIX = IH * IW;
    constexpr int PSTR = 80;                      // bytes per staged pixel: [half 0: 8 hi | 8 lo][half 1: 8 hi | 8 lo] + 16
    constexpr int IN_ITEMS = IN_PIX * 2;          // item = 8 channels of a pixel
    constexpr int IN_PER_T = (IN_ITEMS + 255) / 256;
    constexpr int IN_BYTES_P = (IN_PER_T * 256 / 2) * PSTR;
    constexpr int W_BYTES = TAPS * 2 * 2 * NB * 16;   // [tap][half][hi | lo][cout][8]: 36 pieces of 1 KiB
    constexpr int W_PIECES = W_BYTES / 1024;
    constexpr int STAGE = IN_BYTES_P + W_BYTES;
    constexpr int PH = IH + 2, PW = IW + 2;
    constexpr int PATCH_FLOATS = FUSE1A ? PH * PW : 0;
    constexpr int O_PATCH = 2 * STAGE;                                  // [2][PH][PW] floats (two tiles in flight)
    constexpr int O_W1S = O_PATCH + 2 * PATCH_FLOATS * 4;               // [9][64] conv1a weights
    constexpr int O_B1S = O_W1S + (FUSE1A ? 9 * 64 * 4 : 0);            // [64]
    constexpr int O_EPI = O_B1S + (FUSE1A ? 64 * 4 : 0);                // [2][3][64]: bias, alpha, beta of the tile's cout block
    constexpr int O_RING = O_EPI + 2 * 3 * 64 * 4;                      // [4][16] records of the tiles in flight (see TileRec)
    constexpr int TOTAL = O_RING + 4 * 64;
    static_assert(W_PIECES % 4 == 0, "the weight pieces are dealt evenly to the four producer waves");
    static_assert(TOTAL <= 160 * 1024, "one work-group per CU");
    // (ONE shared object: a second one beside an LDS-DMA destination makes the compiler wait for every load in flight
    // before each LDS read)
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];
    typedef __attribute__((address_space(3))) volatile int ws_lds_vint;   // (an LDS pointer type: a generic volatile pointer would be read by flat loads)
    ws_lds_vint* ring = (ws_lds_vint*)(smem + O_RING);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7;
#ifdef PPG_WS_TIMING
    long long tm_total = 0, tm_bar = 0, tm_epi = 0, tm_tiles = 0;
    const long long tm_start = clock64();
#endif
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    const int nchunks = a.Cin / KC;               // >= 4 (checked by the launch function)
    const int upx = a.xcd_tiles8 * a.nblk;        // units (tile, cout block) of one XCD's band, padding included
    int* ctr = a.tile_counter;                    // [0..7] per-XCD hand-out counters, [8] finished work-groups; all zero at launch

    struct Tile { int k, b, x0, y0, nb; };
    // What the lane that draws a tile leaves in LDS for everybody else (ints): the tile's coordinates and everything its
    // epilogue needs, so that neither the integer divisions of the decoding nor the look-up of the layer's group in the
    // kernel arguments sit on the consumers' path between two tiles.
    enum { R_K = 0, R_B, R_X0, R_Y0, R_NB, R_VARIANT, R_CB, R_COUT, R_RELU, R_CPS_SHIFT, R_OUT_LO, R_OUT_HI, R_FRAME_BYTES };
    enum { EPI_NONE = 0, EPI_POOL = 1, EPI_PLAIN = 2, EPI_PS2BN = 3 };
    auto decode = [&](int k, int& bx, int& nb) -> bool {   // the work-group numbering of ConvArgs::xcd_group with L = 8 k + xcd
        const int r = k / a.xcd_group;
        nb = (r / a.xcd_tiles8) * a.xcd_group + k % a.xcd_group;
        bx = a.xcd_band ? xcd * a.xcd_tiles8 + r % a.xcd_tiles8 : (r % a.xcd_tiles8) * 8 + xcd;
        return bx < ntiles;
    };
    auto tile_of = [&](int k) -> Tile {
        Tile t;
        t.k = k;
        int bx = 0;
        t.nb = 0;
        if (k >= 0) decode(k, bx, t.nb);
        const int tx = bx % a.tiles_x, ty = (bx / a.tiles_x) % a.tiles_y;
        t.b = bx / (a.tiles_x * a.tiles_y);
        t.x0 = tx * TW; t.y0 = ty * TH;
        return t;
    };
    auto draw = [&]() -> int {    // one lane: the next unit of this XCD's band that is a real tile, or -1
        for (;;) {
            const int k = atomicAdd(ctr + xcd, 1);
            if (k >= upx) return -1;
            int bx, nb;
            if (decode(k, bx, nb)) return k;
        }
    };
    auto draw_record = [&](int slot) {   // one lane: draws a unit and files its record
        ws_lds_vint* r = ring + slot * 16;
        const int k = draw();
        if (k >= 0) {
            const Tile T = tile_of(k);
            const ConvGroup& G = a.g[conv_group_of(a, T.nb)];
            const int Cps = G.Cout >> 2;
            int variant = EPI_NONE;
            if (G.Cout % NB == 0 && WS_FAST_EPILOGUE) {
                if (G.mode == CONV_POOL2 && MW == 32 && !G.alpha) variant = EPI_POOL;
                else if (G.mode == CONV_PLAIN && !G.alpha) variant = EPI_PLAIN;
                else if (G.mode == CONV_PS2 && G.alpha && (Cps & (Cps - 1)) == 0) variant = EPI_PS2BN;
            }
            const size_t frame = (size_t)(G.mode == CONV_POOL2 ? (a.H / 2) * (a.W / 2) : a.H * a.W) * G.Cout;
            const unsigned long long out = (unsigned long long)(G.out + (size_t)T.b * frame);
            r[R_B] = T.b; r[R_X0] = T.x0; r[R_Y0] = T.y0; r[R_NB] = T.nb; r[R_VARIANT] = variant;
            r[R_CB] = (T.nb - G.nblk0) * NB; r[R_COUT] = G.Cout; r[R_RELU] = G.relu;
            r[R_CPS_SHIFT] = 31 - __builtin_clz(Cps > 0 ? Cps : 1);
            r[R_OUT_LO] = (int)(unsigned)out; r[R_OUT_HI] = (int)(unsigned)(out >> 32); r[R_FRAME_BYTES] = (int)(frame * 4);
        }
        r[R_K] = k;
    };
    auto tile_from_record = [&](int slot) -> Tile {   // (uniform values: kept in scalar registers)
        ws_lds_vint* r = ring + slot * 16;
        Tile t;
        t.k = __builtin_amdgcn_readfirstlane(r[R_K]);
        t.b = __builtin_amdgcn_readfirstlane(r[R_B]); t.x0 = __builtin_amdgcn_readfirstlane(r[R_X0]);
        t.y0 = __builtin_amdgcn_readfirstlane(r[R_Y0]); t.nb = __builtin_amdgcn_readfirstlane(r[R_NB]);
        return t;
    };

    // ---------------- prologue (every wave passes the same three barriers) ----------------
    if (tid == 256) draw_record(0);
    if (FUSE1A && wave >= 4) {
        float* w1s = (float*)(smem + O_W1S);
        float* b1s = (float*)(smem + O_B1S);
        for (int i = tid - 256; i < 9 * 64; i += 256) w1s[i] = a.w1a[i];
        if (tid - 256 < 64) b1s[tid - 256] = a.b1a[tid - 256];
    }
    __syncthreads();
    const int k_first = __builtin_amdgcn_readfirstlane(ring[R_K]);

    if (wave < 4) {
        // =========================== consumers ===========================
        const int li = lane & 31, lh = lane >> 5;
        constexpr bool ROT = MW == 16;   // column <-> pixel assignment of the 16 x 16 tile (conv_pcol)
        auto prow = [&](int m, int i) { return conv_prow<MW, 2, ROT>(wave, m, i); };
        auto pcol = [&](int i) { return conv_pcol<MW, ROT>(i); };
        struct Ops { bf16x8 xh[2], xl[2], wh[NT], wl[NT]; };
        auto load_ops = [&](const unsigned char* ins, const unsigned char* ws, int tap, Ops& o) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const unsigned char* p = ins + ((prow(m, li) + ky) * IW + (pcol(li) + kx)) * PSTR + lh * 32;
                o.xh[m] = *(const bf16x8*)p;
                if (TERMS == 3) o.xl[m] = *(const bf16x8*)(p + 16);
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const unsigned char* p = ws + (((tap * 2 + lh) * 2) * NB + n * 32 + li) * 16;
                o.wh[n] = *(const bf16x8*)p;
                if (TERMS == 3) o.wl[n] = *(const bf16x8*)(p + NB * 16);
            }
        };
        auto mma = [](bf16x8 wv, bf16x8 xv, f32x16 c) {
            if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv), __builtin_bit_cast(f16x8, xv), c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, c, 0, 0, 0);
        };
        // ---- the epilogue of a finished tile in its three common shapes, as eight slices (one run of four couts of both
        // pixel tiles each) of straight-line code: 32-bit offsets into the output frame and buffer stores whose offset lies
        // outside the buffer for pixels past the image, which the hardware drops (no divergent branches, no 64-bit address
        // arithmetic; the vectors come from LDS).  Same arithmetic as conv_store_run, which serves every other shape.
        struct Pending {
            int variant = EPI_NONE, t = 0, cb = 0, Cout = 0, cps_shift = 0, relu = 0, x0 = 0, y0 = 0;
            __amdgpu_buffer_rsrc_t out;
        } pend;
        constexpr unsigned OUTSIDE = 0x80000000u;
        auto epi_slice = [&](auto VARIANT, const f32x16 (&prv)[2][NT], int sl) __attribute__((always_inline)) {
            constexpr int V = decltype(VARIANT)::value;
            const int n = sl >> 2, j = sl & 3;
            const int cl = n * 32 + 8 * j + 4 * lh;          // first of the run's four couts inside the block
            const float* ev = (const float*)(smem + O_EPI) + (pend.t & 1) * 192;
            const f32x4 bias = *(const f32x4*)(ev + cl);
            const int co = pend.cb + cl;
            auto val = [&](int m, int e) { return F16 ? prv[m][n][4 * j + e] * kF16AccUnscale : prv[m][n][4 * j + e]; };
            if (V == EPI_POOL) {
                const int y = pend.y0 + 2 * wave, x = pend.x0 + li;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float t2 = fmaxf(val(0, e), val(1, e));
                    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t2), 0xB1, 0xF, 0xF, true));  // lane ^ 1
                    const float r = fmaxf(t2, o) + bias[e];
                    v[e] = pend.relu ? fmaxf(r, 0.f) : r;
                }
                const bool ok = (li & 1) == 0 && y + 1 < a.H && x + 1 < a.W;
#ifdef WS_EXP_NOSTORE
                const unsigned off = OUTSIDE + (ok ? 16u : 0u);
#else
                const unsigned off = ok ? (unsigned)((((y >> 1) * (a.W >> 1) + (x >> 1)) * pend.Cout + co) * 4) : OUTSIDE;
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pend.out, off, 0, 0);
            } else {
                f32x4 al, be;
                if (V == EPI_PS2BN) { al = *(const f32x4*)(ev + 64 + cl); be = *(const f32x4*)(ev + 128 + cl); }
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int y = pend.y0 + prow(m, li), x = pend.x0 + pcol(li);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float r = val(m, e) + bias[e];
                        if (V == EPI_PS2BN) r = r * al[e] + be[e];
                        v[e] = pend.relu ? fmaxf(r, 0.f) : r;
                    }
                    unsigned off;
                    if (V == EPI_PLAIN) {
                        off = (unsigned)(((y * a.W + x) * pend.Cout + co) * 4);
                    } else {   // packed cout = q * Cps + c, q = 2 dy + dx (Cps a power of two here)
                        const int q = co >> pend.cps_shift, cc = co & ((1 << pend.cps_shift) - 1);
                        off = (unsigned)((((2 * y + (q >> 1)) * (2 * a.W) + 2 * x + (q & 1)) << pend.cps_shift) + cc) * 4u;
                    }
                    if (y >= a.H || x >= a.W) off = OUTSIDE;
#ifdef WS_EXP_NOSTORE
                    off |= OUTSIDE;
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pend.out, off, 0, 0);
                }
            }
        };
        // one chunk: operands one tap ahead of the MFMAs that use them (the chain of every accumulator is, as in
        // conv_bf16_kernel: chunks in order, taps in order, w_lo x_hi + w_hi x_lo + w_hi x_hi)
        auto chunk = [&](f32x16 (&acc)[2][NT], int g) __attribute__((always_inline)) {
            const unsigned char* ins = smem + (g & 1) * STAGE;
            const unsigned char* ws = ins + IN_BYTES_P;
            Ops ops[2];
            load_ops(ins, ws, 0, ops[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; tap++) {
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < TAPS) load_ops(ins, ws, tap + 1, ops[(tap + 1) & 1]);
                const Ops& o = ops[tap & 1];
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        if (TERMS == 3) {
                            acc[m][n] = mma(o.wl[n], o.xh[m], acc[m][n]);
                            acc[m][n] = mma(o.wh[n], o.xl[m], acc[m][n]);
                        }
                        acc[m][n] = mma(o.wh[n], o.xh[m], acc[m][n]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __syncthreads();   // (the producers' image patch)
        __syncthreads();   // (stage 0 is filled)
        int g = 0, k = k_first;
        f32x16 acc[2][NT];
        for (int t = 0; k >= 0; t++) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
            for (int c = 0; c < nchunks; c++, g++) {
                chunk(acc, g);
                WS_T0(tb);
                ws_barrier();
                WS_ADD(tm_bar, tb);
            }
            // the next tile's unit (drawn by the producers at least three chunks ago)
            k = __builtin_amdgcn_readfirstlane(ring[((t + 1) & 3) * 16 + R_K]);
            // This tile's epilogue, from its record
            WS_T0(te);
            ws_lds_vint* rec = ring + (t & 3) * 16;
            const int variant = __builtin_amdgcn_readfirstlane(rec[R_VARIANT]);
            if (variant == EPI_NONE) {
                const Tile T = tile_from_record(t & 3);
                if (F16) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++)
#pragma unroll
                            for (int r = 0; r < 16; r++) acc[m][n][r] *= kF16AccUnscale;
                }
                const float* epi = (const float*)(smem + O_EPI) + (t & 1) * 192;
                conv_store<NT, MW, 2, false, ROT>(a, acc, wave, li, lh, T.x0, T.y0, T.b, T.nb, epi, epi + 64, epi + 128);
            } else {
                pend.variant = variant; pend.t = t;
                pend.cb = __builtin_amdgcn_readfirstlane(rec[R_CB]); pend.Cout = __builtin_amdgcn_readfirstlane(rec[R_COUT]);
                pend.relu = __builtin_amdgcn_readfirstlane(rec[R_RELU]); pend.cps_shift = __builtin_amdgcn_readfirstlane(rec[R_CPS_SHIFT]);
                pend.x0 = __builtin_amdgcn_readfirstlane(rec[R_X0]); pend.y0 = __builtin_amdgcn_readfirstlane(rec[R_Y0]);
                const unsigned long long out = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rec[R_OUT_LO]) |
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rec[R_OUT_HI]) << 32);
                pend.out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, __builtin_amdgcn_readfirstlane(rec[R_FRAME_BYTES]), 0x00020000);
                if (variant == EPI_POOL) {
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) epi_slice(std::integral_constant<int, EPI_POOL>{}, acc, sl);
                } else if (variant == EPI_PLAIN) {
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) epi_slice(std::integral_constant<int, EPI_PLAIN>{}, acc, sl);
                } else {
#pragma unroll
                    for (int sl = 0; sl < 8; sl++) epi_slice(std::integral_constant<int, EPI_PS2BN>{}, acc, sl);
                }
            }
            WS_ADD(tm_epi, te);
#ifdef PPG_WS_TIMING
            tm_tiles++;
#endif
        }
#ifdef PPG_WS_TIMING
        if (tid == 0 && g_ws_timing) {
            atomicAdd(&g_ws_timing[0], (unsigned long long)(clock64() - tm_start));
            atomicAdd(&g_ws_timing[1], (unsigned long long)tm_bar);
            atomicAdd(&g_ws_timing[2], (unsigned long long)tm_epi);
            atomicAdd(&g_ws_timing[5], 1ull);
            atomicAdd(&g_ws_timing[6], (unsigned long long)tm_tiles);
        }
#endif
    } else {
        // =========================== producers ===========================
        const int pt = tid - 256, pw = wave - 4;
#ifdef PPG_WS_PRODUCER_PRIO
        __builtin_amdgcn_s_setprio(PPG_WS_PRODUCER_PRIO);
#endif
        float* patch = (float*)(smem + O_PATCH);
        const float* w1s = (const float*)(smem + O_W1S);
        const float* b1s = (const float*)(smem + O_B1S);
        constexpr unsigned NOWHERE = 0x80000000u;
        // fetch slots (pixel, half) of this thread: where they go in a stage never changes
        int lofs[IN_PER_T], spix[IN_PER_T];
#pragma unroll
        for (int s = 0; s < IN_PER_T; s++) {
            const int item = pt + 256 * s;
            // FUSE1A: a wave evaluates ONE half (8 channels) for 64 consecutive pixels per slot, so that the conv1a weights it
            // needs are the same in every lane (waves 4, 6: channels 0-7 of the chunk; waves 5, 7: channels 8-15)
            spix[s] = FUSE1A ? 128 * s + (pw >> 1) * 64 + lane : item / 2;
            const int hh = FUSE1A ? pw & 1 : item % 2;
            lofs[s] = spix[s] * PSTR + hh * 32;
        }
        unsigned gofs[IN_PER_T];     // where they come from: byte offset inside the frame (FUSE1A: float offset inside the patch)
        auto slot_sources = [&](const Tile& T) {
#pragma unroll
            for (int s = 0; s < IN_PER_T; s++) {
                const int hh = FUSE1A ? pw & 1 : (pt + 256 * s) % 2;
                const int py = spix[s] / IW, px = spix[s] % IW;
                const int gy = T.y0 + py - 1, gx = T.x0 + px - 1;
                gofs[s] = NOWHERE;
                if (spix[s] < IN_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    gofs[s] = FUSE1A ? (unsigned)(py * PW + px) : (unsigned)(((gy * a.W + gx) * a.Cin + hh * 8) * 4);
            }
        };
        f32x4 vin[IN_PER_T][2];
        bool saturated = false;
        Tile L = tile_from_record(0), P = L;    // L: the tile whose input is being fetched; P: the tile whose chunk is being staged
        int tL = 0, cL = 0, tP = 0, cP = 0;  // their positions in the stream of (tile, chunk)
        __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);

        // input stage, first half: what has to be in flight two chunks ahead
        auto fetch_tile_setup = [&]() {      // L has just become a new tile
            slot_sources(L);
            if (!FUSE1A)
                rin = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)L.b * a.H * a.W * a.Cin), 0, a.H * a.W * a.Cin * 4, 0x00020000);
            if (FUSE1A) {
                // patch pixel (py, px) = image pixel (y0 - 2 + py, x0 - 2 + px); u8 -> f32 by a true division
                // (feature/src/PPGExtractor.cpp:151), zero outside the image (conv1a's own zero padding)
                const uint8_t* im = a.img + (size_t)L.b * a.H * a.W;
                float* pp = patch + (tL & 1) * PATCH_FLOATS;
                for (int i = pt; i < PH * PW; i += 256) {
                    const int gy = L.y0 - 2 + i / PW, gx = L.x0 - 2 + i % PW;
                    float v = 0.f;
                    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = (float)im[(size_t)gy * a.W + gx] / 255.0f;
                    pp[i] = v;
                }
            }
        };
        auto fetch = [&]() {                 // chunk cL of tile L into registers (nothing to do for FUSE1A: the patch is the input)
            if (FUSE1A || L.k < 0) return;
#pragma unroll
            for (int s = 0; s < IN_PER_T; s++)
#pragma unroll
                for (int q = 0; q < 2; q++)
                    vin[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, gofs[s] + 16 * q, cL * KC * 4, 0));
        };
        // input stage, second half + weights: chunk cP of tile P into LDS stage `st`
        auto stage_tile_setup = [&]() {      // P has just become a new tile: its cout block's epilogue vectors
            if (P.k < 0) return;
            int gi = 0;
            if (a.ngroups > 1 && P.nb >= a.g[1].nblk0) gi = 1;
            if (a.ngroups > 2 && P.nb >= a.g[2].nblk0) gi = 2;
            const ConvGroup& G = a.g[gi];
            const int cb = (P.nb - G.nblk0) * NB;
            float* epi = (float*)(smem + O_EPI) + (tP & 1) * 192;
            if (pt < 64) epi[pt] = G.bias[cb + pt];
            else if (pt < 128) epi[pt] = G.alpha ? G.alpha[cb + pt - 64] : 1.f;
            else if (pt < 192) epi[pt] = G.alpha ? G.beta[cb + pt - 128] : 0.f;
        };
        auto stage = [&](int st) {
            if (P.k < 0) return;
            unsigned char* dst = smem + st * STAGE;
            if (!FUSE1A) {
                // the input registers (requested a step ago, hence older than anything below) are waited for HERE: with an
                // LDS-DMA in flight the compiler waits for every outstanding vector-memory operation at the first use of a
                // loaded register, which would put the pieces' whole latency in front of the split
#pragma unroll
                for (int s = 0; s < IN_PER_T; s++) asm volatile("" : "+v"(vin[s][0]), "+v"(vin[s][1]));
            }
            // the chunk's packed weights, 1 KiB per wave instruction, straight into the stage: requested first, so that they travel
            // while this wave evaluates and splits the input tile (they must have landed by the end of the step)
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const unsigned char*)a.wpk + ((size_t)P.nb * nchunks + cP) * W_BYTES), 0, W_BYTES, 0x00020000);
#pragma unroll
            for (int j = 0; j < W_PIECES / 4; j++) {
                const int piece = pw * (W_PIECES / 4) + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (ws_lds_ptr)(dst + IN_BYTES_P + piece * 1024), 16, lane * 16 + piece * 1024, 0, 0, 0);
            }
            if (FUSE1A) {
                // conv1a + ReLU in fp32 for the slot's 8 channels: bias, then the nine taps in (ky, kx) order, one fused
                // multiply-add each (backbone.py:28): the chain of conv1a_kernel, on the vector ALU, whose instructions
                // issue in the shadow of the consumer wave's MFMAs (on the matrix pipe they would queue behind them)
                const float* pp = patch + (tP & 1) * PATCH_FLOATS;
                const int ch = cP * KC + (pw & 1) * 8;      // the same in every lane of the wave
                int po[IN_PER_T];
#pragma unroll
                for (int s = 0; s < IN_PER_T; s++) po[s] = gofs[s] != NOWHERE ? (int)gofs[s] : 0;
                f32x4 o[IN_PER_T][2];
                {
                    const f32x4 b0 = *(const f32x4*)(b1s + ch), b1 = *(const f32x4*)(b1s + ch + 4);
#pragma unroll
                    for (int s = 0; s < IN_PER_T; s++) { o[s][0] = b0; o[s][1] = b1; }
                }
#pragma unroll
                for (int tp = 0; tp < 9; tp++) {
                    const f32x4 w0 = *(const f32x4*)(w1s + tp * 64 + ch), w1 = *(const f32x4*)(w1s + tp * 64 + ch + 4);
#pragma unroll
                    for (int s = 0; s < IN_PER_T; s++) {
                        const float pv = pp[po[s] + (tp / 3) * PW + (tp % 3)];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            o[s][0][e] = __builtin_fmaf(w0[e], pv, o[s][0][e]);
                            o[s][1][e] = __builtin_fmaf(w1[e], pv, o[s][1][e]);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < IN_PER_T; s++) {
                    const bool in = gofs[s] != NOWHERE;
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int e = 0; e < 4; e++) vin[s][q][e] = in ? fmaxf(o[s][q][e], 0.f) : 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < IN_PER_T; s++) {
                u32x4 hi, lo;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    unsigned h0, l0, h1, l1;
                    if (F16) {
                        split_f16(vin[s][q][0], vin[s][q][1], h0, l0, saturated);
                        split_f16(vin[s][q][2], vin[s][q][3], h1, l1, saturated);
                    } else {
                        split_bf16(vin[s][q][0], vin[s][q][1], h0, l0);
                        split_bf16(vin[s][q][2], vin[s][q][3], h1, l1);
                    }
                    hi[2 * q] = h0; hi[2 * q + 1] = h1;
                    lo[2 * q] = l0; lo[2 * q + 1] = l1;
                }
                *(u32x4*)(dst + lofs[s]) = hi;
                if (TERMS == 3) *(u32x4*)(dst + lofs[s] + 16) = lo;
            }
        };
        auto advance = [&](int& t, int& c) { if (++c == nchunks) { c = 0; t++; } };

        if (k_first >= 0) fetch_tile_setup();
        __syncthreads();   // the image patch of the first tile is in LDS
        if (k_first >= 0) {
            stage_tile_setup();
            fetch();                                            // (tile 0, chunk 0)
            stage(0);                                           // the compiler waits for those registers here
            advance(tL, cL);                                    // (tile 0, chunk 1): nchunks >= 4, still tile 0
            asm volatile("" ::: "memory");                      // (the loads below are issued after the LDS-DMA pieces above)
            fetch();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FUSE1A ? 0 : 2 * IN_PER_T) : "memory");   // the weight pieces have landed
        }
        __syncthreads();   // stage 0 is filled
        if (k_first >= 0) {
            // step g: the consumers compute stream position g = (tile, chunk) from stage g & 1; here position g + 1 goes into
            // stage (g + 1) & 1, the input of position g + 2 is requested, and the unit of the tile at g + 3 is drawn
            int tG = 0, cG = 0, tD = 0, cD = 3;   // D: position g + 3
            for (int g = 0;; g++) {
                if (cG == 0 && g > 0 && __builtin_amdgcn_readfirstlane(ring[(tG & 3) * 16 + R_K]) < 0) break;      // the tile the consumers would start now does not exist
                advance(tP, cP);                                     // position g + 1
                if (cP == 0) { P = L; stage_tile_setup(); }          // (L reached this tile one step ago)
                stage((g + 1) & 1);
                asm volatile("" ::: "memory");
                advance(tL, cL);                                     // position g + 2
                if (cL == 0) {
                    L = tile_from_record(tL & 3);                     // drawn at step g - 1
                    if (L.k >= 0) fetch_tile_setup();
                }
                fetch();
                if (cD == 0 && pt == 0) draw_record(tD & 3);         // (tD = tile of position g + 3)
                advance(tD, cD);
                // the weight pieces of this step have landed; the register loads issued after them (when there are any)
                // may stay in flight
                if (FUSE1A || L.k < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IN_PER_T) : "memory");
                WS_T0(tb);
                ws_barrier();
                WS_ADD(tm_bar, tb);
                advance(tG, cG);
            }
        }
        if (F16 && saturated && a.sat) atomicOr(a.sat, 1);
#ifdef PPG_WS_TIMING
        if (tid == 256 && g_ws_timing) {
            atomicAdd(&g_ws_timing[3], (unsigned long long)(clock64() - tm_start));
            atomicAdd(&g_ws_timing[4], (unsigned long long)tm_bar);
        }
#endif
    }
    // the last work-group to finish puts the counters back to zero for the next launch
    __syncthreads();
    if (tid == 0 && atomicAdd(ctr + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; i++) atomicExch(ctr + i, 0);
    }
}

hipError_t launch_conv_ws(const ConvArgs& args, hipStream_t s) {
    ConvArgs a = args;
    if (a.prec < 1 || a.prec > 3 || a.Cin % 16 || a.Cin < 64 || !a.tile_counter || a.ngroups < 1 || a.ngroups > 3 || a.quarter)
        return hipErrorInvalidValue;
    if ((long)a.H * a.W * a.Cin * 4 > 0x7fffffffL) return hipErrorInvalidValue;
    if (a.img && (a.Cin != 64 || a.mw != 32)) return hipErrorInvalidValue;
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.B;
    int G = std::max(1, std::min(a.nblk, a.xcd_group > 0 ? a.xcd_group : 1));
    while (a.nblk % G) G--;
    a.xcd_group = G;
    a.xcd_tiles8 = (int)((ntiles + 7) / 8);
    a.tile_counter += 128;   // its own nine counters, apart from those of the fp32 kernels
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long units = ntiles * a.nblk;
    long wgs = std::min<long>(cus, (units + 7) / 8 * 8);
    wgs = std::max<long>(8, wgs / 8 * 8);     // a multiple of eight: work-group L serves the band of XCD L % 8
    dim3 grid((unsigned)wgs), block(512);
#define WS_LAUNCH(FUSE, MWV, TERMSV, F16V) hipLaunchKernelGGL((conv_ws_kernel<FUSE, MWV, TERMSV, F16V>), grid, block, 0, s, a)
    const int terms = a.prec == 2 ? 1 : 3;
    const bool f16 = a.prec == 3;
    if (a.img) {
        if (f16) WS_LAUNCH(true, 32, 3, true);
        else if (terms == 3) WS_LAUNCH(true, 32, 3, false);
        else WS_LAUNCH(true, 32, 1, false);
    } else if (a.mw == 16) {
        if (f16) WS_LAUNCH(false, 16, 3, true);
        else if (terms == 3) WS_LAUNCH(false, 16, 3, false);
        else WS_LAUNCH(false, 16, 1, false);
    } else {
        if (f16) WS_LAUNCH(false, 32, 3, true);
        else if (terms == 3) WS_LAUNCH(false, 32, 3, false);
        else WS_LAUNCH(false, 32, 1, false);
    }
#undef WS_LAUNCH
    return hipGetLastError();
}

}  // namespace ppg
