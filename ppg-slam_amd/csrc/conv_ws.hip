// Wave-specialised form of the reduced-precision 3x3 convolution (the arithmetic of conv_bf16_kernel in conv_kernels.hip:
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
// stage (g + 1) & 1.  The chunk stream runs on across tile boundaries: what a tile needs before its first chunk (its unit,
// its record, its bias / batch-norm vectors, the image patch, its first input chunk) is requested by the producers five
// to two steps ahead, and its epilogue (bias / batch norm / ReLU / pooling and the stores) goes out in slices behind the
// first taps of the NEXT tile, from a copy of the sums.  The MFMA's rows are the pixels and its columns the couts, so
// that a store instruction writes whole 128-byte lines of the channels-last output.
#include "ppg_kernels.h"
#include "conv_split.h"
#include "conv_store.h"

#include <algorithm>

namespace ppg {

typedef __attribute__((address_space(3))) void* ws_lds_ptr;

// One LDS-DMA instruction: 16 bytes per lane from buffer `rsrc` at byte offset `voff` (per lane) to LDS address
// `lds_base` (the same in every lane) + 16 * lane.  Written as assembly on purpose: the compiler orders every later LDS
// access of the wave behind a DMA it knows about (s_waitcnt vmcnt(0): it cannot tell that the ring, the input image and
// the weights are different parts of one array), which puts the transfer's whole latency into the producers' step.
// Here the kernel waits itself, with a counted s_waitcnt, where the pieces are needed.  The compiler's own counted waits
// stay correct: vector-memory operations complete in order, so operations it does not know about only make a wait
// of its own stronger.
__device__ __forceinline__ void ws_dma_b128(u32x4 rsrc, unsigned lds_base, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc) : "memory");   // (M0 is written; nothing else in this kernel uses it)
}

// Every wave of the work-group, once per chunk.  A raw s_barrier (no vmcnt(0): the producers' register loads for the
// chunk after next stay in flight across it); the asm's memory clobber keeps LDS accesses on their side of it.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef PPG_WS_TIMING
// microbenchmark builds only (tools/ubench/conv_bench.hip): shader cycles summed over work-groups:
// [0] consumer wave 0 in total, [1] of that waiting at the chunk barriers, [2] handing a tile's sums over to the epilogue,
// [3] producer wave 4 in total, [4] of that waiting at the barriers, [5] work-groups, [6] tiles, [7] producers staging
// (weights requested, input split and written, next input requested), [8] their per-tile set-up, [9] waiting for the
// weight pieces, [10] waiting for the input registers
__device__ unsigned long long* g_ws_timing = nullptr;
#define WS_T0(var) const long long var = clock64()
#define WS_ADD(acc, t0) acc += clock64() - (t0)
#else
#define WS_T0(var)
#define WS_ADD(acc, t0)
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
    constexpr int O_EPI = O_B1S + (FUSE1A ? 64 * 4 : 0);                // [4][3][64]: bias, alpha, beta of the cout blocks of the tiles in flight
    constexpr int O_RING = O_EPI + 4 * 3 * 64 * 4;                      // [4][16] records of the tiles in flight
    constexpr int O_GTAB = O_RING + 4 * 64;                             // [3][12] the launch's layers (ConvGroup, as dwords)
    constexpr int TOTAL = O_GTAB + 3 * 48;
    static_assert(sizeof(ConvGroup) == 48, "the layer table in LDS is a copy of ConvArgs::g");
    enum { G_OUT = 0, G_BIAS = 2, G_ALPHA = 4, G_BETA = 6, G_NBLK0 = 8, G_COUT = 9, G_MODE = 10, G_RELU = 11 };   // dwords of a ConvGroup
    static_assert(W_PIECES % 4 == 0, "the weight pieces are dealt evenly to the four producer waves");
    static_assert(TOTAL <= 160 * 1024, "one work-group per CU");
    // (ONE shared object: a second one beside an LDS-DMA destination makes the compiler wait for every load in flight
    // before each LDS read)
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];
    typedef __attribute__((address_space(3))) volatile int ws_lds_vint;   // (an LDS pointer type: a generic volatile pointer would be read by flat loads)
    ws_lds_vint* ring = (ws_lds_vint*)(smem + O_RING);
    ws_lds_vint* gtab = (ws_lds_vint*)(smem + O_GTAB);

    // (the wave number as a scalar: the split into consumers and producers, and everything that depends on the wave only,
    // are then scalar branches and scalar values instead of masked vector code)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int xcd = blockIdx.x & 7;
#ifdef PPG_WS_TIMING
    long long tm_total = 0, tm_bar = 0, tm_epi = 0, tm_tiles = 0, tm_stage = 0, tm_fetch = 0, tm_vm = 0, tm_touch = 0;
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
    enum { R_K = 0, R_B, R_X0, R_Y0, R_NB, R_MODE, R_CB, R_COUT, R_RELU, R_BN, R_OUT_LO, R_OUT_HI, R_FRAME_BYTES, R_GI };
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
    // One lane draws: the next unit of this XCD's band that is a real tile, or -1.  `k` is a unit already drawn (the atomic
    // that returned it was issued a step earlier, so that nobody waits for its round trip); units that are padding of the
    // numbering are skipped with further, synchronous draws.
    auto draw_from = [&](int k) -> int {
        for (;;) {
            if (k >= upx) return -1;
            int bx, nb;
            if (decode(k, bx, nb)) return k;
            k = atomicAdd(ctr + xcd, 1);
        }
    };
    // ... and files the tile's record.  The layers' descriptions come from their copy in LDS: indexed in the kernel arguments
    // they would be a chain of dependent vector loads, a memory round trip each, once per tile on the producers' path.
    auto file_record = [&](int slot, int k) {
        ws_lds_vint* r = ring + slot * 16;
        if (k >= 0) {
            const Tile T = tile_of(k);
            int gi = 0;
            if (a.ngroups > 1 && T.nb >= gtab[12 + G_NBLK0]) gi = 1;
            if (a.ngroups > 2 && T.nb >= gtab[24 + G_NBLK0]) gi = 2;
            ws_lds_vint* G = gtab + gi * 12;
            const int Cout = G[G_COUT], mode = G[G_MODE];
            const size_t frame = (size_t)(mode == CONV_POOL2 ? (a.H / 2) * (a.W / 2) : a.H * a.W) * Cout;
            const unsigned long long out = ((unsigned long long)(unsigned)G[G_OUT] | ((unsigned long long)(unsigned)G[G_OUT + 1] << 32)) + (size_t)T.b * frame * 4;
            r[R_B] = T.b; r[R_X0] = T.x0; r[R_Y0] = T.y0; r[R_NB] = T.nb; r[R_MODE] = mode;
            r[R_CB] = (T.nb - G[G_NBLK0]) * NB; r[R_COUT] = Cout; r[R_RELU] = G[G_RELU]; r[R_BN] = (G[G_ALPHA] | G[G_ALPHA + 1]) != 0;
            r[R_OUT_LO] = (int)(unsigned)out; r[R_OUT_HI] = (int)(unsigned)(out >> 32); r[R_FRAME_BYTES] = (int)(frame * 4);
            r[R_GI] = gi;
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

    // ---------------- prologue (every wave passes the same four barriers) ----------------
    if (tid < 36) gtab[tid] = ((const int*)a.g)[tid];
    if (FUSE1A && wave >= 4) {
        float* w1s = (float*)(smem + O_W1S);
        float* b1s = (float*)(smem + O_B1S);
        for (int i = tid - 256; i < 9 * 64; i += 256) w1s[i] = a.w1a[i];
        if (tid - 256 < 64) b1s[tid - 256] = a.b1a[tid - 256];
    }
    __syncthreads();
    if (tid == 256) {   // the first two tiles (the others are drawn in the loop, five steps ahead of their first chunk)
        file_record(0, draw_from(atomicAdd(ctr + xcd, 1)));
        file_record(1, draw_from(atomicAdd(ctr + xcd, 1)));
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
        // The pixels are the MFMA's rows and the couts its columns: in the result a lane holds ONE cout (column lane & 31) of
        // sixteen pixels (register r: pixel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-pixel tile), so that a store
        // instruction writes 32 consecutive couts of a pixel, a whole 128-byte line, per half wave.  (With the couts as
        // rows a lane holds four couts of one pixel and a store instruction touches 64 lines with 16 bytes each: the
        // epilogue is then bound by the rate of write requests.)  Every sum is the same either way: the products of an
        // output element and the order they are added in depend on k only.
        auto mma = [](bf16x8 wv, bf16x8 xv, f32x16 c) {
            if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xv), __builtin_bit_cast(f16x8, wv), c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wv, c, 0, 0, 0);
        };
        // ---- the epilogue of a finished tile: bias, (batch norm), (ReLU), then the plain / pooled / pixel-shuffled store, the
        // arithmetic of conv_store_run.  32-bit offsets into the output frame; a store whose pixel or cout does not exist
        // gets an offset past the end of the buffer, which the hardware drops (no divergent branches).
        //
        // The epilogue of tile t runs while tile t + 1 is being multiplied: its sums are copied to `prv` and go out in 32
        // slices of two store instructions each (16 of one for a pooled layer), one slice after each of the next tile's
        // first taps, in the shadow of that tap's last MFMAs.  Done in one go the 64 store instructions of a wave take
        // 2-3 us (the CU's store path moves about 16 bytes a clock), during which the matrix pipe would stand still.
        constexpr unsigned OUTSIDE = 0x80000000u;   // (+ any offset inside a frame: still past its end)
        constexpr auto CR = [](int r) { return (r & 3) + 8 * (r >> 2); };   // pixel of register r, less 4 (lane >> 5)
        f32x16 prv[2][NT];
        struct Epi {
            int live = 0, pool = 0, ps2 = 0, bn = 0, fast = 0, Cout = 0, Cps = 0, pstride = 0, x0 = 0, y0 = 0;
            float floor = 0.f;
            __amdgpu_buffer_rsrc_t out;
            float bias[NT], al[NT], be[NT];
            int co[NT], q[NT];              // the lane's cout per cout tile (-1: past the layer); pixel shuffle: its sub-pixel
            unsigned base[2][NT];           // fast / pooled form: offset of the lane's first pixel of pixel tile m
            unsigned baseA[2][NT];          // 16-wide tiles: the same for the two registers whose column wraps around
            int rowstride = 0;              // bytes from a pixel to the one below it
        } E;
        enum { EK_POOL = 0, EK_FAST = 1, EK_GENERIC = 2 };   // the forms of a tile's epilogue (a chunk is compiled for each)
        auto epi_offset = [&](int n, int y, int x) -> unsigned {   // plain: pixel (y, x), cout co.  Pixel shuffle: packed cout
            // = q Cps + c, q = 2 dy + dx -> pixel (2 y + dy, 2 x + dx), channel c of a frame twice as wide and high
            if (E.ps2) return (unsigned)((((2 * y + (E.q[n] >> 1)) * (2 * a.W) + 2 * x + (E.q[n] & 1)) * E.Cps + E.co[n] - E.q[n] * E.Cps) * 4);
            return (unsigned)(((y * a.W + x) * E.Cout + E.co[n]) * 4);
        };
        auto epi_setup = [&](int t) __attribute__((always_inline)) {
            ws_lds_vint* rec = ring + (t & 3) * 16;
            const int mode = __builtin_amdgcn_readfirstlane(rec[R_MODE]), cb = __builtin_amdgcn_readfirstlane(rec[R_CB]);
            E.Cout = __builtin_amdgcn_readfirstlane(rec[R_COUT]); E.bn = __builtin_amdgcn_readfirstlane(rec[R_BN]);
            E.floor = __builtin_amdgcn_readfirstlane(rec[R_RELU]) ? 0.f : -__builtin_inff();   // (no ReLU: max(r, -inf))
            E.x0 = __builtin_amdgcn_readfirstlane(rec[R_X0]); E.y0 = __builtin_amdgcn_readfirstlane(rec[R_Y0]);
            const unsigned long long outp = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rec[R_OUT_LO]) |
                                            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rec[R_OUT_HI]) << 32);
            E.out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, __builtin_amdgcn_readfirstlane(rec[R_FRAME_BYTES]), 0x00020000);
            E.pool = mode == CONV_POOL2; E.ps2 = mode == CONV_PS2;
            E.Cps = E.Cout >> 2;
            E.pstride = E.ps2 ? 2 * E.Cps * 4 : E.Cout * 4;     // bytes from a pixel to its right neighbour's
            E.rowstride = E.ps2 ? 2 * (2 * a.W) * E.Cps * 4 : a.W * E.Cout * 4;
            // the tile lies inside the image: every pixel exists, and (32-wide tiles) a lane's sixteen pixels of a pixel tile
            // are a fixed number of pixels apart, which goes into the store's scalar offset
            const bool inside = E.x0 + TW <= a.W && E.y0 + TH <= a.H;
            E.fast = inside;
            const float* ev = (const float*)(smem + O_EPI) + (t & 3) * 192;
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const int cl = n * 32 + li, co = cb + cl;
                E.bias[n] = ev[cl]; E.al[n] = ev[64 + cl]; E.be[n] = ev[128 + cl];
                E.co[n] = co < E.Cout ? co : -1;
                E.q[n] = E.ps2 ? (co >= E.Cps) + (co >= 2 * E.Cps) + (co >= 3 * E.Cps) : 0;
                if (E.pool) {
                    // (32-wide tiles only: the two rows of an output pixel are the wave's two pixel tiles, its two columns
                    // two registers of the lane)
                    const int y = E.y0 + 2 * wave, xl = E.x0 + 4 * lh;
                    E.base[0][n] = E.co[n] >= 0 && y + 1 < a.H ? (unsigned)((((y >> 1) * (a.W >> 1) + (xl >> 1)) * E.Cout + co) * 4) : OUTSIDE;
                    E.base[1][n] = OUTSIDE;
                } else {
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        E.base[m][n] = E.co[n] >= 0 ? epi_offset(n, E.y0 + prow(m, 0), E.x0 + 4 * lh) : OUTSIDE;
                        // (an offset of its own rather than base - 16 pixels: the range check of a buffer store looks at the
                        // vector offset alone, which must therefore lie inside the frame by itself)
                        E.baseA[m][n] = MW == 16 && E.co[n] >= 0 ? epi_offset(n, E.y0 + prow(m, 0), E.x0 + (lh ? 2 : 14)) : OUTSIDE;
                    }
                }
            }
            E.live = 1;
        };
        // Slice s = 0 .. 31 (a compile-time constant where it is called) of the pending epilogue in form KIND: straight-line
        // code.  The batch norm is applied in every case, with alpha = 1 and beta = -0 when the layer has none (v * 1 and
        // v + -0 are v, bit for bit).
        auto epi_slice = [&](auto KIND, int s) __attribute__((always_inline)) {
            constexpr int K = decltype(KIND)::value;
            auto val = [&](int m, int n, int r) { return F16 ? prv[m][n][r] * kF16AccUnscale : prv[m][n][r]; };
            if (K == EK_POOL) {
                if (s >= 16) return;
                const int n = s >> 3, r = 2 * (s & 7);
                const float v = fmaxf(fmaxf(fmaxf(val(0, n, r), val(1, n, r)), fmaxf(val(0, n, r + 1), val(1, n, r + 1))) + E.bias[n], E.floor);
                const unsigned off = E.fast || E.x0 + 4 * lh + CR(r) + 1 < a.W ? E.base[0][n] : OUTSIDE;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.out, off, (CR(r) >> 1) * E.Cout * 4, 0);
                return;
            }
            const int m = s >> 4, n = (s >> 3) & 1;
#pragma unroll
            for (int r = 2 * (s & 7); r < 2 * (s & 7) + 2; r++) {
                const float v = fmaxf((val(m, n, r) + E.bias[n]) * E.al[n] + E.be[n], E.floor);
                if (K == EK_FAST) {
                    if (MW == 32) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.out, E.base[m][n], CR(r) * E.pstride, 0);
                    } else {
                        // 16 x 16 tile: registers 0-7 are the pixel tile's first row, columns CR(r) + 4 (lane >> 5); registers
                        // 8-15 its second row with the columns rotated by two (conv_pcol): columns 14, 15, 0, 1, 6, 7, 8, 9 in
                        // the lower half wave, 2, 3, 4, 5, 10, 11, 12, 13 in the upper = the same + 4, but for the first two
                        // (baseA: column 14 resp. 2)
                        constexpr int ROTC[8] = {0, 1, 0, 1, 6, 7, 8, 9};
                        if (r < 8) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.out, E.base[m][n], CR(r) * E.pstride, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.out, r < 10 ? E.baseA[m][n] : E.base[m][n],
                                                                   E.rowstride + ROTC[r & 7] * E.pstride, 0);
                    }
                } else {
                    const int pix = CR(r) + 4 * lh;
                    const int y = E.y0 + prow(m, pix), x = E.x0 + pcol(pix);
                    const unsigned off = E.co[n] >= 0 && y < a.H && x < a.W ? epi_offset(n, y, x) : OUTSIDE;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), E.out, off, 0, 0);
                }
            }
        };
        // one chunk: operands one tap ahead of the MFMAs that use them (the chain of every accumulator is, as in
        // conv_bf16_kernel: chunks in order, taps in order, w_lo x_hi + w_hi x_lo + w_hi x_hi)
        // C: the chunk's number inside its tile when the slices 9 C .. 9 C + 8 of the previous tile's epilogue go with it
        // (a compile-time constant), or -1
        auto chunk = [&](f32x16 (&acc)[2][NT], int g, auto C, auto KIND) __attribute__((always_inline)) {
            const unsigned char* ins = smem + (g & 1) * STAGE;
            const unsigned char* ws = ins + IN_BYTES_P;
            Ops ops[2];
            load_ops(ins, ws, 0, ops[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; tap++) {
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < TAPS) load_ops(ins, ws, tap + 1, ops[(tap + 1) & 1]);
                const Ops& o = ops[tap & 1];
                // (every accumulator still sees its three terms in the same order; consecutive MFMAs write different ones)
#pragma unroll
                for (int term = (TERMS == 3 ? 0 : 2); term < 3; term++)
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++)
                            acc[m][n] = term == 0 ? mma(o.wl[n], o.xh[m], acc[m][n]) : term == 1 ? mma(o.wh[n], o.xl[m], acc[m][n]) : mma(o.wh[n], o.xh[m], acc[m][n]);
                // one operand read of the next tap in the shadow of each of the first MFMAs
                if (tap + 1 < TAPS) {
                    constexpr int READS = TERMS == 3 ? 4 + 2 * NT : 2 + NT, MFMAS = 2 * NT * TERMS;
#pragma unroll
                    for (int i = 0; i < (READS < MFMAS ? READS : MFMAS); i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    if (MFMAS > READS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - READS, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (decltype(C)::value >= 0 && decltype(C)::value * TAPS + tap < 32) {
                    epi_slice(KIND, decltype(C)::value * TAPS + tap);
                    __builtin_amdgcn_sched_barrier(0);
                }
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
            auto sync = [&]() {
                WS_T0(tb);
                ws_barrier();
                WS_ADD(tm_bar, tb);
                g++;
            };
            // (nchunks >= 4: the first four chunks carry the previous tile's epilogue, in code compiled for its form)
            auto four = [&](auto KIND) __attribute__((always_inline)) {
                chunk(acc, g, std::integral_constant<int, 0>{}, KIND); sync();
                chunk(acc, g, std::integral_constant<int, 1>{}, KIND); sync();
                chunk(acc, g, std::integral_constant<int, 2>{}, KIND); sync();
                chunk(acc, g, std::integral_constant<int, 3>{}, KIND); sync();
            };
            int c0 = 4;
            if (!E.live) c0 = 0;
            else if (E.pool) four(std::integral_constant<int, EK_POOL>{});
            else if (E.fast) four(std::integral_constant<int, EK_FAST>{});
            else four(std::integral_constant<int, EK_GENERIC>{});
            for (int c = c0; c < nchunks; c++) { chunk(acc, g, std::integral_constant<int, -1>{}, std::integral_constant<int, EK_GENERIC>{}); sync(); }
            // the next tile's unit (drawn by the producers several chunks ago)
            k = __builtin_amdgcn_readfirstlane(ring[((t + 1) & 3) * 16 + R_K]);
            WS_T0(te);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) prv[m][n] = acc[m][n];
            epi_setup(t);
            WS_ADD(tm_epi, te);
#ifdef PPG_WS_TIMING
            tm_tiles++;
#endif
        }
        {   // the last tile's epilogue
            WS_T0(te);
            if (E.live) {
                if (E.pool) {
#pragma unroll
                    for (int s2 = 0; s2 < 32; s2++) epi_slice(std::integral_constant<int, EK_POOL>{}, s2);
                } else if (E.fast) {
#pragma unroll
                    for (int s2 = 0; s2 < 32; s2++) epi_slice(std::integral_constant<int, EK_FAST>{}, s2);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < 32; s2++) epi_slice(std::integral_constant<int, EK_GENERIC>{}, s2);
                }
            }
            WS_ADD(tm_epi, te);
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
        unsigned gofsP[IN_PER_T];    // FUSE1A: the same for tile P (L may already be one tile further)
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
        int rin_lo = 0, rin_hi = 0;          // the frame being fetched (address halves, scalar)

        // What a tile needs from memory before its input can be fetched is requested one step before it is used (its
        // cout block's bias / batch-norm vectors: one per wave; FUSE1A: the image patch), so that nobody waits for it.
        constexpr int PP = (PH * PW + 255) / 256;
        float evreg = 0.f;
        unsigned char praw[PP > 0 ? PP : 1];
        auto request_tile = [&](int slot) {
            ws_lds_vint* r = ring + slot * 16;
            if (__builtin_amdgcn_readfirstlane(r[R_K]) < 0) return;
            const int gi = __builtin_amdgcn_readfirstlane(r[R_GI]), cb = __builtin_amdgcn_readfirstlane(r[R_CB]);
            if (pw < 3) {    // bias, alpha, beta
                const unsigned long long v = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(gtab[gi * 12 + G_BIAS + 2 * pw]) |
                                             ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(gtab[gi * 12 + G_BIAS + 2 * pw + 1]) << 32);
                evreg = pw == 1 ? 1.f : pw == 2 ? -0.f : 0.f;    // (no batch norm: v * 1 + -0 is v, bit for bit)
                // (the bias arrays are padded to whole cout blocks, the batch-norm vectors are not: couts past the layer, whose
                // results are never stored, read its last entry)
                const int Cout = __builtin_amdgcn_readfirstlane(r[R_COUT]);
                if (v) evreg = ((const float*)v)[pw == 0 ? cb + lane : min(cb + lane, Cout - 1)];
            }
            if (FUSE1A) {
                // patch pixel (py, px) = image pixel (y0 - 2 + py, x0 - 2 + px); outside the image the load is given an offset
                // past the end of the frame and returns 0 (conv1a's own zero padding)
                const int b = __builtin_amdgcn_readfirstlane(r[R_B]), x0 = __builtin_amdgcn_readfirstlane(r[R_X0]), y0 = __builtin_amdgcn_readfirstlane(r[R_Y0]);
                const __amdgpu_buffer_rsrc_t rim = __builtin_amdgcn_make_buffer_rsrc((void*)(a.img + (size_t)b * a.H * a.W), 0, a.H * a.W, 0x00020000);
#pragma unroll
                for (int i = 0; i < PP; i++) {
                    const int idx = pt + 256 * i;
                    const int gy = y0 - 2 + idx / PW, gx = x0 - 2 + idx % PW;
                    const bool in = idx < PH * PW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    praw[i] = __builtin_amdgcn_raw_buffer_load_b8(rim, in ? (unsigned)(gy * a.W + gx) : NOWHERE, 0, 0);
                }
            }
        };
        // input stage, first half: what has to be in flight two chunks ahead
        auto fetch_tile_setup = [&]() {      // L has just become a new tile (request_tile saw it a step ago)
            slot_sources(L);
            if (pw < 3) ((float*)(smem + O_EPI))[(tL & 3) * 192 + pw * 64 + lane] = evreg;
            if (!FUSE1A) {
                const unsigned long long f = (unsigned long long)(a.in + (size_t)L.b * a.H * a.W * a.Cin);
                rin_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)f);
                rin_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(f >> 32));
            }
            if (FUSE1A) {
                // u8 -> f32 by a true division (feature/src/PPGExtractor.cpp:151)
                float* pp = patch + (tL & 1) * PATCH_FLOATS;
#pragma unroll
                for (int i = 0; i < PP; i++)
                    if (pt + 256 * i < PH * PW) pp[pt + 256 * i] = (float)praw[i] / 255.0f;
            }
        };
        auto fetch = [&]() {                 // chunk cL of tile L into registers (nothing to do for FUSE1A: the patch is the input)
            if (FUSE1A || L.k < 0) return;
            // (descriptor and chunk offset rebuilt from scalars: as values that live across the loop they end up in vector
            // registers, and every load becomes a loop over the distinct descriptors of the wave)
            const unsigned long long f = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rin_lo) |
                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rin_hi) << 32);
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)f, 0, a.H * a.W * a.Cin * 4, 0x00020000);
#pragma unroll
            for (int s = 0; s < IN_PER_T; s++)
#pragma unroll
                for (int q = 0; q < 2; q++)
                    vin[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, gofs[s] + 16 * q, __builtin_amdgcn_readfirstlane(cL) * KC * 4, 0));
        };
        // input stage, second half + weights: chunk cP of tile P into LDS stage `st`
        auto stage = [&](int st) {
            if (P.k < 0) return;
            unsigned char* dst = smem + st * STAGE;
            if (!FUSE1A) {
                // the input registers (requested a step ago, hence older than anything below) are waited for HERE: with an
                // LDS-DMA in flight the compiler waits for every outstanding vector-memory operation at the first use of a
                // loaded register, which would put the pieces' whole latency in front of the split
                WS_T0(tw);
#pragma unroll
                for (int s = 0; s < IN_PER_T; s++) asm volatile("" : "+v"(vin[s][0]), "+v"(vin[s][1]));
                WS_ADD(tm_touch, tw);
            }
            // the chunk's packed weights, 1 KiB per wave instruction, straight into the stage: requested first, so that they travel
            // while this wave evaluates and splits the input tile (they must have landed by the end of the step)
            const unsigned long long wsrc = (unsigned long long)((const unsigned char*)a.wpk + ((size_t)P.nb * nchunks + cP) * W_BYTES);
            const u32x4 rw = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wsrc),
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wsrc >> 32)) & 0xffffu, (unsigned)W_BYTES, 0x00020000u};
            const unsigned wdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(ws_lds_ptr)(dst + IN_BYTES_P));
#pragma unroll
            for (int j = 0; j < W_PIECES / 4; j++) {
                const int piece = pw * (W_PIECES / 4) + j;
                ws_dma_b128(rw, (unsigned)__builtin_amdgcn_readfirstlane((int)(wdst + piece * 1024)), lane * 16 + piece * 1024);
            }
            if (FUSE1A) {
                // conv1a + ReLU in fp32 for the slot's 8 channels: bias, then the nine taps in (ky, kx) order, one fused
                // multiply-add each (backbone.py:28): the chain of conv1a_kernel, on the vector ALU, whose instructions
                // issue in the shadow of the consumer wave's MFMAs (on the matrix pipe they would queue behind them)
                const float* pp = patch + (tP & 1) * PATCH_FLOATS;
                const int ch = cP * KC + (pw & 1) * 8;      // the same in every lane of the wave
                int po[IN_PER_T];
#pragma unroll
                for (int s = 0; s < IN_PER_T; s++) po[s] = gofsP[s] != NOWHERE ? (int)gofsP[s] : 0;
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
                    const bool in = gofsP[s] != NOWHERE;
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
        auto advance = [&](int& t, int& c) {   // (kept in scalar registers)
            if (++c == nchunks) { c = 0; t++; }
            c = __builtin_amdgcn_readfirstlane(c); t = __builtin_amdgcn_readfirstlane(t);
        };

        if (k_first >= 0) { request_tile(0); fetch_tile_setup(); }
        __syncthreads();   // the image patch of the first tile is in LDS
        if (k_first >= 0) {
#pragma unroll
            for (int s = 0; s < IN_PER_T; s++) gofsP[s] = gofs[s];
            fetch();                                            // (tile 0, chunk 0)
            stage(0);                                           // the compiler waits for those registers here
            advance(tL, cL);                                    // (tile 0, chunk 1): nchunks >= 4, still tile 0
            asm volatile("" ::: "memory");                      // (the loads below are issued after the LDS-DMA pieces above)
            fetch();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FUSE1A ? 0 : 2 * IN_PER_T) : "memory");   // the weight pieces have landed
        }
        __syncthreads();   // stage 0 is filled
        if (k_first >= 0) {
            // Step g: the consumers compute stream position g = (tile, chunk) from stage g & 1.  Here position g + 1 goes
            // into stage (g + 1) & 1 and the input of position g + 2 is requested; a tile whose first chunk is position p is
            // drawn at step p - 5 (the atomic is issued), filed at p - 4 (its record), its vectors and patch are requested
            // at p - 3 and put into LDS at p - 2, when its first input chunk is requested.
            int tG = 0, cG = 0;
            int tN = 0, cN = 2, tD4 = 0, cD4 = 3, tD5 = nchunks == 4 ? 1 : 0, cD5 = nchunks == 4 ? 0 : 4;
            int kq = -1;     // (lane 0 of the first producer wave: the unit drawn last)
            for (int g = 0;; g++) {
                if (cG == 0 && g > 0 && __builtin_amdgcn_readfirstlane(ring[(tG & 3) * 16 + R_K]) < 0) break;      // the tile the consumers would start now does not exist
                // What may wait for memory comes before the weight pieces are requested: a wait the compiler places after
                // that would wait for them as well.
                WS_T0(tf);
                advance(tD5, cD5);                                   // position g + 5
                advance(tD4, cD4);                                   // position g + 4
                if (pt == 0) {
                    if (cD4 == 0 && tD4 >= 2) file_record(tD4 & 3, draw_from(kq));
                    if (cD5 == 0 && tD5 >= 2) kq = atomicAdd(ctr + xcd, 1);
                }
                advance(tP, cP);                                     // position g + 1
                if (cP == 0) {                                       // (L reached this tile one step ago)
                    P = L;
#pragma unroll
                    for (int s = 0; s < IN_PER_T; s++) gofsP[s] = gofs[s];
                }
                advance(tL, cL);                                     // position g + 2
                if (cL == 0) {
                    L = tile_from_record(tL & 3);
                    if (L.k >= 0) fetch_tile_setup();
                }
                advance(tN, cN);                                     // position g + 3
                if (cN == 0) request_tile(tN & 3);                   // (filed at step g - 1)
                WS_ADD(tm_fetch, tf);
                WS_T0(ts);
                stage((g + 1) & 1);
                asm volatile("" ::: "memory");
                fetch();
                WS_ADD(tm_stage, ts);
                WS_T0(tv);
                // the weight pieces of this step have landed; the register loads issued after them (when there are any)
                // may stay in flight
                if (FUSE1A || L.k < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IN_PER_T) : "memory");
                WS_ADD(tm_vm, tv);
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
            atomicAdd(&g_ws_timing[7], (unsigned long long)tm_stage);
            atomicAdd(&g_ws_timing[8], (unsigned long long)tm_fetch);
            atomicAdd(&g_ws_timing[9], (unsigned long long)tm_vm);
            atomicAdd(&g_ws_timing[10], (unsigned long long)tm_touch);
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
    for (int i = 0; i < a.ngroups; i++)
        if (a.g[i].mode == CONV_POOL2 && a.mw != 32) return hipErrorInvalidValue;   // (the epilogue pools inside a 32-wide tile only)
    const long ntiles = (long)a.tiles_x * a.tiles_y * a.B;
    int G = std::max(1, std::min(a.nblk, a.xcd_group > 0 ? a.xcd_group : 1));
    while (a.nblk % G) G--;
    a.xcd_group = G;
    a.xcd_tiles8 = (int)((ntiles + 7) / 8);
    a.tile_counter += 128;   // its own nine counters, apart from those of the fp32 kernels
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long units = ntiles * a.nblk;
    static const int reserve = [] { const char* e = getenv("PPG_CONV_WS_RESERVE"); return e ? std::max(0, atoi(e)) / 8 * 8 : 0; }();
    cus = std::max(8, cus - reserve);       // CUs left to the kernels of the other streams while this one runs
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
