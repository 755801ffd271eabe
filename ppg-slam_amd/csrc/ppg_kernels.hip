// gfx950 (MI355X, CDNA4) kernels of the PPG front-end.  Wave = 64 lanes everywhere.
//
// Built with -ffp-contract=off: every float expression below rounds exactly as written, and fused
// multiply-adds appear only where fmaf()/MFMA is spelled out.  That is what lets the integer outputs
// (keypoints, line candidates, match indices) agree bit for bit with a strict-IEEE CPU evaluation of
// the reference's formulas.
#include "ppg_kernels.h"
#include "conv_split.h"
#include "conv_store.h"

#include <math.h>
#include <algorithm>
#include <type_traits>

namespace ppg {


// =====================================================================================
// Convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: each product is one fmaf into
// the accumulator, so the result is a k-ordered fmaf chain).
//
// Reference op: every `torch._convolution` of net/*.pt (3x3 stride 1 pad 1, or 1x1), followed by
// the ReLU / max_pool2d(2,2) / batch_norm+ReLU+pixel_shuffle(2) that the graphs apply next
// (net/Backbone.pt code/__torch__/model/nets/backbone.py:28-37,
//  net/EdgeHeatmap.pt .../heatmap_decoder.py:19-22).
//
// Work-group = 4 waves, output tile 8 rows x 32 columns x NB couts (NB = 32*NT).
// Wave w owns rows 2w, 2w+1 (two 32-pixel tiles) x NT 32-cout tiles.
//   A operand (lane l): cout = l&31,                 cin = 4*(l>>5)+s for the s-th MFMA of a group
//   B operand (lane l): pixel x = l&31 of the row,   same cin
//   D: lane holds pixel l&31; register r holds cout (r&3) + 8*(r>>2) + 4*(l>>5)
// so a lane stores runs of four consecutive couts (16 bytes of the channels-last output), and the 2x2
// max-pool is: vertical = the two pixel tiles, horizontal = the neighbouring lane (one DPP move).
// Input is streamed through LDS in chunks of KC input channels (pixel stride KC+4 floats keeps the
// 16-byte fragment reads of a 16-lane group on distinct banks); weights arrive pre-packed in
// exactly the LDS image order.  One LDS buffer (39 KiB, so three work-groups share a CU): the next chunk is
// fetched into registers by branch-free buffer loads before the MFMAs of the current one (padding pixels
// and slots past the tile carry an offset outside the buffer and load zeros), and written to LDS between
// two barriers after them.  Straight-line staging matters: with per-load branches the compiler guards the
// register reuse with vmcnt waits that stall every wave on its own prefetch.
// =====================================================================================
// Two tile shapes (MW = width of one 32-pixel M tile): MW = 32 -> 8 rows x 32 columns per work-group (an M
// tile is one row segment), MW = 16 -> 16 x 16 (an M tile is 2 rows x 16 columns), which wastes less of
// feature maps whose width is a multiple of 16 but not of 32 (e.g. 80 columns at 640/8).
// MT = pixel tiles per wave.  MT = 1 halves the tile (4 x 32 or 8 x 16 pixels); with HALF the work-group also takes
// only 32 of the 64 couts of a packed weight block (NT = 1): a quarter of the work per work-group, for launches that
// would otherwise leave most of the chip idle (one frame per call: 20 to 300 big tiles on 256 CUs).  The chain of
// every output is the same in all shapes.
#ifdef PPG_CONV_TIMING
// microbenchmark builds only (tools/ubench/conv_bench.hip): per work-group ticks (10 ns) of prologue, main loop and
// epilogue of the convolution kernels, summed, and the number of work-groups
__device__ unsigned long long* g_conv_timing = nullptr;
#endif
template <int KS, int NT, int KC, bool FUSE1A = false, int MW = 32, int MT = 2, bool HALF = false>
__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(ConvArgs a) {
    static_assert(!HALF || NT == 1, "HALF takes one 32-cout tile out of a 64-cout block");
    constexpr int TW = MW, TH = 128 * MT / MW;
    constexpr int PAD = KS / 2;
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1, IN_PIX = IH * IW;
    constexpr int PSTR = KC + 4;
    constexpr int NB = 32 * NT;
    constexpr int TAPS = KS * KS;
    constexpr int W_FLOATS = TAPS * KC * NB;
    constexpr int QPP = KC / 4;
    constexpr int IN_ITEMS = IN_PIX * QPP;
    constexpr int IN_PER_T = (IN_ITEMS + 255) / 256;
    constexpr int W_ITEMS = W_FLOATS / 4;
    constexpr int W_PER_T = (W_ITEMS + 255) / 256;
    // every thread fills IN_PER_T input slots and W_PER_T weight slots without a branch: the slots past the end
    // of the tile land in padding
    constexpr int IN_FLOATS_P = (IN_PER_T * 256 / QPP) * PSTR;
    constexpr int W_FLOATS_P = W_PER_T * 256 * 4;
    constexpr int BUF = IN_FLOATS_P + W_FLOATS_P;  // one staging buffer: three work-groups per CU
    // FUSE1A: the input of this layer is conv1a(image) and is computed on the fly from the u8 patch of
    // the tile (two halo rings) instead of being read from memory: patch as floats + conv1a weights.
    constexpr int PH = IH + 2, PW = IW + 2;
    constexpr int FUSE_FLOATS = FUSE1A ? (PH * PW + 9 * 64 + 64) : 0;
    __shared__ __attribute__((aligned(16))) float smem[BUF + FUSE_FLOATS];
    float* patch = smem + BUF;        // [PH][PW]
    float* w1s = patch + PH * PW;            // [9][64]
    float* b1s = w1s + 9 * 64;               // [64]

#ifdef PPG_CONV_TIMING
    long long tm0 = wall_clock64();
#endif
    if (FUSE1A) {  // conv1a's weights and bias stay in LDS for all tiles (the barrier after the patch load covers them)
        for (int i = threadIdx.x; i < 9 * 64; i += 256) w1s[i] = a.w1a[i];
        if (threadIdx.x < 64) b1s[threadIdx.x] = a.b1a[threadIdx.x];
    }
    // A launch with fewer work-groups than tiles hands the tiles out through a counter (tile_counter[2 * cout block]):
    // a work-group that becomes resident late, because another kernel held its slot, then simply takes fewer tiles.  The
    // index of the next tile is fetched at the start of the current one.  The last work-group to run dry (counted in
    // tile_counter[2 * cout block + 1]) puts both counters back to zero for the next launch.
    __shared__ int next_tile;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    int bx = blockIdx.x, nby_ = blockIdx.y;
    if (a.xcd_group > 0) {   // (never together with the tile counter)
        const int k = blockIdx.x >> 3, r = k / a.xcd_group;
        nby_ = (r / a.xcd_tiles8) * a.xcd_group + k % a.xcd_group;
        bx = a.xcd_band ? (blockIdx.x & 7) * a.xcd_tiles8 + r % a.xcd_tiles8 : (r % a.xcd_tiles8) * 8 + (blockIdx.x & 7);
        if (bx >= ntiles) return;
    }
    int* tctr = a.tile_counter ? a.tile_counter + 2 * nby_ : nullptr;
    if (tctr) {
        if (threadIdx.x == 0) next_tile = atomicAdd(tctr, 1);
        __syncthreads();
        bx = next_tile;
    }
    while (bx < ntiles) {
    int fetched = ntiles;
    if (tctr && threadIdx.x == 0) fetched = atomicAdd(tctr, 1);
    // (the thread index is made opaque per tile: otherwise every tile-invariant offset is hoisted out of the loop and
    // held in registers through it, which spills)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
#ifdef PPG_CONV_TIMING
    tm0 = wall_clock64();
#endif
    const int tx = bx % a.tiles_x, ty = (bx / a.tiles_x) % a.tiles_y, b = bx / (a.tiles_x * a.tiles_y);
    const int nb = nby_;
    const int x0 = tx * TW, y0 = ty * TH;
    const int nchunks = a.Cin / KC;
    // pixel (row, column) inside the tile of A-operand lane li / accumulator row index i of M tile m
    auto prow = [&](int m, int i) { return MW == 32 ? MT * wave + m : 2 * MT * wave + 2 * m + (i >> 4); };
    auto pcol = [&](int i) { return MW == 32 ? i : (i & 15); };

    // Byte offset of each fetch slot inside the image (FUSE1A: float offset inside the patch).  Padding pixels
    // and slots past the tile get an offset no buffer contains, for which a buffer load returns zeros.
    constexpr unsigned NOWHERE = 0xfffffff0u;
    unsigned gofs[IN_PER_T];
    int lofs[IN_PER_T];
#pragma unroll
    for (int k = 0; k < IN_PER_T; k++) {
        const int item = tid + 256 * k;
        // slot -> (pixel, channel quad).  FUSE1A (QPP = 2): the four lanes of an MFMA 4x4 block share a quad
        // and own four different pixels.
        const int pix = FUSE1A ? 128 * k + (tid >> 3) * 4 + (tid & 3) : item / QPP;
        const int q = FUSE1A ? (tid >> 2) & 1 : item % QPP;
        const int py = pix / IW, px = pix % IW;
        const int gy = y0 + py - PAD, gx = x0 + px - PAD;
        lofs[k] = pix * PSTR + q * 4;
        gofs[k] = NOWHERE;
        if (pix < IN_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            gofs[k] = FUSE1A ? (unsigned)(py * PW + px) : (unsigned)(((gy * a.W + gx) * a.Cin + q * 4) * 4);
    }
    int pofs[IN_PER_T];  // FUSE1A: patch offset of the slot's pixel, 0 for the slots that produce zeros
#pragma unroll
    for (int k = 0; k < IN_PER_T; k++) pofs[k] = FUSE1A && gofs[k] != NOWHERE ? (int)gofs[k] : 0;
    // image b of the input and the packed weights of cout block nb, as buffers
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (FUSE1A ? 0 : (size_t)b * a.H * a.W * a.Cin)), 0, FUSE1A ? 0 : a.H * a.W * a.Cin * 4, 0x00020000);
    // HALF: the weights are packed for 64-cout blocks, [group][tap][cin half][64 couts][4]; this work-group's 32 couts
    // are every other run of 32 x 16 bytes
    constexpr int SRC_W = HALF ? 2 * W_FLOATS : W_FLOATS;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.wpk + (size_t)(HALF ? nb >> 1 : nb) * nchunks * SRC_W), 0, nchunks * SRC_W * 4, 0x00020000);
    int wofs[W_PER_T];
#pragma unroll
    for (int k = 0; k < W_PER_T; k++) {
        const int it = tid + 256 * k;
        wofs[k] = HALF ? (((it >> 5) * 64 + (nb & 1) * 32 + (it & 31)) * 16) : it * 16;
    }
    if (FUSE1A) {
        // patch pixel (py, px) = image pixel (y0 - 2 + py, x0 - 2 + px); u8 -> f32 by a true division
        // (feature/src/PPGExtractor.cpp:151), zero outside the image (conv1a's own zero padding)
        const uint8_t* im = a.img + (size_t)b * a.H * a.W;
        for (int i = tid; i < PH * PW; i += 256) {
            const int gy = y0 - 2 + i / PW, gx = x0 - 2 + i % PW;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = (float)im[(size_t)gy * a.W + gx] / 255.0f;
            patch[i] = v;
        }
        __syncthreads();
    }

    f32x4 vin[IN_PER_T], vw[W_PER_T];
    auto fetch = [&](int c) {
        // slots past the chunk read the next chunk's weights (or zeros past the last one) into padding
#pragma unroll
        for (int k = 0; k < W_PER_T; k++)
            vw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wofs[k], c * SRC_W * 4, 0));
        if (!FUSE1A) {
#pragma unroll
            for (int k = 0; k < IN_PER_T; k++)
                vin[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, gofs[k], c * KC * 4, 0));
        } else {
            // conv1a + ReLU for 4 channels of one pixel: bias, then the nine taps in (ky, kx) order, each one
            // fused multiply-add (backbone.py:28) — the same chain as conv1a_kernel, evaluated by
            // v_mfma_f32_4x4x1 (16 independent 4x4 outer products per instruction: rows = the quad's four
            // channels, columns = the block's four pixels), which costs an eighth of the VALU slots.
            // Slots that produce zeros (padding of this layer, slots past the tile) run on patch offset 0.
            const int ch = c * KC + ((tid >> 2) & 1) * 4;
            const f32x4 bias = *(const f32x4*)(b1s + ch);
            f32x4 o[IN_PER_T];
#pragma unroll
            for (int k = 0; k < IN_PER_T; k++) o[k] = bias;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const float wa = w1s[t * 64 + ch + (tid & 3)];
#pragma unroll
                for (int k = 0; k < IN_PER_T; k++)
                    o[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa, patch[pofs[k] + (t / 3) * PW + (t % 3)], o[k], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < IN_PER_T; k++) {
                const bool in = gofs[k] != NOWHERE;
#pragma unroll
                for (int e = 0; e < 4; e++) vin[k][e] = in ? fmaxf(o[k][e], 0.f) : 0.f;
            }
        }
    };
    auto stash = [&]() {
        float* ws = smem + IN_FLOATS_P;
#pragma unroll
        for (int k = 0; k < IN_PER_T; k++) *(f32x4*)(smem + lofs[k]) = vin[k];
#pragma unroll
        for (int k = 0; k < W_PER_T; k++) *(f32x4*)(ws + (tid + 256 * k) * 4) = vw[k];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    fetch(0);
    stash();
    __syncthreads();
#ifdef PPG_CONV_TIMING
    const long long tm1 = wall_clock64();
#endif
    const float* ins = smem;
    const float* ws = smem + IN_FLOATS_P;
    for (int c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks) fetch(c + 1);
#pragma unroll
        for (int g = 0; g < KC / 8; g++) {
#pragma unroll
            for (int tap = 0; tap < TAPS; tap++) {
                const int ky = tap / KS, kx = tap % KS;
                f32x4 av[MT], bv[NT];
#pragma unroll
                for (int m = 0; m < MT; m++)
                    av[m] = *(const f32x4*)(ins + ((prow(m, li) + ky) * IW + (pcol(li) + kx)) * PSTR + g * 8 + lh * 4);
#pragma unroll
                for (int n = 0; n < NT; n++)
                    bv[n] = *(const f32x4*)(ws + (((g * TAPS + tap) * 2 + lh) * NB + n * 32 + li) * 4);
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[n][s], av[m][s], acc[m][n], 0, 0, 0);
            }
        }
        // everybody must be done reading the chunk before it is overwritten
        __syncthreads();
        if (c + 1 < nchunks) stash();
        __syncthreads();
    }

#ifdef PPG_CONV_TIMING
    const long long tm2 = wall_clock64();
#endif
    conv_store<NT, MW, MT, HALF>(a, acc, wave, li, lh, x0, y0, b, nb);
#ifdef PPG_CONV_TIMING
    if (tid == 0 && g_conv_timing) {
        const long long tm3 = wall_clock64();
        atomicAdd(&g_conv_timing[0], (unsigned long long)(tm1 - tm0));
        atomicAdd(&g_conv_timing[1], (unsigned long long)(tm2 - tm1));
        atomicAdd(&g_conv_timing[2], (unsigned long long)(tm3 - tm2));
        atomicAdd(&g_conv_timing[3], 1ull);
    }
#endif
    if (tctr) {
        if (threadIdx.x == 0) next_tile = fetched;
        __syncthreads();
        bx = next_tile;
    } else {
        bx += gridDim.x;
    }
    }
    if (tctr && threadIdx.x == 0 && atomicAdd(tctr + 1, 1) == (int)gridDim.x - 1) {
        atomicExch(tctr, 0);
        atomicExch(tctr + 1, 0);
    }
}

// =====================================================================================
// The same 3x3 convolution on v_mfma_f32_32x32x16_bf16 for the two reduced-precision modes:
//   TERMS = 3  every fp32 operand x is split into bf16 hi = rne(x) and lo = rne(x - hi), and the product is
//              evaluated as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi with fp32 accumulation (about 16 bits of each operand;
//              three MFMAs that run 16x faster than the fp32 one);
//   TERMS = 1  plain bf16 operands (w_hi*x_hi only).
// Activations stay fp32 in memory: the split happens once per staged element on its way into LDS (weights are split
// on the host when they are packed), so every layer reads and writes exactly what the fp32 kernel does and the two
// kinds of kernel can be mixed layer by layer.
//
// Same work-group tile, wave layout, D layout and epilogue as conv_mfma_kernel.  A chunk is 16 input channels = the K
// of one MFMA: lanes 0-31 supply channels 0-7 of their pixel (B operand) / cout (A operand), lanes 32-63 channels
// 8-15.  LDS image of a pixel: [half 0: 8 hi | 8 lo][half 1: 8 hi | 8 lo] + 16 bytes of padding (80-byte stride: the
// 16-byte fragment reads of 16 consecutive pixels fall on distinct banks); weights [tap][half][hi|lo][cout][8 bf16].
// =====================================================================================
// Timing experiments of tools/ubench/conv_bench.hip on conv_bf16_kernel (the results are wrong when any bit is set):
// 1 / 2 = one LDS operand read serves both pixel tiles / both cout tiles, 4 = no hi/lo split arithmetic (the staged
// registers go to LDS as they are), 8 = no input fetch after the first chunk.
#ifndef PPG_EXP_LDS
#define PPG_EXP_LDS 0
#endif
// KS = 1: a 1x1 layer has one tap per chunk, too little work between two barriers, so G chunks of 16 channels are staged
// and consumed together (G = 2: 45 KiB of LDS).
template <int NT, bool FUSE1A, int MW, int TERMS, bool F16 = false, int KS = 3, int G = 1>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(ConvArgs a) {
    static_assert(!FUSE1A || (KS == 3 && G == 1), "the fused first layer is 3x3");
    constexpr int KC = 16 * G, PAD = KS / 2, TAPS = KS * KS;
    constexpr int TW = MW, TH = 256 / MW;
    constexpr int IH = TH + KS - 1, IW = TW + KS - 1, IN_PIX = IH * IW;
    constexpr int PSTR = 64 * G + 16;  // bytes per staged pixel
    constexpr int NB = 32 * NT;
    constexpr int IPP = 2 * G;            // items per pixel; item = 8 channels of a pixel
    constexpr int IN_ITEMS = IN_PIX * IPP;
    constexpr int IN_PER_T = (IN_ITEMS + 255) / 256;
    constexpr int IN_BYTES_P = (IN_PER_T * 256 / IPP) * PSTR;
    constexpr int W_BYTES = G * TAPS * 2 * 2 * NB * 16;  // one stage; the packed image is [16-channel chunk][tap][half][hi|lo][cout]
    constexpr int W_PER_T = (W_BYTES / 16 + 255) / 256;
    constexpr int W_BYTES_P = W_PER_T * 256 * 16;
    constexpr int PH = IH + 2, PW = IW + 2;
    constexpr int FUSE_FLOATS = FUSE1A ? (PH * PW + 9 * 64 + 64) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[IN_BYTES_P + W_BYTES_P + FUSE_FLOATS * 4];
    float* patch = (float*)(smem + IN_BYTES_P + W_BYTES_P);  // [PH][PW]
    float* w1s = patch + PH * PW;                             // [9][64]
    float* b1s = w1s + 9 * 64;                                // [64]

#ifdef PPG_CONV_TIMING
    const long long tm0 = wall_clock64();
#endif
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    int bx = blockIdx.x, nb = blockIdx.y;
    if (a.xcd_group > 0) {
        const int k = blockIdx.x >> 3, r = k / a.xcd_group;
        nb = (r / a.xcd_tiles8) * a.xcd_group + k % a.xcd_group;
        bx = a.xcd_band ? (blockIdx.x & 7) * a.xcd_tiles8 + r % a.xcd_tiles8 : (r % a.xcd_tiles8) * 8 + (blockIdx.x & 7);
        if (bx >= a.tiles_x * a.tiles_y * a.B) return;
    }
    const int tx = bx % a.tiles_x, ty = (bx / a.tiles_x) % a.tiles_y, b = bx / (a.tiles_x * a.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const int nchunks = a.Cin / KC;
    auto prow = [&](int m, int i) { return MW == 32 ? 2 * wave + m : 4 * wave + 2 * m + (i >> 4); };
    auto pcol = [&](int i) { return MW == 32 ? i : (i & 15); };

    // fetch slots: (pixel, half).  Padding pixels and slots past the tile carry an offset outside the buffer (it stays
    // outside when 16 is added for the second load), for which a buffer load returns zeros.
    constexpr unsigned NOWHERE = 0x80000000u;
    unsigned gofs[IN_PER_T];
    int lofs[IN_PER_T];
#pragma unroll
    for (int k = 0; k < IN_PER_T; k++) {
        const int item = tid + 256 * k;
        // FUSE1A: the four lanes of an MFMA 4x4 block share a half and own four different pixels
        const int pix = FUSE1A ? 128 * k + (tid >> 3) * 4 + (tid & 3) : item / IPP;
        const int hh = FUSE1A ? (tid >> 2) & 1 : item % IPP;  // which 8 channels of the stage
        const int py = pix / IW, px = pix % IW;
        const int gy = y0 + py - PAD, gx = x0 + px - PAD;
        lofs[k] = pix * PSTR + hh * 32;
        gofs[k] = NOWHERE;
        if (pix < IN_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            gofs[k] = FUSE1A ? (unsigned)(py * PW + px) : (unsigned)(((gy * a.W + gx) * a.Cin + hh * 8) * 4);
    }
    int pofs[IN_PER_T];
#pragma unroll
    for (int k = 0; k < IN_PER_T; k++) pofs[k] = FUSE1A && gofs[k] != NOWHERE ? (int)gofs[k] : 0;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in + (FUSE1A ? 0 : (size_t)b * a.H * a.W * a.Cin)), 0, FUSE1A ? 0 : a.H * a.W * a.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned char*)a.wpk + (size_t)nb * nchunks * W_BYTES), 0, nchunks * W_BYTES, 0x00020000);
    if (FUSE1A) {
        const uint8_t* im = a.img + (size_t)b * a.H * a.W;
        for (int i = tid; i < PH * PW; i += 256) {
            const int gy = y0 - 2 + i / PW, gx = x0 - 2 + i % PW;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = (float)im[(size_t)gy * a.W + gx] / 255.0f;
            patch[i] = v;
        }
        for (int i = tid; i < 9 * 64; i += 256) w1s[i] = a.w1a[i];
        if (tid < 64) b1s[tid] = a.b1a[tid];
        __syncthreads();
    }

    f32x4 vin[IN_PER_T][2];
    u32x4 vw[W_PER_T];
    bool saturated = false;
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < W_PER_T; k++)
            vw[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + 256 * k) * 16, c * W_BYTES, 0));
        if (!FUSE1A) {
#pragma unroll
            for (int k = 0; k < IN_PER_T; k++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if ((PPG_EXP_LDS & 8) && c) continue;
                    vin[k][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, gofs[k] + 16 * q, c * KC * 4, 0));
                }
        } else {
            // conv1a + ReLU in fp32 for the slot's 8 channels (two quads), as in conv_mfma_kernel
            const int ch = c * KC + ((tid >> 2) & 1) * 8;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f32x4 bias = *(const f32x4*)(b1s + ch + 4 * q);
                f32x4 o[IN_PER_T];
#pragma unroll
                for (int k = 0; k < IN_PER_T; k++) o[k] = bias;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const float wa = w1s[t * 64 + ch + 4 * q + (tid & 3)];
#pragma unroll
                    for (int k = 0; k < IN_PER_T; k++)
                        o[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa, patch[pofs[k] + (t / 3) * PW + (t % 3)], o[k], 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < IN_PER_T; k++) {
                    const bool in = gofs[k] != NOWHERE;
#pragma unroll
                    for (int e = 0; e < 4; e++) vin[k][q][e] = in ? fmaxf(o[k][e], 0.f) : 0.f;
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < IN_PER_T; k++) {
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                unsigned h0, l0, h1, l1;
#if PPG_EXP_LDS & 4
                if (true) {
                    h0 = __float_as_uint(vin[k][q][0]); l0 = __float_as_uint(vin[k][q][1]);
                    h1 = __float_as_uint(vin[k][q][2]); l1 = __float_as_uint(vin[k][q][3]);
                } else
#endif
                if (F16) {
                    split_f16(vin[k][q][0], vin[k][q][1], h0, l0, saturated);
                    split_f16(vin[k][q][2], vin[k][q][3], h1, l1, saturated);
                } else {
                    split_bf16(vin[k][q][0], vin[k][q][1], h0, l0);
                    split_bf16(vin[k][q][2], vin[k][q][3], h1, l1);
                }
                hi[2 * q] = h0; hi[2 * q + 1] = h1;
                lo[2 * q] = l0; lo[2 * q + 1] = l1;
            }
            *(u32x4*)(smem + lofs[k]) = hi;
            if (TERMS == 3) *(u32x4*)(smem + lofs[k] + 16) = lo;
        }
#pragma unroll
        for (int k = 0; k < W_PER_T; k++) *(u32x4*)(smem + IN_BYTES_P + (tid + 256 * k) * 16) = vw[k];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    fetch(0);
    stash();
    __syncthreads();
#ifdef PPG_CONV_TIMING
    const long long tm1 = wall_clock64();
#endif
    const unsigned char* ins = smem;
    const unsigned char* ws = smem + IN_BYTES_P;
    auto mma = [](bf16x8 wv, bf16x8 xv, f32x16 c) {
        if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv), __builtin_bit_cast(f16x8, xv), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, c, 0, 0, 0);
    };
    for (int c = 0; c < nchunks; c++) {
        if (c + 1 < nchunks) fetch(c + 1);
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int tap = 0; tap < TAPS; tap++) {
                const int ky = tap / KS, kx = tap % KS;
                bf16x8 xh[2], xl[2], wh[NT], wl[NT];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    if ((PPG_EXP_LDS & 1) && m) { xh[m] = xh[0]; xl[m] = xl[0]; continue; }
                    const unsigned char* p = ins + ((prow(m, li) + ky) * IW + (pcol(li) + kx)) * PSTR + g * 64 + lh * 32;
                    xh[m] = *(const bf16x8*)p;
                    if (TERMS == 3) xl[m] = *(const bf16x8*)(p + 16);
                }
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    if ((PPG_EXP_LDS & 2) && n) { wh[n] = wh[0]; wl[n] = wl[0]; continue; }
                    const unsigned char* p = ws + ((((g * TAPS + tap) * 2 + lh) * 2) * NB + n * 32 + li) * 16;
                    wh[n] = *(const bf16x8*)p;
                    if (TERMS == 3) wl[n] = *(const bf16x8*)(p + NB * 16);
                }
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        if (TERMS == 3) {
                            acc[m][n] = mma(wl[n], xh[m], acc[m][n]);
                            acc[m][n] = mma(wh[n], xl[m], acc[m][n]);
                        }
                        acc[m][n] = mma(wh[n], xh[m], acc[m][n]);
                    }
            }
        }
        __syncthreads();
        if (c + 1 < nchunks) stash();
        __syncthreads();
    }
#ifdef PPG_CONV_TIMING
    const long long tm2 = wall_clock64();
#endif
    if (F16) {
        if (saturated && a.sat) atomicOr(a.sat, 1);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[m][n][r] *= kF16AccUnscale;
    }
    conv_store<NT, MW>(a, acc, wave, li, lh, x0, y0, b, nb);
#ifdef PPG_CONV_TIMING
    if (tid == 0 && g_conv_timing) {
        const long long tm3 = wall_clock64();
        atomicAdd(&g_conv_timing[0], (unsigned long long)(tm1 - tm0));
        atomicAdd(&g_conv_timing[1], (unsigned long long)(tm2 - tm1));
        atomicAdd(&g_conv_timing[2], (unsigned long long)(tm3 - tm2));
        atomicAdd(&g_conv_timing[3], 1ull);
    }
#endif
}

int conv_chunk(int ks) { return ks == 3 ? 8 : 16; }

hipError_t launch_conv(const ConvArgs& args, int ks, int nt, hipStream_t s) {
    if (args.ws) return (ks == 3 && nt == 2) ? launch_conv_ws(args, s) : hipErrorInvalidValue;
    ConvArgs a = args;
    dim3 grid(a.tiles_x * a.tiles_y * a.B, a.nblk), block(256);
    // fp32 kernels: at most `persist` work-groups per cout block, each walking the tiles with that stride
    if (a.persist > 0 && a.prec == 0 && a.tile_counter && (int)grid.x > a.persist) grid.x = a.persist;
    if ((int)grid.x == a.tiles_x * a.tiles_y * a.B) a.tile_counter = nullptr;  // one work-group per tile: no counter
    // XCD-aware order of the work-groups (see ConvArgs::xcd_group) for launches of one work-group per tile
    const int nby = a.quarter ? 2 * a.nblk : a.nblk;
    const int want = a.xcd_group;
    a.xcd_group = 0;
    if (want > 0 && !a.tile_counter) {
        int G = std::min(nby, want);
        while (nby % G) G--;   // the largest group size up to the asked one that divides the cout blocks
        a.xcd_group = G;
        a.xcd_tiles8 = (int)(((long)a.tiles_x * a.tiles_y * a.B + 7) / 8);
        grid.x = (unsigned)((long)a.xcd_tiles8 * 8 * nby);
        grid.y = 1;
    }
    if (a.quarter) {
        // quarter-size work-groups: fp32 3x3 layers with 64-cout blocks only; tiles_x / tiles_y count 4 x 32 or 8 x 16 tiles
        if (ks != 3 || nt != 2 || a.img || a.prec != 0 || a.ngroups < 1 || a.ngroups > 3) return hipErrorInvalidValue;
        if ((long)a.H * a.W * a.Cin * 4 > 0x7fffffffL) return hipErrorInvalidValue;
        grid.y = a.xcd_group ? 1 : 2 * a.nblk;
        if (a.mw == 16) hipLaunchKernelGGL((conv_mfma_kernel<3, 1, 8, false, 16, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_mfma_kernel<3, 1, 8, false, 32, 1, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.ngroups < 1 || a.ngroups > 3) return hipErrorInvalidValue;
    // one image of the input is addressed as a buffer with 32-bit byte offsets
    if ((long)a.H * a.W * a.Cin * 4 > 0x7fffffffL) return hipErrorInvalidValue;
    if (a.img && (ks != 3 || nt != 2 || a.Cin != 64 || a.mw != 32)) return hipErrorInvalidValue;
    if (a.prec != 0) {
        // reduced-precision modes exist for the 3x3 layers whose input has a multiple of 16 channels (all but conv1a)
        // and the 1x1 layers whose input has a multiple of 32 (convPb, convDb)
        if ((ks != 1 && ks != 3) || (nt != 1 && nt != 2) || a.Cin % 16 || a.prec < 1 || a.prec > 3) return hipErrorInvalidValue;
        if (ks == 1) {
            // 1x1 layers: two 16-channel chunks per stage, 8 x 32 tile
            if (a.img || a.mw != 32 || a.Cin % 32) return hipErrorInvalidValue;
            if (nt == 2) {
                if (a.prec == 1) hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 3, false, 1, 2>), grid, block, 0, s, a);
                else if (a.prec == 2) hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 1, false, 1, 2>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 3, true, 1, 2>), grid, block, 0, s, a);
            } else {
                if (a.prec == 1) hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 3, false, 1, 2>), grid, block, 0, s, a);
                else if (a.prec == 2) hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 1, false, 1, 2>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 3, true, 1, 2>), grid, block, 0, s, a);
            }
            return hipGetLastError();
        }
        if (nt == 1) {
            // 32-cout blocks (the 16 -> 16 EdgeHeatmap block): 8 x 32 tile only
            if (a.img || a.mw != 32) return hipErrorInvalidValue;
            if (a.prec == 1) hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 3>), grid, block, 0, s, a);
            else if (a.prec == 2) hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 1>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv_bf16_kernel<1, false, 32, 3, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        if (a.prec == 1) {
            if (a.img) hipLaunchKernelGGL((conv_bf16_kernel<2, true, 32, 3>), grid, block, 0, s, a);
            else if (a.mw == 16) hipLaunchKernelGGL((conv_bf16_kernel<2, false, 16, 3>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 3>), grid, block, 0, s, a);
        } else if (a.prec == 2) {
            if (a.img) hipLaunchKernelGGL((conv_bf16_kernel<2, true, 32, 1>), grid, block, 0, s, a);
            else if (a.mw == 16) hipLaunchKernelGGL((conv_bf16_kernel<2, false, 16, 1>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 1>), grid, block, 0, s, a);
        } else {
            if (a.img) hipLaunchKernelGGL((conv_bf16_kernel<2, true, 32, 3, true>), grid, block, 0, s, a);
            else if (a.mw == 16) hipLaunchKernelGGL((conv_bf16_kernel<2, false, 16, 3, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv_bf16_kernel<2, false, 32, 3, true>), grid, block, 0, s, a);
        }
        return hipGetLastError();
    }
    if (ks == 3 && nt == 2 && a.img) hipLaunchKernelGGL((conv_mfma_kernel<3, 2, 8, true>), grid, block, 0, s, a);
    else if (ks == 3 && nt == 2 && a.mw == 16) hipLaunchKernelGGL((conv_mfma_kernel<3, 2, 8, false, 16>), grid, block, 0, s, a);
    else if (ks == 3 && nt == 2) hipLaunchKernelGGL((conv_mfma_kernel<3, 2, 8>), grid, block, 0, s, a);
    else if (ks == 3 && nt == 1) hipLaunchKernelGGL((conv_mfma_kernel<3, 1, 8>), grid, block, 0, s, a);
    else if (ks == 1 && nt == 2) hipLaunchKernelGGL((conv_mfma_kernel<1, 2, 16>), grid, block, 0, s, a);
    else if (ks == 1 && nt == 1) hipLaunchKernelGGL((conv_mfma_kernel<1, 1, 16>), grid, block, 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// =====================================================================================
// conv1a: uint8 image -> /255 (true division, feature/src/PPGExtractor.cpp:151) -> 3x3 conv 1->64
// + ReLU (backbone.py:28).  K = 9 is too shallow for MFMA; this is a store-bound VALU kernel.
// 16 threads per pixel, 4 couts each, so one wave store covers 1 KiB of contiguous NHWC output.
// =====================================================================================
__global__ __launch_bounds__(256) void conv1a_kernel(Conv1aArgs a) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long pix = gid >> 4;
    const int cg = (int)(gid & 15);
    const long total = (long)a.B * a.H * a.W;
    if (pix >= total) return;
    const int x = (int)(pix % a.W), y = (int)((pix / a.W) % a.H);
    const int b = (int)(pix / ((long)a.W * a.H));
    const uint8_t* im = a.img + (size_t)b * a.H * a.W;
    f32x4 acc = *(const f32x4*)(a.bias + cg * 4);
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            float v = 0.f;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) v = lut[im[(size_t)yy * a.W + xx]];
            const f32x4 w = *(const f32x4*)(a.w + (ky * 3 + kx) * 64 + cg * 4);
            acc[0] = fmaf(v, w[0], acc[0]);
            acc[1] = fmaf(v, w[1], acc[1]);
            acc[2] = fmaf(v, w[2], acc[2]);
            acc[3] = fmaf(v, w[3], acc[3]);
        }
    acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f);
    acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f);
    *(f32x4*)(a.out + (size_t)pix * 64 + cg * 4) = acc;
}

hipError_t launch_conv1a(const Conv1aArgs& a, hipStream_t s) {
    const long threads = (long)a.B * a.H * a.W * 16;
    hipLaunchKernelGGL(conv1a_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// Image upload by a kernel: the source is pinned host memory that the GPU reads over PCIe, 16 bytes per lane.
// The copy engines serve a context's result copies in order, and those wait for kernels far down the pipeline;
// an upload queued behind them would not start before the previous batch is finished.  A kernel on its own stream
// has no such queue in front of it.
// =====================================================================================
typedef unsigned up_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void upload_kernel(const up_u32x4* __restrict__ src, up_u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_upload(const void* host_src, void* dst, size_t bytes, hipStream_t s) {
    if (bytes % 16 || ((uintptr_t)host_src | (uintptr_t)dst) % 16) return hipErrorInvalidValue;
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 128);
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, s, (const up_u32x4*)host_src, (up_u32x4*)dst, n16);
    return hipGetLastError();
}

// =====================================================================================
// Wave helpers
// =====================================================================================
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// =====================================================================================
// Score map: softmax over the 65 junction channels, drop the dustbin, depth-to-space x8
// (feature/src/PPGExtractor.cpp:161-162).  One wave per 8x8 cell, lane = channel.
// =====================================================================================
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
    part[tid] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) { const int c = rowcnt[r]; rowcnt[r] = run; run += c; }
    }
    if (tid == 1023) cnt[2] = part[1023];
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
    part[tid] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;
    for (int k = 0; k < per; k++) {
        const int r = tid * per + k;
        if (r < n) { const int c = arr[r]; arr[r] = run; run += c; }
    }
    const int total = part[1023];
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
    part[tid] = mine_cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int ebase = part[1023];
    {
        int e = part[tid] - mine_cnt;
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
