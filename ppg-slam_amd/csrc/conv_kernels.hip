// gfx950 (MI355X, CDNA4) kernels of the PPG front-end: the convolutions (fp32 MFMA and the reduced-precision modes), conv1a
// stand-alone, the image upload.  Wave = 64 lanes everywhere.
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

}  // namespace ppg
