// Epilogue shared by the convolution kernels (conv_mfma_kernel, conv_bf16_kernel in conv_kernels.hip; conv_ws_kernel in
// conv_ws.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ppg_kernels.h"

namespace ppg {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load from a 4-byte aligned address
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue of the convolution kernels: bias, (batch norm), (ReLU), then plain / pooled / pixel-shuffled store.
// acc[m][n] is the 32 x 32 result of pixel tile m and cout tile n in the MFMA D layout (lane = pixel l&31,
// register r = cout (r&3) + 8*(r>>2) + 4*(l>>5)).
//
// Pixel of accumulator column i (= lane & 31) of pixel tile m.  MW = 32: the tile is 8 x 32, a pixel tile one row.
// MW = 16: the tile is 16 x 16, a pixel tile two rows of 16; with ROT the second row's columns are rotated by two
// (lanes 16, 17 -> columns 14, 15; lanes 18..31 -> columns 0..13), which makes the operand reads of such a tile from an
// LDS image with an 18-pixel row pitch free of bank conflicts (the column <-> pixel assignment of an MFMA is free).
template <int MW, int MT, bool ROT = false>
__device__ __forceinline__ int conv_prow(int wave, int m, int i) { return MW == 32 ? MT * wave + m : 2 * MT * wave + 2 * m + (i >> 4); }
template <int MW, bool ROT = false>
__device__ __forceinline__ int conv_pcol(int i) { return MW == 32 ? i : (ROT && i >= 16 ? (i - 18) & 15 : i & 15); }

// One run of the epilogue: the four consecutive couts 8 j + 4 lh + 0..3 of cout tile n, for the lane's pixel in every
// pixel tile.  BN = the layer has a batch norm (al, be valid); SCALE = every sum is first multiplied by `scale` (an exact
// power of two: the fp16 split mode).  cb = first cout of this block inside its layer (packed order).
template <int NT, int MW, int MT, bool BN, bool SCALE = false, bool ROT = false>
__device__ __forceinline__ void conv_store_run(const ConvArgs& a, const ConvGroup& G, int cb, const f32x16 (&acc)[MT][NT], int wave, int li,
                                               int lh, int x0, int y0, int b, int n, int j, f32x4 bias, f32x4 al, f32x4 be,
                                               float scale = 1.f) {
    const bool vec = (G.Cout & 3) == 0;  // rows of the output are 16-byte aligned
    const int Cps = G.Cout >> 2;
    const int co = cb + n * 32 + 8 * j + 4 * lh;  // first of the four couts
    if (co >= G.Cout) return;
    auto val = [&](int m, int e) { return SCALE ? acc[m][n][4 * j + e] * scale : acc[m][n][4 * j + e]; };
    if (G.mode == CONV_POOL2) {
        // pooled layers use the 8 x 32 tile with two pixel tiles per wave (rows 2*wave, 2*wave+1 are the two
        // tiles) or the 8 x 16 tile with one (the tile is two rows of 16: the row neighbour is lane ^ 16);
        // the column neighbour is the next lane
        if (!((MW == 32 && MT == 2) || (MW == 16 && MT == 1))) return;
        const int y = y0 + 2 * wave, x = x0 + conv_pcol<MW>(li);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float t = MT == 2 ? fmaxf(val(0, e), val(MT - 1, e)) : fmaxf(val(0, e), __shfl_xor(val(0, e), 16));
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));  // lane ^ 1
            float r = fmaxf(t, o) + bias[e];
            if (G.relu) r = fmaxf(r, 0.f);
            v[e] = r;
        }
        if ((li & 1) == 0 && (MT == 2 || li < 16) && y + 1 < a.H && x + 1 < a.W) {
            float* dst = G.out + ((size_t)(b * (a.H / 2) + (y >> 1)) * (a.W / 2) + (x >> 1)) * G.Cout + co;
            if (vec) *(f32x4*)dst = v;
            else
                for (int e = 0; e < 4; e++)
                    if (co + e < G.Cout) dst[e] = v[e];
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int y = y0 + conv_prow<MW, MT, ROT>(wave, m, li), x = x0 + conv_pcol<MW, ROT>(li);
            if (y >= a.H || x >= a.W) continue;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float r = val(m, e) + bias[e];
                if (BN) r = r * al[e] + be[e];
                if (G.relu) r = fmaxf(r, 0.f);
                v[e] = r;
            }
            float* dst;
            if (G.mode == CONV_PLAIN) {
                dst = G.out + ((size_t)(b * a.H + y) * a.W + x) * G.Cout + co;
            } else {  // CONV_PS2: packed cout = q*Cps + c, q = 2*dy + dx; Cps is a multiple of 4
                const int q = co / Cps, cc = co - q * Cps;
                dst = G.out + ((size_t)(b * 2 * a.H + 2 * y + (q >> 1)) * (2 * a.W) + 2 * x + (q & 1)) * Cps + cc;
            }
            if (vec) *(f32x4*)dst = v;
            else
                for (int e = 0; e < 4; e++)
                    if (co + e < G.Cout) dst[e] = v[e];
        }
    }
}

// which of the launch's (up to three) layers cout block nb belongs to
__device__ __forceinline__ int conv_group_of(const ConvArgs& a, int nb) {
    int gi = 0;
    if (a.ngroups > 1 && nb >= a.g[1].nblk0) gi = 1;
    if (a.ngroups > 2 && nb >= a.g[2].nblk0) gi = 2;
    return gi;
}

template <int NT, int MW, int MT = 2, bool HALF = false, bool ROT = false>
__device__ __forceinline__ void conv_store(const ConvArgs& a, const f32x16 (&acc)[MT][NT], int wave, int li, int lh, int x0,
                                           int y0, int b, int nb_launch, const float* blk_bias = nullptr,
                                           const float* blk_alpha = nullptr, const float* blk_beta = nullptr) {
    // blk_bias / blk_alpha / blk_beta (optional): the bias and batch-norm vectors of THIS cout block, already fetched
    // (e.g. into LDS) and indexed from the block's first cout; otherwise they are read from the layer's arrays.
    constexpr int NB = 32 * NT;
    // HALF: the launch counts 32-cout halves of the 64-cout blocks the layers are described in
    const int nb = HALF ? nb_launch >> 1 : nb_launch;
    // A launch may serve up to three layers that read the same input (cout blocks [nblk0, next nblk0) each).
    // The lane owns one pixel of each pixel tile and, per cout tile, four runs of four consecutive couts
    // (registers 4j..4j+3 = couts 8j + 4*(l>>5) + 0..3): one 16-byte store per run.
    const ConvGroup G = a.g[conv_group_of(a, nb)];
    const int cb = HALF ? (nb - G.nblk0) * 64 + (nb_launch & 1) * 32 : (nb - G.nblk0) * NB;  // first cout of this block inside its layer (packed order)
    const bool bn = G.alpha != nullptr;
    // All bias vectors are fetched before the first store: a store to `out` may alias them as far as the compiler
    // knows, so loads left inside the loops would each wait out a full memory latency behind the stores before them.
    // (The bias arrays are padded to whole cout blocks.)
    f32x4 biasv[NT][4];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            biasv[n][j] = blk_bias ? *(const f32x4*)(blk_bias + n * 32 + 8 * j + 4 * lh) : *(const f32x4*)(G.bias + cb + n * 32 + 8 * j + 4 * lh);
    // No load may follow a store here: the counter a load is waited on counts the stores before it too, so a load
    // in the middle of the epilogue waits out the full write latency of everything stored so far.  Without a batch norm
    // there is none; with one (a single layer) its eight vectors per cout tile are fetched ahead of that tile's stores.
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    if (!bn) {
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                conv_store_run<NT, MW, MT, false, false, ROT>(a, G, cb, acc, wave, li, lh, x0, y0, b, n, j, biasv[n][j], one, zero);
    } else {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            f32x4 al[4], be[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int co = min(cb + n * 32 + 8 * j + 4 * lh, G.Cout - 4);  // (runs past the layer are skipped below)
                al[j] = blk_alpha ? *(const f32x4*)(blk_alpha + n * 32 + 8 * j + 4 * lh) : *(const f32x4*)(G.alpha + co);
                be[j] = blk_beta ? *(const f32x4*)(blk_beta + n * 32 + 8 * j + 4 * lh) : *(const f32x4*)(G.beta + co);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                conv_store_run<NT, MW, MT, true, false, ROT>(a, G, cb, acc, wave, li, lh, x0, y0, b, n, j, biasv[n][j], al[j], be[j]);
        }
    }
}

}  // namespace ppg
