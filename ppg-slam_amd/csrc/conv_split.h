// Operand types and the hi/lo splits shared by the reduced-precision convolution kernels (conv_bf16_kernel in
// conv_kernels.hip, conv_ws_kernel in conv_ws.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ppg {

typedef float cs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The fp16 variant of the split (F16): operands are first scaled by exact powers of two (activations 2^5, weights
// 2^10 on the host) so that hi and lo stay inside fp16's narrow exponent range, and the accumulators are scaled back
// by 2^-15 before the bias.  fp16 keeps 11 bits per half instead of 8: the three products carry about 22 significant
// bits of each operand, which brings the result to within twice the rounding noise of an fp32 evaluation.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float kF16ActScale = 0x1p5f, kF16AccUnscale = 0x1p-15f;  // weights carry 2^10

__device__ __forceinline__ void split_f16(float x0, float x1, unsigned& hi, unsigned& lo, bool& sat) {
    // saturate instead of overflowing to infinity, and report it (activations beyond 2047 do not occur in these
    // networks; NaN compares false and propagates)
    x0 *= kF16ActScale; x1 *= kF16ActScale;
    sat |= fabsf(x0) > 65504.f || fabsf(x1) > 65504.f;
    x0 = fminf(fmaxf(x0, -65504.f), 65504.f);
    x1 = fminf(fmaxf(x1, -65504.f), 65504.f);
    const f16x2 h = {(_Float16)x0, (_Float16)x1};  // round to nearest even
    const f16x2 l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// two floats -> their bf16 hi parts and the bf16 of the remainders, each pair packed into one dword
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const cs_f32x2 x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const cs_f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}


}  // namespace ppg
