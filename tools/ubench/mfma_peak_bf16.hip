// Sustained bf16-MFMA rate of the chip: every SIMD issues v_mfma_f32_32x32x16_bf16 back to back from registers
// (no memory traffic), with random-looking operand bits (operand toggling is what the power limit sees).  Gives the
// practical ceiling, clock included, that the bf16 convolution kernels are measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed, long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    u32x4 ua[2], ub[2];
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 2; i++)
        for (int e = 0; e < 4; e++) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            // two bf16 values of magnitude ~1e-2 with random mantissas
            ua[i][e] = (x & 0x007f007fu) | 0x3c003c00u | ((x >> 7) & 0x80008000u);
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            ub[i][e] = (x & 0x007f007fu) | 0x3c003c00u | ((x >> 7) & 0x80008000u);
        }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua[(u + i) & 1]),
                                                                  __builtin_bit_cast(bf16x8, ub[u & 1]), acc[i], 0, 0, 0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    long long* dclk; hipMalloc(&dclk, 16); long long h[2];
    for (int wpb : {1, 2}) {   // blocks per CU
        const int blocks = 256 * wpb, iters = 40000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1u, dclk);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 7u + rep, dclk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
            double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * 32768.0;
            printf("blocks/CU %d: %.2f ms  %.1f TFLOP/s  clk=%.0f MHz\n", wpb, ms, flops / ms / 1e9, 100.0 * h[0] / h[1]);
        }
    }
    return 0;
}
