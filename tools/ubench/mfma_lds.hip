// Inner loop of the 3x3 convolution kernel rebuilt bottom-up: per "chunk" 9 taps x (4 ds_read_b128 + 16 MFMA),
// long-lived work-groups.  MODE bits: 1 two barriers per chunk, 2 stash 8 x ds_write_b128 between them,
// 4 the stashed registers come from 8 buffer loads issued before the MFMAs, 8 a block of 60 packed FMAs
// (the fused first layer) before the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int IW = 34, PSTR = 12, NB = 64, TAPS = 9, IN_FLOATS = 384 * 12, W_FLOATS = 5 * 1024;
template <int MODE>
__global__ __launch_bounds__(256, 3) void k(float* out, const float* src, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < IN_FLOATS + W_FLOATS; i += 256) smem[i] = (float)(i % 17) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const float* ins = smem; const float* ws = smem + IN_FLOATS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64 << 20, 0x00020000);
    f32x4 st[8];
    for (int k2 = 0; k2 < 8; k2++) st[k2] = (f32x4){1.f, 2.f, 3.f, (float)tid};
    f32x2 fa = {0.5f, 0.25f}, fb = {0.1f, 0.2f};
    for (int c = 0; c < chunks; c++) {
        if (MODE & 4) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++)
                st[k2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256 * k2) * 16, ((c * 7 + blockIdx.x) & 1023) * 32768, 0));
        }
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 30; i++) {
                fa = __builtin_elementwise_fma(fa, fb, (f32x2){ins[(tid + i) & 1023], 1.f});
                fb = __builtin_elementwise_fma(fb, fa, (f32x2){0.5f, 1.f});
            }
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; tap++) {
            const int ky = tap / 3, kx = tap % 3;
            f32x4 av[2], bv[2];
#pragma unroll
            for (int m = 0; m < 2; m++) av[m] = *(const f32x4*)(ins + ((2 * wave + m + ky) * IW + (li + kx)) * PSTR + lh * 4);
#pragma unroll
            for (int n = 0; n < 2; n++) bv[n] = *(const f32x4*)(ws + ((tap * 2 + lh) * NB + n * 32 + li) * 4);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s], bv[n][s], acc[m][n], 0, 0, 0);
        }
        if (MODE & 1) __syncthreads();
        if (MODE & 2) {
#pragma unroll
            for (int k2 = 0; k2 < 3; k2++) *(f32x4*)(smem + ((tid >> 1) + 128 * k2) * PSTR + (tid & 1) * 4) = st[k2];
#pragma unroll
            for (int k2 = 3; k2 < 8; k2++) *(f32x4*)(smem + IN_FLOATS + (tid + 256 * (k2 - 3)) * 4) = st[k2];
        }
        if (MODE & 1) __syncthreads();
        asm volatile("" ::: "memory");
    }
    float s = fa[0] + fb[1];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* d, const float* src, int wpc, int chunks, int rounds) {
    const int lds = wpc == 1 ? 100 * 1024 : wpc == 2 ? 70 * 1024 : wpc == 3 ? 45 * 1024 : 39 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * wpc * rounds;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, src, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, src, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * chunks * 144.0 * 4096.0;
    printf("mode %2d  wg/CU %d  chunks/wg %5d rounds %3d: %8.3f ms  %6.1f TFLOP/s\n", MODE, wpc, chunks, rounds, ms, flops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 4 * 64 * 256 * 4);
    float* src; hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
    for (int wpc : {2, 3}) {
        run<0>(d, src, wpc, 1000, 1);
        run<1>(d, src, wpc, 1000, 1);
        run<3>(d, src, wpc, 1000, 1);
        run<7>(d, src, wpc, 1000, 1);
        run<15>(d, src, wpc, 1000, 1);
        run<9>(d, src, wpc, 1000, 1);
        run<8>(d, src, wpc, 1000, 1);
    }
    return 0;
}
