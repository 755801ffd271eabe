// Inner loop of the 3x3 convolution kernel in isolation: per "chunk" 9 taps x (4 ds_read_b128 + 16 MFMA),
// no global traffic, no barriers, long-lived work-groups.  Sweeps work-groups per CU (via dynamic LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int IW = 34, PSTR = 12, NB = 64, TAPS = 9, IN_FLOATS = 10 * 34 * 12, W_FLOATS = 9 * 8 * 64;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < IN_FLOATS + W_FLOATS; i += 256) smem[i] = (float)(i % 17) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const float* ins = smem; const float* ws = smem + IN_FLOATS;
    for (int c = 0; c < chunks; c++) {
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
        if (MODE == 1) __syncthreads();
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* d, int wpc, int chunks, int rounds) {
    // LDS per work-group chosen so that exactly wpc work-groups fit a CU (160 KiB)
    const int lds = wpc == 1 ? 100 * 1024 : wpc == 2 ? 70 * 1024 : wpc == 3 ? 45 * 1024 : 36 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * wpc * rounds;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * chunks * 144.0 * 4096.0;
    printf("mode %d  wg/CU %d  chunks/wg %5d rounds %3d: %8.3f ms  %6.1f TFLOP/s\n", MODE, wpc, chunks, rounds, ms, flops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 4 * 64 * 256 * 4);
    for (int wpc : {1, 2, 3, 4}) run<0>(d, wpc, 2000, 1);
    for (int wpc : {2, 3, 4}) run<1>(d, wpc, 2000, 1);
    for (int wpc : {2, 3}) run<0>(d, wpc, 8, 25);    // conv1b-like lifetime: 8 chunks per work-group
    for (int wpc : {2, 3}) run<0>(d, wpc, 16, 12);
    for (int wpc : {2, 3}) run<0>(d, wpc, 64, 4);
    return 0;
}
