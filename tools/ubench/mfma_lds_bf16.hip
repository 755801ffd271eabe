// Inner loop of the bf16 hi/lo-split 3x3 convolution kernel rebuilt bottom-up: per chunk 9 taps x (8 ds_read_b128 +
// 12 v_mfma_f32_32x32x16_bf16), long-lived work-groups, two per CU.  MODE bits: 1 two barriers per chunk, 2 stash
// 15 x ds_write_b128 between them, 4 the stashed registers come from 15 buffer loads issued before the MFMAs,
// 8 the bf16 split arithmetic on the staged input registers, 16 operand reads issued one tap ahead, 32 six of the
// loads gather 16-byte pieces at a 256-byte stride (the input tile of a 64-channel layer) instead of streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int IW = 34, PSTR = 80, NB = 64, IN_BYTES = 384 * 80, W_BYTES = 9 * 2 * 2 * NB * 16;
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* src, int chunks, long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[IN_BYTES + W_BYTES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = tid; i < (IN_BYTES + W_BYTES) / 4; i += 256) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        ((unsigned*)smem)[i] = (x & 0x007f007fu) | 0x3c003c00u | ((x >> 7) & 0x80008000u);
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const unsigned char* ins = smem; const unsigned char* ws = smem + IN_BYTES;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64 << 20, 0x00020000);
    f32x4 st[15];
    for (int k2 = 0; k2 < 15; k2++) st[k2] = (f32x4){1.f + tid, 2.f, 3.f * k2, (float)tid * 0.37f};
    struct Ops { bf16x8 xh[2], xl[2], wh[2], wl[2]; };
    auto load_ops = [&](int tap, Ops& o) {
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const unsigned char* p = ins + ((2 * wave + m + ky) * IW + (li + kx)) * PSTR + lh * 32;
            o.xh[m] = *(const bf16x8*)p; o.xl[m] = *(const bf16x8*)(p + 16);
        }
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const unsigned char* p = ws + (((tap * 2 + lh) * 2) * NB + n * 32 + li) * 16;
            o.wh[n] = *(const bf16x8*)p; o.wl[n] = *(const bf16x8*)(p + NB * 16);
        }
    };
    auto mfmas = [&](const Ops& o) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wl[n], o.xh[m], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[n], o.xl[m], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[n], o.xh[m], acc[m][n], 0, 0, 0);
    };
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int c = 0; c < chunks; c++) {
        if (MODE & 4) {
#pragma unroll
            for (int k2 = 0; k2 < 15; k2++) {
                unsigned off = (tid + 256 * k2) * 16;
                if ((MODE & 32) && k2 < 6) off = ((tid + 256 * (k2 >> 1)) >> 1) * 256 + (tid & 1) * 32 + (k2 & 1) * 16 + (c & 3) * 64;
                st[k2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, ((c * 7 + blockIdx.x) & 255) * 131072, 0));
            }
        }
        if (MODE & 16) {
            Ops ops[2];
            load_ops(0, ops[0]);
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                if (tap + 1 < 9) load_ops(tap + 1, ops[(tap + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(ops[tap & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                Ops o;
                load_ops(tap, o);
                mfmas(o);
            }
        }
        if (MODE & 1) __syncthreads();
        if (MODE & 2) {
#pragma unroll
            for (int k2 = 0; k2 < 3; k2++) {
                u32x4 hi = __builtin_bit_cast(u32x4, st[2 * k2]), lo = __builtin_bit_cast(u32x4, st[2 * k2 + 1]);
                if (MODE & 8) {
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        unsigned h0, l0, h1, l1;
                        split_bf16(st[2 * k2 + q][0], st[2 * k2 + q][1], h0, l0);
                        split_bf16(st[2 * k2 + q][2], st[2 * k2 + q][3], h1, l1);
                        hi[2 * q] = h0; hi[2 * q + 1] = h1; lo[2 * q] = l0; lo[2 * q + 1] = l1;
                    }
                }
                const int item = tid + 256 * k2;
                *(u32x4*)(smem + (item >> 1) * PSTR + (item & 1) * 32) = hi;
                *(u32x4*)(smem + (item >> 1) * PSTR + (item & 1) * 32 + 16) = lo;
            }
#pragma unroll
            for (int k2 = 6; k2 < 15; k2++) *(u32x4*)(smem + IN_BYTES + (tid + 256 * (k2 - 6)) * 16) = __builtin_bit_cast(u32x4, st[k2]);
        }
        if (MODE & 1) __syncthreads();
        asm volatile("" ::: "memory");
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(float* d, const float* src, int chunks, long long* dclk) {
    const int blocks = 256 * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, src, 2, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, src, chunks, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
    // algorithmic flops: a third of the executed MFMA flops
    double flops = (double)blocks * 4 * chunks * 108.0 * 32768.0 / 3.0;
    printf("mode %2d  chunks/wg %5d: %8.3f ms  %6.1f TFLOP/s (x3: %6.1f executed)  clk=%.0f MHz\n", MODE, chunks, ms,
           flops / ms / 1e9, 3 * flops / ms / 1e9, 100.0 * h[0] / h[1]);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 4 * 64 * 256 * 4);
    float* src; hipMalloc(&src, 64 << 20); hipMemset(src, 0x3c, 64 << 20);
    long long* dclk; hipMalloc(&dclk, 16);
    run<0>(d, src, 2000, dclk);
    run<16>(d, src, 2000, dclk);
    run<1>(d, src, 2000, dclk);
    run<17>(d, src, 2000, dclk);
    run<3>(d, src, 2000, dclk);
    run<19>(d, src, 2000, dclk);
    run<7>(d, src, 2000, dclk);
    run<23>(d, src, 2000, dclk);
    run<15>(d, src, 2000, dclk);
    run<31>(d, src, 2000, dclk);
    run<47>(d, src, 2000, dclk);
    run<63>(d, src, 2000, dclk);
    run<15>(d, src, 8, dclk);
    run<47>(d, src, 8, dclk);
    return 0;
}
