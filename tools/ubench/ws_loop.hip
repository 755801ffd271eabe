// Bottom-up models of the reduced-precision 3x3 convolution's inner loop, to choose a structure before building it:
//   A  today's structure: 256 threads, two work-groups per CU, one LDS buffer, two barriers per 16-channel chunk, every
//      wave fetches (15 x 16 bytes per thread), splits into bf16 hi/lo and stores the next chunk between the barriers;
//   B  wave-specialised: 512 threads, one work-group per CU, two LDS buffers, ONE barrier per chunk; waves 0-3 only read
//      operands and issue MFMAs (reads one tap ahead), waves 4-7 only fetch, split and store the chunk after next;
//   C  as B with the weights (already in LDS image order in memory) brought in by LDS-DMA instead of through registers;
//   D  as C on v_mfma_f32_16x16x32_bf16: K = two taps x 16 channels, wave tile still 64 pixels x 64 couts.
// Per chunk and consumer wave: 9 taps x 12 MFMAs of 32x32x16 (or 4.5 tap pairs x 48 of 16x16x32), three per product.
// Random bf16 operands (the clock a loop holds depends on the data).  Prints algorithmic TFLOP/s (a third of the
// executed MFMA flops) and the shader clock seen from inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int IW = 34, PSTR = 80, NB = 64, IN_BYTES = 384 * 80, W_BYTES = 9 * 2 * 2 * NB * 16, STAGE = IN_BYTES + W_BYTES;

__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void fill_lds(unsigned char* smem, int bytes, int nthr) {
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < bytes / 4; i += nthr) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        ((unsigned*)smem)[i] = (x & 0x007f007fu) | 0x3c003c00u | ((x >> 7) & 0x80008000u);
    }
}
struct Ops32 { bf16x8 xh[2], xl[2], wh[2], wl[2]; };
__device__ __forceinline__ void load_ops32(const unsigned char* ins, const unsigned char* ws, int wave, int li, int lh, int tap, Ops32& o) {
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
}
__device__ __forceinline__ void mfmas32(const Ops32& o, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wl[n], o.xh[m], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[n], o.xl[m], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[n], o.xh[m], acc[m][n], 0, 0, 0);
        }
}

// ---- A: today's structure ----
__global__ __launch_bounds__(256, 2) void kA(float* out, const float* src, int chunks, long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    fill_lds(smem, STAGE, 256);
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64 << 20, 0x00020000);
    f32x4 st[15];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int c = 0; c < chunks; c++) {
#pragma unroll
        for (int k2 = 0; k2 < 15; k2++)   // the input tile is this work-group's own (streamed); the weights are every work-group's (L2)
            st[k2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256 * k2) * 16,
                                        k2 < 6 ? ((c * 7 + blockIdx.x) & 255) * 131072 : (32 << 20) + (c & 7) * W_BYTES - 6 * 4096, 0));
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            Ops32 o;
            load_ops32(smem, smem + IN_BYTES, wave, li, lh, tap, o);
            mfmas32(o, acc);
        }
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 3; k2++) {
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                unsigned h0, l0, h1, l1;
                split_bf16(st[2 * k2 + q][0], st[2 * k2 + q][1], h0, l0);
                split_bf16(st[2 * k2 + q][2], st[2 * k2 + q][3], h1, l1);
                hi[2 * q] = h0; hi[2 * q + 1] = h1; lo[2 * q] = l0; lo[2 * q + 1] = l1;
            }
            const int item = tid + 256 * k2;
            *(u32x4*)(smem + (item >> 1) * PSTR + (item & 1) * 32) = hi;
            *(u32x4*)(smem + (item >> 1) * PSTR + (item & 1) * 32 + 16) = lo;
        }
#pragma unroll
        for (int k2 = 6; k2 < 15; k2++) *(u32x4*)(smem + IN_BYTES + (tid + 256 * (k2 - 6)) * 16) = __builtin_bit_cast(u32x4, st[k2]);
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- B / C / D: wave-specialised ----
// DMA: weights by LDS-DMA.  S16: consumers on v_mfma_f32_16x16x32_bf16.
template <int PROD, bool S16, bool IL>
__global__ __launch_bounds__(512, 2) void kW(float* out, const float* src, int chunks, long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    fill_lds(smem, 2 * STAGE, 512);
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    if (wave < 4) {
        // ---------------- consumers ----------------
        float s = 0;
        if (!S16) {
            const int li = lane & 31, lh = lane >> 5;
            f32x16 acc[2][2];
            for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
            for (int c = 0; c < chunks; c++) {
                const unsigned char* ins = smem + (c & 1) * STAGE;
                const unsigned char* ws = ins + IN_BYTES;
                Ops32 ops[2];
                load_ops32(ins, ws, wave, li, lh, 0, ops[0]);
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < 9) load_ops32(ins, ws, wave, li, lh, tap + 1, ops[(tap + 1) & 1]);
                    mfmas32(ops[tap & 1], acc);
                    if (IL) {   // one operand read in the shadow of each of the first eight MFMAs
                        if (tap + 1 < 9) {
#pragma unroll
                            for (int i = 0; i < 8; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        }
                    } else {
                        // (reads first, then the MFMAs)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();
            }
            for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) for (int r = 0; r < 16; r++) s += acc[m][n][r];
        } else {
            // K = 32: lanes 0-31 take tap t (channel halves 0, 1), lanes 32-63 tap t'
            const int l16 = lane & 15, q = lane >> 4, half = q & 1, second = q >> 1;
            f32x4 acc[4][4];
            for (int m = 0; m < 4; m++) for (int n = 0; n < 4; n++) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            struct Ops16 { bf16x8 xh[4], xl[4], wh[4], wl[4]; };
            auto load16 = [&](const unsigned char* ins, const unsigned char* ws, int pair, Ops16& o) {
                const int tap = min(2 * pair + second, 8);
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int m = 0; m < 4; m++) {   // pixel tiles: row 2*wave + (m >> 1), columns 16 * (m & 1) ...
                    const unsigned char* p = ins + ((2 * wave + (m >> 1) + ky) * IW + (16 * (m & 1) + l16 + kx)) * PSTR + half * 32;
                    o.xh[m] = *(const bf16x8*)p; o.xl[m] = *(const bf16x8*)(p + 16);
                }
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    const unsigned char* p = ws + (((tap * 2 + half) * 2) * NB + n * 16 + l16) * 16;
                    o.wh[n] = *(const bf16x8*)p; o.wl[n] = *(const bf16x8*)(p + NB * 16);
                }
            };
            auto mf16 = [&](const Ops16& o) {
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.wl[n], o.xh[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.wh[n], o.xl[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.wh[n], o.xh[m], acc[m][n], 0, 0, 0);
                    }
            };
            for (int c = 0; c < chunks; c += 2) {
                // two chunks = 18 taps = 9 pairs: five from the even chunk's buffer, four from the odd one's
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const unsigned char* ins = smem + h * STAGE;
                    const unsigned char* ws = ins + IN_BYTES;
                    const int np = h ? 4 : 5;
                    Ops16 ops[2];
                    load16(ins, ws, 0, ops[0]);
#pragma unroll
                    for (int p = 0; p < 5; p++) {
                        if (p >= np) break;
                        __builtin_amdgcn_sched_barrier(0);
                        if (p + 1 < np) load16(ins, ws, p + 1, ops[(p + 1) & 1]);
                        mf16(ops[p & 1]);
                        if (IL && p + 1 < np) {
#pragma unroll
                            for (int i = 0; i < 16; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_s_barrier();
                }
            }
            for (int m = 0; m < 4; m++) for (int n = 0; n < 4; n++) for (int r = 0; r < 4; r++) s += acc[m][n][r];
        }
        out[blockIdx.x * 256 + tid] = s;
    } else {
        // ---------------- producers ----------------
        constexpr bool DMA = PROD >= 1;
        const int pt = tid - 256;
        if (PROD == 3) {   // no staging at all: what the consumers reach on their own
            for (int c = 0; c < chunks; c++) __builtin_amdgcn_s_barrier();
            if (pt == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
            return;
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64 << 20, 0x00020000);
        f32x4 st[15];
        auto issue = [&](int c) {
            const int so = ((c * 7 + blockIdx.x) & 255) * 131072, sw = (32 << 20) + (c & 7) * W_BYTES - 6 * 4096;
#pragma unroll
            for (int k2 = 0; k2 < (PROD == 2 ? 0 : DMA ? 6 : 15); k2++)
                st[k2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (pt + 256 * k2) * 16, k2 < 6 ? so : sw, 0));
        };
        issue(1);
        for (int c = 0; c < chunks; c++) {
            unsigned char* dst = smem + ((c + 1) & 1) * STAGE;
            // the chunk fetched during the last iteration goes to the buffer the consumers read NEXT
            if (PROD == 2) {   // activations stored already split by their producer: the input tile by LDS-DMA as well (30 pieces per chunk)
                const int so = ((c * 7 + blockIdx.x) & 255) * 131072;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    const int piece = (wave - 4) * 8 + k2;
                    if (piece < 30) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + piece * 1024), 16, lane * 16 + piece * 1024, so, 0, 0);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < (PROD == 2 ? 0 : 3); k2++) {
                u32x4 hi, lo;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    unsigned h0, l0, h1, l1;
                    split_bf16(st[2 * k2 + q][0], st[2 * k2 + q][1], h0, l0);
                    split_bf16(st[2 * k2 + q][2], st[2 * k2 + q][3], h1, l1);
                    hi[2 * q] = h0; hi[2 * q + 1] = h1; lo[2 * q] = l0; lo[2 * q + 1] = l1;
                }
                const int item = pt + 256 * k2;
                *(u32x4*)(dst + (item >> 1) * PSTR + (item & 1) * 32) = hi;
                *(u32x4*)(dst + (item >> 1) * PSTR + (item & 1) * 32 + 16) = lo;
            }
            if (DMA) {
                // 36 KiB of weights: 9 pieces of 1 KiB per wave, straight into the buffer
                const int so = (32 << 20) + ((c + 1) & 7) * W_BYTES;
#pragma unroll
                for (int k2 = 0; k2 < 9; k2++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + IN_BYTES + ((wave - 4) * 9 + k2) * 1024), 16, lane * 16 + ((wave - 4) * 9 + k2) * 1024, so, 0, 0);
            } else {
#pragma unroll
                for (int k2 = 6; k2 < 15; k2++) *(u32x4*)(dst + IN_BYTES + (pt + 256 * (k2 - 6)) * 16) = __builtin_bit_cast(u32x4, st[k2]);
            }
            issue(c + 2);
            if (PROD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DMA) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // the weight pieces have landed; the six input loads stay in flight
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // a raw barrier does not wait for this wave's LDS stores
            __builtin_amdgcn_s_barrier();
        }
        if (pt == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
        float s = 0;
        if (PROD != 2) for (int k2 = 0; k2 < 6; k2++) s += st[k2][0];
        out[blockIdx.x * 256 + pt] += s * 1e-30f;
    }
}

template <class K>
void run(const char* name, K kern, int threads, int blocks, float* d, const float* src, int chunks, long long* dclk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, src, rep ? chunks : 4, dclk);
        if (!rep) hipDeviceSynchronize();
        if (!rep) hipEventRecord(e0);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, dclk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * chunks * 108.0 * 32768.0 / 3.0;   // four MFMA waves per work-group in every variant
    printf("%-44s chunks/wg %5d: %8.3f ms  %6.1f TFLOP/s algorithmic (x3: %6.1f executed)  clk=%.0f MHz  %s\n", name, chunks, ms,
           flops / ms / 1e9, 3 * flops / ms / 1e9, 100.0 * h[0] / h[1], hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 2000;
    float* d; hipMalloc(&d, (size_t)512 * 512 * 4);
    float* src; hipMalloc(&src, 64 << 20);
    {   // random floats in [-1, 1): what the split arithmetic sees
        float* h = (float*)malloc(64 << 20);
        unsigned x = 12345;
        for (size_t i = 0; i < (64u << 20) / 4; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; h[i] = ((int)(x & 0xffff) - 32768) / 32768.f; }
        hipMemcpy(src, h, 64 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    long long* dclk; hipMalloc(&dclk, 16);
    for (int round = 0; round < 2; round++) {
        run("A  256 thr x 2/CU, 2 barriers, all waves stage", kA, 256, 512, d, src, chunks, dclk);
        run("B  specialised, registers, 32x32x16", kW<0, false, false>, 512, 256, d, src, 2 * chunks, dclk);
        run("C  specialised, weights by LDS-DMA, 32x32x16", kW<1, false, false>, 512, 256, d, src, 2 * chunks, dclk);
        run("F  specialised, everything by LDS-DMA, 32x32x16", kW<2, false, false>, 512, 256, d, src, 2 * chunks, dclk);
        run("E  consumers alone, 32x32x16", kW<3, false, false>, 512, 256, d, src, 2 * chunks, dclk);
        run("Ei consumers alone, interleaved, 32x32x16", kW<3, false, true>, 512, 256, d, src, 2 * chunks, dclk);
        run("Di weights by LDS-DMA, interleaved, 16x16x32", kW<1, true, true>, 512, 256, d, src, 2 * chunks, dclk);
        run("Fi everything by LDS-DMA, interleaved, 16x16x32", kW<2, true, true>, 512, 256, d, src, 2 * chunks, dclk);
        run("E6 consumers alone, 16x16x32", kW<3, true, false>, 512, 256, d, src, 2 * chunks, dclk);
        run("E6i consumers alone, interleaved, 16x16x32", kW<3, true, true>, 512, 256, d, src, 2 * chunks, dclk);
    }
    return 0;
}
