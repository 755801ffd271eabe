// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = a subnormal value in k = 0 of every row, B = 1.0 in
// k = 0 of every column: D = value if subnormals are kept, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, unsigned short abits) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    if (threadIdx.x < 32) {
        a[0] = __builtin_bit_cast(_Float16, abits);
        b[0] = (_Float16)1.0f;
    }
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 4); float h;
    for (unsigned short bits : {(unsigned short)0x0001, (unsigned short)0x0200, (unsigned short)0x03ff, (unsigned short)0x0400}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bits);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("fp16 bits 0x%04x -> D = %.9g\n", bits, h);
    }
    return 0;
}
