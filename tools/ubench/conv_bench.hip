// Isolated timing of the convolution kernel on the layer shapes of the 640x480 front-end (batch 16).
// Usage: conv_bench <0 zeros | 1 random | 2 small positive | 3 denormal-range inputs> [batch] [precision 0|1|2|3] [1: wave-specialised kernel]
#define PPG_CONV_TIMING 1
#define PPG_WS_TIMING 1
#include "../../ppg-slam_amd/csrc/conv_kernels.hip"
#include "../../ppg-slam_amd/csrc/conv_ws.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace ppg;
__global__ void clock_probe(long long* out, long long wall_ticks) {
    const long long c0 = clock64(), w0 = wall_clock64();
    while (wall_clock64() - w0 < wall_ticks) __builtin_amdgcn_s_sleep(64);
    out[0] = clock64() - c0;
    out[1] = wall_clock64() - w0;
}
__global__ void fill_rand(float* p, size_t n, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (scale / 32768.f);
    }
}
__global__ void relu_inplace(float* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] > 0.f ? p[i] : 0.f;
}
__global__ void fill_rand8(uint8_t* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; p[i] = (uint8_t)x; }
}
struct L { const char* name; int H, W, Cin, Cout, mode, fused, mw, nt, ks = 3; };
int main(int argc, char** argv) {
    const int B = argc > 2 ? atoi(argv[2]) : 16;
    const int prec = argc > 3 ? atoi(argv[3]) : 0;
    const int use_ws = argc > 4 ? atoi(argv[4]) : 0;
    int* dctr; hipMalloc(&dctr, 256 * sizeof(int)); hipMemset(dctr, 0, 256 * sizeof(int));
    unsigned long long* dwst; hipMalloc(&dwst, 128);
    hipMemcpyToSymbol(HIP_SYMBOL(ppg::g_ws_timing), &dwst, sizeof(dwst));
    std::vector<L> ls = {
        {"conv1b fused pool", 480, 640, 64, 64, CONV_POOL2, 1, 32, 2},
        {"conv2a plain", 240, 320, 64, 64, CONV_PLAIN, 0, 32, 2},
        {"conv2b pool", 240, 320, 64, 64, CONV_POOL2, 0, 32, 2},
        {"conv3a plain", 120, 160, 64, 128, CONV_PLAIN, 0, 32, 2},
        {"conv3b pool", 120, 160, 128, 128, CONV_POOL2, 0, 32, 2},
        {"conv4a plain nt1", 60, 80, 128, 128, CONV_PLAIN, 0, 32, 1},
        {"conv4a plain mw16", 60, 80, 128, 128, CONV_PLAIN, 0, 16, 2},
        {"conv4a plain mw32 nt2", 60, 80, 128, 128, CONV_PLAIN, 0, 32, 2},
        {"h1 ps2-like plain", 120, 160, 64, 64, CONV_PLAIN, 0, 32, 2},
        {"h1 plain nt1", 120, 160, 64, 64, CONV_PLAIN, 0, 32, 1},
        {"h2 16->16 nt1", 240, 320, 16, 32, CONV_PLAIN, 0, 32, 1},
        {"heads 768 mw16", 60, 80, 128, 768, CONV_PLAIN, 0, 16, 2},
        {"db 1x1 256->256", 60, 80, 256, 256, CONV_PLAIN, 0, 32, 2, 1},
        {"pb 1x1 256->96", 60, 80, 256, 96, CONV_PLAIN, 0, 32, 1, 1},
    };
    size_t maxin = (size_t)B * 480 * 640 * 64;
    float *din, *dout, *dw, *dbias; uint8_t* dimg;
    hipMalloc(&din, maxin * 4 / 4); hipMalloc(&dout, maxin * 4); hipMalloc(&dw, 16 << 20); hipMalloc(&dbias, 4096);
    hipMalloc(&dimg, (size_t)B * 480 * 640);
    hipMemset(din, 0, maxin); hipMemset(dw, 0, 16 << 20); hipMemset(dbias, 0, 4096); hipMemset(dimg, 7, (size_t)B * 480 * 640);
    const int rmode = argc > 1 ? atoi(argv[1]) : 0;
    const bool rnd = rmode != 0;
    if (rnd) {
        fill_rand<<<4096, 256>>>(din, maxin / 4, rmode == 2 ? 1e-3f : rmode == 3 ? 1e-38f : 1.f);
        if (rmode >= 2) relu_inplace<<<4096, 256>>>(din, maxin / 4);
        fill_rand<<<4096, 256>>>(dw, (16 << 20) / 4, 0.05f);
        fill_rand8<<<4096, 256>>>(dimg, (size_t)B * 480 * 640);
        hipDeviceSynchronize();
    }
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    long long* dclk; hipMalloc(&dclk, 16); long long hclk[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* dtim; hipMalloc(&dtim, 32);
    hipMemcpyToSymbol(HIP_SYMBOL(ppg::g_conv_timing), &dtim, sizeof(dtim));
    for (const L& l : ls) {
        if (prec && (l.ks != 3 || l.nt != 2)) continue;
        ConvArgs a{};
        a.prec = prec;
        a.in = din; a.wpk = dw; a.B = B; a.H = l.H; a.W = l.W; a.Cin = l.Cin;
        if (l.fused) { a.img = dimg; a.w1a = dw; a.b1a = dbias; }
        const int tw = l.mw, th = 256 / l.mw;
        a.tiles_x = (l.W + tw - 1) / tw; a.tiles_y = (l.H + th - 1) / th; a.nblk = l.Cout / (32 * l.nt);
        a.mw = l.mw; a.ngroups = 1;
        a.ws = use_ws && prec && l.ks == 3 && l.nt == 2 && l.Cin >= 64;
        a.tile_counter = dctr; a.xcd_group = 4; a.xcd_band = 1;
        a.g[0].out = dout; a.g[0].bias = dbias; a.g[0].nblk0 = 0; a.g[0].Cout = l.Cout; a.g[0].mode = l.mode; a.g[0].relu = 1;
        for (int i = 0; i < 3; i++) launch_conv(a, l.ks, l.nt, 0);
        hipDeviceSynchronize();
        const int reps = 10;
        hipMemset(dtim, 0, 32);
        hipMemset(dwst, 0, 128);
        float est; { hipEventRecord(e0); launch_conv(a, l.ks, l.nt, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&est, e0, e1); }
        hipEventRecord(e0);
        clock_probe<<<1, 64, 0, s2>>>(dclk, (long long)(est * reps * 0.8 * 1e5));
        for (int i = 0; i < reps; i++) if (launch_conv(a, l.ks, l.nt, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        double fl = 2.0 * B * l.H * l.W * l.Cin * l.Cout * l.ks * l.ks;
        hipStreamSynchronize(s2);
        hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
        unsigned long long htim[4];
        hipMemcpy(htim, dtim, 32, hipMemcpyDeviceToHost);
        const double nw = htim[3] ? (double)htim[3] : 1.0;
        if (a.ws) {
            unsigned long long w[16];
            hipMemcpy(w, dwst, 128, hipMemcpyDeviceToHost);
            const double wg = w[5] ? (double)w[5] : 1.0, mhz = 100.0 * hclk[0] / hclk[1];
            printf("rnd=%d %-26s %8.1f us  %6.1f TF/s  units=%d  clk=%.0f MHz  WS per work-group: consumer %.1f us (barrier wait %.1f, epilogues %.1f), producer %.1f us (barrier wait %.1f, stage %.1f (input wait %.1f), fetch+draw %.1f, vmcnt wait %.1f), tiles %.1f\n",
                   (int)rnd, l.name, ms * 1e3, fl / ms / 1e9, a.tiles_x * a.tiles_y * B * a.nblk, mhz, w[0] / wg / mhz, w[1] / wg / mhz,
                   w[2] / wg / mhz, w[3] / wg / mhz, w[4] / wg / mhz, w[7] / wg / mhz, w[10] / wg / mhz, w[8] / wg / mhz, w[9] / wg / mhz, w[6] / wg);
        } else
        printf("rnd=%d %-26s %8.1f us  %6.1f TF/s  blocks=%d  clk=%.0f MHz  wg us: prologue %.2f main %.2f epilogue %.2f\n", (int)rnd, l.name, ms * 1e3, fl / ms / 1e9,
               a.tiles_x * a.tiles_y * B * a.nblk, 100.0 * hclk[0] / hclk[1], htim[0] / nw * 0.01, htim[1] / nw * 0.01, htim[2] / nw * 0.01);
    }
    return 0;
}
