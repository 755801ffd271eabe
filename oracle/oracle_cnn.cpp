// CPU oracle, CNN half (TEST INFRASTRUCTURE ONLY — see ppg_oracle.h).
//
// Restates the four TorchScript graphs the reference executes through LibTorch
// (feature/src/PPGExtractor.cpp:149-156):
//   net/Backbone.pt      code/__torch__/model/nets/backbone.py:16-37
//   net/PointHeatmap.pt  .../junction_decoder.py:9-15
//   net/EdgeHeatmap.pt   .../heatmap_decoder.py:8-22
//   net/Descriptor.pt    .../descriptor_decoder.py:9-15
// as direct fp32 convolutions (NHWC internally, NCHW at the API like the reference tensors).
// Pinned by tests/golden/*.npz, which hold outputs of those archives run with torch (CPU fp32).
#include "ppg_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <algorithm>

namespace {

struct Tensor {
    std::vector<uint32_t> dims;
    std::vector<float> v;
};

struct Conv {
    int cin = 0, cout = 0, ks = 0;
    std::vector<float> w;  // [ky][kx][cin][cout]
    std::vector<float> b;  // [cout]
};

struct BN {
    std::vector<float> alpha, beta;  // y = x*alpha + beta
};

}  // namespace

struct ppgo_net {
    std::map<std::string, Tensor> t;
    Conv c1a, c1b, c2a, c2b, c3a, c3b, c4a, c4b, pa, pb, da, db, h0, h1, h2, h3;
    BN bn0, bn1, bn2;
};

namespace {

bool read_blob(const char* path, std::map<std::string, Tensor>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    uint32_t n = 0, rsv = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "PPGW0001", 8) != 0) { fclose(f); return false; }
    if (fread(&n, 4, 1, f) != 1 || fread(&rsv, 4, 1, f) != 1) { fclose(f); return false; }
    struct Entry { char name[56]; uint32_t ndim; uint32_t dims[4]; uint64_t off, nbytes; };
    static_assert(sizeof(Entry) == 96, "entry size");
    std::vector<Entry> es(n);
    if (fread(es.data(), sizeof(Entry), n, f) != n) { fclose(f); return false; }
    for (auto& e : es) {
        Tensor t;
        t.dims.assign(e.dims, e.dims + e.ndim);
        t.v.resize(e.nbytes / 4);
        fseek(f, (long)e.off, SEEK_SET);
        if (fread(t.v.data(), 1, e.nbytes, f) != e.nbytes) { fclose(f); return false; }
        char nm[57];
        memcpy(nm, e.name, 56);
        nm[56] = 0;
        out[nm] = std::move(t);
    }
    fclose(f);
    return true;
}

bool make_conv(const std::map<std::string, Tensor>& t, const std::string& prefix, Conv& c) {
    auto wi = t.find(prefix + ".weight"), bi = t.find(prefix + ".bias");
    if (wi == t.end() || bi == t.end()) return false;
    const Tensor& w = wi->second;  // OIHW
    c.cout = w.dims[0]; c.cin = w.dims[1]; c.ks = w.dims[2];
    c.w.resize(w.v.size());
    for (int o = 0; o < c.cout; o++)
        for (int i = 0; i < c.cin; i++)
            for (int ky = 0; ky < c.ks; ky++)
                for (int kx = 0; kx < c.ks; kx++)
                    c.w[(((size_t)ky * c.ks + kx) * c.cin + i) * c.cout + o] =
                        w.v[(((size_t)o * c.cin + i) * c.ks + ky) * c.ks + kx];
    c.b = bi->second.v;
    return true;
}

// Eval-mode batch norm as ATen's CPU kernel evaluates it: alpha = weight/sqrt(var+eps),
// beta = bias - mean*alpha, y = x*alpha + beta  (net/EdgeHeatmap.pt .../batchnorm.py:17, eps 1e-5).
bool make_bn(const std::map<std::string, Tensor>& t, const std::string& prefix, BN& bn) {
    auto g = t.find(prefix + ".weight"), b = t.find(prefix + ".bias"),
         m = t.find(prefix + ".running_mean"), v = t.find(prefix + ".running_var");
    if (g == t.end() || b == t.end() || m == t.end() || v == t.end()) return false;
    size_t n = g->second.v.size();
    bn.alpha.resize(n); bn.beta.resize(n);
    for (size_t i = 0; i < n; i++) {
        float invstd = 1.0f / std::sqrt(v->second.v[i] + 1e-5f);
        bn.alpha[i] = invstd * g->second.v[i];
        bn.beta[i] = b->second.v[i] - m->second.v[i] * bn.alpha[i];
    }
    return true;
}

// 3x3 stride-1 pad-1 convolution, NHWC, on a zero-padded copy of the input.
template <int XB, int CB>
void conv3_rows(const float* pin, int Wp, int Cin, const Conv& c, float* out, int W, int y, int x0, int x1) {
    const int Cout = c.cout;
    for (int x = x0; x + XB <= x1; x += XB) {
        for (int co = 0; co < Cout; co += CB) {
            float acc[XB][CB];
            for (int xb = 0; xb < XB; xb++)
                for (int k = 0; k < CB; k++) acc[xb][k] = c.b[co + k];
            for (int ky = 0; ky < 3; ky++)
                for (int kx = 0; kx < 3; kx++) {
                    const float* wt = &c.w[(size_t)(ky * 3 + kx) * Cin * Cout + co];
                    const float* ip = pin + ((size_t)(y + ky) * Wp + (x + kx)) * Cin;
                    for (int ci = 0; ci < Cin; ci++) {
                        const float* wr = wt + (size_t)ci * Cout;
                        for (int xb = 0; xb < XB; xb++) {
                            float v = ip[(size_t)xb * Cin + ci];
#pragma omp simd
                            for (int k = 0; k < CB; k++) acc[xb][k] += v * wr[k];
                        }
                    }
                }
            for (int xb = 0; xb < XB; xb++)
                for (int k = 0; k < CB; k++) out[((size_t)y * W + x + xb) * Cout + co + k] = acc[xb][k];
        }
    }
}

void conv3x3(const std::vector<float>& in, int H, int W, const Conv& c, std::vector<float>& out) {
    const int Cin = c.cin, Cout = c.cout, Wp = W + 2;
    std::vector<float> pad((size_t)(H + 2) * Wp * Cin, 0.0f);
    for (int y = 0; y < H; y++)
        memcpy(&pad[((size_t)(y + 1) * Wp + 1) * Cin], &in[(size_t)y * W * Cin], sizeof(float) * W * Cin);
    out.assign((size_t)H * W * Cout, 0.0f);
    const int W4 = W / 4 * 4;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
        if (Cout % 16 == 0) {
            conv3_rows<4, 16>(pad.data(), Wp, Cin, c, out.data(), W, y, 0, W4);
            conv3_rows<1, 16>(pad.data(), Wp, Cin, c, out.data(), W, y, W4, W);
        } else {
            // generic (not used by the shipped nets: every 3x3 conv has Cout % 16 == 0)
            for (int x = 0; x < W; x++)
                for (int co = 0; co < Cout; co++) {
                    float a = c.b[co];
                    for (int ky = 0; ky < 3; ky++)
                        for (int kx = 0; kx < 3; kx++)
                            for (int ci = 0; ci < Cin; ci++)
                                a += pad[((size_t)(y + ky) * Wp + x + kx) * Cin + ci] *
                                     c.w[((size_t)(ky * 3 + kx) * Cin + ci) * Cout + co];
                    out[((size_t)y * W + x) * Cout + co] = a;
                }
        }
    }
}

// Reduced-precision modes of the build's convolution (no counterpart in the reference, whose CUDA path leaves the
// operand precision to cuDNN: every conv of net/*.pt is traced with allow_tf32=True).  Every fp32 operand v is
// split into bf16 hi = rne(v) and lo = rne(v - hi); terms = 3 evaluates w_hi*x_hi + w_hi*x_lo + w_lo*x_hi,
// terms = 1 only w_hi*x_hi, products exact, accumulated in fp32.  Stated as an ordinary convolution over
// concatenated channel planes, so the accumulation order is that of conv3x3.
inline float bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

// float rounded to the nearest fp16 value (ties to even, subnormals kept), returned as float; |x| < 65520
inline float f16_round(float x) {
    if (std::fabs(x) < 0x1p-14f) return std::nearbyint(x * 0x1p24f) * 0x1p-24f;  // fp16 subnormal grid (nearbyint: to even)
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0xFFFu + ((u >> 13) & 1u)) & 0xFFFFE000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

// The fp16 hi/lo split mode (precision 3): operands are first scaled by exact powers of two (activations 2^5, weights
// 2^10) so that hi and lo stay in fp16's normal range, split as hi = f16(v), lo = f16(v - hi), the three products
// accumulated in fp32, and the sum scaled back by 2^-15 before the bias.  About 22 significant bits per operand.
void conv3x3_f16split(const std::vector<float>& in, int H, int W, const Conv& c, std::vector<float>& out) {
    const int Cin = c.cin, Cout = c.cout, C2 = 3 * Cin;
    const float sa = 0x1p5f, sw = 0x1p10f;
    std::vector<float> in2((size_t)H * W * C2);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int ci = 0; ci < Cin; ci++) {
            const float v = in[p * Cin + ci] * sa, hi = f16_round(v), lo = f16_round(v - hi);
            in2[p * C2 + ci] = hi; in2[p * C2 + Cin + ci] = lo; in2[p * C2 + 2 * Cin + ci] = hi;
        }
    if (getenv("PPGO_PRINT_RANGE")) {  // how much of fp16's range (65504) the scaled operands use
        float ma = 0.f, mw = 0.f;
        for (float v : in) ma = std::max(ma, std::fabs(v));
        for (float v : c.w) mw = std::max(mw, std::fabs(v));
        fprintf(stderr, "f16 split conv %dx%d %d->%d: max |activation| %.3f (scaled %.0f), max |weight| %.4f (scaled %.1f)\n",
                H, W, Cin, Cout, ma, ma * sa, mw, mw * sw);
    }
    Conv c2;
    c2.cin = C2; c2.cout = Cout; c2.ks = 3; c2.b.assign(Cout, 0.0f);
    c2.w.resize((size_t)9 * C2 * Cout);
    for (int t = 0; t < 9; t++)
        for (int ci = 0; ci < Cin; ci++)
            for (int o = 0; o < Cout; o++) {
                const float v = c.w[((size_t)t * Cin + ci) * Cout + o] * sw, hi = f16_round(v), lo = f16_round(v - hi);
                c2.w[((size_t)t * C2 + ci) * Cout + o] = hi;
                c2.w[((size_t)t * C2 + Cin + ci) * Cout + o] = hi;
                c2.w[((size_t)t * C2 + 2 * Cin + ci) * Cout + o] = lo;
            }
    conv3x3(in2, H, W, c2, out);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int o = 0; o < Cout; o++) out[p * Cout + o] = out[p * Cout + o] * 0x1p-15f + c.b[o];
}

void conv3x3_lowp(const std::vector<float>& in, int H, int W, const Conv& c, int terms, std::vector<float>& out) {
    const int Cin = c.cin, Cout = c.cout, C2 = terms * Cin;
    std::vector<float> in2((size_t)H * W * C2);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int ci = 0; ci < Cin; ci++) {
            const float v = in[p * Cin + ci], hi = bf16_round(v), lo = bf16_round(v - hi);
            in2[p * C2 + ci] = hi;
            if (terms == 3) { in2[p * C2 + Cin + ci] = lo; in2[p * C2 + 2 * Cin + ci] = hi; }
        }
    Conv c2;
    c2.cin = C2; c2.cout = Cout; c2.ks = 3; c2.b = c.b;
    c2.w.resize((size_t)9 * C2 * Cout);
    for (int t = 0; t < 9; t++)
        for (int ci = 0; ci < Cin; ci++)
            for (int o = 0; o < Cout; o++) {
                const float v = c.w[((size_t)t * Cin + ci) * Cout + o], hi = bf16_round(v), lo = bf16_round(v - hi);
                c2.w[((size_t)t * C2 + ci) * Cout + o] = hi;
                if (terms == 3) {
                    c2.w[((size_t)t * C2 + Cin + ci) * Cout + o] = hi;
                    c2.w[((size_t)t * C2 + 2 * Cin + ci) * Cout + o] = lo;
                }
            }
    conv3x3(in2, H, W, c2, out);
}

void conv1x1(const std::vector<float>& in, int H, int W, const Conv& c, std::vector<float>& out) {
    const int Cin = c.cin, Cout = c.cout;
    out.assign((size_t)H * W * Cout, 0.0f);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < H * W; p++) {
        float* o = &out[(size_t)p * Cout];
        for (int k = 0; k < Cout; k++) o[k] = c.b[k];
        for (int ci = 0; ci < Cin; ci++) {
            float v = in[(size_t)p * Cin + ci];
            const float* wr = &c.w[(size_t)ci * Cout];
#pragma omp simd
            for (int k = 0; k < Cout; k++) o[k] += v * wr[k];
        }
    }
}

// 1x1 layers in the reduced-precision modes: the same operand treatment as conv3x3_lowp / conv3x3_f16split
void conv1x1_lowp(const std::vector<float>& in, int H, int W, const Conv& c, int precision, std::vector<float>& out) {
    const int Cin = c.cin, Cout = c.cout;
    const bool f16 = precision == 3;
    const int terms = precision == 2 ? 1 : 3, C2 = terms * Cin;
    const float sa = f16 ? 0x1p5f : 1.f, sw = f16 ? 0x1p10f : 1.f, us = f16 ? 0x1p-15f : 1.f;
    auto rnd = [&](float v) { return f16 ? f16_round(v) : bf16_round(v); };
    std::vector<float> in2((size_t)H * W * C2);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int ci = 0; ci < Cin; ci++) {
            const float v = in[p * Cin + ci] * sa, hi = rnd(v), lo = rnd(v - hi);
            in2[p * C2 + ci] = hi;
            if (terms == 3) { in2[p * C2 + Cin + ci] = lo; in2[p * C2 + 2 * Cin + ci] = hi; }
        }
    Conv c2;
    c2.cin = C2; c2.cout = Cout; c2.ks = 1; c2.b.assign(Cout, 0.0f);
    c2.w.resize((size_t)C2 * Cout);
    for (int ci = 0; ci < Cin; ci++)
        for (int o = 0; o < Cout; o++) {
            const float v = c.w[(size_t)ci * Cout + o] * sw, hi = rnd(v), lo = rnd(v - hi);
            c2.w[(size_t)ci * Cout + o] = hi;
            if (terms == 3) { c2.w[(size_t)(Cin + ci) * Cout + o] = hi; c2.w[(size_t)(2 * Cin + ci) * Cout + o] = lo; }
        }
    conv1x1(in2, H, W, c2, out);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int o = 0; o < Cout; o++) out[p * Cout + o] = out[p * Cout + o] * us + c.b[o];
}

void relu(std::vector<float>& a) {
    for (auto& v : a) v = v > 0.0f ? v : 0.0f;
}

void bn_relu(std::vector<float>& a, int C, const BN& bn) {
    size_t n = a.size() / C;
    for (size_t p = 0; p < n; p++)
        for (int k = 0; k < C; k++) {
            float v = a[p * C + k] * bn.alpha[k] + bn.beta[k];
            a[p * C + k] = v > 0.0f ? v : 0.0f;
        }
}

void maxpool2(const std::vector<float>& in, int H, int W, int C, std::vector<float>& out) {
    int Ho = H / 2, Wo = W / 2;
    out.resize((size_t)Ho * Wo * C);
    for (int y = 0; y < Ho; y++)
        for (int x = 0; x < Wo; x++)
            for (int k = 0; k < C; k++) {
                auto at = [&](int yy, int xx) { return in[((size_t)yy * W + xx) * C + k]; };
                float m = std::max(std::max(at(2 * y, 2 * x), at(2 * y, 2 * x + 1)),
                                   std::max(at(2 * y + 1, 2 * x), at(2 * y + 1, 2 * x + 1)));
                out[((size_t)y * Wo + x) * C + k] = m;
            }
}

// pixel_shuffle(2) on NHWC: out[2y+dy][2x+dx][c] = in[y][x][c*4 + dy*2 + dx]
void pixel_shuffle2(const std::vector<float>& in, int H, int W, int C, std::vector<float>& out) {
    int Co = C / 4;
    out.resize((size_t)H * W * C);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < Co; c++)
                for (int dy = 0; dy < 2; dy++)
                    for (int dx = 0; dx < 2; dx++)
                        out[((size_t)(2 * y + dy) * (2 * W) + (2 * x + dx)) * Co + c] =
                            in[((size_t)y * W + x) * C + c * 4 + dy * 2 + dx];
}

void to_nchw(const std::vector<float>& in, int H, int W, int C, float* out) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int k = 0; k < C; k++) out[((size_t)k * H + y) * W + x] = in[((size_t)y * W + x) * C + k];
}

}  // namespace

extern "C" {

ppgo_net* ppgo_net_load(const char* blob_path) {
    ppgo_net* n = new ppgo_net();
    bool ok = read_blob(blob_path, n->t);
    ok = ok && make_conv(n->t, "backbone.conv1a", n->c1a) && make_conv(n->t, "backbone.conv1b", n->c1b) &&
         make_conv(n->t, "backbone.conv2a", n->c2a) && make_conv(n->t, "backbone.conv2b", n->c2b) &&
         make_conv(n->t, "backbone.conv3a", n->c3a) && make_conv(n->t, "backbone.conv3b", n->c3b) &&
         make_conv(n->t, "backbone.conv4a", n->c4a) && make_conv(n->t, "backbone.conv4b", n->c4b) &&
         make_conv(n->t, "point.convPa", n->pa) && make_conv(n->t, "point.convPb", n->pb) &&
         make_conv(n->t, "desc.convDa", n->da) && make_conv(n->t, "desc.convDb", n->db) &&
         make_conv(n->t, "edge.conv_block_lst.0.0", n->h0) && make_conv(n->t, "edge.conv_block_lst.1.0", n->h1) &&
         make_conv(n->t, "edge.conv_block_lst.2.0", n->h2) && make_conv(n->t, "edge.conv_block_lst.3", n->h3) &&
         make_bn(n->t, "edge.conv_block_lst.0.1", n->bn0) && make_bn(n->t, "edge.conv_block_lst.1.1", n->bn1) &&
         make_bn(n->t, "edge.conv_block_lst.2.1", n->bn2);
    if (!ok) { delete n; return nullptr; }
    return n;
}

void ppgo_net_free(ppgo_net* n) { delete n; }

int ppgo_inference(const ppgo_net* n, const uint8_t* img, int H, int W,
                   float* feat, float* junc, float* heat, float* desc) {
    return ppgo_inference_prec(n, img, H, W, 0, feat, junc, heat, desc);
}

// precision 0: fp32 (the reference graphs as torch evaluates them on the CPU); 1 / 2: the build's bf16 split /
// plain bf16 / fp16 split modes, applied to the layers the build applies them to (every 3x3 convolution whose input
// has a multiple of 16 channels: all but conv1a; every 1x1 convolution whose input has a multiple of 32: convPb, convDb).
int ppgo_inference_prec(const ppgo_net* n, const uint8_t* img, int H, int W, int precision,
                        float* feat, float* junc, float* heat, float* desc) {
    if (!n || H % 8 || W % 8 || precision < 0 || precision > 3) return -1;
    auto conv3x3 = [&](const std::vector<float>& in, int h_, int w_, const Conv& cv, std::vector<float>& out) {
        if (precision == 3 && cv.cin % 16 == 0) conv3x3_f16split(in, h_, w_, cv, out);
        else if (precision != 0 && cv.cin % 16 == 0) conv3x3_lowp(in, h_, w_, cv, precision == 1 ? 3 : 1, out);
        else ::conv3x3(in, h_, w_, cv, out);
    };
    auto conv1x1 = [&](const std::vector<float>& in, int h_, int w_, const Conv& cv, std::vector<float>& out) {
        if (precision != 0 && cv.cin % 32 == 0) conv1x1_lowp(in, h_, w_, cv, precision, out);
        else ::conv1x1(in, h_, w_, cv, out);
    };
    std::vector<float> a((size_t)H * W), b, c;
    // PPGExtractor.cpp:151 — uint8 -> float, then a true division by 255
    for (size_t i = 0; i < a.size(); i++) a[i] = (float)img[i] / 255.0f;
    conv3x3(a, H, W, n->c1a, b); relu(b);
    conv3x3(b, H, W, n->c1b, a); relu(a);
    maxpool2(a, H, W, 64, b);
    int h = H / 2, w = W / 2;
    conv3x3(b, h, w, n->c2a, a); relu(a);
    conv3x3(a, h, w, n->c2b, b); relu(b);
    maxpool2(b, h, w, 64, a);
    h /= 2; w /= 2;
    conv3x3(a, h, w, n->c3a, b); relu(b);
    conv3x3(b, h, w, n->c3b, a); relu(a);
    maxpool2(a, h, w, 128, b);
    h /= 2; w /= 2;
    conv3x3(b, h, w, n->c4a, a); relu(a);
    conv3x3(a, h, w, n->c4b, b); relu(b);
    const std::vector<float>& fm = b;  // featureMap [h][w][128]
    if (feat) to_nchw(fm, h, w, 128, feat);
    // PointHeatmap
    {
        std::vector<float> t1, t2;
        conv3x3(fm, h, w, n->pa, t1); relu(t1);
        conv1x1(t1, h, w, n->pb, t2);
        if (junc) to_nchw(t2, h, w, 65, junc);
    }
    // EdgeHeatmap
    {
        std::vector<float> t1, t2;
        conv3x3(fm, h, w, n->h0, t1); bn_relu(t1, 256, n->bn0);
        pixel_shuffle2(t1, h, w, 256, t2);              // [2h][2w][64]
        conv3x3(t2, 2 * h, 2 * w, n->h1, t1); bn_relu(t1, 64, n->bn1);
        pixel_shuffle2(t1, 2 * h, 2 * w, 64, t2);       // [4h][4w][16]
        conv3x3(t2, 4 * h, 4 * w, n->h2, t1); bn_relu(t1, 16, n->bn2);
        pixel_shuffle2(t1, 4 * h, 4 * w, 16, t2);       // [8h][8w][4]
        conv1x1(t2, H, W, n->h3, t1);                   // [H][W][2]
        if (heat) to_nchw(t1, H, W, 2, heat);
    }
    // Descriptor
    {
        std::vector<float> t1, t2;
        conv3x3(fm, h, w, n->da, t1); relu(t1);
        conv1x1(t1, h, w, n->db, t2);
        if (desc) to_nchw(t2, h, w, 256, desc);
    }
    return 0;
}

void ppgo_score_map(const float* junc, int h, int w, float* score) {
    const int W = w * 8;
    for (int cy = 0; cy < h; cy++)
        for (int cx = 0; cx < w; cx++) {
            float v[65], m = -INFINITY;
            for (int k = 0; k < 65; k++) { v[k] = junc[((size_t)k * h + cy) * w + cx]; m = std::max(m, v[k]); }
            float s = 0.0f;
            for (int k = 0; k < 65; k++) { v[k] = std::exp(v[k] - m); s += v[k]; }
            for (int k = 0; k < 64; k++)
                score[(size_t)(cy * 8 + k / 8) * W + cx * 8 + k % 8] = v[k] / s;
        }
}

void ppgo_heat_prob(const float* heat, int H, int W, float* prob) {
    const size_t n = (size_t)H * W;
    for (size_t i = 0; i < n; i++) {
        float a = heat[i], b = heat[n + i];
        float m = std::max(a, b);
        float ea = std::exp(a - m), eb = std::exp(b - m);
        prob[i] = eb / (ea + eb);
    }
}

}  // extern "C"
