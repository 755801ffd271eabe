// Host side of libppgfront, first half: context, weight packing, the CNN and the extract pipeline, the order-dependent
// tail of detectLines (host statement, the fallback), profiling.  The matcher entry points are in ppg_match.cpp.
// C ABI in include/ppgfront.h.
//
// Built with -ffp-contract=off so that the host-side float formulas (line geometry) round exactly
// as the reference's -O3 build without FMA code generation does.
#include "ppg_ctx.h"

namespace {

thread_local std::string g_err;  // reason of a failed ppg_create (no context to hold it)

}  // namespace

int fail(ppg_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_err = buf;
    return code;
}

hipEvent_t prof_event(ppg_ctx* c) {
    if (c->ev_free.empty()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_all.push_back(e);
        return e;
    }
    hipEvent_t e = c->ev_free.back();
    c->ev_free.pop_back();
    return e;
}

// Accumulates every timed span whose closing event has completed; later spans stay queued.
void prof_resolve(ppg_ctx* c) {
    size_t keep = 0;
    for (size_t k = 0; k < c->ev_pairs.size(); k++) {
        auto p = c->ev_pairs[k];
        float ms = 0.f;
        if (hipEventQuery(p.second.second) == hipSuccess &&
            hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            c->prof_ms[p.first] += ms;
            c->prof_launches[p.first] += 1;
            c->ev_free.push_back(p.second.first);
            c->ev_free.push_back(p.second.second);
        } else {
            c->ev_pairs[keep++] = p;
        }
    }
    c->ev_pairs.resize(keep);
}

bool any_inflight(const ppg_ctx* c) {
    for (const Slot& sl : c->slot)
        if (sl.inflight) return true;
    return false;
}

namespace {


double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void select_lane(ppg_ctx* c, int li) {
    Lane& l = c->lane[li];
    c->lane_cur = li;
    c->stream = l.stream;
    c->a1 = l.a1; c->p1 = l.p1; c->a2 = l.a2; c->p2 = l.p2; c->a3 = l.a3; c->p3 = l.p3; c->a4 = l.a4;
    c->feat = l.feat; c->pa_o = l.pa_o; c->da_o = l.da_o; c->hs0 = l.hs0; c->hs1 = l.hs1;
}

int inflight_count(const ppg_ctx* c) {
    int k = 0;
    for (const Slot& sl : c->slot) k += sl.inflight;
    return k;
}

void select_slot(ppg_ctx* c, int sidx) {
    Slot& a = c->slot[sidx];
    Slot& o = c->slot[prev_slot(sidx)];
    c->cur = sidx;
    c->counters = a.counters; c->kp_xy = a.kp_xy; c->kp_score = a.kp_score; c->kp_un = a.kp_un; c->kp_out = a.kp_out;
    c->desc = a.desc; c->cand_ij = a.cand_ij; c->cand_sc = a.cand_sc; c->d_match = a.d_match; c->d_mcounts = a.d_mcounts;
    c->h_counters = a.h_counters; c->h_kp_xy = a.h_kp_xy; c->h_kp_score = a.h_kp_score; c->h_kp_un = a.h_kp_un;
    c->h_kp_out = a.h_kp_out; c->h_desc = a.h_desc;
    c->h_match = a.h_match; c->h_mcounts = a.h_mcounts; c->frames = &a.frames; c->h_img = a.h_img;
    c->desc_prev = o.desc; c->counters_prev = o.counters;
    c->junc = a.junc; c->hs2 = a.hs2; c->descmap = a.descmap;
}

bool read_blob(const char* path, std::map<std::string, Tensor>& out, std::string& why) {
    FILE* f = fopen(path, "rb");
    if (!f) { why = std::string("cannot open ") + path; return false; }
    char magic[8];
    uint32_t n = 0, rsv = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "PPGW0001", 8) == 0 && fread(&n, 4, 1, f) == 1 &&
              fread(&rsv, 4, 1, f) == 1 && n < 4096;
    struct Entry { char name[56]; uint32_t ndim; uint32_t dims[4]; uint32_t pad; uint64_t off, nbytes; };
    static_assert(sizeof(Entry) == 96, "blob entry is 96 bytes");
    std::vector<Entry> es(ok ? n : 0);
    ok = ok && fread(es.data(), sizeof(Entry), n, f) == n;
    for (size_t i = 0; ok && i < es.size(); i++) {
        const Entry& e = es[i];
        if (e.ndim > 4 || e.nbytes % 4 || e.nbytes > (1u << 28)) { ok = false; break; }
        {   // the entry's shape must account for exactly its bytes
            uint64_t prod = 1;
            for (uint32_t k = 0; k < e.ndim; k++) prod *= e.dims[k];
            if (prod * 4 != e.nbytes) { ok = false; break; }
        }
        Tensor t;
        t.dims.assign(e.dims, e.dims + e.ndim);
        t.v.resize(e.nbytes / 4);
        ok = fseek(f, (long)e.off, SEEK_SET) == 0 && fread(t.v.data(), 1, e.nbytes, f) == e.nbytes;
        char nm[57];
        memcpy(nm, e.name, 56);
        nm[56] = 0;
        out[nm] = std::move(t);
    }
    fclose(f);
    if (!ok) why = std::string("malformed weight blob ") + path;
    return ok;
}

// float -> bf16 bits, round to nearest even (finite inputs), and back
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// float -> fp16 bits, round to nearest even, subnormals kept (|x| < 65520), and back
inline uint16_t f16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const float ax = std::fabs(x);
    if (ax < 0x1p-14f) return sign | (uint16_t)std::nearbyint(ax * 0x1p24f);  // subnormal grid (0x0400 = 2^-14 when it rounds up)
    u &= 0x7FFFFFFFu;
    u = (u + 0xFFFu + ((u >> 13) & 1u)) >> 13;       // 10-bit mantissa, exponent still biased by 127
    return sign | (uint16_t)(u - ((127 - 15) << 10));
}
inline float f16_float(uint16_t h) {
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    float v = e == 0 ? (float)m * 0x1p-24f : std::ldexp((float)(m | 1024u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}

// Pack OIHW weights into the LDS image order of conv_mfma_kernel:
//   [cout block][cin chunk][group of 8 cin][tap][half h][cout in block][4 cin]
// with cin = chunk*KC + g*8 + h*4 + s.  `ps2` permutes couts so that a pixel-shuffle block's packed
// cout q*Cps + c (q = 2*dy + dx) is the original channel c*4 + q
// (pixel_shuffle: out[c][2y+dy][2x+dx] = in[c*4 + dy*2 + dx][y][x]).
int make_conv(ppg_ctx* c, const std::map<std::string, Tensor>& t, const std::string& name, int nt, bool ps2,
              const std::string& bn_name, DevConv& dc) {
    auto wi = t.find(name + ".weight"), bi = t.find(name + ".bias");
    if (wi == t.end() || bi == t.end() || wi->second.dims.size() != 4)
        return fail(c, PPG_EIO, "weight blob lacks %s", name.c_str());
    const Tensor& w = wi->second;
    const int cout = w.dims[0], cin = w.dims[1], ks = w.dims[2];
    if ((ks != 1 && ks != 3) || (int)w.dims[3] != ks || (int)bi->second.v.size() != cout)
        return fail(c, PPG_EIO, "%s: unsupported shape", name.c_str());
    const int KC = ppg::conv_chunk(ks), NB = 32 * nt, taps = ks * ks, G = KC / 8;
    if (cin % KC) return fail(c, PPG_EIO, "%s: unsupported shape", name.c_str());
    const int nblk = (cout + NB - 1) / NB, nchunks = cin / KC;
    dc.ks = ks; dc.nt = nt; dc.cin = cin; dc.cout = cout; dc.nblk = nblk;
    auto orig = [&](int cp) -> int {
        if (cp >= cout) return -1;
        if (!ps2) return cp;
        const int cps = cout / 4, q = cp / cps, cc = cp % cps;
        return cc * 4 + q;
    };
    std::vector<float> pk((size_t)nblk * nchunks * taps * KC * NB, 0.f);
    for (int nb = 0; nb < nblk; nb++)
        for (int ch = 0; ch < nchunks; ch++)
            for (int g = 0; g < G; g++)
                for (int tap = 0; tap < taps; tap++)
                    for (int h = 0; h < 2; h++)
                        for (int nn = 0; nn < NB; nn++) {
                            const int o = orig(nb * NB + nn);
                            if (o < 0) continue;
                            for (int s = 0; s < 4; s++) {
                                const int ci = ch * KC + g * 8 + h * 4 + s;
                                const size_t dst = ((((((size_t)nb * nchunks + ch) * G + g) * taps + tap) * 2 + h) * NB + nn) * 4 + s;
                                pk[dst] = w.v[((size_t)o * cin + ci) * taps + tap];
                            }
                        }
    // Reduced-precision modes: the 3x3 layers whose input has a multiple of 16 channels and the 1x1 layers whose input
    // has a multiple of 32 run on the bf16 / fp16 MFMA.  Image order of
    // conv_bf16_kernel: [cout block][chunk of 16 cin][tap][half h][hi | lo][cout in block][8 cin], cin = chunk*16 +
    // h*8 + j; hi = bf16(w) rounded to nearest even, lo = bf16(w - hi) (precision 3: fp16 halves of 2^10 * w).  Same byte
    // count as the fp32 image.
    dc.lowp = c->cfg.precision != 0 && ((ks == 3 && cin % 16 == 0) || (ks == 1 && cin % 32 == 0));
    if (dc.lowp) {
        std::vector<uint16_t> pb(pk.size() * 2, 0);
        const int nch16 = cin / 16;
        for (int nb = 0; nb < nblk; nb++)
            for (int ch = 0; ch < nch16; ch++)
                for (int tap = 0; tap < taps; tap++)
                    for (int h = 0; h < 2; h++)
                        for (int nn = 0; nn < NB; nn++) {
                            const int o = orig(nb * NB + nn);
                            if (o < 0) continue;
                            for (int j = 0; j < 8; j++) {
                                const int ci = ch * 16 + h * 8 + j;
                                float wv = w.v[((size_t)o * cin + ci) * taps + tap];
                                uint16_t hi, lo;
                                if (c->cfg.precision == 3) {  // fp16 halves of the weight scaled by 2^10
                                    if (!(std::fabs(wv) < 63.9f))
                                        return fail(c, PPG_ERANGE, "%s: a weight of magnitude %g leaves the range of the fp16 split (< 63.9): use precision 0 or 1", name.c_str(), (double)wv);
                                    wv = wv * 0x1p10f;
                                    hi = f16_rne(wv);
                                    lo = f16_rne(wv - f16_float(hi));
                                } else {
                                    hi = bf16_rne(wv);
                                    lo = bf16_rne(wv - bf16_float(hi));
                                }
                                const size_t base = (((((size_t)nb * nch16 + ch) * taps + tap) * 2 + h) * 2) * NB;
                                pb[((base + nn) * 8) + j] = hi;
                                pb[((base + NB + nn) * 8) + j] = lo;
                            }
                        }
        memcpy(pk.data(), pb.data(), pk.size() * sizeof(float));
    }
    std::vector<float> bias((size_t)nblk * NB, 0.f), alpha, beta;
    for (int cp = 0; cp < nblk * NB; cp++) {
        const int o = orig(cp);
        if (o >= 0) bias[cp] = bi->second.v[o];
    }
    int rc = upload(c, &dc.w, pk);
    if (!rc) rc = upload(c, &dc.bias, bias);
    if (rc) return rc;
    if (!bn_name.empty()) {
        auto g = t.find(bn_name + ".weight"), b = t.find(bn_name + ".bias"), m = t.find(bn_name + ".running_mean"),
             v = t.find(bn_name + ".running_var");
        if (g == t.end() || b == t.end() || m == t.end() || v == t.end())
            return fail(c, PPG_EIO, "weight blob lacks %s", bn_name.c_str());
        if ((int)g->second.v.size() != cout || (int)b->second.v.size() != cout || (int)m->second.v.size() != cout || (int)v->second.v.size() != cout)
            return fail(c, PPG_EIO, "%s: batch-norm vectors do not match %d output channels", bn_name.c_str(), cout);
        alpha.assign((size_t)nblk * NB, 1.f);
        beta.assign((size_t)nblk * NB, 0.f);
        for (int cp = 0; cp < nblk * NB; cp++) {
            const int o = orig(cp);
            if (o < 0) continue;
            // eval-mode batch norm as ATen evaluates it: alpha = weight/sqrt(var+eps), beta = bias - mean*alpha
            const float invstd = 1.0f / std::sqrt(v->second.v[o] + 1e-5f);
            alpha[cp] = invstd * g->second.v[o];
            beta[cp] = b->second.v[o] - m->second.v[o] * alpha[cp];
        }
        rc = upload(c, &dc.alpha, alpha);
        if (!rc) rc = upload(c, &dc.beta, beta);
    }
    return rc;
}

int load_weights(ppg_ctx* c) {
    std::map<std::string, Tensor> t;
    std::string why;
    if (!read_blob(c->weights_path.c_str(), t, why)) return fail(c, PPG_EIO, "%s", why.c_str());
    int rc = 0;
    auto mk = [&](const char* name, int nt, bool ps2, const char* bn, DevConv& dc) {
        if (!rc) rc = make_conv(c, t, name, nt, ps2, bn, dc);
    };
    mk("backbone.conv1b", 2, false, "", c->c1b);
    mk("backbone.conv2a", 2, false, "", c->c2a);
    mk("backbone.conv2b", 2, false, "", c->c2b);
    mk("backbone.conv3a", 2, false, "", c->c3a);
    mk("backbone.conv3b", 2, false, "", c->c3b);
    mk("backbone.conv4a", 2, false, "", c->c4a);
    mk("backbone.conv4b", 2, false, "", c->c4b);
    mk("point.convPa", 2, false, "", c->pa);
    mk("point.convPb", 1, false, "", c->pb);
    mk("desc.convDa", 2, false, "", c->da);
    mk("desc.convDb", 2, false, "", c->db);
    mk("edge.conv_block_lst.0.0", 2, true, "edge.conv_block_lst.0.1", c->h0);
    mk("edge.conv_block_lst.1.0", 2, true, "edge.conv_block_lst.1.1", c->h1);
    mk("edge.conv_block_lst.2.0", 1, true, "edge.conv_block_lst.2.1", c->h2);
    if (rc) return rc;
    {
        const DevConv* hs[3] = {&c->pa, &c->da, &c->h0};
        size_t tot = 0, off = 0;
        for (const DevConv* d : hs) tot += (size_t)d->nblk * 32 * d->nt * d->cin * 9;
        rc = dalloc(c, &c->heads_w, tot);
        if (rc) return rc;
        for (const DevConv* d : hs) {
            const size_t nfl = (size_t)d->nblk * 32 * d->nt * d->cin * 9;
            HIPCHK(c, hipMemcpy(c->heads_w + off, d->w, nfl * sizeof(float), hipMemcpyDeviceToDevice));
            off += nfl;
        }
    }
    auto w1 = t.find("backbone.conv1a.weight"), b1 = t.find("backbone.conv1a.bias");
    auto w3 = t.find("edge.conv_block_lst.3.weight"), b3 = t.find("edge.conv_block_lst.3.bias");
    if (w1 == t.end() || b1 == t.end() || w3 == t.end() || b3 == t.end() || w1->second.v.size() != 576 ||
        w3->second.v.size() != 8 || b1->second.v.size() != 64 || b3->second.v.size() != 2)
        return fail(c, PPG_EIO, "weight blob lacks conv1a / final heat conv");
    std::vector<float> w1p(9 * 64);
    for (int o = 0; o < 64; o++)
        for (int tap = 0; tap < 9; tap++) w1p[tap * 64 + o] = w1->second.v[o * 9 + tap];
    rc = upload(c, &c->c1a_w, w1p);
    if (!rc) rc = upload(c, &c->c1a_b, b1->second.v);
    if (!rc) rc = upload(c, &c->h3_w, w3->second.v);  // [2][4]
    if (!rc) rc = upload(c, &c->h3_b, b3->second.v);
    return rc;
}

// tile shape with the least padded area; the 16 x 16 tile exists for 3x3 layers with 64-cout blocks and no pooling
int pick_tile(const DevConv& dc, int H, int W, int mode) {
    if (dc.ks != 3 || dc.nt != 2 || mode == ppg::CONV_POOL2) return 32;
    const long a32 = (long)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32), a16 = (long)((H + 15) / 16 * 16) * ((W + 15) / 16 * 16);
    return a16 < a32 ? 16 : 32;
}

// cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap (K, D, R = I, P = K, CV_32F maps) of
// feature/src/PPGExtractor.cpp:65-71, after OpenCV's published algorithm: per destination pixel the normalised
// coordinate (accumulated along the row in double, as OpenCV does), the forward distortion model, back to pixels.
void undistort_rectify_maps(const ppg_config& c, std::vector<float>& mx, std::vector<float>& my) {
    const int W = c.width, H = c.height;
    mx.resize((size_t)W * H); my.resize((size_t)W * H);
    const double fx = c.K[0], fy = c.K[4], cx = c.K[2], cy = c.K[5];
    const double ir0 = 1.0 / fx, ir2 = -cx / fx, ir4 = 1.0 / fy, ir5 = -cy / fy;
    for (int i = 0; i < H; i++) {
        double _x = ir2, _y = i * ir4 + ir5;
        for (int j = 0; j < W; j++, _x += ir0) {
            const double x = _x, y = _y;
            double u, v;
            if (!c.fisheye) {
                const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3];
                const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = 1 + ((0.0 * r2 + k2) * r2 + k1) * r2;
                const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
                const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
                u = fx * xd + cx; v = fy * yd + cy;
            } else {
                const double r = std::sqrt(x * x + y * y), theta = std::atan(r);
                const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
                const double theta_d = theta * (1 + c.D[0] * t2 + c.D[1] * t4 + c.D[2] * t6 + c.D[3] * t8);
                const double scale = r == 0 ? 1.0 : theta_d / r;
                u = fx * x * scale + cx; v = fy * y * scale + cy;
            }
            mx[(size_t)i * W + j] = (float)u;
            my[(size_t)i * W + j] = (float)v;
        }
    }
}

// Tile grid of a launch.  A launch of the big tile (256 pixels x 64 couts per work-group) that would put fewer than
// kQuarterBelow work-groups on the chip (256 CUs take three each) goes out in quarter-size work-groups instead: with
// one or two frames per call the deep layers are 20 to 300 big tiles, and the launch lasts as long as one of them.
constexpr int kQuarterBelow = 600;
void conv_tiling(const ppg_ctx* c, ppg::ConvArgs& a, const DevConv& dc, int mode, bool fused) {
    const int H = a.H, W = a.W;
    a.mw = fused ? 32 : pick_tile(dc, H, W, mode);
    a.tiles_x = (W + a.mw - 1) / a.mw; a.tiles_y = (H + 256 / a.mw - 1) / (256 / a.mw);
    a.quarter = 0;
    // (only for launches of many tiles per work-group: shorter ones gain nothing from it)
    a.persist = (fused ? c->persist_fused : c->persist_other);
    if (!c->persist_forced && (long)a.tiles_x * a.tiles_y * a.B < 20L * a.persist) a.persist = 0;
    a.tile_counter = c->lane[c->lane_cur].tile_ctr;
    a.xcd_group = c->xcd_group; a.xcd_band = c->xcd_band;
    // Reduced-precision 3x3 layers of many tiles run wave-specialised: one resident work-group per CU whose producer
    // waves stage the next chunk while its consumer waves multiply (conv_ws.hip).  Launches too small to give every CU a
    // few tiles keep the one-tile-per-work-group kernel.
    a.ws = 0;
    if ((c->conv_ws < 0 ? a.prec == 1 : c->conv_ws != 0) && a.prec != 0 && dc.ks == 3 && dc.nt == 2 && a.Cin >= 64 && a.Cin % 16 == 0 && a.xcd_group > 0 &&
        (long)a.tiles_x * a.tiles_y * a.B * a.nblk >= c->conv_ws_min_units && a.mw == 32)   // (16 x 16 tiles: measured slower)
        a.ws = 1;
    if (c->no_quarter || fused || a.prec != 0 || dc.ks != 3 || dc.nt != 2) return;
    if ((long)a.tiles_x * a.tiles_y * a.B * a.nblk >= kQuarterBelow) return;
    a.quarter = 1;
    if (4 * a.nblk > kTileCounters) a.persist = 0;  // (two counters per launched cout block, twice as many blocks)
    // pooling pairs the two rows of an 8 x 16 tile's pixel tiles; otherwise the shape with the least padded area
    const long a32 = (long)((H + 3) / 4 * 4) * ((W + 31) / 32 * 32), a16 = (long)((H + 7) / 8 * 8) * ((W + 15) / 16 * 16);
    a.mw = (mode == ppg::CONV_POOL2 || a16 < a32) ? 16 : 32;
    a.tiles_x = (W + a.mw - 1) / a.mw; a.tiles_y = (H + 128 / a.mw - 1) / (128 / a.mw);
}

ppg::ConvGroup conv_group(const DevConv& dc, float* out, int mode, int relu, int nblk0) {
    ppg::ConvGroup g;
    g.out = out; g.bias = dc.bias; g.alpha = dc.alpha; g.beta = dc.beta;
    g.nblk0 = nblk0; g.Cout = dc.cout; g.mode = mode; g.relu = relu;
    return g;
}

int run_conv(ppg_ctx* c, const DevConv& dc, const float* in, float* out, int n, int H, int W, int mode, int relu,
             const uint8_t* fused_img = nullptr) {
    ppg::ConvArgs a;
    if (fused_img) { a.img = fused_img; a.w1a = c->c1a_w; a.b1a = c->c1a_b; }
    a.in = in; a.wpk = dc.w; a.prec = dc.lowp ? c->cfg.precision : 0; a.sat = c->slot[c->cur].sat;
    a.B = n; a.H = H; a.W = W; a.Cin = dc.cin;
    a.nblk = dc.nblk;
    conv_tiling(c, a, dc, mode, fused_img != nullptr);
    a.ngroups = 1;
    a.g[0] = conv_group(dc, out, mode, relu, 0);
    const int cls = fused_img ? PPG_PROF_CONV1A : (dc.ks == 3 && dc.nt == 2 ? PPG_PROF_CONV3X3 : PPG_PROF_CONV_OTHER);
    if (c->prof) c->prof_flops[cls] += 2.0 * n * H * W * (double)dc.cin * dc.cout * dc.ks * dc.ks + (fused_img ? 2.0 * n * H * W * 9.0 * 64 : 0.0);
    ProfScope ps(c, cls, c->stream);
    HIPCHK(c, ppg::launch_conv(a, dc.ks, dc.nt, c->stream));
    return PPG_OK;
}

// convPa, convDa and the first EdgeHeatmap conv all read featureMap (128 -> 256, 3x3): one launch with
// 12 cout blocks instead of three with 4, so the small 1/8-resolution grid fills the chip better.
int run_heads(ppg_ctx* c, int n, int H, int W) {
    ppg::ConvArgs a;
    a.in = c->feat; a.wpk = c->heads_w; a.prec = c->pa.lowp ? c->cfg.precision : 0; a.sat = c->slot[c->cur].sat;
    a.B = n; a.H = H; a.W = W; a.Cin = c->pa.cin;
    a.nblk = c->pa.nblk + c->da.nblk + c->h0.nblk;
    conv_tiling(c, a, c->pa, ppg::CONV_PLAIN, false);
    a.ngroups = 3;
    a.g[0] = conv_group(c->pa, c->pa_o, ppg::CONV_PLAIN, 1, 0);
    a.g[1] = conv_group(c->da, c->da_o, ppg::CONV_PLAIN, 1, c->pa.nblk);
    a.g[2] = conv_group(c->h0, c->hs0, ppg::CONV_PS2, 1, c->pa.nblk + c->da.nblk);
    if (c->prof) c->prof_flops[PPG_PROF_CONV3X3] += 2.0 * n * H * W * 9.0 * c->pa.cin * (c->pa.cout + c->da.cout + c->h0.cout);
    ProfScope ps(c, PPG_PROF_CONV3X3, c->stream);
    HIPCHK(c, ppg::launch_conv(a, 3, 2, c->stream));
    return PPG_OK;
}

ppg::NmsArgs nms_args(ppg_ctx* c, int n) {
    ppg::NmsArgs a;
    a.junc = c->junc; a.score = c->score; a.state = c->state; a.resid = c->resid; a.kept = c->kept;
    a.kept_key = c->kept_key; a.counters = c->counters; a.resid_cap = c->resid_cap; a.kept_cap = c->kept_cap;
    a.B = n; a.H = c->H; a.W = c->W; a.thresh = c->cfg.junction_thresh; a.max_num = c->maxn;
    a.fx = c->cfg.K[0]; a.fy = c->cfg.K[4]; a.cx = c->cfg.K[2]; a.cy = c->cfg.K[5];
    a.d0 = c->cfg.D[0]; a.d1 = c->cfg.D[1]; a.d2 = c->cfg.D[2]; a.d3 = c->cfg.D[3];
    a.fisheye = c->cfg.fisheye;
    a.kp_xy = c->kp_xy; a.kp_score = c->kp_score; a.kp_un = c->kp_un; a.kp_out = c->kp_out;
    return a;
}

ppg::HeatArgs heat_args(ppg_ctx* c, int n) {
    ppg::HeatArgs a;
    a.ps = c->hs2; a.w = c->h3_w; a.bias = c->h3_b; a.raw = c->heat_raw; a.prob = c->prob; a.refined = c->refined;
    a.B = n; a.H = c->H; a.W = c->W; a.refine_sz = c->cfg.heatmap_refine_sz;
    a.valid_thresh = c->cfg.line_valid_thresh; a.valid_ratio = c->cfg.line_valid_ratio;
    return a;
}

size_t tensor_elems(const ppg_ctx* c, int which) {
    const size_t hw = (size_t)c->h8 * c->w8, HW = (size_t)c->H * c->W;
    switch (which) {
        case PPG_T_FEAT: return hw * 128;
        case PPG_T_JUNC: return hw * 65;
        case PPG_T_HEAT_RAW: return HW * 2;
        case PPG_T_DESCMAP: return hw * 256;
        case PPG_T_SCORE: case PPG_T_HEAT: case PPG_T_HEAT_REFINED: return HW;
        case PPG_T_HEAT_REMAPPED: return c->remapped ? HW : 0;
        default: return 0;
    }
}
float* tensor_ptr(ppg_ctx* c, int which) {
    switch (which) {
        case PPG_T_FEAT: return c->feat;
        case PPG_T_JUNC: return c->junc;
        case PPG_T_HEAT_RAW: return c->heat_raw;
        case PPG_T_DESCMAP: return c->descmap;
        case PPG_T_SCORE: return c->score;
        case PPG_T_HEAT: return c->prob;
        case PPG_T_HEAT_REFINED: return c->refined;
        case PPG_T_HEAT_REMAPPED: return c->remapped;
        default: return nullptr;
    }
}

}  // namespace

int ppg_set_error(ppg_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_err = buf;
    return code;
}
int ppg_ctx_device(const ppg_ctx* c) { return c->cfg.device_id; }
int ppg_ctx_max_num(const ppg_ctx* c) { return c->maxn; }
const ppg_config* ppg_ctx_config(const ppg_ctx* c) { return &c->cfg; }
void* ppg_ctx_match_stream(ppg_ctx* c) { return (void*)c->match_stream; }
int ppg_ctx_raise_match_stream(ppg_ctx* c) {
    int lo = 0, hi = 0;
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(c, hipStreamSynchronize(c->match_stream));
    HIPCHK(c, hipStreamDestroy(c->match_stream));
    c->match_stream = nullptr;
    HIPCHK(c, hipStreamCreateWithPriority(&c->match_stream, hipStreamNonBlocking, hi));
    return PPG_OK;
}
int ppg_ctx_last_desc(ppg_ctx* c, int frame, const float** d_desc, int* n) {
    if (c->matcher_only || c->last_out_slot < 0)
        return fail(c, PPG_EINVAL, "no extracted batch is resident: call ppg_extract* or ppg_wait_batch first");
    if (frame < 0 || frame >= c->last_out_n) return fail(c, PPG_EINVAL, "frame %d outside the %d frames of the last batch", frame, c->last_out_n);
    const Slot& sl = c->slot[c->last_out_slot];
    if (sl.uses != c->last_out_use)
        return fail(c, PPG_EINVAL, "the descriptors of that batch are gone: a later ppg_submit_batch* has taken its slot");
    *d_desc = sl.desc + (size_t)frame * c->maxn * 256;
    *n = sl.nkp[frame];
    return PPG_OK;
}

extern "C" {

void ppg_default_config(ppg_config* c, int width, int height) {
    memset(c, 0, sizeof(*c));
    c->width = width; c->height = height;
    c->K[0] = 458.654f; c->K[4] = 457.296f; c->K[2] = width * 0.5f; c->K[5] = height * 0.5f; c->K[8] = 1.f;
    c->max_batch = 1;
    c->desc_dim = 256;                    // feature/src/PPGExtractor.cpp:44-53
    c->junction_thresh = 1.0f / 128.0f;
    c->junction_nms_radius = 4;
    c->junction_max_num = 500;
    c->line_valid_thresh = 1.0e-2f;
    c->line_valid_ratio = 0.3f;
    c->line_dist_thresh = 2.0f;
    c->heatmap_refine_sz = 16;
    c->line_heatmap_thresh = 0.2f;
    c->line_inlier_rate = 0.8f;
}

const char* ppg_last_error(const ppg_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

void ppg_destroy(ppg_ctx* c) {
    if (!c) return;
    delete c->pool;
    c->pool = nullptr;
    (void)hipSetDevice(c->cfg.device_id);
    for (Lane& l : c->lane)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (c->post) (void)hipStreamSynchronize(c->post);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->match_stream) (void)hipStreamSynchronize(c->match_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (void* p : c->dev_allocs) (void)hipFree(p);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    if (c->xPin) (void)hipHostFree(c->xPin);
    void* ms[] = {c->mA.p, c->mB.p, c->mD.p, c->mBest.p, c->mSecond.p, c->mTopD.p, c->mOff.p,
                  c->mIdx.p, c->mMatch.p, c->mTopI.p, c->mSkip.p, c->mClaim.p, c->mNorm.p, c->mListCnt.p, c->mListCol.p, c->mHalf.p, c->mBt.p, c->mExt.p, c->mNearLen.p, c->mListRad.p, c->bWord.p, c->bNode.p, c->bWeight.p, c->mSkipOut.p, c->mRowClaims.p, c->xArena.p,
                  c->voc_desc, c->voc_child_off, c->voc_child_idx, c->voc_word, c->voc_weight};
    for (void* p : ms)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev_all) (void)hipEventDestroy(e);
    for (Slot& sl : c->slot) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.cnn_done) (void)hipEventDestroy(sl.cnn_done);
        if (sl.junc_done) (void)hipEventDestroy(sl.junc_done);
        if (sl.heat_done) (void)hipEventDestroy(sl.heat_done);
        if (sl.match_done) (void)hipEventDestroy(sl.match_done);
    }
    if (c->ev_desc_ready) (void)hipEventDestroy(c->ev_desc_ready);
    for (Lane& l : c->lane)
        if (l.stream) (void)hipStreamDestroy(l.stream);
    if (c->post) (void)hipStreamDestroy(c->post);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->match_stream) (void)hipStreamDestroy(c->match_stream);
    delete c;
}

static int create_impl(ppg_ctx* c) {
    const ppg_config& f = c->cfg;
    if (f.width <= 0 || f.height <= 0 || f.width % 8 || f.height % 8)
        return fail(c, PPG_EINVAL, "image size %dx%d must be positive multiples of 8", f.width, f.height);
    if (f.width < 16 || f.height < 16) return fail(c, PPG_EINVAL, "image too small");
    if (f.junction_nms_radius != 4) return fail(c, PPG_EINVAL, "only JUNCTION_NMS_RADIUS = 4 is built");
    if (f.desc_dim != 256) return fail(c, PPG_EINVAL, "only DESC_DIM_SIZE = 256 is built");
    if (f.junction_max_num < 1 || f.junction_max_num > 4096) return fail(c, PPG_EINVAL, "junction_max_num must be 1..4096");
    if (c->matcher_only) {
        c->cfg.weights_path = nullptr;
        c->cfg.max_batch = 0;
    } else {
        if (f.max_batch < 1 || f.max_batch > 64) return fail(c, PPG_EINVAL, "max_batch must be 1..64");
        if (f.heatmap_refine_sz < 1 || f.heatmap_refine_sz > 16) return fail(c, PPG_EINVAL, "heatmap_refine_sz must be 1..16");
        if (f.precision < 0 || f.precision > 3) return fail(c, PPG_EINVAL, "precision %d is not built", f.precision);
        if (!f.weights_path) return fail(c, PPG_EINVAL, "weights_path is NULL");
        c->weights_path = f.weights_path;
        c->cfg.weights_path = c->weights_path.c_str();
    }
    c->W = f.width; c->H = f.height; c->w8 = c->W / 8; c->h8 = c->H / 8; c->B = c->cfg.max_batch; c->maxn = (int)f.junction_max_num;
    int ndev = 0;
    HIPCHK(c, hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(c, PPG_EHIP, "no HIP device: libppgfront has no CPU path");
    if (f.device_id < 0 || f.device_id >= ndev) return fail(c, PPG_EINVAL, "device_id %d out of range (%d devices)", f.device_id, ndev);
    HIPCHK(c, hipSetDevice(f.device_id));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, f.device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, PPG_EHIP, "device %d is %s; this library is built for gfx950 only", f.device_id, prop.gcnArchName);
    // the fused first convolution keeps one work-group per resident slot (three per CU) walking the tiles
    c->persist_fused = 3 * prop.multiProcessorCount;
    {
        const int rc0 = scratch_setup(c, f.device_id);
        if (rc0) return rc0;
    }
    // the MFMA-bound CNN gets dispatch priority; the latency-bound post-processing fills what is left
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->post, hipStreamNonBlocking, prio_lo));
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    // same priority as the CNN: at a lower one the short upload kernel would starve behind every pending convolution work-group
    HIPCHK(c, hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->match_stream, hipStreamNonBlocking, prio_lo));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_desc_ready, hipEventDisableTiming));
    {
        const char* u = getenv("PPG_UNFUSED_CONV1A");
        c->unfused_conv1a = u && atoi(u) != 0;
        const char* nl = getenv("PPG_CNN_LANES");
        if (nl && atoi(nl) == 2) c->nlanes = 2;
        c->dump_tail = getenv("PPG_DUMP_TAIL");
        const char* mm = getenv("PPG_MATCH_MATRIX");
        c->match_matrix = mm && atoi(mm) != 0;
        if (const char* pe = getenv("PPG_CONV_PERSIST")) c->persist_forced = sscanf(pe, "%d,%d", &c->persist_fused, &c->persist_other) == 2;
        if (const char* xm = getenv("PPG_XCD_MAP")) c->xcd_group = std::max(0, atoi(xm));
        if (const char* xb = getenv("PPG_XCD_BAND")) c->xcd_band = atoi(xb) != 0;
        if (const char* w = getenv("PPG_CONV_WS")) c->conv_ws = atoi(w) != 0;
        if (const char* w = getenv("PPG_CONV_WS_MIN")) c->conv_ws_min_units = std::max(8L, atol(w));
        const char* nq = getenv("PPG_NO_QUARTER_TILES");
        c->no_quarter = nq && atoi(nq) != 0;
        const char* eh = getenv("PPG_EXTEND_HOST");
        c->extend_on_host = eh && atoi(eh) != 0;
        const char* lh = getenv("PPG_LINES_HOST");
        c->lines_on_host = (lh && atoi(lh) != 0) || c->dump_tail;
        const char* e = getenv("PPG_HOST_THREADS");
        // Default: the host's hardware threads divided by the ranks sharing it (torchrun exports LOCAL_WORLD_SIZE), at
        // most 8 — the pool only stages image batches into pinned memory and redoes the rare line-graph tails the
        // device hands over, so eight ranks must not start 8 x 16 threads.
        const char* lw = getenv("LOCAL_WORLD_SIZE");
        const int ranks = std::max(1, lw ? atoi(lw) : 1);
        int t = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / (unsigned)ranks));
        c->host_threads = std::max(1, std::min(t, 64));
    }
    int rc = PPG_OK;
#define DA(ptr, n) do { rc = dalloc(c, &c->ptr, (n)); if (rc) return rc; } while (0)
#define HA(ptr, n) do { rc = halloc(c, &c->ptr, (n)); if (rc) return rc; } while (0)
    if (c->matcher_only) {   // the streams above, the match counter, and work arrays that grow with the first calls
        DA(mCount, 16);
        HA(h_mcount, 4);
        return PPG_OK;
    }
    rc = load_weights(c);
    if (rc) return rc;
    const size_t B = c->B, HW = (size_t)c->H * c->W, hw = (size_t)c->h8 * c->w8;
    for (int li = 0; li < c->nlanes; li++) {
        Lane& l = c->lane[li];
#define DL(ptr, n) do { rc = dalloc(c, &l.ptr, (n)); if (rc) return rc; } while (0)
        if (c->unfused_conv1a) DL(a1, B * HW * 64);
        DL(p1, B * HW / 4 * 64);
        DL(a2, B * HW / 4 * 64);
        DL(p2, B * HW / 16 * 64);
        DL(a3, B * HW / 16 * 128);
        DL(p3, B * hw * 128);
        DL(a4, B * hw * 128);
        DL(feat, B * hw * 128);
        DL(pa_o, B * hw * 256);
        DL(da_o, B * hw * 256);
        DL(hs0, B * hw * 4 * 64);
        DL(hs1, B * hw * 16 * 16);
        DL(tile_ctr, kTileCounters);
        HIPCHK(c, hipMemset(l.tile_ctr, 0, kTileCounters * sizeof(int)));
#undef DL
    }
    select_lane(c, 0);
    DA(heat_raw, B * HW * 2);
    DA(prob, B * HW);
    DA(refined, B * HW);
    if (c->cfg.D[0] != 0.0f) {  // feature/src/PPGExtractor.cpp:261: the heat map is remapped only then
        std::vector<float> mx, my;
        undistort_rectify_maps(c->cfg, mx, my);
        rc = upload(c, &c->mapx, mx);
        if (!rc) rc = upload(c, &c->mapy, my);
        if (rc) return rc;
        DA(remapped, B * HW);
    }
    DA(score, B * HW);
    DA(state, B * HW);
    c->resid_cap = (int)HW;
    c->kept_cap = (int)(HW / 16 + 64);
    c->cand_cap = std::max(1, c->maxn * (c->maxn - 1) / 2);
    c->edge_cap = std::max(1, std::min(c->cand_cap, 4 * c->maxn));
    c->tail_cap = std::min(std::max(c->cand_cap, 2 * c->edge_cap + 2), 65536);
    c->pair_cap = std::min(8 * c->tail_cap, 262144);
    DA(resid, B * c->resid_cap);
    DA(kept, B * c->kept_cap);
    DA(kept_key, B * c->kept_cap);
    DA(row_cnt, B * (c->maxn + 1));
    c->mask_words = (c->maxn + 255) / 256 * 4;
    DA(pair_mask, B * (size_t)c->maxn * c->mask_words);
    DA(lt_geo, B * c->tail_cap * 3);
    DA(lt_col_off, B * (c->maxn + 1));
    HIPCHK(c, hipMemset(c->lt_col_off, 0, sizeof(int) * B * (c->maxn + 1)));   // (line_resolve_kernel leaves it zeroed after every use)
    DA(lt_col_cur, B * c->maxn);
    DA(lt_col_ent, B * c->tail_cap * 4);
    DA(lt_pairs, B * (size_t)c->pair_cap * 2);
    DA(lt_kill, B * c->tail_cap);
    DA(lt_pend, B * c->tail_cap);
    DA(lt_cst, B * c->tail_cap);
    DA(mCount, 16);
    HA(h_mcount, 4);
    for (Slot& sl : c->slot) {
#define DS(ptr, n) do { rc = dalloc(c, &sl.ptr, (n)); if (rc) return rc; } while (0)
#define HS(ptr, n) do { rc = halloc(c, &sl.ptr, (n)); if (rc) return rc; } while (0)
        {
            size_t off = 0;
            auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
            const size_t E = c->edge_cap, M1 = c->maxn + 1;
            const size_t o_cnt = take(sizeof(int) * 8 * B), o_xy = take(sizeof(int32_t) * 2 * B * c->maxn),
                         o_sc = take(sizeof(float) * B * c->maxn), o_un = take(sizeof(float) * 2 * B * c->maxn),
                         o_out = take((size_t)B * c->maxn), o_desc = take(sizeof(float) * 256 * B * c->maxn),
                         o_ei = take(sizeof(uint32_t) * 2 * E * B), o_es = take(sizeof(float) * E * B),
                         o_co = take(sizeof(int32_t) * M1 * B), o_ci = take(sizeof(uint32_t) * 2 * E * B),
                         o_lo = take(sizeof(int32_t) * M1 * B), o_lp = take(sizeof(uint32_t) * 2 * E * B);
            sl.res_bytes = off;
            DS(d_res, off);
            HS(h_res, off);
#define RES(dev, host, T, o) do { sl.dev = (T*)(sl.d_res + (o)); sl.host = (T*)(sl.h_res + (o)); } while (0)
            RES(counters, h_counters, int, o_cnt);
            RES(kp_xy, h_kp_xy, int32_t, o_xy);
            RES(kp_score, h_kp_score, float, o_sc);
            RES(kp_un, h_kp_un, float, o_un);
            RES(kp_out, h_kp_out, uint8_t, o_out);
            RES(desc, h_desc, float, o_desc);
            RES(edge_idx, h_edge_idx, uint32_t, o_ei);
            RES(edge_score, h_edge_score, float, o_es);
            RES(conn_off, h_conn_off, int32_t, o_co);
            RES(conn_idx, h_conn_idx, uint32_t, o_ci);
            RES(coline_off, h_coline_off, int32_t, o_lo);
            RES(coline_pairs, h_coline_pairs, uint32_t, o_lp);
#undef RES
        }
        DS(sat, 1);
        DS(cand_ij, B * c->cand_cap);
        DS(cand_sc, B * c->cand_cap * 2);
        DS(d_match, B * c->maxn);
        DS(d_mcounts, B);
        DS(junc, B * hw * 65);
        DS(hs2, B * HW * 4);
        DS(descmap, B * hw * 256);
        HS(h_sat, 1);
        sl.h_sat[0] = 0;
        HS(h_match, B * c->maxn);
        HS(h_mcounts, B);
        HS(h_img, B * HW);
        DS(d_img, B * HW);
#undef DS
#undef HS
        HIPCHK(c, hipMemset(sl.counters, 0, sizeof(int) * 8 * B));
        HIPCHK(c, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.cnn_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.junc_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.heat_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&sl.match_done, hipEventDisableTiming));
        sl.frames.resize(B);
        sl.nkp.assign(B, 0);
    }
#undef DA
#undef HA
    select_slot(c, 0);
    c->tails.resize(B);
    // warm-up on an all-ones image, as the reference constructor does (PPGExtractor.cpp:100-106)
    // (at the full batch size, so that every kernel has run once at its production shape)
    std::vector<uint8_t> warm(HW, 1);
    std::vector<const uint8_t*> wp(B, warm.data());
    std::vector<ppg_frame_out> tmp(B);
    rc = ppg_extract_batch(c, wp.data(), (int)B, c->W, tmp.data());
    c->last_out_slot = -1;   // the warm-up frame is nobody's result
    return rc;
}

static ppg_ctx* create_common(const ppg_config* cfg, bool matcher_only) {
    if (!cfg) { fail(nullptr, PPG_EINVAL, "cfg is NULL"); return nullptr; }
    ppg_ctx* c = new (std::nothrow) ppg_ctx();
    if (!c) { fail(nullptr, PPG_ENOMEM, "out of host memory"); return nullptr; }
    c->cfg = *cfg;
    c->matcher_only = matcher_only;
    const int rc = create_impl(c);
    if (rc != PPG_OK) {
        g_err = c->err;
        ppg_destroy(c);
        return nullptr;
    }
    return c;
}

ppg_ctx* ppg_create(const ppg_config* cfg) { return create_common(cfg, false); }

ppg_ctx* ppg_create_matcher(const ppg_config* cfg) { return create_common(cfg, true); }

int ppg_set_tunables(ppg_ctx* c, const ppg_config* f) {
    if (!c) return PPG_EINVAL;
    if (!f) return fail(c, PPG_EINVAL, "ppg_set_tunables: cfg is NULL");
    NEED_EXTRACTOR(c, "ppg_set_tunables");
    if (any_inflight(c)) return fail(c, PPG_EINVAL, "ppg_set_tunables: a submitted batch is pending: call ppg_wait_batch first");
    if (f->desc_dim != c->cfg.desc_dim || f->junction_nms_radius != c->cfg.junction_nms_radius || f->junction_max_num != c->cfg.junction_max_num)
        return fail(c, PPG_ECAP, "ppg_set_tunables: desc_dim, junction_nms_radius and junction_max_num size buffers and kernels: create a new context");
    if (f->heatmap_refine_sz < 1 || f->heatmap_refine_sz > 16) return fail(c, PPG_EINVAL, "heatmap_refine_sz must be 1..16");
    c->cfg.junction_thresh = f->junction_thresh;
    c->cfg.line_valid_thresh = f->line_valid_thresh;
    c->cfg.line_valid_ratio = f->line_valid_ratio;
    c->cfg.line_dist_thresh = f->line_dist_thresh;
    c->cfg.heatmap_refine_sz = f->heatmap_refine_sz;
    c->cfg.line_heatmap_thresh = f->line_heatmap_thresh;
    c->cfg.line_inlier_rate = f->line_inlier_rate;
    return PPG_OK;
}

void* ppg_stream(ppg_ctx* c) { return c ? (void*)c->stream : nullptr; }

size_t ppg_tensor_elems(const ppg_ctx* c, int which) { return c ? tensor_elems(c, which) : 0; }

int ppg_read_tensor(ppg_ctx* c, int which, int frame, float* dst, size_t n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_read_tensor");
    const size_t e = tensor_elems(c, which);
    if (!e || !dst || n != e || frame < 0 || frame >= c->B) return fail(c, PPG_EINVAL, "ppg_read_tensor: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamSynchronize(l.stream));
    HIPCHK(c, hipStreamSynchronize(c->post));
    HIPCHK(c, hipMemcpy(dst, tensor_ptr(c, which) + (size_t)frame * e, e * sizeof(float), hipMemcpyDeviceToHost));
    return PPG_OK;
}

int ppg_write_tensor(ppg_ctx* c, int which, int frame, const float* src, size_t n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_write_tensor");
    const size_t e = tensor_elems(c, which);
    if (!e || !src || n != e || frame < 0 || frame >= c->B) return fail(c, PPG_EINVAL, "ppg_write_tensor: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    for (Lane& l : c->lane) HIPCHK(c, hipStreamSynchronize(l.stream));
    HIPCHK(c, hipStreamSynchronize(c->post));
    HIPCHK(c, hipMemcpy(tensor_ptr(c, which) + (size_t)frame * e, src, e * sizeof(float), hipMemcpyHostToDevice));
    // an injected map replaces the one the next stage call would compute from the network outputs
    if (which == PPG_T_SCORE) c->skip_score = true;
    if (which == PPG_T_HEAT || which == PPG_T_HEAT_RAW) c->skip_heat = true;
    if (which == PPG_T_HEAT_REFINED) { c->skip_heat = true; c->skip_refine = true; }
    return PPG_OK;
}

bool early_post(const ppg_ctx* c, int n);

int ppg_inference(ppg_ctx* c, const uint8_t* imgs, int n, int on_device) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_inference");
    if (!imgs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_inference: n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const size_t HW = (size_t)c->H * c->W;
    const uint8_t* dimg = imgs;
    c->skip_score = c->skip_heat = c->skip_refine = false;
    {
        const int ns = next_slot(c->cur);
        if (c->slot[ns].inflight || inflight_count(c) >= kMaxInflight)
            return fail(c, PPG_EINVAL, "four batches are in flight: call ppg_wait_batch first");
        select_lane(c, (c->lane_cur + 1) % c->nlanes);
        select_slot(c, ns);
        c->slot[ns].uses++;
        c->slot[ns].n = n;
        c->slot[ns].matched = false;
        // nkp keeps the counts of the slot's last collected batch: a later ppg_wait_batch of the OTHER slot
        // still needs them to size its match rows
    }
    if (!on_device) {
        // The upload has a stream of its own and the host waits for it here, before the convolutions are enqueued:
        // the copy of batch k+1 then runs beside the CNN of batch k, which is still queued or running, and the CNN
        // stream itself never waits for a copy.  The copy is a kernel reading the pinned buffer (see launch_upload).
        // (An event wait between the two streams instead of the host wait was
        // tried: with two batches in flight the runtime left the second copy unsubmitted until a device-wide
        // synchronisation.)  The slot's buffers are free: a slot is reused only after its batch was waited for.
        Slot& sl = c->slot[c->cur];
        if (imgs != c->h_img) memcpy(c->h_img, imgs, n * HW);  // pinned staging buffer of this slot
        HIPCHK(c, ppg::launch_upload(c->h_img, sl.d_img, n * HW, c->up_stream));
        HIPCHK(c, hipStreamSynchronize(c->up_stream));
        dimg = sl.d_img;
    }
    if (c->cfg.precision == 3) HIPCHK(c, hipMemsetAsync(c->slot[c->cur].sat, 0, sizeof(int), c->stream));
    const int H = c->H, W = c->W;
    // conv1a (u8 -> /255 -> conv 1->64 -> ReLU, backbone.py:28) is evaluated inside conv1b's staging, so the
    // full-resolution 64-channel tensor never goes to memory; PPG_UNFUSED_CONV1A=1 keeps the separate kernel.
    const bool fuse1a = !c->unfused_conv1a;
    if (!fuse1a) {
        ppg::Conv1aArgs a1;
        a1.img = dimg; a1.w = c->c1a_w; a1.bias = c->c1a_b; a1.out = c->a1; a1.B = n; a1.H = H; a1.W = W;
        ProfScope ps(c, PPG_PROF_CONV1A, c->stream);
        HIPCHK(c, ppg::launch_conv1a(a1, c->stream));
    }
    int rc;
#define RC(x) do { rc = (x); if (rc) return rc; } while (0)
    // Backbone (net/Backbone.pt backbone.py:28-37)
    RC(run_conv(c, c->c1b, c->a1, c->p1, n, H, W, ppg::CONV_POOL2, 1, fuse1a ? dimg : nullptr));
    RC(run_conv(c, c->c2a, c->p1, c->a2, n, H / 2, W / 2, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c2b, c->a2, c->p2, n, H / 2, W / 2, ppg::CONV_POOL2, 1));
    RC(run_conv(c, c->c3a, c->p2, c->a3, n, H / 4, W / 4, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c3b, c->a3, c->p3, n, H / 4, W / 4, ppg::CONV_POOL2, 1));
    RC(run_conv(c, c->c4a, c->p3, c->a4, n, H / 8, W / 8, ppg::CONV_PLAIN, 1));
    RC(run_conv(c, c->c4b, c->a4, c->feat, n, H / 8, W / 8, ppg::CONV_PLAIN, 1));
    // PointHeatmap (junction_decoder.py:14-15)
    RC(run_heads(c, n, H / 8, W / 8));  // convPa + convDa + EdgeHeatmap block 0
    RC(run_conv(c, c->pb, c->pa_o, c->junc, n, H / 8, W / 8, ppg::CONV_PLAIN, 0));
    if (early_post(c, n)) HIPCHK(c, hipEventRecord(c->slot[c->cur].junc_done, c->stream));
    // EdgeHeatmap (heatmap_decoder.py:19-22); the last 1x1 conv runs in ppg_detect_lines' first kernel
    RC(run_conv(c, c->h1, c->hs0, c->hs1, n, H / 4, W / 4, ppg::CONV_PS2, 1));
    RC(run_conv(c, c->h2, c->hs1, c->hs2, n, H / 2, W / 2, ppg::CONV_PS2, 1));
    if (early_post(c, n)) HIPCHK(c, hipEventRecord(c->slot[c->cur].heat_done, c->stream));
    // Descriptor (descriptor_decoder.py:14-15)
    RC(run_conv(c, c->db, c->da_o, c->descmap, n, H / 8, W / 8, ppg::CONV_PLAIN, 0));
#undef RC
    HIPCHK(c, hipEventRecord(c->slot[c->cur].cnn_done, c->stream));
    c->last_n = n;
    return PPG_OK;
}

// Small batches start the key-point kernels as soon as the junction logits are written and the line kernels as soon as
// EdgeHeatmap is through, beside the last convolutions: the call returns 5 % sooner.  Large batches wait for the whole CNN:
// started early, the post-processing of a 32-frame batch lands on the next batch's longest convolution instead of
// spreading over the later ones (measured: +0.7 % frames/s, but that launch 5 % longer), and nobody waits for a frame there.
// PPG_EARLY_POST=0 / 1 forces either.
bool early_post(const ppg_ctx* c, int n) {
    static const char* force = getenv("PPG_EARLY_POST");
    if (force) return atoi(force) != 0;
    return (long)n * c->H * c->W <= 16L * 640 * 480;
}

int ppg_detect_keypoints(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    // the slot's counters are cleared before the stream waits for the networks: the clear then runs beside them instead
    // of on the chain behind the junction logits (the slot is free: its last batch was waited for)
    HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(int) * 8 * n, c->post));
    HIPCHK(c, hipStreamWaitEvent(c->post, early_post(c, n) ? c->slot[c->cur].junc_done : c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_POST, c->post);
    ppg::NmsArgs a = nms_args(c, n);
    if (!c->skip_score) HIPCHK(c, ppg::launch_score_map(a, c->post));
    HIPCHK(c, ppg::launch_nms(a, c->post));
    return PPG_OK;
}

int ppg_detect_lines(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamWaitEvent(c->post, early_post(c, n) ? c->slot[c->cur].heat_done : c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_LINES, c->post);
    ppg::HeatArgs ha = heat_args(c, n);
    // last layer of EdgeHeatmap (1x1 conv 4->2, heatmap_decoder.py:22) fused with the 2-way softmax of :242
    if (!c->skip_heat) HIPCHK(c, ppg::launch_heat_tail(ha, c->post));
    if (!c->skip_refine) HIPCHK(c, ppg::launch_refine(ha, c->post));
    if (c->remapped) {  // undistort the heat map (PPGExtractor.cpp:261-262)
        ppg::RemapArgs ra;
        ra.src = c->refined; ra.dst = c->remapped; ra.mapx = c->mapx; ra.mapy = c->mapy; ra.B = n; ra.H = c->H; ra.W = c->W;
        HIPCHK(c, ppg::launch_remap(ra, c->post));
    }
    ppg::LineArgs la;
    la.heat = c->remapped ? c->remapped : c->refined; la.kp_un = c->kp_un; la.kp_out = c->kp_out; la.counters = c->counters;
    la.row_cnt = c->row_cnt; la.pair_mask = c->pair_mask; la.mask_words = c->mask_words; la.cand_ij = c->cand_ij; la.cand_sc = c->cand_sc; la.cand_cap = c->cand_cap;
    la.B = n; la.H = c->H; la.W = c->W; la.max_num = c->maxn;
    la.heat_thresh = c->cfg.line_heatmap_thresh;
    la.inv_scale = 1.0f / std::sqrt((float)(c->H * c->H + c->W * c->W));  // PPGExtractor.cpp:74
    HIPCHK(c, ppg::launch_line_candidates(la, c->post));
    if (!c->lines_on_host) {
        // overlap / score / collinearity logic of :315-441 on the device; frames it flags are redone on the host
        Slot& sl = c->slot[c->cur];
        ppg::LineTailArgs t;
        t.kp_un = c->kp_un; t.counters = c->counters; t.row_off = c->row_cnt; t.cand_ij = c->cand_ij; t.cand_sc = c->cand_sc;
        t.cand_cap = c->cand_cap; t.geo = c->lt_geo; t.col_off = c->lt_col_off; t.col_cur = c->lt_col_cur; t.col_ent = c->lt_col_ent;
        t.pairs = c->lt_pairs; t.pair_cap = c->pair_cap; t.kill_t = c->lt_kill; t.pend_t = c->lt_pend; t.cst = c->lt_cst;
        t.tail_cap = c->tail_cap; t.edge_idx = sl.edge_idx; t.edge_score = sl.edge_score; t.conn_off = sl.conn_off;
        t.conn_idx = sl.conn_idx; t.coline_off = sl.coline_off; t.coline_pairs = sl.coline_pairs; t.edge_cap = c->edge_cap;
        t.B = n; t.max_num = c->maxn; t.dist_thresh = c->cfg.line_dist_thresh; t.inlier_rate = c->cfg.line_inlier_rate;
        t.heat_thresh = c->cfg.line_heatmap_thresh;
        HIPCHK(c, ppg::launch_line_tail(t, c->post));
    }
    return PPG_OK;
}

int ppg_gen_descriptors(ppg_ctx* c, int n) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "this stage");
    if (n < 1 || n > c->B) return fail(c, PPG_EINVAL, "n = %d outside 1..%d", n, c->B);
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    HIPCHK(c, hipStreamWaitEvent(c->post, c->slot[c->cur].cnn_done, 0));
    ProfScope ps(c, PPG_PROF_DESC, c->post);
    ppg::DescArgs a;
    a.descmap = c->descmap; a.kp_xy = c->kp_xy; a.counters = c->counters; a.desc = c->desc;
    a.B = n; a.H = c->H; a.W = c->W; a.max_num = c->maxn;
    HIPCHK(c, ppg::launch_desc_sample(a, c->post));
    return PPG_OK;
}

// Device-to-host copies of everything the host needs from slot `si`, then the slot's done event.
static int enqueue_d2h(ppg_ctx* c, int si, int n) {
    Slot& sl = c->slot[si];
    const int M = c->maxn;
    hipStream_t s = c->post;
    if (c->cfg.precision == 3) HIPCHK(c, hipMemcpyAsync(sl.h_sat, sl.sat, sizeof(int), hipMemcpyDeviceToHost, s));
    if (n == c->B && !c->lines_on_host) {
        HIPCHK(c, hipMemcpyAsync(sl.h_res, sl.d_res, sl.res_bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK(c, hipMemcpyAsync(sl.h_counters, sl.counters, sizeof(int) * 8 * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_xy, sl.kp_xy, sizeof(int32_t) * 2 * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_score, sl.kp_score, sizeof(float) * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_un, sl.kp_un, sizeof(float) * 2 * M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_kp_out, sl.kp_out, (size_t)M * n, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(sl.h_desc, sl.desc, sizeof(float) * 256 * M * n, hipMemcpyDeviceToHost, s));
        if (!c->lines_on_host) {
            const size_t E = c->edge_cap;
            HIPCHK(c, hipMemcpyAsync(sl.h_edge_idx, sl.edge_idx, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_edge_score, sl.edge_score, sizeof(float) * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_conn_off, sl.conn_off, sizeof(int32_t) * (M + 1) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_conn_idx, sl.conn_idx, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_coline_off, sl.coline_off, sizeof(int32_t) * (M + 1) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(sl.h_coline_pairs, sl.coline_pairs, sizeof(uint32_t) * 2 * E * n, hipMemcpyDeviceToHost, s));
        }
    }
    if (sl.matched) {
        hipStream_t ms = c->match_stream;
        HIPCHK(c, hipMemcpyAsync(sl.h_match, sl.d_match, sizeof(int32_t) * M * n, hipMemcpyDeviceToHost, ms));
        HIPCHK(c, hipMemcpyAsync(sl.h_mcounts, sl.d_mcounts, sizeof(int) * n, hipMemcpyDeviceToHost, ms));
        HIPCHK(c, hipEventRecord(sl.match_done, ms));
    }
    HIPCHK(c, hipEventRecord(sl.done, s));
    return PPG_OK;
}

// Waits for slot `si`, runs the host tails of its frames and fills `outs`.
static int finish_slot(ppg_ctx* c, int si, int n, ppg_frame_out* outs) {
    Slot& sl = c->slot[si];
    const int M = c->maxn;
    // the slot's data is complete once `done` fired; late copies use their own stream so that they do
    // not queue behind the next batch already running on the main one
    hipStream_t s = c->copy_stream;
    const double t_sync0 = now_ms();
    HIPCHK(c, hipEventSynchronize(sl.done));
    if (sl.matched) HIPCHK(c, hipEventSynchronize(sl.match_done));
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_WAIT] += now_ms() - t_sync0; c->prof_launches[PPG_PROF_HOST_WAIT] += 1; }
    if (c->cfg.precision == 3 && sl.h_sat[0])
        return fail(c, PPG_ERANGE, "an activation left the range of the fp16 split (|x| > 2047): use precision 0 or 1");
    for (int b = 0; b < n; b++) {
        const int* cnt = sl.h_counters + b * 8;
        if (cnt[0] > c->resid_cap || cnt[1] > c->kept_cap || cnt[2] > c->cand_cap)
            return fail(c, PPG_ECAP, "frame %d: work list overflow (resid %d kept %d cand %d)", b, cnt[0], cnt[1], cnt[2]);
    }
    if (c->prof) prof_resolve(c);
    static const bool dbg_counters = getenv("PPG_DEBUG_COUNTERS") != nullptr;   // measurement aid: work-list sizes of frame 0
    if (dbg_counters) {
        const int* k = sl.h_counters;
        fprintf(stderr, "[ppg] frame 0: %d residual pixels (%d rounds), %d kept, %d line candidates, %d key points, %d edges, %d pairs (%d rounds)\n",
                k[0], k[7] & 255, k[1], k[2], k[3], k[4], k[6], k[7] >> 8);
    }
    for (int b = 0; b < n; b++) sl.nkp[b] = sl.h_counters[b * 8 + 3];
    c->last_out_slot = si; c->last_out_n = n; c->last_out_use = sl.uses;
    const double t_tail0 = now_ms();
    const int M1 = M + 1;
    // Frames whose line graph the device tail finished point straight into the pinned mirrors.  The others (a value
    // too close to a rounding boundary, a work array too small, or PPG_LINES_HOST) get the host statement of the
    // tail, lines_tail(), on their candidate list, which is fetched only now.
    std::vector<int> redo;
    for (int b = 0; b < n; b++) {
        const int* cnt = sl.h_counters + b * 8;
        const int N = cnt[3];
        ppg_frame_out& o = outs[b];
        o.n = N;
        o.kp_xy = sl.h_kp_xy + (size_t)b * M * 2;
        o.kp_score = sl.h_kp_score + (size_t)b * M;
        o.kp_un = sl.h_kp_un + (size_t)b * M * 2;
        o.kp_out = sl.h_kp_out + (size_t)b * M;
        o.desc = sl.h_desc + (size_t)b * M * 256;
        if (N > 0 && (c->lines_on_host || cnt[5] != 0)) { redo.push_back(b); continue; }
        FrameHost& fh = sl.frames[b];
        if (N == 0) {
            // PPGExtractor.cpp:239-240 returns before touching mvKeyEdges (stale edges of the previous
            // frame stay in the reference's member); with no key points they index nothing, so this
            // boundary reports an empty graph.
            fh.conn_off.assign(1, 0);
            fh.coline_off.assign(1, 0);
            o.n_edges = 0;
            o.edge_idx = nullptr; o.edge_score = nullptr; o.conn_idx = nullptr; o.coline_pairs = nullptr;
            o.conn_off = fh.conn_off.data(); o.coline_off = fh.coline_off.data();
            continue;
        }
        o.n_edges = cnt[4];
        o.edge_idx = sl.h_edge_idx + (size_t)b * c->edge_cap * 2;
        o.edge_score = sl.h_edge_score + (size_t)b * c->edge_cap;
        o.conn_off = sl.h_conn_off + (size_t)b * M1;
        o.conn_idx = sl.h_conn_idx + (size_t)b * c->edge_cap * 2;
        o.coline_off = sl.h_coline_off + (size_t)b * M1;
        o.coline_pairs = sl.h_coline_pairs + (size_t)b * c->edge_cap * 2;
    }
    c->tail_frames += n;
    c->tail_fallbacks += (long long)redo.size();
    for (int b : redo) {
        const int* cnt = sl.h_counters + b * 8;
        const int N = cnt[3], nc = cnt[2];
        c->fb_ij.resize((size_t)std::max(nc, 1));
        c->fb_sc.resize((size_t)std::max(nc, 1) * 2);
        if (nc > 0) {
            HIPCHK(c, hipMemcpyAsync(c->fb_ij.data(), sl.cand_ij + (size_t)b * c->cand_cap, sizeof(uint32_t) * nc, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->fb_sc.data(), sl.cand_sc + (size_t)b * c->cand_cap * 2, sizeof(float) * 2 * nc, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
        ppg_frame_out& o = outs[b];
        if (c->dump_tail && !c->dump_tail_done && nc > 0) {
            // debugging aid (PPG_DUMP_TAIL=<file>): the first frame's input of lines_tail(), in the format that
            // tools/bench_tail.cpp reads
            c->dump_tail_done = true;
            if (FILE* f = fopen(c->dump_tail, "wb")) {
                const int32_t hdr[2] = {N, nc};
                fwrite(hdr, 4, 2, f);
                fwrite(o.kp_un, 4, (size_t)2 * N, f);
                fwrite(c->fb_ij.data(), 4, (size_t)nc, f);
                fwrite(c->fb_sc.data(), 4, (size_t)2 * nc, f);
                fclose(f);
            }
        }
        FrameHost& fh = sl.frames[b];
        lines_tail(c->cfg, N, o.kp_un, c->fb_ij.data(), c->fb_sc.data(), nc, fh, c->tails[b]);
        o.n_edges = (int)fh.edge_score.size();
        o.edge_idx = fh.edge_idx.data();
        o.edge_score = fh.edge_score.data();
        o.conn_off = fh.conn_off.data();
        o.conn_idx = fh.conn_idx.data();
        o.coline_off = fh.coline_off.data();
        o.coline_pairs = fh.coline_pairs.data();
    }
    const int redone = (int)redo.size();
    if (c->prof) {
        c->prof_ms[PPG_PROF_HOST_TAIL] += now_ms() - t_tail0;
        c->prof_launches[PPG_PROF_HOST_TAIL] += 1;
        c->prof_launches[PPG_PROF_HOST_TAIL_FRAMES] += redone;   // frames the host had to redo
    }
    return PPG_OK;
}

int ppg_collect(ppg_ctx* c, int n, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_collect");
    if (n < 1 || n > c->B || !outs) return fail(c, PPG_EINVAL, "ppg_collect: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    int rc = enqueue_d2h(c, c->cur, n);
    if (!rc) rc = finish_slot(c, c->cur, n, outs);
    return rc;
}


static int enqueue_extract(ppg_ctx* c, const uint8_t* imgs, int n, int on_device) {
    static const int skip = getenv("PPG_DEBUG_SKIP") ? atoi(getenv("PPG_DEBUG_SKIP")) : 0;  // measurement aid only
    int rc = ppg_inference(c, imgs, n, on_device);
    if (!rc && !(skip & 1)) rc = ppg_detect_keypoints(c, n);
    if (!rc && !(skip & 2)) rc = ppg_detect_lines(c, n);
    if (!rc && !(skip & 4)) rc = ppg_gen_descriptors(c, n);
    return rc;
}

static int extract_common(ppg_ctx* c, const uint8_t* imgs, int n, int on_device, ppg_frame_out* outs) {
    if (any_inflight(c))
        return fail(c, PPG_EINVAL, "a submitted batch is pending: call ppg_wait_batch first");
    int rc = enqueue_extract(c, imgs, n, on_device);
    if (!rc) rc = ppg_collect(c, n, outs);
    return rc;
}

static int stage_host_images(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, const uint8_t** staged);

static int submit_common(ppg_ctx* c, const uint8_t* imgs, int n, int on_device, int match_prev, float nnratio) {
    const double t0 = now_ms();
    if (match_prev && c->slot[c->cur].n < n)
        return fail(c, PPG_EINVAL, "match_prev needs a previous batch of at least %d frames", n);
    int rc = enqueue_extract(c, imgs, n, on_device);  // switches to the other slot or fails if it is in flight
    if (rc) return rc;
    Slot& sl = c->slot[c->cur];
    if (match_prev) {
        rc = match_resident(c, c->desc_prev, c->counters_prev + 3, c->desc, c->counters + 3, n, nnratio, sl.d_match, sl.d_mcounts);
        if (rc) return rc;
        sl.matched = true;
    }
    rc = enqueue_d2h(c, c->cur, n);
    if (rc) return rc;
    sl.inflight = true;
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_SUBMIT] += now_ms() - t0; c->prof_launches[PPG_PROF_HOST_SUBMIT] += 1; }
    return PPG_OK;
}

int ppg_submit_batch_device(ppg_ctx* c, const uint8_t* d_imgs, int n, int match_prev, float nnratio) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_submit_batch_device");
    if (!d_imgs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_submit_batch_device: bad argument");
    return submit_common(c, d_imgs, n, 1, match_prev, nnratio);
}

int ppg_submit_batch(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, int match_prev, float nnratio) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_submit_batch");
    const uint8_t* staged = nullptr;
    int rc = stage_host_images(c, imgs, n, row_stride, &staged);
    if (rc) return rc;
    return submit_common(c, staged, n, 0, match_prev, nnratio);
}

int ppg_wait_batch(ppg_ctx* c, ppg_frame_out* outs, int32_t* m12, int* n_matches) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_wait_batch");
    if (!outs) return fail(c, PPG_EINVAL, "ppg_wait_batch: outs is NULL");
    // slots alternate, so the older pending batch is the one that is not current (if it is pending)
    int si = c->cur;
    for (int k = 1; k <= kSlots; k++) {
        si = (c->cur + k) % kSlots;
        if (c->slot[si].inflight) break;
    }
    Slot& sl = c->slot[si];
    if (!sl.inflight) return fail(c, PPG_EINVAL, "ppg_wait_batch: nothing was submitted");
    HIPCHK(c, hipSetDevice(c->cfg.device_id));
    const int n = sl.n, M = c->maxn;
    const double t0 = now_ms();
    int rc = finish_slot(c, si, n, outs);
    sl.inflight = false;
    if (rc) return rc;
    if (c->prof) { c->prof_ms[PPG_PROF_HOST_FINISH] += now_ms() - t0; c->prof_launches[PPG_PROF_HOST_FINISH] += 1; }
    if (sl.matched) {
        const Slot& prev = c->slot[prev_slot(si)];
        for (int b = 0; b < n; b++) {
            if (m12) {
                memcpy(m12 + (size_t)b * M, sl.h_match + (size_t)b * M, sizeof(int32_t) * prev.nkp[b]);
                for (int k = prev.nkp[b]; k < M; k++) m12[(size_t)b * M + k] = -1;
            }
            if (n_matches) n_matches[b] = sl.h_mcounts[b];
        }
    } else {
        if (m12) for (size_t k = 0; k < (size_t)n * M; k++) m12[k] = -1;
        if (n_matches) for (int b = 0; b < n; b++) n_matches[b] = 0;
    }
    return n;
}

// copies n host images (row_stride >= W) into the pinned staging buffer of the slot the next extract uses
static int stage_host_images(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, const uint8_t** staged) {
    if (!imgs || n < 1 || n > c->B || row_stride < c->W)
        return fail(c, PPG_EINVAL, "bad image batch (n = %d, max_batch = %d, row_stride = %d)", n, c->B, row_stride);
    Slot& nx = c->slot[next_slot(c->cur)];
    if (nx.inflight || inflight_count(c) >= kMaxInflight) return fail(c, PPG_EINVAL, "four batches are in flight: call ppg_wait_batch first");
    const size_t HW = (size_t)c->H * c->W;
    for (int b = 0; b < n; b++)
        if (!imgs[b]) return fail(c, PPG_EINVAL, "image %d is NULL", b);
    auto copy_one = [&](int b) {
        if (row_stride == c->W) memcpy(nx.h_img + b * HW, imgs[b], HW);
        else
            for (int y = 0; y < c->H; y++) memcpy(nx.h_img + b * HW + (size_t)y * c->W, imgs[b] + (size_t)y * row_stride, c->W);
    };
    // a batch of VGA frames is ~10 MB: spread the copy into pinned memory over the host pool
    if (n >= 4 && c->host_threads > 1 && (size_t)n * HW >= (1u << 20)) {
        if (!c->pool) c->pool = new HostPool(c->host_threads - 1);
        c->pool->run(n, copy_one);
    } else {
        for (int b = 0; b < n; b++) copy_one(b);
    }
    *staged = nx.h_img;
    return PPG_OK;
}

int ppg_extract_batch(ppg_ctx* c, const uint8_t* const* imgs, int n, int row_stride, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_extract_batch");
    if (!outs) return fail(c, PPG_EINVAL, "ppg_extract_batch: outs is NULL");
    if (any_inflight(c))
        return fail(c, PPG_EINVAL, "a submitted batch is pending: call ppg_wait_batch first");
    const uint8_t* staged = nullptr;
    int rc = stage_host_images(c, imgs, n, row_stride, &staged);
    if (rc) return rc;
    return extract_common(c, staged, n, 0, outs);
}

int ppg_extract(ppg_ctx* c, const uint8_t* img, int row_stride, ppg_frame_out* out) {
    return ppg_extract_batch(c, &img, 1, row_stride, out);
}

int ppg_extract_batch_device(ppg_ctx* c, const uint8_t* d_imgs, int n, ppg_frame_out* outs) {
    if (!c) return PPG_EINVAL;
    NEED_EXTRACTOR(c, "ppg_extract_batch_device");
    if (!d_imgs || !outs || n < 1 || n > c->B) return fail(c, PPG_EINVAL, "ppg_extract_batch_device: bad argument");
    return extract_common(c, d_imgs, n, 1, outs);
}

// ------------------------------------------------------------------------------------------
// matching/
// ------------------------------------------------------------------------------------------

int ppg_profile(ppg_ctx* c, int enable) {
    if (!c) return PPG_EINVAL;
    c->prof = enable == 2 ? 2 : (enable != 0);
    for (int k = 0; k < PPG_PROF_N; k++) { c->prof_ms[k] = 0; c->prof_launches[k] = 0; c->prof_flops[k] = 0; }
    return PPG_OK;
}

int ppg_tail_stats(ppg_ctx* c, long long* frames, long long* host_frames) {
    if (!c || !frames || !host_frames) return PPG_EINVAL;
    *frames = c->tail_frames;
    *host_frames = c->tail_fallbacks;
    return PPG_OK;
}

int ppg_profile_flops(ppg_ctx* c, double* flops) {
    if (!c || !flops) return PPG_EINVAL;
    for (int k = 0; k < PPG_PROF_N; k++) flops[k] = c->prof_flops[k];
    return PPG_OK;
}

int ppg_profile_read(ppg_ctx* c, double* ms, long long* launches) {
    if (!c || !ms || !launches) return PPG_EINVAL;
    for (int k = 0; k < PPG_PROF_N; k++) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_launches[k]; }
    return PPG_OK;
}

}  // extern "C"
