// Host micro-benchmark of lines_tail() (the order-dependent tail of detectLines that runs on host threads).
// Input: a binary file { int32 n, nc; float un[2n]; uint32 ij[nc]; float sc[2nc] }.
//   g++ -O3 -std=c++17 -ffp-contract=off -o bench_tail bench_tail.cpp && ./bench_tail cands.bin
#include "../ppg-slam_amd/csrc/lines_tail.h"

#include <chrono>
#include <cstdio>
#include <cstring>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s cands.bin [reps]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 1;
    const int n = hdr[0], nc = hdr[1];
    std::vector<float> un(2 * n), sc(2 * (size_t)nc);
    std::vector<uint32_t> ij(nc);
    if (fread(un.data(), 4, un.size(), f) != un.size() || fread(ij.data(), 4, ij.size(), f) != ij.size() ||
        fread(sc.data(), 4, sc.size(), f) != sc.size()) return 1;
    fclose(f);
    ppg_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.line_dist_thresh = 2.0f; cfg.line_inlier_rate = 0.8f; cfg.line_heatmap_thresh = 0.2f;
    FrameHost fh;
    TailScratch ts;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    lines_tail(cfg, n, un.data(), ij.data(), sc.data(), nc, fh, ts);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) lines_tail(cfg, n, un.data(), ij.data(), sc.data(), nc, fh, ts);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    unsigned long long h = 1469598103934665603ull;
    for (uint32_t v : fh.edge_idx) h = (h ^ v) * 1099511628211ull;
    for (uint32_t v : fh.coline_pairs) h = (h ^ v) * 1099511628211ull;
    printf("n=%d candidates=%d edges=%zu coline=%zu hash=%016llx  %.3f ms per frame\n", n, nc, fh.edge_score.size(),
           fh.coline_pairs.size() / 2, h, ms);
    return 0;
}
