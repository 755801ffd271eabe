// Exercises the host-side mirror (PPGExtractor::run, Matcher::SearchForInitialization) the way
// map/src/Frame.cpp:62 and system/src/Tracking.cpp:524-525 use the reference classes.
// Usage: shim_demo <weights dir> <raw u8 image A> <raw u8 image B> <width> <height>
// Prints one line of integers that tests/test_gpu_parity.py compares with the C-ABI results.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Matcher.h"
#include "PPGExtractor.h"

using namespace ppg_host;

static std::vector<uint8_t> read_raw(const char* path, size_t n) {
    std::vector<uint8_t> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s weights_dir imgA imgB W H\n", argv[0]); return 2; }
    const int W = atoi(argv[4]), H = atoi(argv[5]);
    CameraModel cam;
    cam.width = W; cam.height = H;
    cam.K[0] = 458.654f; cam.K[4] = 457.296f; cam.K[2] = W * 0.5f; cam.K[5] = H * 0.5f;
    try {
        PPGExtractor ext(&cam, argv[1]);
        FrameView F[2];
        for (int k = 0; k < 2; k++) {
            std::vector<uint8_t> img = read_raw(argv[2 + k], (size_t)W * H);
            GrayImage g;
            g.data = img.data(); g.rows = H; g.cols = W; g.step = W;
            std::vector<KeyPointEx> keys;
            ext.run(g, keys, F[k].mvKeysUn, F[k].mvKeyEdges, F[k].mDescriptors);
        }
        Matcher matcher(ext.context(), &cam, 0.9f);
        std::vector<Point2f> prev;
        for (auto& kp : F[0].mvKeysUn) prev.emplace_back(kp.mPos[0], kp.mPos[1]);
        std::vector<int> m12;
        const int nm = matcher.SearchForInitialization(F[0], F[1], prev, m12, 50);
        long long chk = 0;
        for (size_t i = 0; i < m12.size(); i++) chk = chk * 31 + (m12[i] + 2);
        size_t conn = 0;
        for (auto& kp : F[0].mvKeysUn) conn += kp.mvConnected.size();
        printf("%zu %zu %zu %zu %d %lld\n", F[0].mvKeysUn.size(), F[0].mvKeyEdges.size(), F[1].mvKeysUn.size(), conn, nm, chk);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
