// Stress test of HostPool (built and run by tests/test_host.py): many back-to-back jobs of changing size; every index of
// every job must run exactly once, and no index of a finished job may run later.
#include "../ppg-slam_amd/csrc/host_pool.h"

#include <cstdio>

int main() {
    HostPool pool(7);
    long total = 0, expect = 0;
    for (int job = 0; job < 20000; job++) {
        const int n = 1 + (job * 7919) % 64;
        std::vector<std::atomic<int>> hits(n);
        for (auto& h : hits) h.store(0);
        std::atomic<long> sum{0};
        pool.run(n, [&](int i) {
            hits[i].fetch_add(1);
            sum.fetch_add(i + 1);
            if ((i & 7) == 0) std::this_thread::yield();
        });
        for (int i = 0; i < n; i++)
            if (hits[i].load() != 1) { printf("job %d index %d ran %d times\n", job, i, hits[i].load()); return 1; }
        total += sum.load();
        expect += (long)n * (n + 1) / 2;
    }
    if (total != expect) { printf("sum mismatch\n"); return 1; }
    printf("ok\n");
    return 0;
}
