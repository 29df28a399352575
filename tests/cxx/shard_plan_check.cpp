// Test driver for host/shard_plan.h (the C++ twin of shard.py): reads "world n row_ptr[0..n]" from stdin,
// prints the bounds of the edge-balanced plan and, per rank, the local CSR's row count and edge count.
#include <cstdio>
#include <cstdint>
#include <vector>

#include "shard_plan.h"

int main() {
    int world = 0;
    long long n = 0;
    if (std::scanf("%d %lld", &world, &n) != 2) return 2;
    std::vector<int32_t> rp((size_t)n + 1);
    for (auto& v : rp) { int x; if (std::scanf("%d", &x) != 1) return 2; v = x; }
    std::vector<int32_t> ci((size_t)rp[(size_t)n], 0);
    try {
        for (int r = 0; r < world; ++r) {
            gatshard::Plan p = gatshard::make_plan(rp.data(), n, world, r);
            if (r == 0) {
                std::printf("bounds");
                for (auto b : p.bounds) std::printf(" %lld", (long long)b);
                std::printf("\n");
            }
            std::vector<int32_t> lrp, lci;
            gatshard::local_csr(p, rp.data(), ci.data(), lrp, lci);
            std::printf("rank %d rows %lld edges %zu last %d\n", r, (long long)p.n_rows(), lci.size(), lrp.back());
        }
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
