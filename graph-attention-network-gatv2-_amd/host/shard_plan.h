// shard_plan.h — destination-range partition of a CSR graph over the GPUs of a node:
// the C++ twin of shard.py (same split points, same table layout), used by train_edge --ranks N.
//
// Rows are split on row_ptr so every rank holds ~E/P edges (power-law graphs are edge-, not
// node-balanced).  Every rank is padded to max_rows rows, so the exchange tables are plain
// [P][max_rows][H*D] arrays and source ids become table rows  owner*max_rows + local.
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace gatshard {

struct Plan {
    int world = 1, rank = 0;
    std::vector<int64_t> bounds;      // [world+1] global row boundaries
    int64_t max_rows = 0;
    int64_t row0() const { return bounds[rank]; }
    int64_t n_rows() const { return bounds[rank + 1] - bounds[rank]; }
    int64_t n_table() const { return (int64_t)world * max_rows; }
    int64_t table_row0() const { return (int64_t)rank * max_rows; }
    int32_t table_id(int64_t src) const {
        const int owner = (int)(std::upper_bound(bounds.begin(), bounds.end(), src) - bounds.begin()) - 1;
        return (int32_t)((int64_t)owner * max_rows + (src - bounds[owner]));
    }
};

// Row boundaries such that every shard holds ~E/world edges and at least one row.
inline std::vector<int64_t> edge_balanced_bounds(const int32_t* row_ptr, int64_t n, int world) {
    if (world > n) throw std::invalid_argument("more ranks than rows");
    const double e = (double)row_ptr[n];
    std::vector<int64_t> b(world + 1);
    b[0] = 0; b[world] = n;
    for (int p = 1; p < world; ++p) {
        const double target = (double)p * e / (double)world;
        // first i with row_ptr[i] >= target  (numpy.searchsorted(..., side="left"))
        b[p] = std::lower_bound(row_ptr, row_ptr + n + 1, target,
                                [](int32_t v, double t) { return (double)v < t; }) - row_ptr;
    }
    // b[0] = 0 and b[world] = n stay fixed: leave room on the right first (a hub in the last rows puts several
    // cuts at n), then make the cuts strictly increasing (world <= n makes both satisfiable)
    for (int p = world - 1; p >= 1; --p) b[p] = std::min(b[p], n - (int64_t)(world - p));
    for (int p = 1; p < world; ++p) b[p] = std::max(b[p], b[p - 1] + 1);
    return b;
}

inline Plan make_plan(const int32_t* row_ptr, int64_t n, int world, int rank) {
    Plan p;
    p.world = world; p.rank = rank;
    p.bounds = edge_balanced_bounds(row_ptr, n, world);
    for (int q = 0; q < world; ++q) p.max_rows = std::max(p.max_rows, p.bounds[q + 1] - p.bounds[q]);
    return p;
}

// The rank's rows as a local CSR whose column ids are TABLE rows.
inline void local_csr(const Plan& p, const int32_t* row_ptr, const int32_t* col_idx, std::vector<int32_t>& rp,
                      std::vector<int32_t>& ci) {
    const int64_t s = p.row0(), t = s + p.n_rows();
    const int64_t e0 = row_ptr[s], e1 = row_ptr[t];
    rp.resize((size_t)(t - s + 1));
    for (int64_t i = s; i <= t; ++i) rp[(size_t)(i - s)] = (int32_t)(row_ptr[i] - e0);
    ci.resize((size_t)(e1 - e0));
    for (int64_t e = e0; e < e1; ++e) ci[(size_t)(e - e0)] = p.table_id(col_idx[e]);
}

// Global [n][F] features -> the padded source-table layout [world*max_rows][F] (padding rows zero).
inline std::vector<float> table_features(const Plan& p, const float* x, int64_t F) {
    std::vector<float> out((size_t)(p.n_table() * F), 0.0f);
    for (int q = 0; q < p.world; ++q) {
        const int64_t lo = p.bounds[q], hi = p.bounds[q + 1];
        std::copy(x + lo * F, x + hi * F, out.begin() + (size_t)((int64_t)q * p.max_rows * F));
    }
    return out;
}

}  // namespace gatshard
