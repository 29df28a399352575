// gat_synth.hip — the E-sized and N*F-sized parts of the synthetic benchmark datasets, generated ON the device
// (SURVEY §8 f3 "synthetic generator on device"; the law is SURVEY §8d's, implemented on the host in synth.py).
//
// The real datasets of the reference are a download link (README R:21) and there is no network, so every BASELINE
// workload is a deterministic synthetic graph of the stated shape.  The host keeps the cheap N-sized tables
// (apportioned in-degrees -> row_ptr, the rank-weight CDF, the rank -> node table: a few argsorts of N values); this
// file does what is proportional to E or N*F: one counter-based hash + inverse-CDF search per edge, the per-row sort
// of the sources (one global radix sort of dst*N + src), the feature matrix and the labels.  Bit-for-bit the arrays
// synth.powerlaw_graph / features / labels produce (tests/test_synth_device.py), so parity fixtures generated on the
// CPU and workloads generated on the GPU are the same graphs.  Products shape: 14.8 s of numpy -> ~1.5 s.
#include "gat_internal.h"

#include <hipcub/hipcub.hpp>

namespace gat {
namespace {

constexpr uint64_t kGold = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t mix64(uint64_t x) {          // synth._mix (splitmix64 finaliser)
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
inline uint64_t hash_base(uint64_t seed, uint64_t stream) { return mix64(seed ^ (stream * 0xD1342543DE82EF95ull)); }   // synth._hash
__device__ inline uint64_t hash_at(uint64_t base, uint64_t idx) { return mix64(idx * kGold + base); }
__device__ inline double uniform01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }          // synth._uniform01

// key[e] = dst(e) * n + node_of_rank[ upper_bound(cdf, u_e) ]      (synth.powerlaw_graph: searchsorted side="right")
__global__ __launch_bounds__(256) void source_keys_kernel(const double* __restrict__ cdf, const int32_t* __restrict__ node_of_rank,
                                                          const int32_t* __restrict__ row_ptr, int64_t n, int64_t e_total,
                                                          uint64_t base, int64_t* __restrict__ keys) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_total; e += stride) {
        const double u = uniform01(hash_at(base, (uint64_t)e));
        int64_t lo = 0, hi = n;                       // first r with cdf[r] > u
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int64_t r = lo < n - 1 ? lo : n - 1;
        int64_t a = 0, b = n;                         // row containing edge e: row_ptr[a] <= e < row_ptr[b]
        while (b - a > 1) {
            const int64_t mid = (a + b) >> 1;
            if ((int64_t)row_ptr[mid] <= e) a = mid; else b = mid;
        }
        keys[e] = a * n + (int64_t)node_of_rank[r];
    }
}
__global__ __launch_bounds__(256) void keys_to_cols_kernel(const int64_t* __restrict__ keys, int64_t n, int64_t e_total,
                                                           int32_t* __restrict__ col) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_total; e += stride) col[e] = (int32_t)(keys[e] % n);
}

// kind 0: U[-1,1) fp32 (synth.features "uniform").  One thread per element; the double -> float rounding is numpy's.
__global__ __launch_bounds__(256) void features_uniform_kernel(uint64_t base, int64_t row0, int64_t rows, int32_t f,
                                                               float* __restrict__ x) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, total = rows * f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t idx = (uint64_t)(row0 + i / f) * (uint64_t)f + (uint64_t)(i % f);
        x[i] = (float)(uniform01(hash_at(base, idx)) * 2.0 - 1.0);
    }
}
// kind 1: "bow" — sparse binary rows (~18 nnz) normalised by their count (Cora-like).  One wave per row.
__global__ __launch_bounds__(256) void features_bow_kernel(uint64_t base, int64_t row0, int64_t rows, int32_t f,
                                                           float* __restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double thr = 18.0 / (double)f;
    float cnt = 0.f;
    for (int j = lane; j < f; j += 64)
        cnt += uniform01(hash_at(base, (uint64_t)(row0 + r) * (uint64_t)f + (uint64_t)j)) < thr ? 1.0f : 0.0f;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);        // integers < 2^24: exact in any order
    const float den = cnt > 1.0f ? cnt : 1.0f;
    for (int j = lane; j < f; j += 64) {
        const float m = uniform01(hash_at(base, (uint64_t)(row0 + r) * (uint64_t)f + (uint64_t)j)) < thr ? 1.0f : 0.0f;
        x[r * f + j] = m / den;
    }
}
__global__ __launch_bounds__(256) void labels_kernel(uint64_t base, int64_t row0, int64_t rows, int32_t c, int32_t* __restrict__ lab) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        int32_t v = (int32_t)(hash_at(base, (uint64_t)(row0 + i)) % (uint64_t)c);
        if (row0 + i == 0) v = c - 1;                  // C = max(label)+1 (E:1107) must come out as requested
        lab[i] = v;
    }
}

__global__ __launch_bounds__(256) void iota32_kernel(int32_t* v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = (int32_t)i;
}

int64_t grid_for(int64_t n) { return std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), 65536); }

}  // namespace
}  // namespace gat

using namespace gat;

extern "C" {

int gat_synth_sources_device(const double* h_cdf, const int32_t* h_node_of_rank, const int32_t* h_row_ptr, int64_t n, int64_t e,
                             uint64_t seed, int32_t* d_col_idx, void* stream) {
    if (!h_cdf || !h_node_of_rank || !h_row_ptr || (!d_col_idx && e > 0)) return fail(GAT_E_INVALID, "gat_synth_sources_device: null argument");
    if (n <= 0 || e < 0 || e > 0x7fffffffLL || n > 0x7fffffffLL) return fail(GAT_E_UNSUPPORTED, "gat_synth_sources_device: sizes beyond int32");
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    double* d_cdf = nullptr; int32_t *d_nor = nullptr, *d_rp = nullptr; int64_t *k0 = nullptr, *k1 = nullptr; void* temp = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_cdf); (void)hipFree(d_nor); (void)hipFree(d_rp); (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(temp); };
#define SYN_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); return fail((int)e__, std::string(#x) + ": " + hipGetErrorString(e__)); } } while (0)
    SYN_HIP(hipMalloc((void**)&d_cdf, n * sizeof(double)));
    SYN_HIP(hipMalloc((void**)&d_nor, n * sizeof(int32_t)));
    SYN_HIP(hipMalloc((void**)&d_rp, (n + 1) * sizeof(int32_t)));
    SYN_HIP(hipMalloc((void**)&k0, e * sizeof(int64_t)));
    SYN_HIP(hipMalloc((void**)&k1, e * sizeof(int64_t)));
    SYN_HIP(hipMemcpyAsync(d_cdf, h_cdf, n * sizeof(double), hipMemcpyHostToDevice, s));
    SYN_HIP(hipMemcpyAsync(d_nor, h_node_of_rank, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    SYN_HIP(hipMemcpyAsync(d_rp, h_row_ptr, (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(source_keys_kernel, dim3((unsigned)grid_for(e)), dim3(256), 0, s, d_cdf, d_nor, d_rp, n, e, hash_base(seed, 3), k0);
    int end_bit = 1;
    while (end_bit < 63 && ((int64_t)1 << end_bit) < n * n) ++end_bit;
    size_t temp_bytes = 0;
    SYN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, k0, k1, (int)e, 0, end_bit, s));
    SYN_HIP(hipMalloc(&temp, temp_bytes));
    SYN_HIP(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, k0, k1, (int)e, 0, end_bit, s));
    hipLaunchKernelGGL(keys_to_cols_kernel, dim3((unsigned)grid_for(e)), dim3(256), 0, s, k1, n, e, d_col_idx);
    SYN_HIP(hipGetLastError());
    SYN_HIP(hipStreamSynchronize(s));
#undef SYN_HIP
    cleanup();
    return 0;
}

// order[i] = index of the i-th smallest key, ties in index order (== numpy.argsort(keys, kind="stable") for uint64
// keys): the three N-sized sorts of the host tables (two node permutations, the largest-remainder order).
int gat_synth_argsort_u64(const uint64_t* h_keys, int64_t n, int32_t* h_order, void* stream) {
    if (!h_keys || !h_order || n <= 0 || n > 0x7fffffffLL) return fail(GAT_E_INVALID, "gat_synth_argsort_u64: bad argument");
    hipStream_t s = (hipStream_t)stream;
    uint64_t *k0 = nullptr, *k1 = nullptr; int32_t *v0 = nullptr, *v1 = nullptr; void* temp = nullptr;
    auto cleanup = [&]() { (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(temp); };
#define SYN_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); return fail((int)e__, std::string(#x) + ": " + hipGetErrorString(e__)); } } while (0)
    SYN_HIP(hipMalloc((void**)&k0, n * sizeof(uint64_t))); SYN_HIP(hipMalloc((void**)&k1, n * sizeof(uint64_t)));
    SYN_HIP(hipMalloc((void**)&v0, n * sizeof(int32_t))); SYN_HIP(hipMalloc((void**)&v1, n * sizeof(int32_t)));
    SYN_HIP(hipMemcpyAsync(k0, h_keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(iota32_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, s, v0, n);
    size_t temp_bytes = 0;
    SYN_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, k0, k1, v0, v1, (int)n, 0, 64, s));
    SYN_HIP(hipMalloc(&temp, temp_bytes));
    SYN_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k0, k1, v0, v1, (int)n, 0, 64, s));
    SYN_HIP(hipMemcpyAsync(h_order, v1, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SYN_HIP(hipStreamSynchronize(s));
#undef SYN_HIP
    cleanup();
    return 0;
}

int gat_synth_features_device(uint64_t seed, int64_t row0, int64_t rows, int32_t f, int32_t kind, float* d_x, void* stream) {
    if (!d_x || rows < 0 || f <= 0 || (kind != 0 && kind != 1)) return fail(GAT_E_INVALID, "gat_synth_features_device: bad argument");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t base = hash_base(seed, 7);
    if (kind == 0) hipLaunchKernelGGL(features_uniform_kernel, dim3((unsigned)grid_for(rows * f)), dim3(256), 0, s, base, row0, rows, f, d_x);
    else hipLaunchKernelGGL(features_bow_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, base, row0, rows, f, d_x);
    GAT_HIP(hipGetLastError());
    return 0;
}

int gat_synth_labels_device(uint64_t seed, int64_t row0, int64_t rows, int32_t num_classes, int32_t* d_labels, void* stream) {
    if (!d_labels || rows < 0 || num_classes <= 0) return fail(GAT_E_INVALID, "gat_synth_labels_device: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(labels_kernel, dim3((unsigned)grid_for(rows)), dim3(256), 0, (hipStream_t)stream, hash_base(seed, 11), row0, rows,
                       num_classes, d_labels);
    GAT_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
