// Read-only HBM streaming ceiling on this box, for gpl_sum's roofline: float4 loads, U in flight per lane,
// (a) persistent grid-stride over one 16 GB array, (b) one wave per contiguous 6.4 KB segment (gpl_sum's shape).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_read_probe tools/hbm_read_probe.hip && /tmp/hbm_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void stream_sum(const float4* __restrict__ x, size_t n4, float* out) {
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n4) ? x[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
// one wave per segment of SEG float4 rows-of-16-lanes (256 B rows, 4 rows per wave instruction)
template <int U>
__global__ __launch_bounds__(256) void segment_sum(const float4* __restrict__ x, int rows_per_seg, size_t n_seg, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t s = (size_t)blockIdx.x * 4 + wave;
    if (s >= n_seg) return;
    const int q = lane & 15, r = lane >> 4;
    const float4* base = x + s * rows_per_seg * 16;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i0 = 0; i0 < rows_per_seg; i0 += 4 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + u * 4 + r; v[u] = i < rows_per_seg ? base[(size_t)i * 16 + q] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
// gpl_sum's exact structure: segment bounds from a pointer array (dependent loads), cross-lane reduce, row written
template <int U>
__global__ __launch_bounds__(256) void csc_sum(const int* __restrict__ ptr, const float4* __restrict__ m4, float4* __restrict__ out, long n) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long s = (long)blockIdx.x * 4 + wave;
    if (s >= n) return;
    const int b = ptr[s], e = ptr[s + 1];
    const int q = lane & 15, r = lane >> 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i0 = b; i0 < e; i0 += 4 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = i0 + u * 4 + r; v[u] = i < e ? m4[(size_t)i * 16 + q] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    }
    if (r == 0) out[s * 16 + q] = acc;
}
// persistent variant: a wave walks sources s, s+W, ... and loads the NEXT source's bounds before summing the current
template <int U>
__global__ __launch_bounds__(256) void csc_sum_persistent(const int* __restrict__ ptr, const float4* __restrict__ m4, float4* __restrict__ out, long n) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long W = (long)gridDim.x * 4;
    long s = (long)blockIdx.x * 4 + wave;
    if (s >= n) return;
    const int q = lane & 15, r = lane >> 4;
    int b = ptr[s], e = ptr[s + 1];
    while (true) {
        const long sn = s + W;
        int bn = 0, en = 0;
        if (sn < n) { bn = ptr[sn]; en = ptr[sn + 1]; }
        float4 acc = make_float4(0, 0, 0, 0);
        for (int i0 = b; i0 < e; i0 += 4 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int i = i0 + u * 4 + r; v[u] = i < e ? m4[(size_t)i * 16 + q] : make_float4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        if (r == 0) out[s * 16 + q] = acc;
        if (sn >= n) break;
        s = sn; b = bn; e = en;
    }
}
// 4 consecutive sources per wave: bounds fetched with one load, the first batch (<= 16 rows) of ALL four lists issued
// before any is reduced; longer lists continue with the plain loop
__global__ __launch_bounds__(256) void csc_sum4(const int* __restrict__ ptr, const float4* __restrict__ m4, float4* __restrict__ out, long n) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long s0 = ((long)blockIdx.x * 4 + wave) * 4;
    if (s0 >= n) return;
    const int q = lane & 15, r = lane >> 4;
    int pv = 0;
    if (lane < 5) pv = ptr[s0 + lane < n ? s0 + lane : n];
    int p[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) p[k] = __builtin_amdgcn_readlane(pv, k);
    float4 v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = p[k] + u * 4 + r; v[k][u] = i < p[k + 1] ? m4[(size_t)i * 16 + q] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[k][u].x; acc.y += v[k][u].y; acc.z += v[k][u].z; acc.w += v[k][u].w; }
        for (int i0 = p[k] + 16; i0 < p[k + 1]; i0 += 16) {
            float4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 4 + r; w[u] = i < p[k + 1] ? m4[(size_t)i * 16 + q] : make_float4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += w[u].x; acc.y += w[u].y; acc.z += w[u].z; acc.w += w[u].w; }
        }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        if (r == 0 && s0 + k < n) out[(s0 + k) * 16 + q] = acc;
    }
}
// mixed read + write ceiling: y[i] = x[i] (the backward edge kernel reads ~19 GB and writes ~16.5 GB per launch)
template <int U>
__global__ __launch_bounds__(256) void stream_copy(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n4) ? x[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n4) y[i + u * 256] = v[u];
    }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}
int main() {
    const size_t bytes = 15846400000ull;             // one layer's message buffer
    const size_t n4 = bytes / 16;
    float4* x; float* out; CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(x, 0, bytes));
    for (int blocks : {2048, 4096, 8192, 16384}) {
        float m4 = timeit([&] { stream_sum<4><<<blocks, 256>>>(x, n4, out); });
        float m8 = timeit([&] { stream_sum<8><<<blocks, 256>>>(x, n4, out); });
        printf("grid-stride blocks=%5d  U=4 %.2f ms %.2f TB/s | U=8 %.2f ms %.2f TB/s\n", blocks, m4, bytes / m4 / 1e9, m8, bytes / m8 / 1e9);
    }
    {
        float4* y; CK(hipMalloc(&y, bytes));
        for (int blocks : {2048, 8192}) {
            float m4 = timeit([&] { stream_copy<4><<<blocks, 256>>>(x, y, n4); });
            float m8 = timeit([&] { stream_copy<8><<<blocks, 256>>>(x, y, n4); });
            printf("copy blocks=%5d  U=4 %.2f ms %.2f TB/s (r+w) | U=8 %.2f ms %.2f TB/s\n", blocks, m4, 2.0 * bytes / m4 / 1e9, m8, 2.0 * bytes / m8 / 1e9);
        }
        CK(hipFree(y));
    }
    for (int rows : {25, 32, 128}) {
        const size_t n_seg = n4 / 16 / rows;
        float m4 = timeit([&] { segment_sum<4><<<(unsigned)((n_seg + 3) / 4), 256>>>(x, rows, n_seg, out); });
        float m8 = timeit([&] { segment_sum<8><<<(unsigned)((n_seg + 3) / 4), 256>>>(x, rows, n_seg, out); });
        printf("wave/segment rows=%3d  U=4 %.2f ms %.2f TB/s | U=8 %.2f ms %.2f TB/s\n", rows, m4, bytes / m4 / 1e9, m8, bytes / m8 / 1e9);
    }
    // CSC-shaped: n sources, E slots; uniform lengths, then a power-law like the benchmark graph's out-degrees
    const long n0 = 2450000, E = (long)(bytes / 256);
    float4* out4; CK(hipMalloc(&out4, (n0 + 100000) * 256));
    int* dptr; CK(hipMalloc(&dptr, (n0 + 100001) * 4));
    std::vector<int> hp(n0 + 1);
    for (int mode = 0; mode < 3; ++mode) {
        hp.assign(n0 + 1, 0);
        std::vector<double> w(n0);
        double tot = 0;
        unsigned long long st = 88172645463325252ull;
        for (long i = 0; i < n0; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;                     // random placement of the heavy sources
            const double rank = (double)(st % (unsigned long long)n0);
            w[i] = mode == 0 ? 1.0 : pow(rank + 100.0, -0.75);
            tot += w[i];
        }
        double accw = 0; hp[0] = 0;
        for (long i = 0; i < n0; ++i) { accw += w[i]; hp[i + 1] = (int)(accw / tot * (double)E); }
        if (mode == 2) {                                     // power-law with the long lists cut at 256 (as gpl_sum does)
            std::vector<int> cut; cut.push_back(0);
            for (long i = 0; i < n0; ++i) { int b = hp[i]; const int e = hp[i + 1]; while (e - b > 256) { b += 256; cut.push_back(b); } cut.push_back(e); }
            hp = cut;
        }
        const long n = (long)hp.size() - 1;
        CK(hipMemcpy(dptr, hp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
        const float c4 = timeit([&] { csc_sum4<<<(unsigned)((n + 15) / 16), 256>>>(dptr, x, out4, n); });
        int mx = 0; for (long i = 0; i < n; ++i) mx = hp[i + 1] - hp[i] > mx ? hp[i + 1] - hp[i] : mx;
        const unsigned g = (unsigned)((n + 3) / 4);
        float a4 = timeit([&] { csc_sum<4><<<g, 256>>>(dptr, x, out4, n); });
        float a8 = timeit([&] { csc_sum<8><<<g, 256>>>(dptr, x, out4, n); });
        float p4 = timeit([&] { csc_sum_persistent<4><<<2048, 256>>>(dptr, x, out4, n); });
        float p8 = timeit([&] { csc_sum_persistent<4><<<4096, 256>>>(dptr, x, out4, n); });
        printf("csc %s (max len %d): wave/source U=4 %.2f ms  U=8 %.2f ms | persistent 2048 blocks %.2f ms, 4096 blocks %.2f ms | 4 sources/wave %.2f ms\n",
               mode == 0 ? "uniform  " : mode == 1 ? "power-law" : "pl cut256", mx, a4, a8, p4, p8, c4);
    }
    return 0;
}
