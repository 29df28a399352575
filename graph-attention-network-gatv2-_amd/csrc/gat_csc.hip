// gat_csc.hip — source-major (CSC) slot index for the backward scatter, built once per graph.
//
// The reference scatters per-edge gradient contributions with float atomicAdd (E:868-869 into
// gx[src], E:772-786 into grad_W).  On gfx950 float atomics run at ≈1.3 TB/s chip-wide against
// ≈6 TB/s for plain stores, so the backward instead STORES per edge, into its slot of a source-sorted
// scratch array, either a 64-byte record (alpha, ge, LeakyReLU' decisions: the training path) or the
// whole message row, and a second pass walks each source's contiguous slots: gpl_pull_* rebuilds the
// messages from the records and one gathered row of g[dst] while it sums, gpl_sum_* just sums rows
// (cdna_hip_programming.md Appendix B "Scatter / gather": store pass + per-destination sum pass).
//   pos[e]      = slot of CSR edge e in (src, e)-sorted order   (stable radix sort: fixed order
//                 inside every source's list => bitwise reproducible sums)
//   src_ptr[s]  = first slot of table row s, src_ptr[n_table] = E
//   csc_src[j]  = table row of slot j (the sorted keys), csc_dst[j] = destination row of slot j
// Two families of second-pass kernels: LIST kernels (a source's slot list per wave or lane group: gpl_pull_kernel, gpl_pull3_kernel,
// gpl_sum_*; long lists chunked) for graphs with ~25 slots per list, and the SLOT-PARALLEL form (runs of consecutive slots per
// lane group, segmented by csc_src: gpl_pull_runs_kernel, gpl_sum_runs_kernel) where lists are short — destination-range shards
// and sparse graphs (round 4).  run_pull / launch_gpl_sum hold the selection rules and the measurements behind them.
#include "gat_internal.h"

#include <hipcub/hipcub.hpp>
#include <utility>

namespace gat {
namespace {

__global__ __launch_bounds__(256) void iota_kernel(int32_t* v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void invert_perm_kernel(const int32_t* __restrict__ perm, int32_t* __restrict__ pos,
                                                         int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) pos[perm[i]] = (int32_t)i;
}
// src_ptr[s] = lower_bound(sorted_keys, s)
__global__ __launch_bounds__(256) void segment_offsets_kernel(const int32_t* __restrict__ keys, int64_t n_keys,
                                                             int32_t* __restrict__ ptr, int64_t n_seg) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= n_seg; s += stride) {
        int64_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < s) lo = mid + 1; else hi = mid;
        }
        ptr[s] = (int32_t)lo;
    }
}

// Sources with more than kHeavySlots slots (power-law hubs: one wave would stream megabytes with a few KB in
// flight and become the tail of the launch — 13 k slots = 1.7 ms alone) are cut into chunks of kHeavySlots,
// one wave each, partial rows summed per source in chunk order by a small second kernel (deterministic).
constexpr int kHeavySlotsDefault = 512;    // swept on the Products shape: 64..1024, 512 best (5.83 -> 5.56 ms per step vs 256)
// GAT_GPL_HEAVY=<n> overrides (tests: huge = never chunk).  Graphs with few edges chunk earlier: there the pass is as long
// as its longest list (a 500-slot list is ~30 dependent gather steps of one wave; Arxiv shape: 0.13 -> 0.05 ms per launch)
static int heavy_slots(int64_t n_edges) {
    static const int v = [] { const char* e = choice_env("GAT_GPL_HEAVY"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 0; }();
    return v ? v : (n_edges < (16 << 20) ? 128 : kHeavySlotsDefault);
}

// chunks: {first slot, end slot, partial row index, -}; row = HD floats (BF: HD bf16) per slot
template <int HD, bool BF>
__global__ __launch_bounds__(256) void gpl_chunk_kernel(const int4* __restrict__ chunks, int32_t n_chunks,
                                                        const float* __restrict__ msg, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * 4 + wave;
    if (k >= n_chunks) return;
    const int4 ch = chunks[k];
    const int b = ch.x, e = ch.y;
    constexpr int U = 8;
    if constexpr (!BF) {
        constexpr int LPR = HD / 4, RPI = 64 / LPR;
        const int q = lane % LPR, r = lane / LPR;
        const float4* m4 = reinterpret_cast<const float4*>(msg);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = b; i0 < e; i0 += RPI * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * RPI + r;
                v[u] = (i < e) ? m4[(int64_t)i * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
        }
        if (r == 0) reinterpret_cast<float4*>(part)[(int64_t)ch.z * LPR + q] = acc;
    } else {
        constexpr int LPR = HD / 8, RPI = 64 / LPR;
        const int q = lane % LPR, r = lane / LPR;
        const uint4* m8 = reinterpret_cast<const uint4*>(msg);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int i0 = b; i0 < e; i0 += RPI * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * RPI + r;
                v[u] = (i < e) ? m8[(int64_t)i * LPR + q] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] += __builtin_bit_cast(float, w[j] << 16);
                    acc[2 * j + 1] += __builtin_bit_cast(float, w[j] & 0xFFFF0000u);
                }
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off);
        if (r == 0) {
            float4* out = reinterpret_cast<float4*>(part + (int64_t)ch.z * HD + q * 8);
            out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}
// heavy: {source, first partial, partial count, -}; one thread per (source, channel), ascending chunk order
__global__ __launch_bounds__(256) void gpl_heavy_fix_kernel(const int4* __restrict__ heavy, int32_t n_heavy,
                                                            const float* __restrict__ part, float* __restrict__ gPL,
                                                            int32_t HD) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = t / HD;
    const int c = (int)(t % HD);
    if (k >= n_heavy) return;
    const int4 h = heavy[k];
    float s = 0.f;
    int p = h.y;
    for (; p + 8 <= h.y + h.z; p += 8) {            // eight loads in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = part[(int64_t)(p + i) * HD + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; p < h.y + h.z; ++p) s += part[(int64_t)p * HD + c];
    gPL[(int64_t)h.x * HD + c] = s;
}

// gPL[s][:] = sum over slots of s.  Row = HD floats read as float4 by HD/4 lanes, 64/(HD/4) rows
// per wave-instruction (1 KiB), U instructions in flight.
template <int HD>
__global__ __launch_bounds__(256) void gpl_sum_kernel(const int32_t* __restrict__ src_ptr,
                                                      const float* __restrict__ msg, float* __restrict__ gPL,
                                                      int64_t n_table, int32_t kHeavySlots) {
    constexpr int LPR = HD / 4;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per wave-instruction
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (s >= n_table) return;
    const int b = src_ptr[s], e = src_ptr[s + 1];
    if (e - b > kHeavySlots) return;                    // long lists: gpl_chunk_kernel + gpl_heavy_fix_kernel
    const int q = lane % LPR, r = lane / LPR;
    const float4* m4 = reinterpret_cast<const float4*>(msg);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = b; i0 < e; i0 += RPI * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * RPI + r;
            v[u] = (i < e) ? m4[(int64_t)i * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    }
    if (r == 0) reinterpret_cast<float4*>(gPL)[s * LPR + q] = acc;
}

// bf16 message rows (cfg.storage_dtype): a row is HD*2 bytes, one 16-byte load = 8 channels, HD/8 lanes per
// row; accumulation and the gPL output stay fp32.  One source per group of HD/8 lanes walking its slots
// (U loads in flight); sources with long lists are finished by the whole wave like in the group kernel.
template <int HD>
__global__ __launch_bounds__(256) void gpl_sum_bf16_kernel(const int32_t* __restrict__ src_ptr,
                                                           const float* __restrict__ msg, float* __restrict__ gPL,
                                                           int64_t n_table, int32_t kHeavySlots) {
    constexpr int LPR = HD / 8, RPI = 64 / LPR, U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane % LPR, r = lane / LPR;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (s >= n_table) return;
    const int b = src_ptr[s], e = src_ptr[s + 1];
    if (e - b > kHeavySlots) return;
    const uint4* m8 = reinterpret_cast<const uint4*>(msg);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int i0 = b; i0 < e; i0 += RPI * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * RPI + r;
            v[u] = (i < e) ? m8[(int64_t)i * LPR + q] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += __builtin_bit_cast(float, w[k] << 16);
                acc[2 * k + 1] += __builtin_bit_cast(float, w[k] & 0xFFFF0000u);
            }
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], off);
    if (r == 0) {
        float4* out = reinterpret_cast<float4*>(gPL + s * HD + q * 8);
        out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// Same sum for short lists (a shard sees ~deg/P slots per source): one source per GROUP of HD/4
// lanes, 64/(HD/4) sources per wave, each group walking its own slots — no cross-lane reduction and
// 64/(HD/4) times fewer waves.  Sources with more than kGroupMax slots are handed to the whole
// wave afterwards (same scheme as gpl_sum_kernel).
constexpr int kGroupMax = 32;
template <int HD>
__global__ __launch_bounds__(256) void gpl_sum_group_kernel(const int32_t* __restrict__ src_ptr,
                                                            const float* __restrict__ msg, float* __restrict__ gPL,
                                                            int64_t n_table, int32_t kHeavySlots) {
    constexpr int LPR = HD / 4, RPI = 64 / LPR, U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane % LPR, r = lane / LPR;
    const int64_t s = ((int64_t)blockIdx.x * 4 + wave) * RPI + r;
    int b = 0, e = 0;
    if (s < n_table) { b = src_ptr[s]; e = src_ptr[s + 1]; }
    const bool heavy = (e - b) > kHeavySlots;          // handled by gpl_chunk_kernel + gpl_heavy_fix_kernel
    const bool big = !heavy && (e - b) > kGroupMax;
    const float4* m4 = reinterpret_cast<const float4*>(msg);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!big && !heavy) {
        for (int i0 = b; i0 < e; i0 += U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = (i0 + u < e) ? m4[(int64_t)(i0 + u) * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    uint64_t todo = __ballot(big && q == 0);
    while (todo) {                                     // wave-uniform: every lane sees the same mask
        const int g = (__ffsll((unsigned long long)todo) - 1) / LPR;
        todo &= todo - 1;
        const int bg = __shfl(b, g * LPR), eg = __shfl(e, g * LPR);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = bg; i0 < eg; i0 += RPI * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * RPI + r;
                v[u] = (i < eg) ? m4[(int64_t)i * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            t.x += __shfl_xor(t.x, off); t.y += __shfl_xor(t.y, off);
            t.z += __shfl_xor(t.z, off); t.w += __shfl_xor(t.w, off);
        }
        if (r == g) acc = t;
    }
    if (s < n_table && !heavy) reinterpret_cast<float4*>(gPL)[s * LPR + q] = acc;
}

// bf16 message rows, group per source (see gpl_sum_group_kernel): HD/8 lanes own one source's list (a 64-byte row at
// H*D = 32 is FOUR lanes: one wave per source left 15 of its 16 row slots idle on the ~25-slot lists of BASELINE config 5
// and paid a 4-level cross-lane reduction per source).  Consecutive sources' lists are contiguous in slot order, so a wave's
// 64/(HD/8) groups together stream one contiguous region.  Lists beyond kGroupMax go to the whole wave afterwards.
template <int HD>
__global__ __launch_bounds__(256) void gpl_sum_bf16_group_kernel(const int32_t* __restrict__ src_ptr,
                                                                 const float* __restrict__ msg, float* __restrict__ gPL,
                                                                 int64_t n_table, int32_t kHeavySlots) {
    constexpr int LPR = HD / 8, RPI = 64 / LPR, U = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane % LPR, r = lane / LPR;
    const int64_t s = ((int64_t)blockIdx.x * 4 + wave) * RPI + r;
    int b = 0, e = 0;
    if (s < n_table) { b = src_ptr[s]; e = src_ptr[s + 1]; }
    const bool heavy = (e - b) > kHeavySlots;          // handled by gpl_chunk_kernel + gpl_heavy_fix_kernel
    const bool big = !heavy && (e - b) > kGroupMax;
    const uint4* m8 = reinterpret_cast<const uint4*>(msg);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    auto add = [&](float (&a8)[8], const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a8[2 * k] += __builtin_bit_cast(float, w[k] << 16);
            a8[2 * k + 1] += __builtin_bit_cast(float, w[k] & 0xFFFF0000u);
        }
    };
    if (!big && !heavy) {
        for (int i0 = b; i0 < e; i0 += U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = (i0 + u < e) ? m8[(int64_t)(i0 + u) * LPR + q] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < U; ++u) add(acc, v[u]);
        }
    }
    uint64_t todo = __ballot(big && q == 0);
    while (todo) {                                     // wave-uniform: every lane sees the same mask
        const int g = (__ffsll((unsigned long long)todo) - 1) / LPR;
        todo &= todo - 1;
        const int bg = __shfl(b, g * LPR), eg = __shfl(e, g * LPR);
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = 0.f;
        for (int i0 = bg; i0 < eg; i0 += RPI * U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * RPI + r;
                v[u] = (i < eg) ? m8[(int64_t)i * LPR + q] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add(t, v[u]);
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += __shfl_xor(t[k], off);
        if (r == g) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = t[k];
        }
    }
    if (s < n_table && !heavy) {
        float4* out = reinterpret_cast<float4*>(gPL + s * HD + q * 8);
        out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Stash path (edge_bwd2_kernel<..., STASH>): the destination-major backward leaves a record of HD/N words per edge
// in its source-major slot — per head, alpha in the even lane's word and grad_attn_score in the odd lane's, each
// with the N LeakyReLU'(s) decisions of that lane's channels in its N low bits — instead of the HD-float message
// row.  This pass walks a source's slots, gathers ONE row g[dst] per edge and rebuilds
//     msg[c] = g[dst][c] * alpha[h] + ge[h] * a[c] * (bit_c ? 1 : slope)                       (E:859-869)
// while it sums: 64 B + 4 B streamed and 256 B gathered per edge (H*D = 64), against 256 B written and 256 B
// re-read by the message-row path.  Same fixed summation order per source => bitwise reproducible.
// Lane layout as in the edge kernels: N adjacent channels per lane, LPE = HD/N lanes per edge, G = 64/LPE edges per
// wave-instruction; the chunk's destination indices come from one coalesced load and reach their lanes via shuffles.
template <int N> struct PullVec;
template <> struct PullVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct PullVec<4> { typedef float T __attribute__((ext_vector_type(4))); };

// write-once output rows: streaming stores (see gat_edge_kernels.hip; GAT_NT_STORES: experiment)
template <class T>
__device__ __forceinline__ void stream_store(T* p, const T& v) {
#ifdef GAT_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// LASTD > 0 (last layer, D = LASTD): g[dst][c] = gh[dst][c % D] * LReLU'(h_pre[dst][c]) / H  (E:598-603) is rebuilt from
// gh [n_rows][D] and the per-lane decision byte hbits [n_rows][HD/N] — 48 B gathered per edge from two small tables
// (98 MB + 39 MB at the Products shape: cache-resident) instead of a 256-B row of g.
template <int HD, int N, bool BF, int LASTD>
__device__ __forceinline__ typename PullVec<N>::T pull_range(const uint32_t* __restrict__ stash,
                                                             const int32_t* __restrict__ cdst,
                                                             const float* __restrict__ gfull, const uint8_t* __restrict__ hbits, int gh_stride, int hb_stride,
                                                             float slope, int b, int e, int lane,
                                                             typename PullVec<N>::T ac, typename PullVec<N>::T acs) {
    using V = typename PullVec<N>::T;
    constexpr int LPE = HD / N, G = 64 / LPE;
    constexpr int CH = 16, U = CH / G;
    static_assert(U >= 1, "lane layout");
    const int cp = lane % LPE, gidx = lane / LPE;
    V acc;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    auto load_dst = [&](int i0) {
        const int j = i0 + (lane & (CH - 1));
        return cdst[j < e ? j : e - 1];
    };
    int dv = load_dst(b);                                   // callers pass b < e
    for (int i0 = b; i0 < e; i0 += CH) {
        const int dn = (i0 + CH < e) ? load_dst(i0 + CH) : 0;
        uint32_t w[U];
        V g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * G + gidx;
            const int ic = i < e ? i : e - 1;                   // clamped: loads need no predicate
            w[u] = stash[(uint64_t)(uint32_t)ic * LPE + cp];
            const uint32_t drow = (uint32_t)__shfl(dv, u * G + gidx);
            if constexpr (LASTD > 0) {                            // gfull = gh here
                constexpr float inv_heads = 1.0f / (float)(HD / LASTD);
                const V gh4 = *reinterpret_cast<const V*>(gfull + (size_t)drow * gh_stride + (cp * N) % LASTD);
                const uint32_t nib = hbits[(size_t)drow * hb_stride + cp];
#pragma unroll
                for (int k = 0; k < N; ++k) g[u][k] = gh4[k] * ((nib >> k) & 1u ? 1.0f : slope) * inv_heads;
            } else if constexpr (BF) {                                   // g rows stored as bf16 (cfg.storage_dtype): 2*N bytes per lane
                const char* p = reinterpret_cast<const char*>(gfull) + drow * (uint32_t)(HD * 2) + (uint32_t)cp * (uint32_t)(N * 2);
                if constexpr (N == 2) {
                    const uint32_t w2 = *reinterpret_cast<const uint32_t*>(p);
                    g[u][0] = __builtin_bit_cast(float, w2 << 16); g[u][1] = __builtin_bit_cast(float, w2 & 0xFFFF0000u);
                } else {
                    const uint2 w2 = *reinterpret_cast<const uint2*>(p);
                    g[u][0] = __builtin_bit_cast(float, w2.x << 16); g[u][1] = __builtin_bit_cast(float, w2.x & 0xFFFF0000u);
                    g[u][2] = __builtin_bit_cast(float, w2.y << 16); g[u][3] = __builtin_bit_cast(float, w2.y & 0xFFFF0000u);
                }
            } else {
                const uint32_t off = drow * (uint32_t)(HD * 4) + (uint32_t)cp * (uint32_t)(N * 4);
#ifdef GAT_NT_GATHER
                g[u] = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + off));
#else
                g[u] = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + off);
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * G + gidx;
            const float val = __builtin_bit_cast(float, w[u] & ~((1u << N) - 1u));
            const float oth = __shfl_xor(val, 1);               // the head's other lane: alpha <-> ge
            const float al = (cp & 1) ? oth : val, ge = (cp & 1) ? val : oth;
            V asel;
#pragma unroll
            for (int k = 0; k < N; ++k) asel[k] = (w[u] >> k) & 1u ? ac[k] : acs[k];
            const float keep = i < e ? 1.0f : 0.0f;             // padded slots contribute nothing
            acc += (g[u] * al + asel * ge) * keep;
        }
        dv = dn;
    }
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += __shfl_xor(acc[k], off);
    return acc;
}

// ---- pull_range, second form (fp32; GAT_PULL_V2=0 selects the first one, A/B) --------------------------------------------
// The last-layer pass measured VALU-bound (rocprofv3: vector ALU busy on every cycle of the kernel, matrix pipe idle, 44 %
// of wave time waiting): 62 vector instructions per slot, 45 % of them the and / compare / select triples that turn a
// decision bit into its multiplier (1 or slope), 12 % 64-bit address arithmetic for the clamped slot index.  Here
//   * the N decision bits of a word index a table of multiplier vectors in LDS ({1 | slope}^N, 2^N entries, the same for
//     every lane: lanes with equal bits read the same address, lanes with different bits different banks — conflict-free);
//   * slots are addressed as a wave-uniform base (advanced per chunk on the scalar unit) + a per-lane byte offset that
//     is loop-invariant; nothing is clamped: the record and destination arrays carry kPullPad entries of padding, a
//     slot beyond the list is zeroed after its load (and its gather redirected to the chunk's first row);
//   * the head's other word comes from a DPP quad permute instead of a cross-lane LDS shuffle.
// Same sums in the same slot order; the two products of a slot are added one after the other (fma, fma) instead of as one sum.
template <int N> struct PullTabs { alignas(16) float m[2][1 << N][N]; };     // [0]: bit ? 1 : slope; [1]: the same times 1 / heads
template <int N>
__device__ __forceinline__ void pull_tabs_init(PullTabs<N>& T, float slope, float inv_heads) {
    for (int t = threadIdx.x; t < 2 * (1 << N) * N; t += blockDim.x) {
        const int tab = t / ((1 << N) * N), nib = (t / N) % (1 << N), k = t % N;
        const float v = (nib >> k) & 1 ? 1.0f : slope;
        T.m[tab][nib][k] = tab ? v * inv_heads : v;
    }
    __syncthreads();
}
// bit K of `bits` set ? a : b without a compare: v_bfe_i32 (one bit, sign-extended = the select mask) + v_bfi_b32.  Written
// as asm, one channel per call: hipcc folds the C form back into v_and + v_cmp + v_cndmask (three instructions), and an asm
// statement inside a loop over the channel index took element 0 of both vectors for every channel (ROCm 7.2).
template <int K>
__device__ __forceinline__ float select_bit(uint32_t bits, float a, float b) {
    uint32_t mask, sel;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mask) : "v"(bits), "n"(K));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(mask), "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, b)));
    return __builtin_bit_cast(float, sel);
}
template <int N>
__device__ __forceinline__ typename PullVec<N>::T select_bits(uint32_t bits, typename PullVec<N>::T a, typename PullVec<N>::T b) {
    typename PullVec<N>::T r;
    r[0] = select_bit<0>(bits, a[0], b[0]);
    r[1] = select_bit<1>(bits, a[1], b[1]);
    if constexpr (N > 2) { r[2] = select_bit<2>(bits, a[2], b[2]); r[3] = select_bit<3>(bits, a[3], b[3]); }
    return r;
}
// TAB: multipliers from the LDS tables (T) — else from select_bits (no LDS, no per-block table set-up)
template <int HD, int N, int LASTD, bool TAB>
__device__ __forceinline__ typename PullVec<N>::T pull_range2(const PullTabs<N>& T, const uint32_t* __restrict__ stash,
                                                              const int32_t* __restrict__ cdst, const float* __restrict__ gfull,
                                                              const uint8_t* __restrict__ hbits, int b, int e, int lane,
                                                              typename PullVec<N>::T ac, float slope) {
    using V = typename PullVec<N>::T;
    [[maybe_unused]] const V acs = ac * slope;
    [[maybe_unused]] V mhi, mlo;                      // last layer: 1 / heads and slope / heads
    if constexpr (LASTD > 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) { mhi[k] = 1.0f / (float)(HD / LASTD); mlo[k] = slope * (1.0f / (float)(HD / LASTD)); }
    }
    constexpr int LPE = HD / N, G = 64 / LPE;
    constexpr int CH = 16, U = CH / G;
    constexpr uint32_t kMask = (1u << N) - 1u;
    static_assert(U >= 1, "lane layout");
    const int cp = lane % LPE, gidx = lane / LPE;
    V acc;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    const char* sp = reinterpret_cast<const char*>(stash + (uint64_t)(uint32_t)b * LPE);      // wave-uniform
    const char* dp = reinterpret_cast<const char*>(cdst + b);
    const uint32_t doff = (uint32_t)(lane & (CH - 1)) * 4u;
    int dv = *reinterpret_cast<const int32_t*>(dp + doff);
    for (int rem = e - b; rem > 0; rem -= CH, sp += CH * LPE * 4, dp += CH * 4) {
        const int dn = *reinterpret_cast<const int32_t*>(dp + CH * 4 + doff);          // next chunk's destinations, unconditionally (padding)
        const int d0 = __builtin_amdgcn_readfirstlane(dv);                              // the chunk's first slot: always inside the list
        uint32_t w[U];
        V g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = u * G + gidx < rem;
            // unclamped (the padding keeps it legal; a slot beyond the list reads the next source's record and is zeroed below).
            // Redirecting such lanes to the chunk's first record instead cost the last layer's launch +1.1 ms (1.71 -> 2.79):
            // the per-lane offset stops being loop-invariant
            const uint32_t ww = *reinterpret_cast<const uint32_t*>(sp + (uint32_t)((u * G + gidx) * LPE + cp) * 4u);
            int drow = __shfl(dv, u * G + gidx);
            drow = valid ? drow : d0;
            w[u] = valid ? ww : 0u;
            if constexpr (LASTD > 0) {               // 64-byte node records {gH[8] | - | decision bytes at +32}: one line per slot
                const uint32_t ro = (uint32_t)drow << 6;
                const V gh4 = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + (ro + (uint32_t)((cp * N) % LASTD) * 4u));
                const uint32_t nib = *reinterpret_cast<const uint8_t*>(reinterpret_cast<const char*>(hbits) + (ro + (uint32_t)cp));
                if constexpr (TAB) g[u] = gh4 * *reinterpret_cast<const V*>(&T.m[1][nib & kMask][0]);
                else g[u] = gh4 * select_bits<N>(nib, mhi, mlo);
            } else {
                g[u] = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + ((uint32_t)drow * (uint32_t)(HD * 4) + (uint32_t)cp * (uint32_t)(N * 4)));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float val = __builtin_bit_cast(float, w[u] & ~kMask);
            const float oth = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, val), 0xB1, 0xF, 0xF, true));   // lane ^ 1
            const float al = (cp & 1) ? oth : val, ge = (cp & 1) ? val : oth;
            acc += g[u] * al;
            if constexpr (TAB) acc += (ac * ge) * *reinterpret_cast<const V*>(&T.m[0][w[u] & kMask][0]);
            else acc += select_bits<N>(w[u], ac, acs) * ge;
        }
        dv = dn;
    }
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += __shfl_xor(acc[k], off);
    return acc;
}

template <int HD, int N, bool BF, int LASTD, int V2 = 0>          // V2: 0 first form, 1 second form with LDS tables, 2 second form with bit selects
__global__ __launch_bounds__(256) void gpl_pull_kernel(const int32_t* __restrict__ src_ptr, const uint32_t* __restrict__ stash,
                                                       const int32_t* __restrict__ cdst, const float* __restrict__ gfull,
                                                       const uint8_t* __restrict__ hbits, int gh_stride, int hb_stride,
                                                       const float* __restrict__ a, float slope, float* __restrict__ gPL,
                                                       int64_t n_table, int32_t kHeavySlots) {
    using V = typename PullVec<N>::T;
    constexpr int LPE = HD / N;
    __shared__ PullTabs<N> tabs;
    if constexpr (V2 == 1) pull_tabs_init<N>(tabs, slope, LASTD > 0 ? 1.0f / (float)(HD / (LASTD > 0 ? LASTD : 1)) : 1.0f);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (s >= n_table) return;
    const int b = __builtin_amdgcn_readfirstlane(src_ptr[s]), e = __builtin_amdgcn_readfirstlane(src_ptr[s + 1]);
    if (e - b > kHeavySlots) return;                    // long lists: gpl_pull_chunk_kernel + gpl_heavy_fix_kernel
    const int cp = lane % LPE;
    V acc;
    if (b < e) {
        const V ac = *reinterpret_cast<const V*>(a + cp * N);
        if constexpr (V2 != 0) acc = pull_range2<HD, N, LASTD, V2 == 1>(tabs, stash, cdst, gfull, hbits, b, e, lane, ac, slope);
        else acc = pull_range<HD, N, BF, LASTD>(stash, cdst, gfull, hbits, gh_stride, hb_stride, slope, b, e, lane, ac, ac * slope);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.f;
    }
    if (lane < LPE) stream_store(reinterpret_cast<V*>(gPL + s * HD + cp * N), acc);
}

template <int HD, int N, bool BF, int LASTD, int V2 = 0>
__global__ __launch_bounds__(256) void gpl_pull_chunk_kernel(const int4* __restrict__ chunks, int32_t n_chunks,
                                                             const uint32_t* __restrict__ stash, const int32_t* __restrict__ cdst,
                                                             const float* __restrict__ gfull, const uint8_t* __restrict__ hbits,
                                                             int gh_stride, int hb_stride, const float* __restrict__ a,
                                                             float slope, float* __restrict__ part) {
    using V = typename PullVec<N>::T;
    constexpr int LPE = HD / N;
    __shared__ PullTabs<N> tabs;
    if constexpr (V2 == 1) pull_tabs_init<N>(tabs, slope, LASTD > 0 ? 1.0f / (float)(HD / (LASTD > 0 ? LASTD : 1)) : 1.0f);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * 4 + wave;
    if (k >= n_chunks) return;
    const int4 ch = chunks[k];                          // {first slot, end slot, partial row, -}, never empty
    const int cp = lane % LPE;
    const V ac = *reinterpret_cast<const V*>(a + cp * N);
    V acc;
    if constexpr (V2 != 0) acc = pull_range2<HD, N, LASTD, V2 == 1>(tabs, stash, cdst, gfull, hbits, __builtin_amdgcn_readfirstlane(ch.x), __builtin_amdgcn_readfirstlane(ch.y), lane, ac, slope);
    else acc = pull_range<HD, N, BF, LASTD>(stash, cdst, gfull, hbits, gh_stride, hb_stride, slope, ch.x, ch.y, lane, ac, ac * slope);
    if (lane < LPE) *reinterpret_cast<V*>(part + (int64_t)ch.z * HD + cp * N) = acc;
}

// Group-per-source form of the same pass (see edge_fwd3_kernel): a wave carries G = 64/(HD/N) source lists side by side,
// one per lane group, from a length-sorted item list {source, first slot, end slot, partial row | -1} (lists longer than
// the heavy threshold are cut into chunk items whose partial rows gpl_heavy_fix_kernel adds in chunk order).  One wave
// per source was latency-bound: four dependent memory latencies per ~25 slots (gathering 48 B instead of 256 B per edge
// did not change its time).
template <int HD, int N, bool BF, int LASTD>
__global__ __launch_bounds__(256) void gpl_pull3_kernel(const int4* __restrict__ items, int64_t n_items,
                                                        const uint32_t* __restrict__ stash, const int32_t* __restrict__ cdst,
                                                        const float* __restrict__ gfull, const uint8_t* __restrict__ hbits,
                                                        int gh_stride, int hb_stride,
                                                        const float* __restrict__ a, float slope, float* __restrict__ gPL,
                                                        float* __restrict__ part) {
    using V = typename PullVec<N>::T;
    constexpr int LPE = HD / N, G = 64 / LPE, U = 4;
    static_assert(LPE >= U, "lane layout");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int64_t it = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * G + gidx;
    int src = -1, b = 0, e = 0, pslot = -1;
    if (it < n_items) { const int4 item = items[it]; src = item.x; b = item.y; e = item.z; pslot = item.w; }
    const V ac = *reinterpret_cast<const V*>(a + cp * N);
    const V acs = ac * slope;
    V acc;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    int nst = (e - b + U - 1) / U;
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) { const int o = __shfl_xor(nst, off); nst = nst > o ? nst : o; }
    nst = __builtin_amdgcn_readfirstlane(nst);
    auto load_dst = [&](int st) {                    // lanes 0..U-1 of the group: the step's destination rows (clamped; slot 0 always exists)
        int j = b + st * U + (cp & (U - 1));
        j = j < e ? j : e - 1;
        return cdst[j > 0 ? j : 0];
    };
    int dv = load_dst(0);
    for (int st = 0; st < nst; ++st) {
        const int dn = load_dst(st + 1);
        uint32_t w[U];
        V g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int j = b + st * U + u;
            j = j < e ? j : e - 1;
            j = j > 0 ? j : 0;
            w[u] = stash[(uint64_t)(uint32_t)j * LPE + cp];
            const uint32_t drow = (uint32_t)__shfl(dv, gidx * LPE + u);
            if constexpr (LASTD > 0) {
                constexpr float inv_heads = 1.0f / (float)(HD / LASTD);
                const V gh4 = *reinterpret_cast<const V*>(gfull + (size_t)drow * gh_stride + (cp * N) % LASTD);
                const uint32_t nib = hbits[(size_t)drow * hb_stride + cp];
#pragma unroll
                for (int k = 0; k < N; ++k) g[u][k] = gh4[k] * ((nib >> k) & 1u ? 1.0f : slope) * inv_heads;
            } else if constexpr (BF) {
                const char* p = reinterpret_cast<const char*>(gfull) + drow * (uint32_t)(HD * 2) + (uint32_t)cp * (uint32_t)(N * 2);
                if constexpr (N == 2) {
                    const uint32_t w2 = *reinterpret_cast<const uint32_t*>(p);
                    g[u][0] = __builtin_bit_cast(float, w2 << 16); g[u][1] = __builtin_bit_cast(float, w2 & 0xFFFF0000u);
                } else {
                    const uint2 w2 = *reinterpret_cast<const uint2*>(p);
                    g[u][0] = __builtin_bit_cast(float, w2.x << 16); g[u][1] = __builtin_bit_cast(float, w2.x & 0xFFFF0000u);
                    g[u][2] = __builtin_bit_cast(float, w2.y << 16); g[u][3] = __builtin_bit_cast(float, w2.y & 0xFFFF0000u);
                }
            } else {
                const uint32_t off = drow * (uint32_t)(HD * 4) + (uint32_t)cp * (uint32_t)(N * 4);
#ifdef GAT_NT_GATHER
                g[u] = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + off));
#else
                g[u] = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + off);
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float val = __builtin_bit_cast(float, w[u] & ~((1u << N) - 1u));
            const float oth = __shfl_xor(val, 1);               // the head's other lane: alpha <-> ge
            const float al = (cp & 1) ? oth : val, ge = (cp & 1) ? val : oth;
            V asel;
#pragma unroll
            for (int k = 0; k < N; ++k) asel[k] = (w[u] >> k) & 1u ? ac[k] : acs[k];
            const float keep = (b + st * U + u < e) ? 1.0f : 0.0f;          // padded slots contribute nothing
            acc += (g[u] * al + asel * ge) * keep;
        }
        dv = dn;
    }
    if (src < 0) return;
    float* dst = pslot < 0 ? gPL + (int64_t)src * HD : part + (int64_t)pslot * HD;
    stream_store(reinterpret_cast<V*>(dst + cp * N), acc);
}

// ------------------------------------------------------------------------------------------------------------
// Slot-parallel form of the source-major pass ("runs").  Every kernel above hands a source LIST to a wave or a lane
// group: per list one descriptor load, then the destination indices, then the gathers — three dependent memory
// latencies per list, and a pass over a destination-range shard (deg/P slots per source: ~4 at P = 8) is as long
// as (number of lists) x (latency) / (waves in flight), whatever it moves.  Here the FLAT slot stream is cut into runs of
// `run` consecutive slots, one run per lane group (64/(HD/N) runs side by side per wave), whatever sources they belong
// to:
//   * no per-list descriptor: the group streams csc_src[slot] (4 B per slot) beside the records and destinations;
//     a slot whose successor has another source ends a segment;
//   * a segment that lies inside its run is stored straight to gPL[source]; a run's first segment when it began in
//     an earlier run goes to partial row 2r, its last segment when it continues goes to partial row 2r + 1 (a run
//     inside one long list: one partial, 2r + 1); gpl_runs_fix_kernel adds the partial rows of every list that
//     crosses a run boundary in ascending run order.  Fixed cut points, fixed order: bitwise reproducible;
//   * table rows without slots are zeroed by extra blocks at the front of the same grid (list built once per graph);
//   * balance is perfect at any degree distribution: no heavy-list chunks, no length-sorted item list.
// Per slot the messages are rebuilt exactly as in pull_range2 (E:859-869).  Index words reach their lanes by DPP
// row broadcasts (16-lane groups) instead of LDS shuffles.
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <int U, int LPE>
__device__ __forceinline__ int group_bcast(int v, int gidx) {       // lane U of the caller's group
    if constexpr (LPE == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, false);        // row_newbcast:U
    else if constexpr (LPE == 4) return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xF, 0xF, false);     // quad_perm:[U,U,U,U]
    else return __shfl(v, gidx * LPE + U);
}

// below this many slots the runs cannot fill the chip (Pubmed shape: 44 k slots = 173 waves; 0.244 -> 0.268 ms per step with
// runs, Cora shape 0.190 -> 0.216) and the list-per-wave kernels stay; Arxiv shape (1.17 M slots): 0.256 -> 0.195 ms per step
constexpr int64_t kRunsMinSlots = 512 << 10;
struct RunsDev {
    const int32_t* csrc;       // [n_slots + kPullPad] table row of every slot; -1 behind the last
    const int32_t* empty;      // [n_empty] table rows without slots
    float* part;               // [2 n_runs][HD]
    int64_t n_empty, n_slots, n_table;
    int32_t run, zero_blocks;
};

// index words of slot t of the group's run (one lane per slot): d = destination row (gather index; cdst may be null),
// code = bit 31: a segment ends here | row it is stored to (gPL row, or n_table + partial row)
__device__ __forceinline__ void runs_index(const RunsDev& rd, const int32_t* __restrict__ cdst, int64_t jb, int64_t r, int cnt, int sb, int dfirst,
                                           int t, int& d, int& code) {
    const bool valid = t < cnt;
    const int64_t j = jb + (valid ? t : 0);
    const int dv = cdst ? cdst[j] : 0, s0 = rd.csrc[j], s1 = rd.csrc[j + 1];
    const bool last = t == cnt - 1;
    const bool endf = valid && (s0 != s1 || last);
    const int prow = (int)rd.n_table + 2 * (int)r;
    const int dest = (last && s0 == s1) ? prow + 1 : (s0 == sb ? prow : s0);
    d = valid ? dv : dfirst;
    code = endf ? (int)(0x80000000u | (uint32_t)dest) : 0;
}

template <int HD, int N, bool BF, int LASTD, int U>
__global__ __launch_bounds__(256) void gpl_pull_runs_kernel(const uint32_t* __restrict__ stash, const int32_t* __restrict__ cdst,
                                                            const float* __restrict__ gfull, const uint8_t* __restrict__ hbits,
                                                            const float* __restrict__ a, float slope, float* __restrict__ gPL, RunsDev rd) {
    using V = typename PullVec<N>::T;
    constexpr int LPE = HD / N, G = 64 / LPE;
    constexpr uint32_t kMask = (1u << N) - 1u;
    static_assert(LPE % U == 0, "batches of U slots");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    V acc;
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    if ((int)blockIdx.x < rd.zero_blocks) {                 // rows without slots: one row per lane group and step
        constexpr int kSteps = 8;
        const int64_t k0 = (int64_t)blockIdx.x * (4 * G * kSteps) + (wave * G + gidx);
#pragma unroll
        for (int it = 0; it < kSteps; ++it) {
            const int64_t k = k0 + (int64_t)it * 4 * G;
            if (k < rd.n_empty) stream_store(reinterpret_cast<V*>(gPL + (int64_t)rd.empty[k] * HD + cp * N), acc);
        }
        return;
    }
    const int64_t wv = (int64_t)((int)blockIdx.x - rd.zero_blocks) * 4 + wave;
    const int64_t r = wv * G + gidx;                        // this group's run
    const int64_t j0 = r * rd.run, left = rd.n_slots - j0;
    const int cnt = left <= 0 ? 0 : (left < rd.run ? (int)left : rd.run);
    const int nT = (__builtin_amdgcn_readfirstlane(cnt) + LPE - 1) / LPE;     // group 0 holds the wave's longest run
    if (nT == 0) return;
    const int64_t jb = cnt > 0 ? j0 : 0;                    // a group without slots reads (and ignores) slot 0
    const int sb = (cnt > 0 && j0 > 0) ? rd.csrc[j0 - 1] : -1;               // source of the slot before the run
    const int dfirst = cdst[jb];
    const V ac = *reinterpret_cast<const V*>(a + cp * N);
    [[maybe_unused]] const V acs = ac * slope;
    [[maybe_unused]] V mhi, mlo;                              // last layer: 1 / heads and slope / heads
    if constexpr (LASTD > 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) { mhi[k] = 1.0f / (float)(HD / LASTD); mlo[k] = slope * (1.0f / (float)(HD / LASTD)); }
    }
    const char* sp = reinterpret_cast<const char*>(stash + ((uint64_t)jb * LPE + cp));
    int d, code;
    runs_index(rd, cdst, jb, r, cnt, sb, dfirst, cp, d, code);
    for (int T = 0; T < nT; ++T, sp += LPE * LPE * 4) {
        int dn = 0, coden = 0;
        if (T + 1 < nT) runs_index(rd, cdst, jb, r, cnt, sb, dfirst, (T + 1) * LPE + cp, dn, coden);
        if (T * LPE < cnt) {                                // only the launch's last wave has groups that skip
            const int rem = cnt - T * LPE;
            static_for(std::make_integer_sequence<int, LPE / U>{}, [&](auto bc) {
                constexpr int b = decltype(bc)::value;
                uint32_t w[U];
                V g[U];
                static_for(std::make_integer_sequence<int, U>{}, [&](auto uc) {
                    constexpr int u = decltype(uc)::value, slot = b * U + u;
                    const uint32_t ww = *reinterpret_cast<const uint32_t*>(sp + slot * LPE * 4);     // unclamped: kPullPad
                    w[u] = slot < rem ? ww : 0u;
                    const uint32_t drow = (uint32_t)group_bcast<slot, LPE>(d, gidx);
                    if constexpr (LASTD > 0) {               // 64-byte node records {gH[8] | - | decision bytes at +32}
                        const uint32_t ro = drow << 6;
                        const V gh4 = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + (ro + (uint32_t)((cp * N) % LASTD) * 4u));
                        const uint32_t nib = *reinterpret_cast<const uint8_t*>(reinterpret_cast<const char*>(hbits) + (ro + (uint32_t)cp));
                        g[u] = gh4 * select_bits<N>(nib, mhi, mlo);
                    } else if constexpr (BF) {
                        const char* p = reinterpret_cast<const char*>(gfull) + (drow * (uint32_t)(HD * 2) + (uint32_t)cp * (uint32_t)(N * 2));
                        if constexpr (N == 2) {
                            const uint32_t w2 = *reinterpret_cast<const uint32_t*>(p);
                            g[u][0] = __builtin_bit_cast(float, w2 << 16); g[u][1] = __builtin_bit_cast(float, w2 & 0xFFFF0000u);
                        } else {
                            const uint2 w2 = *reinterpret_cast<const uint2*>(p);
                            g[u][0] = __builtin_bit_cast(float, w2.x << 16); g[u][1] = __builtin_bit_cast(float, w2.x & 0xFFFF0000u);
                            g[u][2] = __builtin_bit_cast(float, w2.y << 16); g[u][3] = __builtin_bit_cast(float, w2.y & 0xFFFF0000u);
                        }
                    } else {
                        g[u] = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(gfull) + (drow * (uint32_t)(HD * 4) + (uint32_t)cp * (uint32_t)(N * 4)));
                    }
                });
                static_for(std::make_integer_sequence<int, U>{}, [&](auto uc) {
                    constexpr int u = decltype(uc)::value, slot = b * U + u;
                    const float val = __builtin_bit_cast(float, w[u] & ~kMask);
                    const float oth = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, val), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                    const float al = (cp & 1) ? oth : val, ge = (cp & 1) ? val : oth;
                    acc += g[u] * al;
                    acc += select_bits<N>(w[u], ac, acs) * ge;
                    const int cw = group_bcast<slot, LPE>(code, gidx);
                    if (cw < 0) {                           // the segment ends with this slot
                        const uint32_t row = (uint32_t)cw & 0x7fffffffu;
                        float* dstp = row >= (uint32_t)rd.n_table ? rd.part + (size_t)(row - (uint32_t)rd.n_table) * HD : gPL + (size_t)row * HD;
                        stream_store(reinterpret_cast<V*>(dstp + cp * N), acc);
#pragma unroll
                        for (int i = 0; i < N; ++i) acc[i] = 0.f;
                    }
                });
            });
        }
        d = dn; code = coden;
    }
}

// The same slot-parallel walk for MESSAGE rows (layers without a record path, bf16 storage at H*D < 64: BASELINE config 5): no
// gather, the group just streams its run's rows and sums per segment.  A row is LPR lanes x 16 bytes (4 fp32 / 8 bf16 channels
// per lane); a lane holds the index words of K slots per step so that even a 4-lane group has K*LPR loads in flight.
template <int HD, bool BF>
__global__ __launch_bounds__(256) void gpl_sum_runs_kernel(const float* __restrict__ msg, float* __restrict__ gPL, RunsDev rd) {
    constexpr int CPL = BF ? 8 : 4, LPR = HD / CPL, G = 64 / LPR;
    constexpr int SS = LPR >= 8 ? LPR : (LPR == 4 ? 16 : 8), K = SS / LPR, U = SS < 8 ? SS : 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane % LPR, gidx = lane / LPR;
    float acc[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
    auto store_row = [&](float* rowp) {
        float4* o = reinterpret_cast<float4*>(rowp + q * CPL);
        stream_store(o, make_float4(acc[0], acc[1], acc[2], acc[3]));
        if constexpr (BF) stream_store(o + 1, make_float4(acc[4], acc[5], acc[6], acc[7]));
    };
    if ((int)blockIdx.x < rd.zero_blocks) {
        constexpr int kSteps = 8;
        const int64_t k0 = (int64_t)blockIdx.x * (4 * G * kSteps) + (wave * G + gidx);
#pragma unroll
        for (int it = 0; it < kSteps; ++it) {
            const int64_t k = k0 + (int64_t)it * 4 * G;
            if (k < rd.n_empty) store_row(gPL + (int64_t)rd.empty[k] * HD);
        }
        return;
    }
    const int64_t wv = (int64_t)((int)blockIdx.x - rd.zero_blocks) * 4 + wave;
    const int64_t r = wv * G + gidx;
    const int64_t j0 = r * rd.run, left = rd.n_slots - j0;
    const int cnt = left <= 0 ? 0 : (left < rd.run ? (int)left : rd.run);
    const int nT = (__builtin_amdgcn_readfirstlane(cnt) + SS - 1) / SS;
    if (nT == 0) return;
    const int64_t jb = cnt > 0 ? j0 : 0;
    const int sb = (cnt > 0 && j0 > 0) ? rd.csrc[j0 - 1] : -1;
    const char* mp = reinterpret_cast<const char*>(msg) + ((uint64_t)jb * LPR + q) * 16u;
    int code[K], coden[K], dummy;
#pragma unroll
    for (int k = 0; k < K; ++k) runs_index(rd, nullptr, jb, r, cnt, sb, 0, k * LPR + q, dummy, code[k]);
    for (int T = 0; T < nT; ++T, mp += SS * LPR * 16) {
#pragma unroll
        for (int k = 0; k < K; ++k) coden[k] = 0;
        if (T + 1 < nT) {
#pragma unroll
            for (int k = 0; k < K; ++k) runs_index(rd, nullptr, jb, r, cnt, sb, 0, (T + 1) * SS + k * LPR + q, dummy, coden[k]);
        }
        if (T * SS < cnt) {
            const int rem = cnt - T * SS;
            static_for(std::make_integer_sequence<int, SS / U>{}, [&](auto bc) {
                constexpr int b = decltype(bc)::value;
                uint4 v[U];
                static_for(std::make_integer_sequence<int, U>{}, [&](auto uc) {
                    constexpr int u = decltype(uc)::value, slot = b * U + u;
                    const uint4 vv = *reinterpret_cast<const uint4*>(mp + slot * LPR * 16);      // unclamped: kPullPad rows behind the last
                    v[u] = slot < rem ? vv : make_uint4(0u, 0u, 0u, 0u);
                });
                static_for(std::make_integer_sequence<int, U>{}, [&](auto uc) {
                    constexpr int u = decltype(uc)::value, slot = b * U + u;
                    const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if constexpr (BF) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[2 * k] += __builtin_bit_cast(float, w[k] << 16);
                            acc[2 * k + 1] += __builtin_bit_cast(float, w[k] & 0xFFFF0000u);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += __builtin_bit_cast(float, w[k]);
                    }
                    const int cw = group_bcast<slot % LPR, LPR>(code[slot / LPR], gidx);
                    if (cw < 0) {
                        const uint32_t row = (uint32_t)cw & 0x7fffffffu;
                        store_row(row >= (uint32_t)rd.n_table ? rd.part + (size_t)(row - (uint32_t)rd.n_table) * HD : gPL + (size_t)row * HD);
#pragma unroll
                        for (int i = 0; i < CPL; ++i) acc[i] = 0.f;
                    }
                });
            });
        }
#pragma unroll
        for (int k = 0; k < K; ++k) code[k] = coden[k];
    }
}

// open: {source, first run, last run, -}: partial rows 2*r0 + 1, 2*r + 1 for r0 < r < r1, 2*r1, added in that order
__global__ __launch_bounds__(256) void gpl_runs_fix_kernel(const int4* __restrict__ open, int32_t n_open, const float* __restrict__ part,
                                                           float* __restrict__ gPL, int32_t Q /* HD / 4 */) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = t / Q;
    const int q = (int)(t % Q);
    if (k >= n_open) return;
    const int4 o = open[k];
    const float4* p4 = reinterpret_cast<const float4*>(part);
    float4 acc = p4[(int64_t)(2 * o.y + 1) * Q + q];
    int r = o.y + 1;
    for (; r + 4 <= o.z; r += 4) {                           // loads of four runs in flight, added in run order
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p4[(int64_t)(2 * (r + i) + 1) * Q + q];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    for (; r < o.z; ++r) { const float4 v = p4[(int64_t)(2 * r + 1) * Q + q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    { const float4 v = p4[(int64_t)(2 * o.z) * Q + q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    reinterpret_cast<float4*>(gPL)[(int64_t)o.x * Q + q] = acc;
}

// cdst[pos[e]] = row of CSR edge e (binary search in row_ptr, as csr_to_coo_kernel)
__global__ __launch_bounds__(256) void csc_dst_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ pos,
                                                     int32_t* __restrict__ cdst, int64_t n_rows, int64_t n_edges) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
        int64_t lo = 0, hi = n_rows;          // invariant: row_ptr[lo] <= e < row_ptr[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)row_ptr[mid] <= e) lo = mid; else hi = mid;
        }
        cdst[pos[e]] = (int32_t)lo;
    }
}

}  // namespace

int build_csc_dst(const int32_t* row_ptr, const int32_t* pos, int32_t* cdst, int64_t n_rows, int64_t n_edges, hipStream_t s) {
    if (n_edges <= 0) return 0;
    const int64_t blocks = std::min<int64_t>((n_edges + 255) / 256, 65536);
    hipLaunchKernelGGL(csc_dst_kernel, dim3((unsigned)blocks), dim3(256), 0, s, row_ptr, pos, cdst, n_rows, n_edges);
    GAT_HIP(hipGetLastError());
    return 0;
}

template <int HD, int N, bool BF, int LASTD>
static int run_pull(const int32_t* src_ptr, const uint32_t* stash, const int32_t* cdst, const float* gfull, const uint8_t* hbits,
                    int gh_stride, int hb_stride, const float* a, float slope, float* gPL, int64_t n_table, const int4* chunks, int32_t n_chunks, const int4* heavy,
                    int32_t n_heavy, float* part, int wpb, int64_t n_slots, const int4* items, int64_t n_items, const SlotRuns* runs, bool padded, hipStream_t s) {
    // Slot-parallel form (gpl_pull_runs_kernel): default where the lists are short (a destination-range shard: the list-per-group
    // kernel below pays three dependent latencies per ~4 slots); GAT_PULL_RUNS=0|1 forces.  The last layer's variant reads the
    // 64-byte node records (gh_stride 16, decision bytes at +32).
    static const int runs_env = [] { const char* e = choice_env("GAT_PULL_RUNS"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    static const int groups_env = [] { const char* e = choice_env("GAT_PULL_GROUPS"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool runs_ok = runs != nullptr && runs->csrc != nullptr && runs->run % (HD / N) == 0 && (LASTD == 0 || (gh_stride == 16 && hb_stride == 64));
    if (runs_ok && (runs_env >= 0 ? runs_env == 1 : (groups_env < 0 && n_slots < 8 * n_table && n_slots >= kRunsMinSlots))) {     // an explicit GAT_PULL_GROUPS selects among the list kernels
        constexpr int G = 64 / (HD / N), U = (HD / N) % 8 == 0 ? 8 : 4;
        RunsDev rd{};
        rd.csrc = runs->csrc; rd.empty = runs->empty; rd.part = runs->part; rd.n_empty = runs->n_empty; rd.n_slots = n_slots; rd.n_table = n_table;
        rd.run = runs->run; rd.zero_blocks = (int32_t)((runs->n_empty + 4 * G * 8 - 1) / (4 * G * 8));
        const int64_t waves = (runs->n_runs + G - 1) / G, blocks = rd.zero_blocks + (waves + 3) / 4;
        static const int u_env = [] { const char* e = choice_env("GAT_PULL_U"); return e ? atoi(e) : 0; }();      // experiment: slots per batch (16-lane groups)
        if (blocks > 0) {
            if constexpr (HD / N == 16 && !BF) {
                if (u_env == 4) hipLaunchKernelGGL((gpl_pull_runs_kernel<HD, N, BF, LASTD, 4>), dim3((unsigned)blocks), dim3(256), 0, s, stash, cdst, gfull, hbits, a, slope, gPL, rd);
                else if (u_env == 16) hipLaunchKernelGGL((gpl_pull_runs_kernel<HD, N, BF, LASTD, 16>), dim3((unsigned)blocks), dim3(256), 0, s, stash, cdst, gfull, hbits, a, slope, gPL, rd);
                else hipLaunchKernelGGL((gpl_pull_runs_kernel<HD, N, BF, LASTD, U>), dim3((unsigned)blocks), dim3(256), 0, s, stash, cdst, gfull, hbits, a, slope, gPL, rd);
            } else {
                hipLaunchKernelGGL((gpl_pull_runs_kernel<HD, N, BF, LASTD, U>), dim3((unsigned)blocks), dim3(256), 0, s, stash, cdst, gfull, hbits, a, slope, gPL, rd);
            }
        }
        if (runs->n_open > 0) {
            const int64_t threads = (int64_t)runs->n_open * (HD / 4);
            hipLaunchKernelGGL(gpl_runs_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, runs->open, runs->n_open, runs->part, gPL, HD / 4);
        }
        GAT_HIP(hipGetLastError());
        return 0;
    }
    // GAT_PULL_GROUPS=1: the group-per-source kernel.  Measured SLOWER on the Products shape (5.8 vs 5.36 ms per step), as was
    // everything else that shortened this pass's dependency chains or shrank its gathers: see DESIGN §4 (random-row rate)
    // default: groups only where the lists are short (a destination-range shard sees ~deg/P slots per source: one wave
    // per 3-slot list wastes 15 of its 16 gather slots) — the same rule the message-row sum used (n_slots < 8 n_table)
    // ... and only with enough lists to give every resident wave several (Pubmed shape, 19,717 lists: one wave per list 0.27 vs 0.30 ms per step)
    const bool groups = groups_env >= 0 ? groups_env == 1 : (n_slots < 8 * n_table && n_items >= 32768);
    if (items != nullptr && groups) {
        constexpr int G = 64 / (HD / N);
        const int64_t quads = (n_items + G - 1) / G;
        hipLaunchKernelGGL((gpl_pull3_kernel<HD, N, BF, LASTD>), dim3((unsigned)((quads + 3) / 4)), dim3(256), 0, s, items, n_items, stash, cdst,
                           gfull, hbits, gh_stride, hb_stride, a, slope, gPL, part);
        if (n_heavy > 0) {
            const int64_t threads = (int64_t)n_heavy * HD;
            hipLaunchKernelGGL(gpl_heavy_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, heavy, n_heavy, part, gPL, HD);
        }
        GAT_HIP(hipGetLastError());
        return 0;
    }
    // second form of the slot walk (pull_range2): fp32 tables; the last-layer variant needs the 64-byte node records.
    // GAT_PULL_V2 = 0 first form | 1 second form, multipliers from LDS tables | 2 second form, multipliers by bit selects (A/B)
    // Per launch on the Products shape (rocprofv3, same box): first form 2.25 ms hidden layer / 1.95 ms last layer; second form
    // with bit selects 2.45 / 1.71; with LDS tables slower on both (every 64-thread block pays the table set-up, and an LDS round
    // trip sits behind every record load).  The last layer's walk is vector-ALU-bound (its rows are one cache line: 80-100 % VALU
    // busy) and gains from the shorter instruction stream; the hidden layer's is bound by its two-line gathers and loses to the
    // second form's unclamped chunk reads (+8 % record lines).  Default: second form for the last layer only; GAT_PULL_V2 forces
    // one form on both (0 | 1 | 2).
    static const int v2_env = [] { const char* e = choice_env("GAT_PULL_V2"); return e ? atoi(e) : -1; }();
    const int v2_want = v2_env >= 0 ? v2_env : (LASTD > 0 ? 2 : 0);
    const int v2 = (padded && !BF && (LASTD == 0 || (gh_stride == 16 && hb_stride == 64))) ? v2_want : 0;
    const dim3 cgrid((unsigned)((n_chunks + 3) / 4)), pgrid((unsigned)((n_table + wpb - 1) / wpb)), pblock(64 * wpb);
#define GAT_PULL_LAUNCH(V2_)                                                                                                          \
    do {                                                                                                                              \
        if (n_heavy > 0) {                             /* long lists first: they are the longest-running waves */                      \
            hipLaunchKernelGGL((gpl_pull_chunk_kernel<HD, N, BF, LASTD, V2_>), cgrid, dim3(256), 0, s, chunks, n_chunks, stash, cdst, gfull, hbits, \
                               gh_stride, hb_stride, a, slope, part);                                                                  \
            const int64_t threads = (int64_t)n_heavy * HD;                                                                              \
            hipLaunchKernelGGL(gpl_heavy_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, heavy, n_heavy, part, gPL, HD); \
        }                                                                                                                             \
        hipLaunchKernelGGL((gpl_pull_kernel<HD, N, BF, LASTD, V2_>), pgrid, pblock, 0, s, src_ptr, stash, cdst, gfull, hbits, gh_stride, hb_stride, \
                           a, slope, gPL, n_table, heavy_slots(n_slots));                                                              \
    } while (0)
    if constexpr (!BF) {
        if (v2 == 1) GAT_PULL_LAUNCH(1);
        else if (v2 == 2) GAT_PULL_LAUNCH(2);
        else GAT_PULL_LAUNCH(0);
    } else {
        GAT_PULL_LAUNCH(0);
    }
#undef GAT_PULL_LAUNCH
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_gpl_pull(const int32_t* src_ptr, const uint32_t* stash, const int32_t* cdst, const float* gfull, bool g_bf16,
                    const float* gh, const uint8_t* hbits, int32_t gh_stride, int32_t hb_stride, const float* a,
                    float slope, float* gPL, int64_t n_table, int64_t n_slots, int32_t H, int32_t D, const int4* chunks,
                    int32_t n_chunks, const int4* heavy, int32_t n_heavy, float* part, const int4* items, int64_t n_items, const SlotRuns* runs,
                    int64_t slot_capacity, hipStream_t s) {
    if (n_table <= 0) return 0;
    // the second form of the slot walk and the slot-parallel form read whole 16-slot chunks without clamping: the record buffer and
    // the destination list must hold kPullPad slots behind the last (gat_internal.h); a caller with exact-size buffers gets the
    // clamped first form on per-list kernels instead of an out-of-bounds read
    const bool padded = slot_capacity >= n_slots + kPullPad;
    if (!padded) runs = nullptr;
    // one wave per block for the pull kernel (GAT_GPL_WAVES=1|2|4): 5.03 / 5.11 / 5.30 ms per step on the Products shape — a
    // 4-wave block lives as long as its longest list; the message-row sum (launch_gpl_sum) measured the other way round
    static const int wpb = [] { const char* e = choice_env("GAT_GPL_WAVES"); const int v = e ? atoi(e) : 1; return (v == 1 || v == 2 || v == 4) ? v : 1; }();
    const int HD = H * D;
    const bool last = gh != nullptr && hbits != nullptr;
#define PULL_ARGS(G_) src_ptr, stash, cdst, G_, hbits, gh_stride, hb_stride, a, slope, gPL, n_table, chunks, n_chunks, heavy, n_heavy, part, wpb, n_slots, items, n_items, runs, padded, s
#define PULL(HD_, N_, D_)                                                                           \
    {                                                                                               \
        if (last) return run_pull<HD_, N_, false, D_>(PULL_ARGS(gh));                               \
        return g_bf16 ? run_pull<HD_, N_, true, 0>(PULL_ARGS(gfull)) : run_pull<HD_, N_, false, 0>(PULL_ARGS(gfull)); \
    }
    if (D == 8 && HD == 64) PULL(64, 4, 8)
    if (D == 8 && HD == 32) PULL(32, 4, 8)
    if (D == 4 && HD == 64) PULL(64, 2, 4)
    if (D == 4 && HD == 32) PULL(32, 2, 4)
    if (D == 4 && HD == 16) PULL(16, 2, 4)
    if (D == 4 && HD == 8) PULL(8, 2, 4)
#undef PULL
#undef PULL_ARGS
    return fail(GAT_E_UNSUPPORTED, "gpl_pull: no stash path for this (H, D)");
}

int build_csc(const int32_t* col_idx, int64_t n_edges, int64_t n_table, int32_t* pos, int32_t* src_ptr,
              int32_t* csrc, hipStream_t s) {
    if (csrc) GAT_HIP(hipMemsetAsync(csrc + std::max<int64_t>(n_edges, 0), 0xFF, (size_t)kPullPad * sizeof(int32_t), s));      // -1: no source
    if (n_edges <= 0) {
        GAT_HIP(hipMemsetAsync(src_ptr, 0, (size_t)(n_table + 1) * sizeof(int32_t), s));
        return 0;
    }
    int32_t *iota = nullptr, *keys_out = nullptr, *perm = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    int end_bit = 1;
    while (((int64_t)1 << end_bit) < n_table) ++end_bit;
    int rc = 0;
    auto cleanup = [&]() { (void)hipFree(iota); (void)hipFree(keys_out); (void)hipFree(perm); (void)hipFree(temp); };
#define CSC_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); return fail((int)e__, std::string(#x) + ": " + hipGetErrorString(e__)); } } while (0)
    CSC_HIP(hipMalloc((void**)&iota, n_edges * sizeof(int32_t)));
    CSC_HIP(hipMalloc((void**)&keys_out, n_edges * sizeof(int32_t)));
    CSC_HIP(hipMalloc((void**)&perm, n_edges * sizeof(int32_t)));
    int64_t blocks = std::min<int64_t>((n_edges + 255) / 256, 65536);
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)blocks), dim3(256), 0, s, iota, n_edges);
    CSC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, col_idx, keys_out, iota, perm, (int)n_edges, 0,
                                               end_bit, s));
    CSC_HIP(hipMalloc(&temp, temp_bytes));
    CSC_HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, col_idx, keys_out, iota, perm, (int)n_edges, 0,
                                               end_bit, s));
    hipLaunchKernelGGL(invert_perm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, perm, pos, n_edges);
    if (csrc) CSC_HIP(hipMemcpyAsync(csrc, keys_out, (size_t)n_edges * sizeof(int32_t), hipMemcpyDeviceToDevice, s));      // the sorted keys = source of every slot
    const int64_t sblocks = std::min<int64_t>((n_table + 1 + 255) / 256, 65536);
    hipLaunchKernelGGL(segment_offsets_kernel, dim3((unsigned)sblocks), dim3(256), 0, s, keys_out, n_edges, src_ptr,
                       n_table);
    CSC_HIP(hipGetLastError());
    CSC_HIP(hipStreamSynchronize(s));
#undef CSC_HIP
    cleanup();
    return rc;
}

int build_heavy_list(const int32_t* d_src_ptr, int64_t n_table, int64_t n_edges, HeavyList* out, hipStream_t s, int32_t run) {
    out->chunks.clear(); out->heavy.clear();
    out->run = 0; out->n_runs = 0; out->open.clear(); out->empty.clear();
    if (n_table <= 0) return 0;
    std::vector<int32_t> ptr((size_t)n_table + 1);
    GAT_HIP(hipMemcpyAsync(ptr.data(), d_src_ptr, ptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    GAT_HIP(hipStreamSynchronize(s));
    // slot-parallel form: runs of `run` consecutive slots — the lists that cross a run boundary (their partial rows are added by
    // gpl_runs_fix_kernel) and the table rows without slots (zeroed by the runs kernel's first blocks)
    if (run > 0 && n_table + 2 * ((n_edges + run - 1) / run) + 2 < 0x7fffffffLL) {
        out->run = run; out->n_runs = (n_edges + run - 1) / run;
        for (int64_t src = 0; src < n_table; ++src) {
            const int32_t b = ptr[(size_t)src], e = ptr[(size_t)src + 1];
            if (e == b) { out->empty.push_back((int32_t)src); continue; }
            const int32_t r0 = b / run, r1 = (e - 1) / run;
            if (r0 != r1) out->open.insert(out->open.end(), {(int32_t)src, r0, r1, 0});
        }
    }
    const int kHeavySlots = heavy_slots(n_edges);
    out->threshold = kHeavySlots;
    out->items.clear();
    for (int64_t src = 0; src < n_table; ++src) {
        const int32_t b = ptr[(size_t)src], e = ptr[(size_t)src + 1];
        if (e - b <= kHeavySlots) continue;
        const int32_t first = (int32_t)(out->chunks.size() / 4);
        int32_t n = 0;
        for (int32_t cb = b; cb < e; cb += kHeavySlots, ++n) {
            out->chunks.insert(out->chunks.end(), {cb, std::min(e, cb + kHeavySlots), first + n, 0});
            out->items.insert(out->items.end(), {(int32_t)src, cb, std::min(e, cb + kHeavySlots), first + n});    // longest items first
        }
        out->heavy.insert(out->heavy.end(), {(int32_t)src, first, n, 0});
    }
    // the other sources: one item each, sorted by list length (longest first) inside windows of consecutive sources
    // (counting sort; the group-per-source kernel puts neighbours of the list side by side in one wave)
    {
        const int64_t win = 4096;
        std::vector<int64_t> cnt((size_t)kHeavySlots + 2);
        for (int64_t s0 = 0; s0 < n_table; s0 += win) {
            const int64_t s1 = std::min(n_table, s0 + win);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t sr = s0; sr < s1; ++sr) { const int32_t d = ptr[(size_t)sr + 1] - ptr[(size_t)sr]; if (d <= kHeavySlots) ++cnt[(size_t)(kHeavySlots - d) + 1]; }
            for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
            const size_t base = out->items.size();
            out->items.resize(base + 4 * (size_t)cnt.back());
            for (int64_t sr = s0; sr < s1; ++sr) {
                const int32_t b = ptr[(size_t)sr], e = ptr[(size_t)sr + 1];
                if (e - b > kHeavySlots) continue;
                int32_t* it = &out->items[base + 4 * (size_t)cnt[(size_t)(kHeavySlots - (e - b))]++];
                it[0] = (int32_t)sr; it[1] = b; it[2] = e; it[3] = -1;
            }
        }
    }
    return 0;
}

template <int HD>
static int run_heavy(const float* msg, float* gPL, bool bf, const int4* chunks, int32_t n_chunks, const int4* heavy,
                     int32_t n_heavy, float* part, hipStream_t s) {
    const dim3 grid((unsigned)((n_chunks + 3) / 4)), block(256);
    if (bf) hipLaunchKernelGGL((gpl_chunk_kernel<HD, true>), grid, block, 0, s, chunks, n_chunks, msg, part);
    else hipLaunchKernelGGL((gpl_chunk_kernel<HD, false>), grid, block, 0, s, chunks, n_chunks, msg, part);
    const int64_t threads = (int64_t)n_heavy * HD;
    hipLaunchKernelGGL(gpl_heavy_fix_kernel, dim3((unsigned)((threads + 255) / 256)), block, 0, s, heavy, n_heavy, part, gPL, HD);
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_gpl_sum(const int32_t* src_ptr, const float* msg, float* gPL, int64_t n_table, int64_t n_slots,
                   int32_t HD, bool msg_bf16, const int4* chunks, int32_t n_chunks, const int4* heavy,
                   int32_t n_heavy, float* part, hipStream_t s, const SlotRuns* runs) {
    if (n_table <= 0) return 0;
    // slot-parallel form (gpl_sum_runs_kernel): the default on short lists (a destination-range shard) of fp32 rows; GAT_PULL_RUNS=0|1
    // forces.  Not for bf16 rows: a 64-byte row is FOUR lanes, and a segment end is a store instruction of its own per group (two
    // half-line requests per 128-byte gPL row), where the list kernel's 16 groups store 16 consecutive rows with one instruction —
    // BASELINE config 5's P = 8 shard measured 1.49-1.54 ms per step against 1.31 (profiles/r04/experiments)
    static const int runs_env = [] { const char* e = choice_env("GAT_PULL_RUNS"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    static const bool group_env = choice_env("GAT_GPL_GROUP") != nullptr || choice_env("GAT_GPL_BF16_GROUP") != nullptr;
    if (runs != nullptr && runs->csrc != nullptr && runs->run % 16 == 0 && (HD == 64 || HD == 32 || HD == 16 || HD == 8) &&
        (runs_env >= 0 ? runs_env == 1 : (!group_env && !msg_bf16 && n_slots < 8 * n_table && n_slots >= kRunsMinSlots))) {
        const int G = 64 / (HD / (msg_bf16 ? 8 : 4));
        RunsDev rd{};
        rd.csrc = runs->csrc; rd.empty = runs->empty; rd.part = runs->part; rd.n_empty = runs->n_empty; rd.n_slots = n_slots; rd.n_table = n_table;
        rd.run = runs->run; rd.zero_blocks = (int32_t)((runs->n_empty + 4 * G * 8 - 1) / (4 * G * 8));
        const int64_t waves = (runs->n_runs + G - 1) / G, blocks = rd.zero_blocks + (waves + 3) / 4;
        const dim3 grid((unsigned)blocks), block(256);
        if (blocks > 0) {
#define SUM_RUNS(HD_) do { if (msg_bf16) hipLaunchKernelGGL((gpl_sum_runs_kernel<HD_, true>), grid, block, 0, s, msg, gPL, rd); \
                           else hipLaunchKernelGGL((gpl_sum_runs_kernel<HD_, false>), grid, block, 0, s, msg, gPL, rd); } while (0)
            switch (HD) { case 64: SUM_RUNS(64); break; case 32: SUM_RUNS(32); break; case 16: SUM_RUNS(16); break; default: SUM_RUNS(8); break; }
#undef SUM_RUNS
        }
        if (runs->n_open > 0) {
            const int64_t threads = (int64_t)runs->n_open * (HD / 4);
            hipLaunchKernelGGL(gpl_runs_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, runs->open, runs->n_open, runs->part, gPL, HD / 4);
        }
        GAT_HIP(hipGetLastError());
        return 0;
    }
    if (n_heavy > 0) {                                  // long lists first: they are the longest-running waves
        switch (HD) {
            case 64: GAT_TRY(run_heavy<64>(msg, gPL, msg_bf16, chunks, n_chunks, heavy, n_heavy, part, s)); break;
            case 32: GAT_TRY(run_heavy<32>(msg, gPL, msg_bf16, chunks, n_chunks, heavy, n_heavy, part, s)); break;
            case 16: GAT_TRY(run_heavy<16>(msg, gPL, msg_bf16, chunks, n_chunks, heavy, n_heavy, part, s)); break;
            case 8: GAT_TRY(run_heavy<8>(msg, gPL, msg_bf16, chunks, n_chunks, heavy, n_heavy, part, s)); break;
            default: return fail(GAT_E_UNSUPPORTED, "gpl_sum: H*D outside the fast path");
        }
    }
    // waves per block of the per-source kernels (GAT_GPL_WAVES, A/B): unlike the edge forward, 4 beats 1 here
    // (6.15 vs 6.31 ms per step on one box)
    static const int wpb = [] { const char* e = choice_env("GAT_GPL_WAVES"); const int v = e ? atoi(e) : 4; return (v == 1 || v == 2 || v == 4) ? v : 4; }();
    if (msg_bf16) {
        // group per source unless GAT_GPL_BF16_GROUP=0 (A/B): 10 M / 250 M shape, H*D = 32
        static const bool bgroup = [] { const char* e = choice_env("GAT_GPL_BF16_GROUP"); return !(e && e[0] == '0'); }();
        if (bgroup) {
            const int rpi = 64 / (HD / 8);
            const dim3 ggrid((unsigned)((n_table + 4 * rpi - 1) / (4 * rpi)));
            switch (HD) {
                case 64: hipLaunchKernelGGL(gpl_sum_bf16_group_kernel<64>, ggrid, dim3(256), 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
                case 32: hipLaunchKernelGGL(gpl_sum_bf16_group_kernel<32>, ggrid, dim3(256), 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
                case 16: hipLaunchKernelGGL(gpl_sum_bf16_group_kernel<16>, ggrid, dim3(256), 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
                case 8: hipLaunchKernelGGL(gpl_sum_bf16_group_kernel<8>, ggrid, dim3(256), 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
                default: return fail(GAT_E_UNSUPPORTED, "gpl_sum: H*D outside the fast path");
            }
            GAT_HIP(hipGetLastError());
            return 0;
        }
        const dim3 grid((unsigned)((n_table + wpb - 1) / wpb)), block(64 * wpb);
        switch (HD) {
            case 64: hipLaunchKernelGGL(gpl_sum_bf16_kernel<64>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 32: hipLaunchKernelGGL(gpl_sum_bf16_kernel<32>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 16: hipLaunchKernelGGL(gpl_sum_bf16_kernel<16>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 8: hipLaunchKernelGGL(gpl_sum_bf16_kernel<8>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            default: return fail(GAT_E_UNSUPPORTED, "gpl_sum: H*D outside the fast path");
        }
        GAT_HIP(hipGetLastError());
        return 0;
    }
    static const char* force = choice_env("GAT_GPL_GROUP");          // A/B switch: 0 = wave per source, 1 = group per source
    const bool group = force ? force[0] == '1' : n_slots < 8 * n_table;   // measured: 3.2 slots/source 1.24 -> 1.00 ms, 12.6: 3.20 -> 3.31
    if (group && HD >= 8 && HD <= 64) {
        const int rpi = 64 / (HD / 4);
        const dim3 grid((unsigned)((n_table + 4 * rpi - 1) / (4 * rpi))), block(256);
        switch (HD) {
            case 64: hipLaunchKernelGGL(gpl_sum_group_kernel<64>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 32: hipLaunchKernelGGL(gpl_sum_group_kernel<32>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 16: hipLaunchKernelGGL(gpl_sum_group_kernel<16>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            case 8: hipLaunchKernelGGL(gpl_sum_group_kernel<8>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
            default: return fail(GAT_E_UNSUPPORTED, "gpl_sum: H*D outside the fast path");
        }
        GAT_HIP(hipGetLastError());
        return 0;
    }
    const dim3 grid((unsigned)((n_table + wpb - 1) / wpb)), block(64 * wpb);
    switch (HD) {
        case 64: hipLaunchKernelGGL(gpl_sum_kernel<64>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
        case 32: hipLaunchKernelGGL(gpl_sum_kernel<32>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
        case 16: hipLaunchKernelGGL(gpl_sum_kernel<16>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
        case 8: hipLaunchKernelGGL(gpl_sum_kernel<8>, grid, block, 0, s, src_ptr, msg, gPL, n_table, heavy_slots(n_slots)); break;
        default: return fail(GAT_E_UNSUPPORTED, "gpl_sum: H*D outside the fast path");
    }
    GAT_HIP(hipGetLastError());
    return 0;
}

}  // namespace gat
