// gat_gemm_kernels.hip — the dense W_l / W_r feature projections of the GATv2 layer and their
// backward, as exact-fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, so the
// numerics are those of the reference's float loops up to summation order).
//
// The reference recomputes W·x inside every per-edge thread (E:303-316, 415-420, 636-640,
// 752-761, 848-853).  Here it is computed once per node:
//   project : [PL | PR] = X · [W_left ; W_right]^T                     (fwd of E:303-316)
//   grad_x  : gX = gPL · W_left + gPR · W_right, fused with E:888-892  (E:859-869 summed over edges)
//   grad_w  : gradW_left += gPL^T · X,  gradW_right += gPR^T · X       (E:770-782 summed over edges)
// W stays in the reference layout [H*D][2F] (row j: cols 0..F-1 left, F..2F-1 right).
//
// Shapes are skinny: M = nodes (millions), K and N <= ~128.  project / grad_x therefore keep the
// whole B operand resident in LDS for the lifetime of a persistent block and stream A straight
// from HBM into MFMA fragments (one float4 per lane covers 4 k-steps: lane (i, half) reads
// A[i][kb+4*half .. +3]; MFMA j' of the step pairs element j' with B row kb+4*half+j').  grad_w
// reduces over the node dimension: node tiles are staged through double-buffered LDS in their
// memory layout (which already is the k-major layout the fragments want), split-K over blocks,
// slabs summed in fixed order.
#include "gat_internal.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace gat {
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

// ---- A sources of the row-streaming kernel --------------------------------------------------------
struct ASrcRows {                 // A(i,k) = X[i*ld + k]
    const float* X; int32_t ld;
    __device__ __forceinline__ float4 load4(int64_t i, int k) const {
        return *reinterpret_cast<const float4*>(X + i * ld + k);
    }
    __device__ __forceinline__ float load1(int64_t i, int k) const { return X[i * ld + k]; }
};
struct ASrcGcat {                 // A(i,k) = k < HD ? gPL[i][k] : gPR[i][k-HD]
    const float* gPL; const float* gPR; int32_t HD;
    __device__ __forceinline__ float4 load4(int64_t i, int k) const {
        return k < HD ? *reinterpret_cast<const float4*>(gPL + i * HD + k)
                      : *reinterpret_cast<const float4*>(gPR + i * HD + (k - HD));
    }
    __device__ __forceinline__ float load1(int64_t i, int k) const {
        return k < HD ? gPL[i * HD + k] : gPR[i * HD + (k - HD)];
    }
};
// ---- B sources (copied to LDS once per block / K chunk) ----------------------------------------------
struct BSrcProject {              // B(k=f, j): jj = j0+j;  jj < HD ? W[jj][f] : W[jj-HD][F+f]
    const float* W; int32_t F, HD, j0;
    __device__ __forceinline__ float at(int k, int j) const {
        j += j0;
        return j < HD ? W[(int64_t)j * 2 * F + k] : W[(int64_t)(j - HD) * 2 * F + F + k];
    }
    static constexpr bool kAlongK = true;     // consecutive threads -> consecutive k (W rows are k-contiguous)
};
struct BSrcGradX {                // B(k=c, j=f): c < HD ? W[c][f] : W[c-HD][F+f]
    const float* W; int32_t F, HD;
    __device__ __forceinline__ float at(int k, int j) const {
        return k < HD ? W[(int64_t)k * 2 * F + j] : W[(int64_t)(k - HD) * 2 * F + F + j];
    }
    static constexpr bool kAlongK = false;
};
// ---- epilogues ---------------------------------------------------------------------------------------
template <bool BF>
struct EpiProject {               // cols j0+j < HD -> PL rows (bf16 rows when BF: round to nearest even), else PR
    static constexpr bool kTwoPhase = false;
    float* PL; float* PR; int32_t HD, j0;
    __device__ __forceinline__ float pre(int64_t, int) const { return 0.f; }
    __device__ __forceinline__ void apply(int64_t i, int j, float v, float) const { (*this)(i, j, v); }
    __device__ __forceinline__ void operator()(int64_t i, int j, float v) const {
        j += j0;
        if (j < HD) {
            if constexpr (BF) {
                const uint32_t u = __builtin_bit_cast(uint32_t, v);
                reinterpret_cast<uint16_t*>(PL)[i * HD + j] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
            } else {
                PL[i * HD + j] = v;
            }
        } else {
            PR[i * HD + (j - HD)] = v;
        }
    }
};
struct EpiGradX {                 // g_prev = gX ⊙ LReLU'(h_pre_prev)   (E:888-892)
    // two-phase: all h_pre_prev operands of a tile are loaded first, then all results are stored — a
    // load/store pair per element is serialised by hipcc (the store may alias the next load)
    static constexpr bool kTwoPhase = true;
    float* out; const float* hpre_prev; int32_t ld; float slope;
    __device__ __forceinline__ float pre(int64_t i, int j) const { return hpre_prev[i * ld + j]; }
    __device__ __forceinline__ void apply(int64_t i, int j, float v, float hv) const {
        out[i * ld + j] = v * (hv > 0.f ? 1.0f : slope);
    }
    __device__ __forceinline__ void operator()(int64_t i, int j, float v) const { apply(i, j, v, pre(i, j)); }
};

struct EpiStore {                 // out[i][j] = v
    static constexpr bool kTwoPhase = false;
    float* out; int32_t ld;
    __device__ __forceinline__ float pre(int64_t, int) const { return 0.f; }
    __device__ __forceinline__ void apply(int64_t i, int j, float v, float) const { out[i * ld + j] = v; }
    __device__ __forceinline__ void operator()(int64_t i, int j, float v) const { out[i * ld + j] = v; }
};

constexpr int kKC = 128;          // K chunk resident in LDS

}  // namespace

// blocks of a 256-thread kernel resident on the whole chip (occupancy API incl. dynamic LDS), cached
int64_t resident_blocks(const void* fn, size_t dyn_lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int64_t> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(fn, dyn_lds);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    return cache[key] = (int64_t)per_cu * (cus > 0 ? cus : 256);
}

namespace {

// C[M][N] = A[M][K] · B[K][N].  256 threads = 4 waves, each wave owns 32 rows x (NT*32) columns of a
// 128-row tile; blocks are persistent over row tiles (grid.x) and column blocks of NT*32 (grid.y).
template <int NT, bool VEC4, class AS, class BS, class EP>
__global__ __launch_bounds__(256) void rowgemm_kernel(AS as, BS bs, EP ep, int64_t M, int32_t N, int32_t K,
                                                      int32_t kc_lds) {
    constexpr int NW = NT * 32;
    extern __shared__ float Bsh[];                    // [kc_lds][NW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const int64_t ntiles = (M + 127) / 128;
    bool b_loaded = false;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * 128 + wave * 32 + li;
        const bool rvalid = row < M;
        v16f acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += kKC) {
            const int kc = (K - k0 < kKC) ? (K - k0) : kKC;
            const int kc8 = (kc + 7) & ~7;
            if (K > kKC || !b_loaded) {               // B chunk -> LDS (once per block when K <= kKC)
                __syncthreads();
                if constexpr (BS::kAlongK) {
                    for (int idx = threadIdx.x; idx < kc8 * NW; idx += 256) {
                        const int kk = idx % kc8, j = idx / kc8;
                        Bsh[kk * NW + j] = (kk < kc && n0 + j < N) ? bs.at(k0 + kk, n0 + j) : 0.f;
                    }
                } else {
                    for (int idx = threadIdx.x; idx < kc8 * NW; idx += 256) {
                        const int kk = idx / NW, j = idx % NW;
                        Bsh[kk * NW + j] = (kk < kc && n0 + j < N) ? bs.at(k0 + kk, n0 + j) : 0.f;
                    }
                }
                b_loaded = true;
                __syncthreads();
            }
            // all A fragments of this chunk in flight at once
            float4 af[kKC / 8];
#pragma unroll
            for (int st = 0; st < kKC / 8; ++st) {
                const int kk = st * 8 + 4 * half;
                af[st] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (st * 8 < kc8) {
                    if constexpr (VEC4) {
                        if (rvalid && kk < kc) af[st] = as.load4(row, k0 + kk);
                    } else {
                        if (rvalid) {
                            if (kk + 0 < kc) af[st].x = as.load1(row, k0 + kk + 0);
                            if (kk + 1 < kc) af[st].y = as.load1(row, k0 + kk + 1);
                            if (kk + 2 < kc) af[st].z = as.load1(row, k0 + kk + 2);
                            if (kk + 3 < kc) af[st].w = as.load1(row, k0 + kk + 3);
                        }
                    }
                }
            }
#pragma unroll
            for (int st = 0; st < kKC / 8; ++st) {
                if (st * 8 < kc8) {
                    const float* brow = Bsh + (st * 8 + 4 * half) * NW + li;
                    const float a4[4] = {af[st].x, af[st].y, af[st].z, af[st].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], brow[j * NW + nt * 32], acc[nt], 0, 0, 0);
                }
            }
        }
        // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        float hv[EP::kTwoPhase ? NT : 1][16];
        if constexpr (EP::kTwoPhase) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t orow = tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int col = n0 + nt * 32 + li;
                    hv[nt][r] = (orow < M && col < N) ? ep.pre(orow, col) : 0.f;
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = n0 + nt * 32 + li;
                if (orow < M && col < N) {
                    if constexpr (EP::kTwoPhase) ep.apply(orow, col, acc[nt][r], hv[nt][r]);
                    else ep(orow, col, acc[nt][r]);
                }
            }
    }
}

// The same product for K <= kKC (B resident in LDS for the whole block) with the A stream software-pipelined:
// k is walked in steps of 32 (4 float4 per lane), the fragments of step q+1 — of this tile or of the block's next
// tile — are in flight while step q is multiplied.  The all-at-once variant above needs 64 VGPRs of A fragments
// next to 64 accumulator registers: 2 waves per SIMD, and every tile starts with an exposed load latency (MFMA
// pipe 47-55 % busy, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES).  Two 4-float4 buffers instead of one 16-float4 one.
template <int NT, class AS, class BS, class EP>
__global__ __launch_bounds__(256) void rowgemm_pipe_kernel(AS as, BS bs, EP ep, int64_t M, int32_t N, int32_t K) {
    constexpr int NW = NT * 32;
    extern __shared__ float Bsh[];                    // [kc8][NW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const int64_t ntiles = (M + 127) / 128;
    const int kc8 = (K + 7) & ~7;
    if constexpr (BS::kAlongK) {
        for (int idx = threadIdx.x; idx < kc8 * NW; idx += 256) {
            const int kk = idx % kc8, j = idx / kc8;
            Bsh[kk * NW + j] = (kk < K && n0 + j < N) ? bs.at(kk, n0 + j) : 0.f;
        }
    } else {
        for (int idx = threadIdx.x; idx < kc8 * NW; idx += 256) {
            const int kk = idx / NW, j = idx % NW;
            Bsh[kk * NW + j] = (kk < K && n0 + j < N) ? bs.at(kk, n0 + j) : 0.f;
        }
    }
    __syncthreads();
    const int nsteps = (kc8 + 31) >> 5;
    auto load_a = [&](int64_t tile, int step, float4 (&buf)[4]) {
        const int64_t row = tile * 128 + wave * 32 + li;
        const bool rvalid = row < M;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kk = step * 32 + st * 8 + 4 * half;
            buf[st] = (rvalid && kk < K) ? as.load4(row, kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    v16f acc[NT];
    auto clear = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    };
    auto mma = [&](int step, const float4 (&buf)[4]) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (step * 32 + st * 8 < kc8) {
                const float* brow = Bsh + (step * 32 + st * 8 + 4 * half) * NW + li;
                const float a4[4] = {buf[st].x, buf[st].y, buf[st].z, buf[st].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], brow[j * NW + nt * 32], acc[nt], 0, 0, 0);
            }
        }
    };
    auto store = [&](int64_t tile) {                  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = n0 + nt * 32 + li;
                if (orow < M && col < N) ep(orow, col, acc[nt][r]);
            }
    };
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    int step = 0;
    float4 b0[4], b1[4];
    clear();
    load_a(tile, 0, b0);
    for (;;) {
        int64_t t1 = tile; int s1 = step + 1;
        if (s1 == nsteps) { s1 = 0; t1 = tile + gridDim.x; }
        const bool more1 = t1 < ntiles;
        if (more1) load_a(t1, s1, b1);
        mma(step, b0);
        if (step == nsteps - 1) { store(tile); clear(); }
        if (!more1) break;
        int64_t t2 = t1; int s2 = s1 + 1;
        if (s2 == nsteps) { s2 = 0; t2 = t1 + gridDim.x; }
        const bool more2 = t2 < ntiles;
        if (more2) load_a(t2, s2, b0);
        mma(s1, b1);
        if (s1 == nsteps - 1) { store(t1); clear(); }
        if (!more2) break;
        tile = t2; step = s2;
    }
}

// ---- projection of FEW rows with a LONG K (Cora-shape: 2,708 x 1,433; Pubmed-shape: 19,717 x 500) ----------------
// The row-streaming kernel above gives such a product 22 (Cora) or 154 (Pubmed) row tiles for 256 CUs, each walking
// K in 128-wide chunks one after the other.  Here a block owns 128 rows x 128 columns x ONE 128-wide K chunk (grid.z =
// K split; its four waves take 32 rows each, as above), and the K partials go to slabs that project_reduce_kernel
// adds in chunk order (deterministic) on its way through the projection's epilogue.
template <bool VEC4>
__global__ __launch_bounds__(256) void project_splitk_kernel(ASrcRows as, BSrcProject bs, float* __restrict__ slabs,
                                                             int64_t M, int32_t N, int32_t K) {
    constexpr int NW = 128, NT = 4, KC = 128;
    extern __shared__ float Bsh[];                    // [KC][NW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const int k0 = blockIdx.z * KC;
    const int kc = (K - k0 < KC) ? (K - k0) : KC;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + li;
    float4 af[KC / 8];                                // the wave's A fragments of the chunk: all in flight behind the B fill
#pragma unroll
    for (int st = 0; st < KC / 8; ++st) {
        const int kk = st * 8 + 4 * half;
        af[st] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < M) {
            if constexpr (VEC4) {
                if (kk < kc) af[st] = as.load4(row, k0 + kk);
            } else {
                if (kk + 0 < kc) af[st].x = as.load1(row, k0 + kk + 0);
                if (kk + 1 < kc) af[st].y = as.load1(row, k0 + kk + 1);
                if (kk + 2 < kc) af[st].z = as.load1(row, k0 + kk + 2);
                if (kk + 3 < kc) af[st].w = as.load1(row, k0 + kk + 3);
            }
        }
    }
    for (int idx = threadIdx.x; idx < KC * NW; idx += 256) {       // consecutive threads -> consecutive k (W rows are k-contiguous)
        const int kk = idx % KC, j = idx / KC;
        Bsh[kk * NW + j] = (kk < kc && n0 + j < N) ? bs.at(k0 + kk, n0 + j) : 0.f;
    }
    __syncthreads();
    v16f acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int st = 0; st < KC / 8; ++st) {
        const float* brow = Bsh + (st * 8 + 4 * half) * NW + li;
        const float a4[4] = {af[st].x, af[st].y, af[st].z, af[st].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], brow[j * NW + nt * 32], acc[nt], 0, 0, 0);
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* out = slabs + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t orow = (int64_t)blockIdx.x * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int col = n0 + nt * 32 + li;
            if (orow < M && col < N) out[orow * N + col] = acc[nt][r];
        }
}
template <class EP>
__global__ __launch_bounds__(256) void project_reduce_kernel(const float* __restrict__ slabs, int32_t ksplit, int64_t M, int32_t N, EP ep) {
    const int64_t total = M * N, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float v = 0.f;
        for (int z = 0; z < ksplit; ++z) v += slabs[(int64_t)z * total + i];
        ep(i / N, (int)(i % N), v);
    }
}
// rows few enough that the streaming kernel would leave most CUs idle, K long enough to be worth splitting
static bool project_wants_splitk(int64_t M, int32_t N, int32_t K) {
    static const bool off = [] { const char* e = getenv("GAT_PROJECT_SPLITK"); return e && e[0] == '0'; }();     // A/B
    return !off && K > 128 && ((M + 127) / 128) * ((N + 127) / 128) < 256;
}

template <class AS, class BS, class EP>
int run_rowgemm(const AS& as, const BS& bs, const EP& ep, int64_t M, int32_t N, int32_t K, bool vec4, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int NT = N > 64 ? 4 : (N > 32 ? 2 : 1);
    const int NW = NT * 32;
    const int kc_lds = ((K < kKC ? K : kKC) + 7) & ~7;
    const size_t lds = (size_t)kc_lds * NW * sizeof(float);
    const int64_t ntiles = (M + 127) / 128;
    if constexpr (!EP::kTwoPhase) {
        static const bool pipe = [] { const char* e = getenv("GAT_GEMM_PIPE"); return !(e && e[0] == '0'); }();   // A/B
        // only up to 64 output columns per block: with 128 (64 accumulator + 148 other registers = 2 waves per SIMD
        // either way) the pipelined form measured 1.31 vs 1.22 ms per step for the projections; grad_x 0.56 -> 0.48
        if (vec4 && K <= kKC && pipe && NT <= 2) {
#define GAT_ROWGEMM_PIPE(NT_)                                                                                 \
    {                                                                                                         \
        auto kern = rowgemm_pipe_kernel<NT_, AS, BS, EP>;                                                     \
        const int64_t res = resident_blocks((const void*)kern, lds);                                          \
        const dim3 grid((unsigned)(ntiles < res ? ntiles : res), (unsigned)((N + NW - 1) / NW));              \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, as, bs, ep, M, N, K);                                \
    }
            if (NT == 2) GAT_ROWGEMM_PIPE(2) else GAT_ROWGEMM_PIPE(1)
#undef GAT_ROWGEMM_PIPE
            GAT_HIP(hipGetLastError());
            return 0;
        }
    }
    // persistent grid = exactly the blocks that are resident at once (registers + LDS, occupancy API):
    // a larger grid would run a second, partly filled round of 25-tile blocks
#define GAT_ROWGEMM(NT_, V_)                                                                                  \
    {                                                                                                         \
        auto kern = rowgemm_kernel<NT_, V_, AS, BS, EP>;                                                      \
        const int64_t res = resident_blocks((const void*)kern, lds);                                          \
        const dim3 grid((unsigned)(ntiles < res ? ntiles : res), (unsigned)((N + NW - 1) / NW));              \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, as, bs, ep, M, N, K, kc_lds);                        \
    }
    if (vec4) { if (NT == 4) GAT_ROWGEMM(4, true) else if (NT == 2) GAT_ROWGEMM(2, true) else GAT_ROWGEMM(1, true) }
    else { if (NT == 4) GAT_ROWGEMM(4, false) else if (NT == 2) GAT_ROWGEMM(2, false) else GAT_ROWGEMM(1, false) }
#undef GAT_ROWGEMM
    GAT_HIP(hipGetLastError());
    return 0;
}

// ---- grad_w: slabs[z][c][f] = sum over the block's nodes of [gPL|gPR][n][c] * X[n][f] ------------------
// Block tile BM (c) x 128 (f): WM=2 -> 128 x 128 as 2x2 waves of 64x64 (both halves, 2*HD = 128
// channels); WM=1 -> 64 x 128 as 1x4 waves of 64x32 (one half, HD = 64 channels: no MFMA spent on
// padding rows).  Node tiles of 32 staged through two LDS buffers in memory layout ([node][c] and
// [node][f] are already k-major); global loads of tile t+1 are in flight while tile t is multiplied.
template <bool VEC4, int WM, int BN = 128>
__global__ __launch_bounds__(256) void gradw_kernel(const float* __restrict__ gPL, const float* __restrict__ gPR,
                                                    const float* __restrict__ X, float* __restrict__ slabs,
                                                    int64_t n_rows, int32_t HD, int32_t F, int64_t kchunk,
                                                    int32_t c_base, int32_t M) {
    // BN = 64 when F <= 64 (hidden layers: F = H*D of the layer below): no MFMA spent on columns beyond F
    constexpr int KT = 32, BM = 64 * WM;
    constexpr int WN = 4 / WM, NY = BN / WN / 32;     // waves along f; 32-col MFMA tiles per wave
    static_assert(NY >= 1, "tile shape");
    __shared__ float As[2][KT][BM];
    __shared__ float Bs[2][KT][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, half = lane >> 5;
    // rows c_base .. c_base+M-1 of the concatenated [gPL | gPR] channel space (both halves, or one)
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    const int64_t kb = (int64_t)blockIdx.z * kchunk;
    const int64_t ke = (kb + kchunk < n_rows) ? kb + kchunk : n_rows;
    v16f acc[2][NY];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NY; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];              // staging registers: 4 float4 of A and of B per thread per tile
    auto load_tile = [&](int64_t n0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + 256 * p;
            const int kk = idx >> 5, c = (idx & 31) * 4;
            const int64_t node = n0 + kk;
            ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (node < ke) {
                const int ci = c_base + i0 + c, cj = j0 + c;
                if constexpr (VEC4) {
                    if (c < BM && i0 + c < M) ra[p] = ci < HD ? *reinterpret_cast<const float4*>(gPL + node * HD + ci)
                                                    : *reinterpret_cast<const float4*>(gPR + node * HD + (ci - HD));
                    if (c < BN && cj < F) rb[p] = *reinterpret_cast<const float4*>(X + node * F + cj);
                } else {
                    float t[4], u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cc = ci + q, jj = cj + q;
                        t[q] = (c < BM && (cc - c_base) < M) ? (cc < HD ? gPL[node * HD + cc] : gPR[node * HD + (cc - HD)]) : 0.f;
                        u[q] = (c < BN && jj < F) ? X[node * F + jj] : 0.f;
                    }
                    ra[p] = make_float4(t[0], t[1], t[2], t[3]);
                    rb[p] = make_float4(u[0], u[1], u[2], u[3]);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + 256 * p;
            const int kk = idx >> 5, c = (idx & 31) * 4;
            if (c < BM) *reinterpret_cast<float4*>(&As[buf][kk][c]) = ra[p];
            if (c < BN) *reinterpret_cast<float4*>(&Bs[buf][kk][c]) = rb[p];
        }
    };

    const int64_t ntile = (ke - kb + KT - 1) / KT;
    if (ntile > 0) {
        load_tile(kb);
        store_tile(0);
    }
    __syncthreads();
    for (int64_t t = 0; t < ntile; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntile) load_tile(kb + (t + 1) * KT);
        // LDS operands of step ks+1 are read before the MFMAs of step ks are issued: hipcc otherwise reloads the
        // same two registers and waits lgkmcnt(0) in front of every MFMA group (LDS latency exposed, pipe 67 % busy)
        float a[2], b[NY];
        a[0] = As[buf][half][wm * 64 + li]; a[1] = As[buf][half][wm * 64 + 32 + li];
#pragma unroll
        for (int y = 0; y < NY; ++y) b[y] = Bs[buf][half][wn * (NY * 32) + y * 32 + li];
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            float an[2] = {0.f, 0.f}, bn[NY];
#pragma unroll
            for (int y = 0; y < NY; ++y) bn[y] = 0.f;
            if (ks + 1 < KT / 2) {
                const int kk = (ks + 1) * 2 + half;
                an[0] = As[buf][kk][wm * 64 + li]; an[1] = As[buf][kk][wm * 64 + 32 + li];
#pragma unroll
                for (int y = 0; y < NY; ++y) bn[y] = Bs[buf][kk][wn * (NY * 32) + y * 32 + li];
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the reads above the MFMAs (the scheduler sinks them to their use)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NY; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a[0] = an[0]; a[1] = an[1];
#pragma unroll
            for (int y = 0; y < NY; ++y) b[y] = bn[y];
        }
        if (t + 1 < ntile) store_tile(buf ^ 1);
        __syncthreads();
    }
    float* out = slabs + (int64_t)blockIdx.z * M * F;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = j0 + wn * (NY * 32) + y * 32 + li;
                if (row < M && col < F) out[(int64_t)row * F + col] = acc[x][y][r];
            }
}

inline int grad_w_bm(int32_t M) { return (M % 128 == 0 || M % 128 > 64) ? 128 : 64; }
inline int grad_w_bn(int32_t M, int32_t F) { return (grad_w_bm(M) == 128 && F <= 64) ? 64 : 128; }   // 64 only in the 2x2-wave shape

int64_t grad_w_kchunk(int64_t n_rows, int32_t F, int32_t M) {
    const int bm = grad_w_bm(M), bn = grad_w_bn(M, F);
    const int64_t tiles = (((int64_t)M + bm - 1) / bm) * (((int64_t)F + bn - 1) / bn);
    // split-K over exactly the blocks that fit at once (LDS-limited: 64 / 48 KiB per block)
    int64_t splits = resident_blocks(bm == 128 ? (bn == 64 ? (const void*)gradw_kernel<true, 2, 64> : (const void*)gradw_kernel<true, 2>)
                                               : (const void*)gradw_kernel<true, 1>, 0) / tiles;
    if (splits < 1) splits = 1;
    int64_t kchunk = (n_rows + splits - 1) / splits;
    kchunk = ((kchunk + 31) / 32) * 32;
    if (kchunk < 256) kchunk = 256;
    return kchunk;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int64_t project_scratch_floats(int64_t n_rows, int32_t F, int32_t HD) {
    return project_wants_splitk(n_rows, 2 * HD, F) ? (int64_t)((F + 127) / 128) * n_rows * 2 * HD : 0;
}

int launch_project(const float* X, const float* W, float* PL_rows, float* PR, int64_t n_rows, int32_t F,
                   int32_t HD, int32_t part, bool pl_bf16, float* scratch, hipStream_t s) {
    const int32_t j0 = part == kPartRight ? HD : 0;
    ASrcRows as{X, F};
    BSrcProject bs{W, F, HD, j0};
    const bool vec4 = (F % 4 == 0) && aligned16(X);
    const int32_t N = part == kPartBoth ? 2 * HD : HD;
    if (scratch != nullptr && n_rows > 0 && project_wants_splitk(n_rows, N, F)) {
        const int ksplit = (F + 127) / 128;
        const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((N + 127) / 128), (unsigned)ksplit);
        const size_t lds = (size_t)(128 * 128) * sizeof(float);
        if (vec4) hipLaunchKernelGGL(project_splitk_kernel<true>, grid, dim3(256), lds, s, as, bs, scratch, n_rows, N, F);
        else hipLaunchKernelGGL(project_splitk_kernel<false>, grid, dim3(256), lds, s, as, bs, scratch, n_rows, N, F);
        const int64_t rb = std::min<int64_t>((n_rows * N + 255) / 256, 4096);
        if (pl_bf16) hipLaunchKernelGGL(project_reduce_kernel<EpiProject<true>>, dim3((unsigned)rb), dim3(256), 0, s, scratch, ksplit, n_rows, N, EpiProject<true>{PL_rows, PR, HD, j0});
        else hipLaunchKernelGGL(project_reduce_kernel<EpiProject<false>>, dim3((unsigned)rb), dim3(256), 0, s, scratch, ksplit, n_rows, N, EpiProject<false>{PL_rows, PR, HD, j0});
        GAT_HIP(hipGetLastError());
        return 0;
    }
    if (pl_bf16) return run_rowgemm(as, bs, EpiProject<true>{PL_rows, PR, HD, j0}, n_rows, N, F, vec4, s);
    return run_rowgemm(as, bs, EpiProject<false>{PL_rows, PR, HD, j0}, n_rows, N, F, vec4, s);
}

int launch_grad_x(const float* gPL_rows, const float* gPR, const float* W, const float* hpre_prev,
                  float* gprev, int64_t n_rows, int32_t F, int32_t HD, float slope, hipStream_t s) {
    ASrcGcat as{gPL_rows, gPR, HD};
    BSrcGradX bs{W, F, HD};
    const bool vec4 = (HD % 4 == 0) && aligned16(gPL_rows) && aligned16(gPR);
    if (hpre_prev == nullptr) return run_rowgemm(as, bs, EpiStore{gprev, F}, n_rows, F, 2 * HD, vec4, s);
    EpiGradX ep{gprev, hpre_prev, F, slope};
    return run_rowgemm(as, bs, ep, n_rows, F, 2 * HD, vec4, s);
}

int64_t grad_w_scratch_floats(int64_t n_rows, int32_t F, int32_t HD) {
    int64_t need = 0;
    for (int32_t M : {2 * HD, HD}) {                 // both halves at once, or one half per launch
        const int64_t kchunk = grad_w_kchunk(n_rows, F, M);
        const int64_t ksplit = (n_rows + kchunk - 1) / kchunk;
        need = std::max<int64_t>(need, (ksplit < 1 ? 1 : ksplit) * M * (int64_t)F);
    }
    return need;
}

int launch_grad_w(const float* gPL_rows, const float* gPR, const float* X, float* gradW, float* scratch,
                  int64_t n_rows, int32_t F, int32_t HD, int32_t part, hipStream_t s) {
    if (n_rows <= 0) return 0;
    const int M = part == kPartBoth ? 2 * HD : HD;
    const int c_base = part == kPartRight ? HD : 0;
    if (part == kPartLeft) gPR = gPL_rows;           // never dereferenced; keeps the alignment test meaningful
    if (part == kPartRight) gPL_rows = gPR;
    const int64_t kchunk = grad_w_kchunk(n_rows, F, M);
    const int64_t ksplit = (n_rows + kchunk - 1) / kchunk;
    const int bm = grad_w_bm(M), bn = grad_w_bn(M, F);
    const dim3 grid((unsigned)((F + bn - 1) / bn), (unsigned)((M + bm - 1) / bm), (unsigned)ksplit);
    const bool vec4 = (F % 4 == 0) && (HD % 4 == 0) && aligned16(X) && aligned16(gPL_rows) && aligned16(gPR);
#define GAT_GRADW(V_, WM_, BN_) hipLaunchKernelGGL((gradw_kernel<V_, WM_, BN_>), grid, dim3(256), 0, s, gPL_rows, gPR, X, scratch, n_rows, HD, F, kchunk, c_base, M)
    if (vec4) { if (bm == 128 && bn == 64) GAT_GRADW(true, 2, 64); else if (bm == 128) GAT_GRADW(true, 2, 128); else GAT_GRADW(true, 1, 128); }
    else { if (bm == 128 && bn == 64) GAT_GRADW(false, 2, 64); else if (bm == 128) GAT_GRADW(false, 2, 128); else GAT_GRADW(false, 1, 128); }
#undef GAT_GRADW
    GAT_HIP(hipGetLastError());
    return launch_reduce_gradw(scratch, (int32_t)ksplit, HD, F, c_base, M, gradW, s);
}

}  // namespace gat
