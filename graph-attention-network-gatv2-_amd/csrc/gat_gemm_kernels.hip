// gat_gemm_kernels.hip — the dense W_l / W_r feature projections of the GATv2 layer and their
// backward as MFMA GEMMs with fp32 accuracy.
//
// The reference recomputes W·x inside every per-edge thread (E:303-316, 415-420, 636-640,
// 752-761, 848-853).  Here it is computed once per node:
//   project : [PL | PR] = X · [W_left ; W_right]^T                     (fwd of E:303-316)
//   grad_x  : gX = gPL · W_left + gPR · W_right, fused with E:888-892  (E:859-869 summed over edges)
//   grad_w  : gradW_left += gPL^T · X,  gradW_right += gPR^T · X       (E:770-782 summed over edges)
// W stays in the reference layout [H*D][2F] (row j: cols 0..F-1 left, F..2F-1 right).
//
// Arithmetic (round 2): the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) runs at 1/16 of
// the bf16 rate and made every kernel here MFMA-bound at Products size.  The default kernels (*_x3_*) cut each fp32
// operand EXACTLY into three bf16 pieces (rounded to nearest) and run six bf16 MFMAs per product with fp32 accumulation:
// error per product < 2^-24 — below the rounding of an fp32 multiply — in the worst case (tests/test_split_pieces.py);
// a K-term dot product comes out at or below the error of the reference's own fp32 fma chain (tests/test_dense_precision.py).  The fp32-MFMA kernels stay as the A/B (GAT_GEMM_X3=0, GAT_GRADW_X3=0).
//
// Shapes are skinny: M = nodes (millions), K and N <= ~128.  project / grad_x keep the whole B operand resident in
// LDS for the lifetime of a persistent block and stream A straight from HBM into MFMA fragments.  grad_w reduces
// over the node dimension: node tiles are staged through LDS in their memory layout and read back transposed
// (ds_read_b64_tr_b16), split-K over the resident blocks, slabs summed in fixed order.
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

// blocks of a kernel (256 threads unless said otherwise) resident on the whole chip (occupancy API incl. dynamic LDS), cached
int64_t resident_blocks(const void* fn, size_t dyn_lds, int threads) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int64_t> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(fn, dyn_lds);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    return cache[key] = (int64_t)per_cu * (cus > 0 ? cus : 256);
}

// dynamic LDS beyond 64 KiB must be allowed per kernel AND per device (a process may drive several); cheap after the first call
static void allow_big_lds(const void* fn) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, bool> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    bool& d = done[std::make_pair(fn, dev)];
    if (!d) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); d = true; }
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

// ---- the same product on the bf16 matrix pipe, fp32 operands cut into three bf16 pieces -----------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate, and at Products size every dense kernel of the step is bound
// by it, not by HBM.  An fp32 value is EXACTLY hi + mid + lo with three bf16 pieces of 8 significant bits
// each, cut by rounding to nearest (the remainders x - hi and (x - hi) - mid are exact in fp32, and the second has at most
// 8 significant bits), a bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32.  Of the nine piece
// products of a·b the six with (piece index of a) + (piece index of b) <= 2 are kept: the dropped ones (mid·lo, lo·mid,
// lo·lo) sum to < 2^-24 |a·b| in the worst case, 2^-28 on average — less than the rounding of one fp32 multiply
// (tests/test_split_pieces.py); what remains is the fp32 accumulation, as in an fma chain.
// -DGAT_X3_EIGHT_TERMS adds mid·lo and lo·mid (< 2^-33; +0.2 ms per Products step: not the default).  Six
// v_mfma_f32_32x32x16_bf16 (6 x 32 cycles for K = 16) replace eight 32x32x2_f32 (8 x 64 cycles), and the kernel
// becomes HBM-bound.  Small terms are accumulated first.  Non-finite operands (and |x| within half a bf16 ulp of
// FLT_MAX, which rounds to inf) give NaN: inf - inf in the first remainder.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two floats -> the packed (x0 | x1 << 16) bf16 pairs of their hi / mid / lo pieces.  Pieces are cut by ROUNDING to
// nearest (v_cvt_pk_bf16_f32): hi = bf16(x), mid = bf16(x - hi), lo = (x - hi) - mid; both remainders are exact in fp32 and
// the last one has <= 8 significant bits, so hi + mid + lo == x exactly, with |mid| <= 2^-9 |x|, |lo| <= 2^-17 |x|.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
    lo = cvt_pk_bf16(q0, q1);
}
struct Pieces { uint4 hi, mid, lo; };                                // 8 consecutive k of one row / column
__device__ __forceinline__ Pieces split8(const float4& a, const float4& b) {
    Pieces p;
    split_pair(a.x, a.y, p.hi.x, p.mid.x, p.lo.x);
    split_pair(a.z, a.w, p.hi.y, p.mid.y, p.lo.y);
    split_pair(b.x, b.y, p.hi.z, p.mid.z, p.lo.z);
    split_pair(b.z, b.w, p.hi.w, p.mid.w, p.lo.w);
    return p;
}
__device__ __forceinline__ v16f mfma_bf16(const uint4& a, const uint4& b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// C[M][N] = A[M][K] · B[K][N] for K <= 128.  WAVES waves, each owning 32 rows x (NT*32) columns of a (WAVES*32)-row
// tile; blocks persistent over row tiles.  B lives in LDS for the lifetime of the block, already cut into pieces and
// in fragment order: plane p (hi, mid, lo), k-step ks (16 k), half h, column n -> the 8 bf16 of k = 16 ks + 8 h .. + 7
// (lane (n, h) of the 32x32x16 B operand reads 16 contiguous bytes, lanes of a half 16 B apart: conflict-free
// ds_read_b128).  A streams from HBM straight into the operand layout (lane (row, h) holds k = 8h .. 8h+7: two float4)
// through a ring of kRing k-steps per wave, kRing-1 of them in flight across tile boundaries.
template <int NT, int WAVES, bool VEC4, class AS, class BS, class EP, int kRing = 4>
__global__ __launch_bounds__(WAVES * 64) void rowgemm_x3_kernel(AS as, BS bs, EP ep, int64_t M, int32_t N, int32_t K) {
    constexpr int NW = NT * 32, TR = WAVES * 32;
    extern __shared__ uint4 Bq[];                     // [3][KS][2][NW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const int KS = (K + 15) >> 4;
    const int plane = KS * 2 * NW;
    for (int idx = threadIdx.x; idx < KS * 2 * NW; idx += WAVES * 64) {
        int n, kh;
        if constexpr (BS::kAlongK) { kh = idx % (KS * 2); n = idx / (KS * 2); }
        else { n = idx % NW; kh = idx / NW; }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kh * 8 + j;
            v[j] = (k < K && n0 + n < N) ? bs.at(k, n0 + n) : 0.f;
        }
        const Pieces p = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        Bq[kh * NW + n] = p.hi; Bq[plane + kh * NW + n] = p.mid; Bq[2 * plane + kh * NW + n] = p.lo;
    }
    __syncthreads();
    const int64_t ntiles = (M + TR - 1) / TR;
    if ((int64_t)blockIdx.x >= ntiles) return;
    const int64_t my_tiles = (ntiles - 1 - blockIdx.x) / gridDim.x + 1;
    const int64_t total = my_tiles * KS;              // k-steps this wave walks

    auto load_a = [&](int64_t tile, int step, float4 (&buf)[2]) {
        int64_t row = tile * TR + wave * 32 + li;
        row = row < M ? row : M - 1;                  // rows past the end: a valid address, results never stored
        const int kk = step * 16 + 8 * half;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            buf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (VEC4) {
                if (kk + 4 * q < K) buf[q] = as.load4(row, kk + 4 * q);
            } else {
                if (kk + 4 * q + 0 < K) buf[q].x = as.load1(row, kk + 4 * q + 0);
                if (kk + 4 * q + 1 < K) buf[q].y = as.load1(row, kk + 4 * q + 1);
                if (kk + 4 * q + 2 < K) buf[q].z = as.load1(row, kk + 4 * q + 2);
                if (kk + 4 * q + 3 < K) buf[q].w = as.load1(row, kk + 4 * q + 3);
            }
        }
    };
    v16f acc[NT];
    auto clear = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    };
    auto mma = [&](int step, const float4 (&buf)[2]) {
        const Pieces a = split8(buf[0], buf[1]);
        const uint4* bp = Bq + (step * 2 + half) * NW + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint4 bh = bp[nt * 32], bm = bp[plane + nt * 32], bl = bp[2 * plane + nt * 32];
#ifdef GAT_X3_EIGHT_TERMS
            acc[nt] = mfma_bf16(a.lo, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bl, acc[nt]);
#endif
            acc[nt] = mfma_bf16(a.lo, bh, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bl, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bh, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bh, acc[nt]);
        }
    };
    auto store = [&](int64_t tile) {                  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        float hv[EP::kTwoPhase ? NT : 1][16];
        if constexpr (EP::kTwoPhase) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t orow = tile * TR + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int col = n0 + nt * 32 + li;
                    hv[nt][r] = (orow < M && col < N) ? ep.pre(orow, col) : 0.f;
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = tile * TR + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int col = n0 + nt * 32 + li;
                if (orow < M && col < N) {
                    if constexpr (EP::kTwoPhase) ep.apply(orow, col, acc[nt][r], hv[nt][r]);
                    else ep(orow, col, acc[nt][r]);
                }
            }
    };
    // two cursors over the wave's (tile, step) sequence: the loader runs kRing-1 steps ahead of the multiplier
    int64_t lt = blockIdx.x, ct = blockIdx.x;
    int ls = 0, cs = 0;
    float4 ring[kRing][2];
    clear();
#pragma unroll
    for (int r = 0; r < kRing - 1; ++r) {
        if (r < total) {
            load_a(lt, ls, ring[r]);
            if (++ls == KS) { ls = 0; lt += gridDim.x; }
        }
    }
    for (int64_t q = 0; q < total; q += kRing) {
#pragma unroll
        for (int r = 0; r < kRing; ++r) {
            if (q + r + kRing - 1 < total) {
                load_a(lt, ls, ring[(r + kRing - 1) % kRing]);
                if (++ls == KS) { ls = 0; lt += gridDim.x; }
            }
            if (q + r < total) {
                mma(cs, ring[r]);
                if (++cs == KS) { store(ct); clear(); cs = 0; ct += gridDim.x; }
            }
        }
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
// the same block shape on the bf16 pipe (three-piece operands: rowgemm_x3_kernel); B chunk = 3 piece planes in LDS (96 KiB)
#ifdef GAT_EXPERIMENTS
#define GAT_ABLATE_PARAM , int ablate
#define GAT_ABLATE(bit) ((ablate & (bit)) != 0)
#else
#define GAT_ABLATE_PARAM
#define GAT_ABLATE(bit) false
#endif
// (experiment library: GAT_DBG_SPLITK = bit mask of parts to SKIP — 1 A loads, 2 B fill, 4 MFMA loop, 8 slab stores — wrong results,
//  where the block's time goes: DESIGN §4 "Round 4")
template <bool VEC4, int NT>
__global__ __launch_bounds__(256) void project_splitk_x3_kernel(ASrcRows as, BSrcProject bs, float* __restrict__ slabs,
                                                                int64_t M, int32_t N, int32_t K GAT_ABLATE_PARAM) {
    constexpr int NW = NT * 32, KC = 128, KS = KC / 16, plane = KS * 2 * NW;
    extern __shared__ uint4 Bq[];                     // [3][KS][2][NW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const int k0 = blockIdx.z * KC;
    const int kc = (K - k0 < KC) ? (K - k0) : KC;
    int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + li;
    row = row < M ? row : M - 1;
    float4 af[KS][2];                                 // the wave's A fragments of the chunk: all in flight behind the B fill
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kk = st * 16 + 8 * half + 4 * q;
            af[st][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (GAT_ABLATE(1)) continue;
            if constexpr (VEC4) {
                if (kk < kc) af[st][q] = as.load4(row, k0 + kk);
            } else {
                if (kk + 0 < kc) af[st][q].x = as.load1(row, k0 + kk + 0);
                if (kk + 1 < kc) af[st][q].y = as.load1(row, k0 + kk + 1);
                if (kk + 2 < kc) af[st][q].z = as.load1(row, k0 + kk + 2);
                if (kk + 3 < kc) af[st][q].w = as.load1(row, k0 + kk + 3);
            }
        }
    for (int idx = threadIdx.x; idx < KS * 2 * NW; idx += 256) {   // consecutive threads -> consecutive k-octets of one column
        if (GAT_ABLATE(2)) break;
        const int kh = idx % (KS * 2), n = idx / (KS * 2);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = kh * 8 + j;
            if (GAT_ABLATE(16)) v[j] = (float)(kk + n) * 0.001f;          // (experiment: no loads)
            else v[j] = (kk < kc && n0 + n < N) ? bs.at(k0 + kk, n0 + n) : 0.f;
        }
        if (GAT_ABLATE(32)) { if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] == 12345.678f) Bq[0] = make_uint4(1u, 2u, 3u, 4u); continue; }   // (experiment: loads only)
        const Pieces p = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        Bq[kh * NW + n] = p.hi; Bq[plane + kh * NW + n] = p.mid; Bq[2 * plane + kh * NW + n] = p.lo;
    }
    __syncthreads();
    v16f acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        if (GAT_ABLATE(4)) break;
        const Pieces a = split8(af[st][0], af[st][1]);
        const uint4* bp = Bq + (st * 2 + half) * NW + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint4 bh = bp[nt * 32], bm = bp[plane + nt * 32], bl = bp[2 * plane + nt * 32];
#ifdef GAT_X3_EIGHT_TERMS
            acc[nt] = mfma_bf16(a.lo, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bl, acc[nt]);
#endif
            acc[nt] = mfma_bf16(a.lo, bh, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bl, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.mid, bh, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bm, acc[nt]);
            acc[nt] = mfma_bf16(a.hi, bh, acc[nt]);
        }
    }
    float* out = slabs + (int64_t)blockIdx.z * M * N;
    if (GAT_ABLATE(8)) { if (acc[0][0] == 12345.678f) out[0] = acc[0][1]; return; }
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
    static const bool off = [] { const char* e = choice_env("GAT_PROJECT_SPLITK"); return e && e[0] == '0'; }();     // A/B
    static const int x3 = [] { const char* e = choice_env("GAT_GEMM_X3"); return e ? atoi(e) : -1; }();
    static const int kmax_env = [] { const char* e = choice_env("GAT_X3_KMAX"); return e ? atoi(e) : 128; }();            // A/B: 512 = row-streaming kernel up to K = 512
    const int kmax = x3 != 0 ? kmax_env : 128;       // up to here the row-streaming kernel keeps all of K in LDS (run_rowgemm)
    return !off && K > kmax && ((M + 127) / 128) * ((N + 127) / 128) < 256;
}

template <class AS, class BS, class EP>
int run_rowgemm(const AS& as, const BS& bs, const EP& ep, int64_t M, int32_t N, int32_t K, bool vec4, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int NT = N > 64 ? 4 : (N > 32 ? 2 : 1);
    static const int x3 = [] { const char* e = choice_env("GAT_GEMM_X3"); return e ? atoi(e) : -1; }();  // A/B: 0 = fp32 MFMA; 4 / 8 = force waves per block
    // B (all of K x the block's columns, three piece planes) must fit in LDS: K <= 128 with 128 columns per block.  Narrower
    // column slices would take K up to 512 (the Pubmed shape's 500: four slices, A re-read per slice from L2), but measured
    // there it only ties the split-K kernel (0.245-0.259 vs 0.252 ms per step) and one 500-long accumulation chain is less
    // accurate than four 128-long ones: split-K stays the default beyond K = 128 (GAT_X3_KMAX=512 is the A/B).
    int NT3 = NT;
    while (NT3 > 1 && (size_t)3 * ((K + 15) / 16) * 2 * (NT3 * 32) * sizeof(uint4) > 100 * 1024) NT3 >>= 1;
    if (x3 != 0 && (size_t)3 * ((K + 15) / 16) * 2 * (NT3 * 32) * sizeof(uint4) <= 100 * 1024) {
        NT = NT3;
        const int NW = NT * 32;
        const int KS = (K + 15) / 16;
        const size_t lds3 = (size_t)3 * KS * 2 * NW * sizeof(uint4);
        const int ny = (N + NW - 1) / NW;
#define GAT_ROWGEMM_X3(NT_, W_, V_)                                                                           \
    {                                                                                                         \
        auto kern = rowgemm_x3_kernel<NT_, W_, V_, AS, BS, EP>;                                               \
        if (lds3 > 64 * 1024) allow_big_lds((const void*)kern);                                               \
        const int64_t nt_ = (M + W_ * 32 - 1) / (W_ * 32);                                                    \
        int64_t res = resident_blocks((const void*)kern, lds3, W_ * 64) / ny;     /* persistent: one round */  \
        if (res < 1) res = 1;                                                                                 \
        const dim3 grid((unsigned)(nt_ < res ? nt_ : res), (unsigned)ny);                                     \
        hipLaunchKernelGGL(kern, grid, dim3(W_ * 64), lds3, s, as, bs, ep, M, N, K);                           \
    }
#define GAT_ROWGEMM_X3_NT(W_, V_)                                                                             \
    { if (NT == 4) GAT_ROWGEMM_X3(4, W_, V_) else if (NT == 2) GAT_ROWGEMM_X3(2, W_, V_) else GAT_ROWGEMM_X3(1, W_, V_) }
        if (x3 == 4) { if (vec4) GAT_ROWGEMM_X3_NT(4, true) else GAT_ROWGEMM_X3_NT(4, false) }
        else { if (vec4) GAT_ROWGEMM_X3_NT(8, true) else GAT_ROWGEMM_X3_NT(8, false) }
#undef GAT_ROWGEMM_X3_NT
#undef GAT_ROWGEMM_X3
        GAT_HIP(hipGetLastError());
        return 0;
    }
    const int NW = NT * 32;
    const int kc_lds = ((K < kKC ? K : kKC) + 7) & ~7;
    const size_t lds = (size_t)kc_lds * NW * sizeof(float);
    const int64_t ntiles = (M + 127) / 128;
    if constexpr (!EP::kTwoPhase) {
        static const bool pipe = [] { const char* e = choice_env("GAT_GEMM_PIPE"); return !(e && e[0] == '0'); }();   // A/B
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
                                                    int32_t c_base, int32_t M, int32_t ldx) {
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
                    if (c < BN && cj < F) rb[p] = *reinterpret_cast<const float4*>(X + node * ldx + cj);
                } else {
                    float t[4], u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cc = ci + q, jj = cj + q;
                        t[q] = (c < BM && (cc - c_base) < M) ? (cc < HD ? gPL[node * HD + cc] : gPR[node * HD + (cc - HD)]) : 0.f;
                        u[q] = (c < BN && jj < F) ? X[node * ldx + jj] : 0.f;
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

// ---- grad_w on the bf16 pipe (three-piece operands, as rowgemm_x3_kernel) ---------------------------------------------
// Both operands are k-strided in memory (k = node is the row index of [gPL|gPR] and of X), and the 32x32x16 operand wants
// 8 consecutive k per lane: node tiles of 16 are cut into pieces on their way into LDS, kept there in memory orientation
// ([node][channel] bf16, one image per piece) and read back transposed by ds_read_b64_tr_b16 (a 16-lane group reads 4
// rows x 16 columns, each lane receiving one column).  Image layout: 8-row x 32-column subtiles of 512 B, the four 16-B
// chunks of a subtile row XORed with (row>>2)&3 (conflict-free for the 8-B piece writes and the transposed reads).
// Global loads run kDepth node tiles ahead in registers; one __syncthreads per 16 nodes.
template <int COLS>
__device__ __forceinline__ int img_off(int row, int col4) {                     // byte offset of columns col4*4 .. +3 of a row
    const int ch = col4 >> 1;
    return (COLS / 32) * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3)) + 8 * (col4 & 1);
}
typedef short tr4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 tr_frag(const char* img, int off0, int off1) {  // rows 8h..8h+3 | 8h+4..8h+7 of one column
    const tr4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(img + off0));
    const tr4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(img + off1));
    const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
    return make_uint4(ua.x, ua.y, ub.x, ub.y);
}

template <bool VEC4, int WM, int BN, int kDepth = 4>
__global__ __launch_bounds__(256) void gradw_x3_kernel(const float* __restrict__ gPL, const float* __restrict__ gPR,
                                                       const float* __restrict__ X, float* __restrict__ slabs,
                                                       int64_t n_rows, int32_t HD, int32_t F, int64_t kchunk,
                                                       int32_t c_base, int32_t M, int32_t ldx /* floats between rows of X (>= F) */) {
    constexpr int KT = 16, BM = 64 * WM;
    constexpr int WN = 4 / WM, NY = BN / WN / 32;
    static_assert(NY >= 1, "tile shape");
    constexpr int A_IMG = KT * BM * 2, B_IMG = KT * BN * 2;          // bytes of one piece image
    constexpr int STAGE = 3 * (A_IMG + B_IMG);
    constexpr int NA = KT * BM / 4 / 256, NB = KT * BN / 4 / 256;    // float4 per thread per tile
    static_assert(NA >= 1 && NB >= 1, "staging shape");
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, half = lane >> 5;
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

    struct Raw { float4 a[NA], b[NB]; };
    auto load_tile = [&](int64_t n0, Raw& t) {
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int idx = tid + 256 * p;
            const int kk = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
            const int64_t node = n0 + kk;
            t.a[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ci = c_base + i0 + c;
            if (node < ke && i0 + c < M) {
                if constexpr (VEC4) {
                    t.a[p] = ci < HD ? *reinterpret_cast<const float4*>(gPL + node * HD + ci)
                                     : *reinterpret_cast<const float4*>(gPR + node * HD + (ci - HD));
                } else {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cc = ci + q;
                        v[q] = (cc - c_base) < M ? (cc < HD ? gPL[node * HD + cc] : gPR[node * HD + (cc - HD)]) : 0.f;
                    }
                    t.a[p] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int idx = tid + 256 * p;
            const int kk = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
            const int64_t node = n0 + kk;
            t.b[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int cj = j0 + c;
            if (node < ke && cj < F) {
                if constexpr (VEC4) {
                    t.b[p] = *reinterpret_cast<const float4*>(X + node * ldx + cj);
                } else {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (cj + q < F) ? X[node * ldx + cj + q] : 0.f;
                    t.b[p] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };
    auto put = [&](char* img, int img_bytes, int off, const float4& v) {
        uint2 h, m, l;
        split_pair(v.x, v.y, h.x, m.x, l.x);
        split_pair(v.z, v.w, h.y, m.y, l.y);
        *reinterpret_cast<uint2*>(img + off) = h;
        *reinterpret_cast<uint2*>(img + img_bytes + off) = m;
        *reinterpret_cast<uint2*>(img + 2 * img_bytes + off) = l;
    };
    auto store_tile = [&](int buf, const Raw& t) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < NA; ++p) {
            const int idx = tid + 256 * p;
            put(base, A_IMG, img_off<BM>(idx / (BM / 4), idx % (BM / 4)), t.a[p]);
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int idx = tid + 256 * p;
            put(base + 3 * A_IMG, B_IMG, img_off<BN>(idx / (BN / 4), idx % (BN / 4)), t.b[p]);
        }
    };
    // transposed-read addresses of this lane (Mechanism of ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q,
    // columns 4p..4p+3 of the group's 4 x 16 block); group g = lane>>4 serves operand lanes (col 16(g&1) + i, half g>>1)
    const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int trow = 8 * (g >> 1) + tq;               // + 4 for the second read
    int offA[2][2], offB[NY][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            offA[x][j] = img_off<BM>(trow + 4 * j, (wm * 64 % BM + x * 32 + 16 * (g & 1)) / 4 + tp);
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            offB[y][j] = img_off<BN>(trow + 4 * j, (wn * (NY * 32) + y * 32 + 16 * (g & 1)) / 4 + tp);
    auto mma_tile = [&](int buf) {
        const char* base = lds + buf * STAGE;
        uint4 a[2][3], b[NY][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int x = 0; x < 2; ++x) a[x][pl] = tr_frag(base + pl * A_IMG, offA[x][0], offA[x][1]);
#pragma unroll
            for (int y = 0; y < NY; ++y) b[y][pl] = tr_frag(base + 3 * A_IMG + pl * B_IMG, offB[y][0], offB[y][1]);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < NY; ++y) {
#ifdef GAT_X3_EIGHT_TERMS
                acc[x][y] = mfma_bf16(a[x][2], b[y][1], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][1], b[y][2], acc[x][y]);
#endif
                acc[x][y] = mfma_bf16(a[x][2], b[y][0], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][0], b[y][2], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][1], b[y][1], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][1], b[y][0], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][0], b[y][1], acc[x][y]);
                acc[x][y] = mfma_bf16(a[x][0], b[y][0], acc[x][y]);
            }
    };

    const int64_t ntile = (ke - kb + KT - 1) / KT;
    Raw ring[kDepth];                                 // ring[t % kDepth] holds node tile t until it is cut into LDS
    if (ntile > 0) {
        load_tile(kb, ring[0]);
        store_tile(0, ring[0]);
#pragma unroll
        for (int d = 1; d < kDepth; ++d) load_tile(kb + (int64_t)d * KT, ring[d]);      // past ke: zeros, no loads
        load_tile(kb + (int64_t)kDepth * KT, ring[0]);
    }
    __syncthreads();
    for (int64_t t0 = 0; t0 < ntile; t0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int64_t t = t0 + d;
            if (t < ntile) {
                mma_tile((int)(t & 1));
                if (t + 1 < ntile) {
                    store_tile((int)((t + 1) & 1), ring[(d + 1) % kDepth]);
                    load_tile(kb + (t + 1 + kDepth) * KT, ring[(d + 1) % kDepth]);
                }
                __syncthreads();
            }
        }
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

static bool grad_w_x3() {
    static const bool on = [] { const char* e = choice_env("GAT_GRADW_X3"); return !(e && e[0] == '0'); }();     // A/B: 0 = fp32 MFMA
    return on;
}
int64_t grad_w_kchunk(int64_t n_rows, int32_t F, int32_t M) {
    const int bm = grad_w_bm(M), bn = grad_w_bn(M, F);
    const int64_t tiles = (((int64_t)M + bm - 1) / bm) * (((int64_t)F + bn - 1) / bn);
    // split-K over exactly the blocks that fit at once (LDS-limited: 64 / 48 KiB per block)
    const void* fn = grad_w_x3() ? (bm == 128 ? (bn == 64 ? (const void*)gradw_x3_kernel<true, 2, 64> : (const void*)gradw_x3_kernel<true, 2, 128>)
                                              : (const void*)gradw_x3_kernel<true, 1, 128>)
                                 : (bm == 128 ? (bn == 64 ? (const void*)gradw_kernel<true, 2, 64> : (const void*)gradw_kernel<true, 2>)
                                              : (const void*)gradw_kernel<true, 1>);
    int64_t splits = resident_blocks(fn, 0) / tiles;
    if (splits < 1) splits = 1;
    int64_t kchunk = (n_rows + splits - 1) / splits;
    kchunk = ((kchunk + 31) / 32) * 32;
    // shortest node chunk per block (GAT_GRADW_KMIN, A/B): a block's time is its chain of 16-node tiles (load -> cut into LDS ->
    // barrier -> MFMA), so few rows want short chains and more slabs; many rows never reach this bound
    static const int kmin = [] { const char* e = choice_env("GAT_GRADW_KMIN"); const int v = e ? atoi(e) : 0; return v >= 32 ? (v / 32) * 32 : 128; }();       // 128: Cora-shape 187.7 -> 182.6 us, Pubmed-shape 248 -> 242 us per step (256 before; 64: the same)
    if (kchunk < kmin) kchunk = kmin;
    return kchunk;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int64_t project_scratch_floats(int64_t n_rows, int32_t F, int32_t HD, int32_t part) {
    const int32_t N = part == kPartBoth ? 2 * HD : HD;
    return project_wants_splitk(n_rows, N, F) ? (int64_t)((F + 127) / 128) * n_rows * N : 0;
}

int launch_project(const float* X, const float* W, float* PL_rows, float* PR, int64_t n_rows, int32_t F,
                   int32_t HD, int32_t part, bool pl_bf16, float* scratch, int64_t scratch_floats, hipStream_t s, int32_t ldx) {
    const int32_t j0 = part == kPartRight ? HD : 0;
    if (ldx < F) ldx = F;
    ASrcRows as{X, ldx};
    BSrcProject bs{W, F, HD, j0};
    const bool vec4 = (ldx % 4 == 0) && aligned16(X);      // F itself, or a padded pitch with zeros behind column F (see launch_grad_w)
    const int32_t N = part == kPartBoth ? 2 * HD : HD;
    // split-K only when the caller's scratch holds this launch's slabs (the capacity is the caller's to state: a buffer
    // sized for another (rows, part) pair must not be overrun) — otherwise the streaming kernel, same results up to summation order
    if (scratch != nullptr && n_rows > 0 && project_wants_splitk(n_rows, N, F) && scratch_floats >= (int64_t)((F + 127) / 128) * n_rows * N) {
        const int ksplit = (F + 127) / 128;
        const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((N + 127) / 128), (unsigned)ksplit);
        static const bool x3 = [] { const char* e = choice_env("GAT_GEMM_X3"); return !(e && e[0] == '0'); }();       // A/B: 0 = fp32 MFMA
        // MEASURED AND NOT KEPT (profiles/r04/experiments/small_shapes; DESIGN §4 "Round 4"): a latency-sized form of this launch —
        // 64-row blocks, both operand chunks staged ROW-MAJOR through LDS (whole lines per load instruction instead of 32 rows per
        // instruction), same slab bits — Cora-shape 186 -> 192 us, Pubmed-shape 246 -> 272 us per step under replay (twice the
        // blocks, each re-cutting its W chunk; the operand fetch was not the bound: the chunk's unique lines are the same either way
        // and the CUs' miss throughput caps both); with the slabs added by the LAST block of a tile (arrival counter + device-
        // scope fences, no second launch) 352 / 610 us: a release / acquire pair per block is an L2 write-back + invalidate on this
        // 8-XCD part; all loads of the B fill issued before the first is cut: 190 / 247 us (no change); the W chunk loaded with two
        // 16-byte loads per k-octet instead of eight scalar ones: 41.1 vs 40.5-42.1 us for the kernel (no change).  Where the kernel's
        // 40.5 us go on the Pubmed shape (tools/splitk_ablate.sh, experiment library, parts of the block skipped): nothing but the launch
        // of its 1,240 blocks 5.2; without A loads 38.9, without the W fill 27.5 (fill from constants: 32.7; loads without the cut: 41.3),
        // without the MFMA loop 35.8, without the slab stores 34.0 — the block is a chain of its W-chunk load latency, the cut, the
        // barrier, 96 MFMAs and 32 store instructions per lane at 12 waves per CU, not a throughput problem of any one part.
        if (x3) {
            // 64-column blocks (48 KiB of LDS: three per CU) unless GAT_SPLITK_NT=4: Pubmed-shape 0.311 -> 0.302 ms per step
            static const int nt = [] { const char* e = choice_env("GAT_SPLITK_NT"); return e && atoi(e) == 4 ? 4 : 2; }();
            const size_t lds = (size_t)3 * 8 * 2 * (nt * 32) * sizeof(uint4);
            if (nt == 4) { allow_big_lds((const void*)project_splitk_x3_kernel<true, 4>); allow_big_lds((const void*)project_splitk_x3_kernel<false, 4>); }
            const dim3 grid3((unsigned)((n_rows + 127) / 128), (unsigned)((N + nt * 32 - 1) / (nt * 32)), (unsigned)ksplit);
#ifdef GAT_EXPERIMENTS
            static const int ablate = [] { const char* e = getenv("GAT_DBG_SPLITK"); return e ? atoi(e) : 0; }();
#define GAT_ABLATE_ARG , ablate
#else
#define GAT_ABLATE_ARG
#endif
            if (nt == 4) {
                if (vec4) hipLaunchKernelGGL((project_splitk_x3_kernel<true, 4>), grid3, dim3(256), lds, s, as, bs, scratch, n_rows, N, F GAT_ABLATE_ARG);
                else hipLaunchKernelGGL((project_splitk_x3_kernel<false, 4>), grid3, dim3(256), lds, s, as, bs, scratch, n_rows, N, F GAT_ABLATE_ARG);
            } else {
                if (vec4) hipLaunchKernelGGL((project_splitk_x3_kernel<true, 2>), grid3, dim3(256), lds, s, as, bs, scratch, n_rows, N, F GAT_ABLATE_ARG);
                else hipLaunchKernelGGL((project_splitk_x3_kernel<false, 2>), grid3, dim3(256), lds, s, as, bs, scratch, n_rows, N, F GAT_ABLATE_ARG);
            }
#undef GAT_ABLATE_ARG
        } else {
            const size_t lds = (size_t)(128 * 128) * sizeof(float);
            if (vec4) hipLaunchKernelGGL(project_splitk_kernel<true>, grid, dim3(256), lds, s, as, bs, scratch, n_rows, N, F);
            else hipLaunchKernelGGL(project_splitk_kernel<false>, grid, dim3(256), lds, s, as, bs, scratch, n_rows, N, F);
        }
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
                  int64_t n_rows, int32_t F, int32_t HD, int32_t part, hipStream_t s, int32_t ldx) {
    if (n_rows <= 0) return 0;
    if (ldx < F) ldx = F;
    const int M = part == kPartBoth ? 2 * HD : HD;
    const int c_base = part == kPartRight ? HD : 0;
    if (part == kPartLeft) gPR = gPL_rows;           // never dereferenced; keeps the alignment test meaningful
    if (part == kPartRight) gPL_rows = gPR;
    const int64_t kchunk = grad_w_kchunk(n_rows, F, M);
    const int64_t ksplit = (n_rows + kchunk - 1) / kchunk;
    const int bm = grad_w_bm(M), bn = grad_w_bn(M, F);
    const dim3 grid((unsigned)((F + bn - 1) / bn), (unsigned)((M + bm - 1) / bm), (unsigned)ksplit);
    // 16-byte loads need a row pitch of whole float4s: F itself, or the caller's padded pitch (zeros behind column F: the context
    // pads an odd in_dim — Cora's 1,433 — once at gat_set_features)
    const bool vec4 = (ldx % 4 == 0) && (HD % 4 == 0) && aligned16(X) && aligned16(gPL_rows) && aligned16(gPR);
#define GAT_GRADW(V_, WM_, BN_)                                                                                             \
    do {                                                                                                                    \
        if (grad_w_x3()) hipLaunchKernelGGL((gradw_x3_kernel<V_, WM_, BN_>), grid, dim3(256), 0, s, gPL_rows, gPR, X, scratch, n_rows, HD, F, kchunk, c_base, M, ldx); \
        else hipLaunchKernelGGL((gradw_kernel<V_, WM_, BN_>), grid, dim3(256), 0, s, gPL_rows, gPR, X, scratch, n_rows, HD, F, kchunk, c_base, M, ldx); \
    } while (0)
    if (vec4) { if (bm == 128 && bn == 64) GAT_GRADW(true, 2, 64); else if (bm == 128) GAT_GRADW(true, 2, 128); else GAT_GRADW(true, 1, 128); }
    else { if (bm == 128 && bn == 64) GAT_GRADW(false, 2, 64); else if (bm == 128) GAT_GRADW(false, 2, 128); else GAT_GRADW(false, 1, 128); }
#undef GAT_GRADW
    GAT_HIP(hipGetLastError());
    return launch_reduce_gradw(scratch, (int32_t)ksplit, HD, F, c_base, M, gradW, s);
}

}  // namespace gat
