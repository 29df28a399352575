// gat_dense_kernels.hip — dense pieces of the GATv2 step on gfx950: the W_l/W_r feature
// projections and their backward as exact-fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), the output
// head (W_o + softmax + loss) forward/backward, optimizer and clip kernels.
//
// The reference recomputes W·x inside every per-edge thread (E:303-316, 415-420, 636-640,
// 752-761, 848-853).  Here it is computed once per node:
//   project : [PL | PR] = X · [W_left ; W_right]^T                     (fwd of E:303-316)
//   grad_w  : gradW_left += gPL^T · X,  gradW_right += gPR^T · X       (E:770-782 summed over edges)
//   grad_x  : gX = gPL · W_left + gPR · W_right, fused with E:888-892  (E:859-869 summed over edges)
// W stays in the reference layout [H*D][2F] (row j: cols 0..F-1 left, F..2F-1 right).
#include "gat_internal.h"

namespace gat {
namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

// ---- operand accessors ------------------------------------------------------------------------------
struct RowMajorK {            // element (i,k) = p[i*ld + k]   (contiguous along k)
    static constexpr bool kContigK = true;
    const float* p; int64_t ld;
    __device__ __forceinline__ float operator()(int64_t i, int64_t k) const { return p[i * ld + k]; }
};
struct RowMajorMN {           // element (k,j) = p[k*ld + j]   (contiguous along the M/N index)
    static constexpr bool kContigK = false;
    const float* p; int64_t ld;
    __device__ __forceinline__ float operator()(int64_t k, int64_t j) const { return p[k * ld + j]; }
};
// B(k=f, j) of the projection: j < HD -> W[j][f], else W[j-HD][F+f]; contiguous along k.
struct WcatK {
    static constexpr bool kContigK = true;
    const float* W; int32_t F, HD;
    __device__ __forceinline__ float operator()(int64_t k, int64_t j) const {
        const int64_t r = j % HD, half = j / HD;
        return W[r * 2 * F + half * F + k];
    }
};
// B(k=c, j=f) of grad_x: Wcat[c][f]; contiguous along j.
struct WcatMN {
    static constexpr bool kContigK = false;
    const float* W; int32_t F, HD;
    __device__ __forceinline__ float operator()(int64_t k, int64_t j) const {
        const int64_t r = k % HD, half = k / HD;
        return W[r * 2 * F + half * F + j];
    }
};
// A(i=node, k=c) of grad_x: [gPL | gPR] concatenated along k; contiguous along k.
struct GcatK {
    static constexpr bool kContigK = true;
    const float* gPL; const float* gPR; int32_t HD;
    __device__ __forceinline__ float operator()(int64_t i, int64_t k) const {
        return k < HD ? gPL[i * HD + k] : gPR[i * HD + (k - HD)];
    }
};
// A(i=c, k=node) of grad_w: [gPL | gPR]^T; contiguous along i.
struct GcatT {
    static constexpr bool kContigK = false;
    const float* gPL; const float* gPR; int32_t HD;
    __device__ __forceinline__ float operator()(int64_t k, int64_t i) const {   // (k=node, i=c)
        return i < HD ? gPL[k * HD + i] : gPR[k * HD + (i - HD)];
    }
};

// ---- epilogues ------------------------------------------------------------------------------------------
struct EpiProject {           // cols < HD -> PL rows, else PR
    float* PL; float* PR; int32_t HD;
    __device__ __forceinline__ void operator()(int64_t i, int64_t j, float v, int) const {
        if (j < HD) PL[i * HD + j] = v; else PR[i * HD + (j - HD)] = v;
    }
};
struct EpiSlab {              // split-K partial slabs [z][M][N]
    float* out; int64_t M, N;
    __device__ __forceinline__ void operator()(int64_t i, int64_t j, float v, int z) const {
        out[((int64_t)z * M + i) * N + j] = v;
    }
};
struct EpiGradX {             // g_prev = gX ⊙ LReLU'(h_pre_prev)   (E:888-892)
    float* out; const float* hpre_prev; int64_t ld; float slope;
    __device__ __forceinline__ void operator()(int64_t i, int64_t j, float v, int) const {
        const float hv = hpre_prev[i * ld + j];
        out[i * ld + j] = v * (hv > 0.f ? 1.0f : slope);
    }
};

// ---- LDS-tiled fp32 MFMA GEMM:  C(i,j) = sum_k A(i,k) B(k,j) over k in this block's K slice ----
// 256 threads = 2x2 waves, each wave (BM/2)x(BN/2) in 32x32 MFMA tiles.  Operands are staged
// k-major in LDS (As[k][i], Bs[k][j], +1 pad) so that the A/B fragments of
// v_mfma_f32_32x32x2_f32 (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) are conflict-free reads.
// AL/BL accessors take (row-of-their-matrix, col) as documented on each struct; the kContigK flag
// picks the thread->element map that keeps global loads coalesced.
template <int BM, int BN, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void gemm_f32_kernel(AL la, BL lb, EP ep, int64_t M, int64_t N,
                                                       int64_t K, int64_t kchunk) {
    constexpr int BK = 32, PAD = 1;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t i0 = (int64_t)blockIdx.y * BM, j0 = (int64_t)blockIdx.x * BN;
    const int64_t kb = (int64_t)blockIdx.z * kchunk;
    const int64_t ke = (kb + kchunk < K) ? kb + kchunk : K;
    v16f acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    for (int64_t k0 = kb; k0 < ke; k0 += BK) {
        if constexpr (AL::kContigK) {
            const int kk = tid % BK, ii = tid / BK;
#pragma unroll 4
            for (int i = ii; i < BM; i += 256 / BK)
                As[kk][i] = (i0 + i < M && k0 + kk < ke) ? la(i0 + i, k0 + kk) : 0.f;
        } else {
            const int ii = tid % BM, kq = tid / BM;
#pragma unroll 4
            for (int kk = kq; kk < BK; kk += 256 / BM)
                As[kk][ii] = (i0 + ii < M && k0 + kk < ke) ? la(k0 + kk, i0 + ii) : 0.f;
        }
        if constexpr (BL::kContigK) {
            const int kk = tid % BK, jj = tid / BK;
#pragma unroll 4
            for (int j = jj; j < BN; j += 256 / BK)
                Bs[kk][j] = (j0 + j < N && k0 + kk < ke) ? lb(k0 + kk, j0 + j) : 0.f;
        } else {
            const int jj = tid % BN, kq = tid / BN;
#pragma unroll 4
            for (int kk = kq; kk < BK; kk += 256 / BN)
                Bs[kk][jj] = (j0 + jj < N && k0 + kk < ke) ? lb(k0 + kk, j0 + jj) : 0.f;
        }
        __syncthreads();
        const int klen = (ke - k0 < BK) ? (int)(ke - k0) : BK;
        const int ksteps = (klen + 1) >> 1;
        for (int ks = 0; ks < ksteps; ++ks) {
            const int kk = ks * 2 + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = As[kk][wm * WM + tm * 32 + (lane & 31)];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[kk][wn * WN + tn * 32 + (lane & 31)];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = i0 + wm * WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t col = j0 + wn * WN + tn * 32 + (lane & 31);
                if (row < M && col < N) ep(row, col, acc[tm][tn][r], (int)blockIdx.z);
            }
}

template <int BM, int BN, class AL, class BL, class EP>
int run_gemm(const AL& la, const BL& lb, const EP& ep, int64_t M, int64_t N, int64_t K, int64_t kchunk,
             hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    const int64_t ksplit = (K + kchunk - 1) / kchunk;
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)(ksplit < 1 ? 1 : ksplit));
    if (grid.y > 65535u * 1024u) return fail(GAT_E_UNSUPPORTED, "gemm: M too large");
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AL, BL, EP>), grid, dim3(256), 0, s, la, lb, ep, M, N, K, kchunk);
    GAT_HIP(hipGetLastError());
    return 0;
}

// out[map(idx)] += sum_z slabs[z][idx].  Block = 64 columns x 16 z-slices, 4 independent partial
// sums per thread, combined in a fixed order: bitwise reproducible, and short dependent chains
// (nslabs/64 rounds) instead of one thread walking all slabs.
struct MapIdentity {
    __device__ __forceinline__ int64_t operator()(int64_t idx) const { return idx; }
};
struct MapGradW {             // idx = c*F + f over [2HD][F]  ->  reference layout [HD][2F]
    int32_t HD, F;
    __device__ __forceinline__ int64_t operator()(int64_t idx) const {
        const int64_t c = idx / F, f = idx % F;
        return (c % HD) * 2 * F + (c / HD) * F + f;
    }
};
template <class MAP>
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, int32_t nslabs,
                                                           int64_t width, float* __restrict__ out, MAP map) {
    __shared__ float part[16][64];
    const int q = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t idx = (int64_t)blockIdx.x * 64 + q;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < width) {
        int z = sl;
        for (; z + 48 < nslabs; z += 64) {
            s0 += slabs[(int64_t)z * width + idx];
            s1 += slabs[(int64_t)(z + 16) * width + idx];
            s2 += slabs[(int64_t)(z + 32) * width + idx];
            s3 += slabs[(int64_t)(z + 48) * width + idx];
        }
        for (; z < nslabs; z += 16) s0 += slabs[(int64_t)z * width + idx];
    }
    part[sl][q] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && idx < width) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += part[w][q];
        out[map(idx)] += tot;
    }
}

// ---- output head ----------------------------------------------------------------------------------------------
// C12+C13 (E:463-537): z = Wo·H_L[n]; y = exp(z-max)/(sum+1e-8) (double divide, E:140);
// loss_n = -log(max(y[label],1e-12)); correct_n = (argmax == label).  One thread per node, the
// block's y tile lives in LDS ([NB][ldz], ldz odd) and is written back coalesced.
__global__ __launch_bounds__(256) void head_forward_kernel(HeadArgs A, int32_t NB, int32_t ldz) {
    extern __shared__ float lds[];
    float* s_wo = lds;                               // [C*DL]
    float* s_z = lds + (int64_t)A.C * A.DL;          // [NB][ldz]
    __shared__ double s_loss[4];
    __shared__ int32_t s_corr[4];
    const int C = A.C, DL = A.DL;
    for (int i = threadIdx.x; i < C * DL; i += blockDim.x) s_wo[i] = A.Wo[i];
    __syncthreads();
    double loss_acc = 0.0;
    int corr_acc = 0;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int t = threadIdx.x;
        const int64_t n = n0 + t;
        if (t < NB && n < A.n_rows) {
            float* zr = s_z + (int64_t)t * ldz;
            const float* x = A.HL + n * DL;
            float mv = -INFINITY;
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int j = 0; j < DL; ++j) acc += s_wo[c * DL + j] * x[j];
                zr[c] = acc;
                mv = fmaxf(mv, acc);
            }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) { const float ev = expf(zr[c] - mv); zr[c] = ev; sum += ev; }
            const double den = (double)sum + 1e-8;
            const int lab = A.labels[n];
            float best = -1.f; int pred = 0; float plab = 0.f;
            for (int c = 0; c < C; ++c) {
                const float yv = (float)((double)zr[c] / den);
                zr[c] = yv;
                if (c == 0 || yv > best) { best = yv; pred = c; }     // strict >, first max wins
                if (c == lab) plab = yv;
            }
            loss_acc += (double)(-logf(fmaxf(plab, 1e-12f)));
            corr_acc += (pred == lab) ? 1 : 0;
        }
        __syncthreads();
        const int64_t rows_here = (A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB;
        const int64_t tot = rows_here * C;
        for (int64_t q = threadIdx.x; q < tot; q += blockDim.x)
            A.y[n0 * C + q] = s_z[(q / C) * ldz + (q % C)];
        __syncthreads();
    }
    // block reduction (fixed order)
    for (int off = 32; off > 0; off >>= 1) {
        loss_acc += __shfl_down(loss_acc, off);
        corr_acc += __shfl_down(corr_acc, off);
    }
    if ((threadIdx.x & 63) == 0) { s_loss[threadIdx.x >> 6] = loss_acc; s_corr[threadIdx.x >> 6] = corr_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.loss_partial[blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        A.correct_partial[blockIdx.x] = s_corr[0] + s_corr[1] + s_corr[2] + s_corr[3];
    }
}

__global__ __launch_bounds__(64) void head_finalize_kernel(const double* __restrict__ lp,
                                                          const int32_t* __restrict__ cp, int32_t nb,
                                                          float* loss_out, int32_t* correct_out) {
    double l = 0.0; int c = 0;
    for (int i = threadIdx.x; i < nb; i += 64) { l += lp[i]; c += cp[i]; }
    for (int off = 32; off > 0; off >>= 1) { l += __shfl_down(l, off); c += __shfl_down(c, off); }
    if (threadIdx.x == 0) { *loss_out = (float)l; *correct_out = c; }
}

// C14 (E:553-608): dz = y - onehot; gradWo += dz^T·H_L; g[n,h,d] = (Wo^T dz)[d]·LReLU'(h_pre)/H.
// Tile of NB nodes in LDS; grad_Wo as a per-block register partial (thread = one (c,d) entry),
// g written with a coalesced sweep.
__global__ __launch_bounds__(256) void head_backward_kernel(HeadBwdArgs A, int32_t NB, int32_t ldz, int32_t per_thread) {
    extern __shared__ float lds[];
    const int C = A.C, DL = A.DL, H = A.H;
    float* s_wo = lds;                                   // [C*DL]
    float* s_dz = s_wo + (int64_t)C * DL;                // [NB][ldz]
    float* s_hl = s_dz + (int64_t)NB * ldz;              // [NB][DL]
    float* s_gh = s_hl + (int64_t)NB * DL;               // [NB][DL]
    for (int i = threadIdx.x; i < C * DL; i += blockDim.x) s_wo[i] = A.Wo[i];
    float wacc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) wacc[p] = 0.f;
    const float inv_heads = 1.0f / (float)H;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int64_t rows_here = (A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB;
        for (int64_t q = threadIdx.x; q < rows_here * C; q += blockDim.x) {
            const int64_t r = q / C; const int c = (int)(q % C);
            s_dz[r * ldz + c] = A.y[n0 * C + q] - (c == A.labels[n0 + r] ? 1.0f : 0.0f);
        }
        for (int64_t q = threadIdx.x; q < rows_here * DL; q += blockDim.x) s_hl[q] = A.HL[n0 * DL + q];
        __syncthreads();
        for (int64_t q = threadIdx.x; q < rows_here * DL; q += blockDim.x) {
            const int64_t r = q / DL; const int d = (int)(q % DL);
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += s_wo[c * DL + d] * s_dz[r * ldz + c];
            s_gh[q] = sum;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int t = threadIdx.x + p * 256;
            if (p < per_thread && t < C * DL) {
                const int c = t / DL, d = t % DL;
                float acc = wacc[p];
                for (int r = 0; r < (int)rows_here; ++r) acc += s_dz[(int64_t)r * ldz + c] * s_hl[(int64_t)r * DL + d];
                wacc[p] = acc;
            }
        }
        __syncthreads();
        const int64_t hd = (int64_t)H * DL;
        for (int64_t q = threadIdx.x; q < rows_here * hd; q += blockDim.x) {
            const int64_t r = q / hd; const int d = (int)(q % DL);
            const int64_t hi = A.flat_index ? ((n0 + r) * DL + d) : (n0 * hd + q);
            const float hv = A.hpre[hi];
            A.g[n0 * hd + q] = s_gh[r * DL + d] * (hv > 0.f ? 1.0f : A.slope) * inv_heads;
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int t = threadIdx.x + p * 256;
        if (p < per_thread && t < C * DL) A.partial[(int64_t)blockIdx.x * C * DL + t] = wacc[p];
    }
}

// ---- optimizer / clip (E:146-177, 250-278, 896-923) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float lr, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] -= lr * g[i];
}
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, float lr,
                                                  int64_t n, float b1, float b2, float eps, int32_t t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * (gi * gi);
        m[i] = mi; v[i] = vi;
        const float mh = mi / (1.0f - powf(b1, (float)t));
        const float vh = vi / (1.0f - powf(b2, (float)t));
        p[i] -= lr * mh / (sqrtf(vh) + eps);
    }
}
// one block: sum of squares in a fixed order, result (the squared norm) to *out
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* out) {
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        *out = t;
    }
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float* g, int64_t n, const float* sumsq, float thresh) {
    const float norm = sqrtf(*sumsq);
    float scale = 1.0f;
    if (norm > thresh) scale = thresh / (norm + 1e-9f);
    if (!(scale < 1.0f)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] *= scale;
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------------------------------
int launch_project(const float* X, const float* W, float* PL_rows, float* PR, int64_t n_rows, int32_t F,
                   int32_t HD, hipStream_t s) {
    RowMajorK la{X, F};
    WcatK lb{W, F, HD};
    EpiProject ep{PL_rows, PR, HD};
    const int64_t N = 2 * (int64_t)HD;
    if (N <= 64) return run_gemm<128, 64>(la, lb, ep, n_rows, N, F, F, s);
    return run_gemm<128, 128>(la, lb, ep, n_rows, N, F, F, s);
}

static int64_t grad_w_kchunk(int64_t n_rows, int32_t F, int32_t HD) {
    const int64_t tiles = ((2 * (int64_t)HD + 127) / 128) * (((int64_t)F + 127) / 128);
    int64_t splits = 1024 / tiles;
    if (splits < 1) splits = 1;
    int64_t kchunk = (n_rows + splits - 1) / splits;
    kchunk = ((kchunk + 31) / 32) * 32;
    if (kchunk < 256) kchunk = 256;
    return kchunk;
}
int64_t grad_w_scratch_floats(int64_t n_rows, int32_t F, int32_t HD) {
    const int64_t kchunk = grad_w_kchunk(n_rows, F, HD);
    const int64_t ksplit = (n_rows + kchunk - 1) / kchunk;
    return (ksplit < 1 ? 1 : ksplit) * 2 * HD * (int64_t)F;
}
int launch_grad_w(const float* gPL_rows, const float* gPR, const float* X, float* gradW, float* scratch,
                  int64_t n_rows, int32_t F, int32_t HD, hipStream_t s) {
    if (n_rows <= 0) return 0;
    const int64_t kchunk = grad_w_kchunk(n_rows, F, HD);
    const int64_t ksplit = (n_rows + kchunk - 1) / kchunk;
    GcatT la{gPL_rows, gPR, HD};
    RowMajorMN lb{X, F};
    const int64_t M = 2 * (int64_t)HD;
    EpiSlab ep{scratch, M, F};
    if (F <= 64) { GAT_TRY((run_gemm<128, 64>(la, lb, ep, M, F, n_rows, kchunk, s))); }
    else { GAT_TRY((run_gemm<128, 128>(la, lb, ep, M, F, n_rows, kchunk, s))); }
    const int64_t width = M * F;
    hipLaunchKernelGGL((reduce_slabs_kernel<MapGradW>), dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, s,
                       scratch, (int32_t)ksplit, width, gradW, MapGradW{HD, F});
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_reduce_partials_add(const float* partial, int32_t nblocks, int64_t width, float* out, hipStream_t s) {
    if (width <= 0) return 0;
    hipLaunchKernelGGL((reduce_slabs_kernel<MapIdentity>), dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, s,
                       partial, nblocks, width, out, MapIdentity{});
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_grad_x(const float* gPL_rows, const float* gPR, const float* W, const float* hpre_prev,
                  float* gprev, int64_t n_rows, int32_t F, int32_t HD, float slope, hipStream_t s) {
    GcatK la{gPL_rows, gPR, HD};
    WcatMN lb{W, F, HD};
    EpiGradX ep{gprev, hpre_prev, F, slope};
    const int64_t K = 2 * (int64_t)HD;
    if (F <= 64) return run_gemm<128, 64>(la, lb, ep, n_rows, F, K, K, s);
    return run_gemm<128, 128>(la, lb, ep, n_rows, F, K, K, s);
}

static int head_tile(int32_t C, int32_t DL, int32_t extra_per_row, int32_t* NB, int32_t* ldz) {
    const int l = (C % 2 == 0) ? C + 1 : C;               // odd row stride: conflict-free rows
    int64_t budget = 60 * 1024 / 4 - (int64_t)C * DL;     // floats
    int nb = 256;
    while (nb >= 32 && (int64_t)nb * (l + extra_per_row) > budget) nb >>= 1;
    if (nb < 32) return fail(GAT_E_UNSUPPORTED, "output head: num_classes*D_last too large for the LDS tile");
    *NB = nb; *ldz = l;
    return 0;
}
int head_blocks(int64_t n_rows) {
    const int64_t b = (n_rows + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
int launch_head_forward(const HeadArgs& a, hipStream_t s) {
    int32_t NB, ldz;
    GAT_TRY(head_tile(a.C, a.DL, 0, &NB, &ldz));
    const int blocks = head_blocks(a.n_rows);
    const size_t lds = ((size_t)a.C * a.DL + (size_t)NB * ldz) * sizeof(float);
    hipLaunchKernelGGL(head_forward_kernel, dim3(blocks), dim3(256), lds, s, a, NB, ldz);
    GAT_HIP(hipGetLastError());
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(64), 0, s, a.loss_partial, a.correct_partial, blocks,
                       a.loss_out, a.correct_out);
    GAT_HIP(hipGetLastError());
    return 0;
}
int head_bwd_blocks(int64_t n_rows, int32_t C, int32_t DL) {
    (void)C; (void)DL;
    const int64_t b = (n_rows + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
int launch_head_backward(const HeadBwdArgs& a, hipStream_t s) {
    int32_t NB, ldz;
    GAT_TRY(head_tile(a.C, a.DL, 2 * a.DL, &NB, &ldz));
    const int per_thread = (a.C * a.DL + 255) / 256;
    if (per_thread > 8) return fail(GAT_E_UNSUPPORTED, "output head: num_classes*D_last > 2048");
    const int blocks = head_bwd_blocks(a.n_rows, a.C, a.DL);
    const size_t lds = ((size_t)a.C * a.DL + (size_t)NB * (ldz + 2 * a.DL)) * sizeof(float);
    hipLaunchKernelGGL(head_backward_kernel, dim3(blocks), dim3(256), lds, s, a, NB, ldz, per_thread);
    GAT_HIP(hipGetLastError());
    return launch_reduce_partials_add(a.partial, blocks, (int64_t)a.C * a.DL, a.gradWo, s);
}

int launch_sgd(float* p, const float* g, float lr, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, lr, n);
    GAT_HIP(hipGetLastError());
    return 0;
}
int launch_adam(float* p, const float* g, float* m, float* v, float lr, int64_t n, float b1, float b2, float eps,
                int32_t t, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, lr, n, b1, b2,
                       eps, t);
    GAT_HIP(hipGetLastError());
    return 0;
}
int launch_clip(float* g, int64_t n, float thresh, float* scratch, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, s, g, n, scratch);
    GAT_HIP(hipGetLastError());
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, n, scratch, thresh);
    GAT_HIP(hipGetLastError());
    return 0;
}

}  // namespace gat
