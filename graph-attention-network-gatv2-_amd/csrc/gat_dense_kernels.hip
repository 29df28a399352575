// gat_dense_kernels.hip — node-level pieces of the GATv2 step on gfx950 that are not GEMMs:
// the output head (W_o + softmax + loss) forward/backward, the fixed-order slab reductions,
// optimizer and clip kernels.  (Projection GEMMs: gat_gemm_kernels.hip.)
#include "gat_internal.h"

#include <algorithm>

#include <cstdlib>

namespace gat {
namespace {

// out[map(idx)] += sum_z slabs[z][idx].  Block = 64 columns x 16 z-slices, 4 independent partial
// sums per thread, combined in a fixed order: bitwise reproducible, and short dependent chains
// (nslabs/64 rounds) instead of one thread walking all slabs.
struct MapIdentity {
    __device__ __forceinline__ int64_t operator()(int64_t idx) const { return idx; }
};
struct MapGradW {             // idx = (c-c_base)*F + f over rows c of [2HD][F]  ->  reference layout [HD][2F]
    int32_t HD, F, c_base;
    __device__ __forceinline__ int64_t operator()(int64_t idx) const {
        const int32_t c = c_base + (int32_t)idx / F, f = (int32_t)idx % F;
        return (int64_t)(c % HD) * 2 * F + (c / HD) * F + f;
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

// every queued reduction in one launch (ReduceBatch); block -> job by the jobs' first-block prefix; one extra block
// packs {loss, correct} like pack_result_kernel
__global__ __launch_bounds__(1024) void reduce_batch_kernel(ReduceBatch B) {
    __shared__ float part[16][64];
    if ((int)blockIdx.x >= B.blocks) {                   // the extra block: head finalize (as head_finalize_kernel), then the pack
        if (threadIdx.x >= 64) return;
        float lossv = 0.f; int32_t corrv = 0;
        if (B.fin_loss != nullptr) {
            double l = 0.0; int c = 0;
            for (int i = threadIdx.x; i < B.fin_n; i += 64) { l += B.fin_loss[i]; c += B.fin_corr[i]; }
            for (int off = 32; off > 0; off >>= 1) { l += __shfl_down(l, off); c += __shfl_down(c, off); }
            lossv = (float)l; corrv = c;
            if (threadIdx.x == 0) { *B.fin_loss_out = lossv; *B.fin_corr_out = corrv; }
        } else if (threadIdx.x == 0 && B.dst3 != nullptr) {
            lossv = *B.loss; corrv = *B.correct;
        }
        if (threadIdx.x == 0 && B.dst3 != nullptr) {
            const float v0 = lossv, v1 = (float)(corrv & 4095), v2 = (float)(corrv >> 12);
            B.dst3[0] = v0; B.dst3[1] = v1; B.dst3[2] = v2;
            if (B.dst3_host != nullptr) { B.dst3_host[0] = v0; B.dst3_host[1] = v1; B.dst3_host[2] = v2; }
        }
        return;
    }
    int j = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i < B.n && (int)blockIdx.x >= B.jobs[i].block0) j = i;
    const ReduceJob J = B.jobs[j];
    const int q = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t idx = (int64_t)((int)blockIdx.x - J.block0) * 64 + q;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < J.width) {
        int z = sl;
        for (; z + 48 < J.nslabs; z += 64) {
            s0 += J.slabs[(int64_t)z * J.width + idx];
            s1 += J.slabs[(int64_t)(z + 16) * J.width + idx];
            s2 += J.slabs[(int64_t)(z + 32) * J.width + idx];
            s3 += J.slabs[(int64_t)(z + 48) * J.width + idx];
        }
        for (; z < J.nslabs; z += 16) s0 += J.slabs[(int64_t)z * J.width + idx];
    }
    part[sl][q] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && idx < J.width) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += part[w][q];
        const int64_t o = J.HD != 0 ? MapGradW{J.HD, J.F, J.c_base}(idx) : idx;
        J.out[o] += tot;
    }
}

// ---- output head ----------------------------------------------------------------------------------------------
// C12+C13 (E:463-537): z = Wo·H_L[n]; y = exp(z-max)/(sum+1e-8) (double divide, E:140);
// loss_n = -log(max(y[label],1e-12)); correct_n = (argmax == label).  One thread per node, the
// block's y tile lives in LDS ([NB][ldz], ldz odd) and is written back coalesced.  All tile index
// arithmetic is 32-bit and incremental (no per-element division).
// DLP = D_last padded to 8/16/32/64 (the node's H_L row is held in registers, W_o rows are padded
// with zeros in LDS); DLP = 0 is the any-size fallback that re-reads the row from memory.
template <int DLP>
__global__ __launch_bounds__(256) void head_forward_kernel(HeadArgs A, int32_t NB, int32_t ldz) {
    extern __shared__ float lds[];
    constexpr int DLS = DLP > 0 ? DLP : 1;
    const int C = A.C, DL = A.DL;
    const int wstride = DLP > 0 ? DLP : DL;
    float* s_wo = lds;                               // [C][wstride]
    float* s_z = lds + C * wstride;                  // [NB][ldz]
    __shared__ double s_loss[4];
    __shared__ int32_t s_corr[4];
    for (int i = threadIdx.x; i < C * wstride; i += blockDim.x) {
        const int c = i / wstride, j = i % wstride;
        s_wo[i] = j < DL ? A.Wo[c * DL + j] : 0.f;
    }
    __syncthreads();
    double loss_acc = 0.0;
    int corr_acc = 0;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    const int dq = 256 / C, dr = 256 % C;            // advance of (row, col) per 256 linear elements
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int t = threadIdx.x;
        const int64_t n = n0 + t;
        if (t < NB && n < A.n_rows) {
            float* zr = s_z + t * ldz;
            const float* x = A.HL + n * DL;
            float mv = -INFINITY;
            if constexpr (DLP > 0) {
                float xr[DLS];
#pragma unroll
                for (int j = 0; j < DLS; ++j) xr[j] = j < DL ? x[j] : 0.f;
                for (int c = 0; c < C; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < DLS; ++j) acc += s_wo[c * DLS + j] * xr[j];   // ascending j (E:496-498)
                    zr[c] = acc;
                    mv = fmaxf(mv, acc);
                }
            } else {
                for (int c = 0; c < C; ++c) {
                    float acc = 0.f;
                    for (int j = 0; j < DL; ++j) acc += s_wo[c * DL + j] * x[j];
                    zr[c] = acc;
                    mv = fmaxf(mv, acc);
                }
            }
            float sum = 0.f;
            // E:514 uses __expf (= ex2.approx(x * log2 e)): the hardware exp2 here, not the 10-instruction libm expf
            for (int c = 0; c < C; ++c) { const float ev = __builtin_amdgcn_exp2f((zr[c] - mv) * 1.4426950408889634f); zr[c] = ev; sum += ev; }
            const double den = (double)sum + 1e-8;
            const double rden = 1.0 / den;
            const int lab = A.labels[n];
            float best = -1.f; int pred = 0; float plab = 0.f;
            for (int c = 0; c < C; ++c) {
                const float yv = (float)((double)zr[c] * rden);       // E:140 divides in double: one reciprocal + a product per class
                                                                         // (47 fp64 divisions per node made this kernel VALU-bound)
                zr[c] = yv;
                if (c == 0 || yv > best) { best = yv; pred = c; }     // strict >, first max wins
                if (c == lab) plab = yv;
            }
            // lab < 0: the node is outside the training mask (gat_set_train_mask stores ~label): no loss, no count
            loss_acc += lab >= 0 ? (double)(-logf(fmaxf(plab, 1e-12f))) : 0.0;
            corr_acc += (lab >= 0 && pred == lab) ? 1 : 0;
        }
        __syncthreads();
        const int rows_here = (int)((A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB);
        const int tot = rows_here * C;
        float* yt = A.y + n0 * C;
        int r = threadIdx.x / C, cc = threadIdx.x % C;
        for (int q = threadIdx.x; q < tot; q += 256) {
            yt[q] = s_z[r * ldz + cc];
            r += dq; cc += dr;
            if (cc >= C) { cc -= C; ++r; }
        }
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

// loss sum and #correct as three floats that survive a float all-reduce exactly: the count is split
// into 12-bit halves (each half's sum over ranks stays below 2^24 for any realistic world size)
__global__ void pack_result_kernel(const float* __restrict__ loss, const int32_t* __restrict__ correct,
                                   float* __restrict__ dst) {
    if (threadIdx.x == 0) {
        const int32_t c = *correct;
        dst[0] = *loss; dst[1] = (float)(c & 4095); dst[2] = (float)(c >> 12);
    }
}

// C14 (E:553-608): dz = y - onehot; gradWo += dz^T·H_L; g[n,h,d] = (Wo^T dz)[d]·LReLU'(h_pre)/H.
// Tile of NB nodes in LDS; grad_Wo as a per-block register partial (thread = one (c,d) entry),
// g written with a coalesced sweep.
__global__ __launch_bounds__(256) void head_backward_kernel(HeadBwdArgs A, int32_t NB, int32_t ldz, int32_t per_thread) {
    extern __shared__ float lds[];
    const int C = A.C, DL = A.DL, H = A.H;
    float* s_wo = lds;                                   // [C*DL]
    float* s_dz = s_wo + C * DL;                         // [NB][ldz]
    float* s_hl = s_dz + NB * ldz;                       // [NB][DL]
    float* s_gh = s_hl + NB * DL;                        // [NB][DL]
    for (int i = threadIdx.x; i < C * DL; i += blockDim.x) s_wo[i] = A.Wo[i];
    float wacc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) wacc[p] = 0.f;
    const float inv_heads = 1.0f / (float)H;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    const int hd = H * DL;
    const int dqC = 256 / C, drC = 256 % C, dqD = 256 / DL, drD = 256 % DL, dqH = 256 / hd, drH = 256 % hd;
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int rows_here = (int)((A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB);
        {
            const float* yt = A.y + n0 * C;
            int r = threadIdx.x / C, c = threadIdx.x % C;
            for (int q = threadIdx.x; q < rows_here * C; q += 256) {
                const int lab_r = A.labels[n0 + r];                    // < 0: outside the training mask => dz = 0
                s_dz[r * ldz + c] = lab_r >= 0 ? yt[q] - (c == lab_r ? 1.0f : 0.0f) : 0.0f;
                r += dqC; c += drC;
                if (c >= C) { c -= C; ++r; }
            }
        }
        for (int q = threadIdx.x; q < rows_here * DL; q += 256) s_hl[q] = A.HL[n0 * DL + q];
        __syncthreads();
        {
            int r = threadIdx.x / DL, d = threadIdx.x % DL;
            for (int q = threadIdx.x; q < rows_here * DL; q += 256) {
                float sum = 0.f;
                for (int c = 0; c < C; ++c) sum += s_wo[c * DL + d] * s_dz[r * ldz + c];
                s_gh[q] = sum;
                r += dqD; d += drD;
                if (d >= DL) { d -= DL; ++r; }
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int t = threadIdx.x + p * 256;
            if (p < per_thread && t < C * DL) {
                const int c = t / DL, d = t % DL;
                float acc = wacc[p];
                for (int r = 0; r < rows_here; ++r) acc += s_dz[r * ldz + c] * s_hl[r * DL + d];
                wacc[p] = acc;
            }
        }
        __syncthreads();
        if (A.gh_out != nullptr)
            for (int q = threadIdx.x; q < rows_here * DL; q += 256) A.gh_out[(n0 + q / DL) * (A.gh_stride ? A.gh_stride : DL) + q % DL] = s_gh[q];
        if (A.g != nullptr) {
            const float* hp = A.hpre + n0 * hd;
            float* gt = A.g + n0 * hd;
            int r = threadIdx.x / hd, x = threadIdx.x % hd;     // x = h*DL + d
            for (int q = threadIdx.x; q < rows_here * hd; q += 256) {
                const int d = x % DL;
                const float hv = A.flat_index ? A.hpre[(n0 + r) * DL + d] : hp[q];
                gt[q] = s_gh[r * DL + d] * (hv > 0.f ? 1.0f : A.slope) * inv_heads;
                r += dqH; x += drH;
                if (x >= hd) { x -= hd; ++r; }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int t = threadIdx.x + p * 256;
        if (p < per_thread && t < C * DL) A.partial[(int64_t)blockIdx.x * C * DL + t] = wacc[p];
    }
}

// The same head backward for the training path (only gH is written: HeadBwdArgs::g == nullptr), DL <= 16,
// C <= 64, register-blocked over the DL outputs so that an FMA costs 3/8 LDS reads instead of 2:
//   waves 0-1: one thread per node of the 128-node tile:  gH[r][:] = sum_c dz[r][c] Wo[c][:]   -> global
//   waves 2-3: thread (c, half):  gradWo[c][:] += sum over the half's 64 nodes of dz[r][c] H_L[r][:]
// (the two halves run concurrently on different waves; per-thread partials live across all tiles of the block)
template <int DL>
__global__ __launch_bounds__(256) void head_backward2_kernel(HeadBwdArgs A, int32_t ldz) {
    constexpr int NB = 128;
    extern __shared__ float lds[];
    const int C = A.C;
    float* s_wo = lds;                                   // [C][DL]
    float* s_dz = s_wo + C * DL;                         // [NB][ldz]
    float* s_hl = s_dz + NB * ldz;                       // [NB][DL]
    for (int i = threadIdx.x; i < C * DL; i += 256) s_wo[i] = A.Wo[i];
    const int tid = threadIdx.x;
    const int t2 = tid - 128;                            // gradWo thread: class t2 % C ... only t2 < 2*C work
    const int wc = t2 >= 0 ? t2 % C : 0, whalf = t2 >= 0 ? t2 / C : 2;
    float wacc[DL];
#pragma unroll
    for (int d = 0; d < DL; ++d) wacc[d] = 0.f;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    const int dqC = 256 / C, drC = 256 % C;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int rows_here = (int)((A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB);
        __syncthreads();                                 // previous tile fully consumed
        {
            const float* yt = A.y + n0 * C;
            int r = tid / C, c = tid % C;
            for (int q = tid; q < rows_here * C; q += 256) {
                const int lab_r = A.labels[n0 + r];                    // < 0: outside the training mask => dz = 0
                s_dz[r * ldz + c] = lab_r >= 0 ? yt[q] - (c == lab_r ? 1.0f : 0.0f) : 0.0f;
                r += dqC; c += drC;
                if (c >= C) { c -= C; ++r; }
            }
        }
        for (int q = tid; q < rows_here * DL; q += 256) s_hl[q] = A.HL[n0 * DL + q];
        __syncthreads();
        if (tid < 128) {
            if (tid < rows_here) {
                float acc[DL];
#pragma unroll
                for (int d = 0; d < DL; ++d) acc[d] = 0.f;
                const float* zr = s_dz + tid * ldz;
                for (int c = 0; c < C; ++c) {            // ascending c: the order of the reference's loop (E:585-590)
                    const float z = zr[c];
#pragma unroll
                    for (int d = 0; d < DL; ++d) acc[d] += s_wo[c * DL + d] * z;
                }
                float* out = A.gh_out + (n0 + tid) * (A.gh_stride ? A.gh_stride : DL);
#pragma unroll
                for (int d = 0; d < DL; ++d) out[d] = acc[d];
            }
        } else if (whalf < 2) {
            const int r0 = whalf * 64, r1 = (r0 + 64 < rows_here) ? r0 + 64 : rows_here;
            for (int r = r0; r < r1; ++r) {
                const float z = s_dz[r * ldz + wc];
#pragma unroll
                for (int d = 0; d < DL; ++d) wacc[d] += z * s_hl[r * DL + d];
            }
        }
    }
    __syncthreads();
    float* s_tmp = s_dz;                                 // [2][C][DL]
    if (whalf < 2) {
#pragma unroll
        for (int d = 0; d < DL; ++d) s_tmp[(whalf * C + wc) * DL + d] = wacc[d];
    }
    __syncthreads();
    for (int i = tid; i < C * DL; i += 256) A.partial[(int64_t)blockIdx.x * C * DL + i] = s_tmp[i] + s_tmp[C * DL + i];
}

// Output head forward AND backward in one pass over the nodes (gat_step): logits, softmax, loss / #correct,
// dz = y - onehot, gH = Wo^T dz, gradWo += dz^T H_L — the class probabilities never reach HBM (0.46 GB written
// by head_forward_kernel and read back by head_backward2_kernel otherwise).  Same expressions in the same order
// as those two kernels.  128-node tiles: waves 0-1 own a node each for the softmax and gH, waves 2-3 own
// (class, half-tile) for gradWo.
template <int DL>
__global__ __launch_bounds__(256) void head_step_kernel(HeadArgs F, HeadBwdArgs A, int32_t ldz) {
    constexpr int NB = 128;
    extern __shared__ float lds[];
    const int C = A.C;
    float* s_wo = lds;                                   // [C][DL]
    float* s_dz = s_wo + C * DL;                         // [NB][ldz]
    float* s_hl = s_dz + NB * ldz;                       // [NB][DL]
    __shared__ double s_loss[2];
    __shared__ int32_t s_corr[2];
    for (int i = threadIdx.x; i < C * DL; i += 256) s_wo[i] = A.Wo[i];
    const int tid = threadIdx.x;
    const int t2 = tid - 128;
    const int wc = t2 >= 0 ? t2 % C : 0, whalf = t2 >= 0 ? t2 / C : 2;
    float wacc[DL];
#pragma unroll
    for (int d = 0; d < DL; ++d) wacc[d] = 0.f;
    double loss_acc = 0.0;
    int corr_acc = 0;
    const int64_t ntiles = (A.n_rows + NB - 1) / NB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n0 = tile * NB;
        const int rows_here = (int)((A.n_rows - n0 < NB) ? (A.n_rows - n0) : NB);
        __syncthreads();                                 // previous tile fully consumed (and s_wo loaded)
        if (tid < rows_here) {
            const int64_t n = n0 + tid;
            float x[DL];
#pragma unroll
            for (int j = 0; j < DL; ++j) { x[j] = A.HL[n * DL + j]; s_hl[tid * DL + j] = x[j]; }
            float* zr = s_dz + tid * ldz;
            float mv = -INFINITY;
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < DL; ++j) acc += s_wo[c * DL + j] * x[j];       // ascending j (E:496-498)
                zr[c] = acc;
                mv = fmaxf(mv, acc);
            }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) { const float ev = __builtin_amdgcn_exp2f((zr[c] - mv) * 1.4426950408889634f); zr[c] = ev; sum += ev; }
            const double rden = 1.0 / ((double)sum + 1e-8);
            const int lab = A.labels[n];
            float best = -1.f; int pred = 0; float plab = 0.f;
            for (int c = 0; c < C; ++c) {
                const float yv = (float)((double)zr[c] * rden);
                if (c == 0 || yv > best) { best = yv; pred = c; }      // strict >, first max wins
                if (c == lab) plab = yv;
                zr[c] = lab >= 0 ? yv - (c == lab ? 1.0f : 0.0f) : 0.0f;   // dz (0 outside the training mask)
            }
            // lab < 0: the node is outside the training mask (gat_set_train_mask stores ~label): no loss, no count
            loss_acc += lab >= 0 ? (double)(-logf(fmaxf(plab, 1e-12f))) : 0.0;
            corr_acc += (lab >= 0 && pred == lab) ? 1 : 0;
        }
        __syncthreads();
        if (tid < 128) {
            if (tid < rows_here) {
                float acc[DL];
#pragma unroll
                for (int d = 0; d < DL; ++d) acc[d] = 0.f;
                const float* zr = s_dz + tid * ldz;
                for (int c = 0; c < C; ++c) {
                    const float z = zr[c];
#pragma unroll
                    for (int d = 0; d < DL; ++d) acc[d] += s_wo[c * DL + d] * z;
                }
                if (A.gh_out != nullptr) {                 // null: loss, #correct and grad_Wo only (fused last layer: gH formed there)
                    float* out = A.gh_out + (n0 + tid) * (A.gh_stride ? A.gh_stride : DL);
#pragma unroll
                    for (int d = 0; d < DL; ++d) out[d] = acc[d];
                }
            }
        } else if (whalf < 2) {
            const int r0 = whalf * 64, r1 = (r0 + 64 < rows_here) ? r0 + 64 : rows_here;
            for (int r = r0; r < r1; ++r) {
                const float z = s_dz[r * ldz + wc];
#pragma unroll
                for (int d = 0; d < DL; ++d) wacc[d] += z * s_hl[r * DL + d];
            }
        }
    }
    __syncthreads();
    float* s_tmp = s_dz;                                 // [2][C][DL]
    if (whalf < 2) {
#pragma unroll
        for (int d = 0; d < DL; ++d) s_tmp[(whalf * C + wc) * DL + d] = wacc[d];
    }
    for (int off = 32; off > 0; off >>= 1) {             // loss / #correct live in waves 0-1
        loss_acc += __shfl_down(loss_acc, off);
        corr_acc += __shfl_down(corr_acc, off);
    }
    if (tid < 128 && (tid & 63) == 0) { s_loss[tid >> 6] = loss_acc; s_corr[tid >> 6] = corr_acc; }
    __syncthreads();
    for (int i = tid; i < C * DL; i += 256) A.partial[(int64_t)blockIdx.x * C * DL + i] = s_tmp[i] + s_tmp[C * DL + i];
    if (tid == 0) {
        F.loss_partial[blockIdx.x] = s_loss[0] + s_loss[1];
        F.correct_partial[blockIdx.x] = s_corr[0] + s_corr[1];
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
static thread_local ReduceBatch t_batch;
static thread_local bool t_batch_on = false;
void reduce_batch_begin() { t_batch = ReduceBatch{}; t_batch_on = true; }
void reduce_batch_abort() { t_batch_on = false; }
void reduce_batch_pack(const float* loss, const int32_t* correct, float* dst3, float* dst3_host) {
    t_batch.loss = loss; t_batch.correct = correct; t_batch.dst3 = dst3; t_batch.dst3_host = dst3_host;
}
bool reduce_batch_finalize(const double* lp, const int32_t* cp, int32_t nb, float* loss_out, int32_t* correct_out) {
    if (!t_batch_on || t_batch.fin_loss != nullptr) return false;
    t_batch.fin_loss = lp; t_batch.fin_corr = cp; t_batch.fin_n = nb; t_batch.fin_loss_out = loss_out; t_batch.fin_corr_out = correct_out;
    return true;
}
static bool reduce_batch_push(const float* slabs, int32_t nslabs, int64_t width, float* out, int32_t HD, int32_t F, int32_t c_base) {
    if (!t_batch_on || t_batch.n >= 8 || width <= 0) return false;
    ReduceJob& j = t_batch.jobs[t_batch.n++];
    j = ReduceJob{slabs, out, width, nslabs, HD, F, c_base, t_batch.blocks};
    t_batch.blocks += (int32_t)((width + 63) / 64);
    return true;
}
int reduce_batch_flush(hipStream_t s) {
    if (!t_batch_on) return 0;
    t_batch_on = false;
    const int extra = (t_batch.dst3 != nullptr || t_batch.fin_loss != nullptr) ? 1 : 0;
    if (t_batch.blocks + extra == 0) return 0;
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)(t_batch.blocks + extra)), dim3(1024), 0, s, t_batch);
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_reduce_gradw(const float* slabs, int32_t ksplit, int32_t HD, int32_t F, int32_t c_base, int32_t M,
                        float* gradW, hipStream_t s) {
    const int64_t width = (int64_t)M * F;
    if (reduce_batch_push(slabs, ksplit, width, gradW, HD, F, c_base)) return 0;
    hipLaunchKernelGGL((reduce_slabs_kernel<MapGradW>), dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, s, slabs,
                       ksplit, width, gradW, MapGradW{HD, F, c_base});
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_reduce_partials_add(const float* partial, int32_t nblocks, int64_t width, float* out, hipStream_t s) {
    if (width <= 0) return 0;
    if (reduce_batch_push(partial, nblocks, width, out, 0, 0, 0)) return 0;
    hipLaunchKernelGGL((reduce_slabs_kernel<MapIdentity>), dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, s,
                       partial, nblocks, width, out, MapIdentity{});
    GAT_HIP(hipGetLastError());
    return 0;
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
    // nodes per block: 128 = one tile of the fused head kernel per block (GAT_HEAD_NODES=256: two tiles in sequence per block, as before)
    static const int per = [] { const char* e = choice_env("GAT_HEAD_NODES"); const int v = e ? atoi(e) : 0; return v == 256 ? 256 : 128; }();
    const int64_t b = (n_rows + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
int launch_head_forward(const HeadArgs& a, hipStream_t s) {
    int32_t NB, ldz;
    const int dlp = a.DL <= 8 ? 8 : (a.DL <= 16 ? 16 : (a.DL <= 32 ? 32 : (a.DL <= 64 ? 64 : 0)));
    GAT_TRY(head_tile(a.C, dlp ? dlp : a.DL, 0, &NB, &ldz));
    int blocks = head_blocks(a.n_rows);
    const size_t lds = ((size_t)a.C * (dlp ? dlp : a.DL) + (size_t)NB * ldz) * sizeof(float);
    {   // persistent over node tiles: launch only what is resident at once
        const void* fn = dlp == 8 ? (const void*)head_forward_kernel<8> : dlp == 16 ? (const void*)head_forward_kernel<16>
                       : dlp == 32 ? (const void*)head_forward_kernel<32> : dlp == 64 ? (const void*)head_forward_kernel<64>
                                                                                      : (const void*)head_forward_kernel<0>;
        const int64_t res = resident_blocks(fn, lds);
        if (res < blocks) blocks = (int)res;
    }
    switch (dlp) {
        case 8: hipLaunchKernelGGL(head_forward_kernel<8>, dim3(blocks), dim3(256), lds, s, a, NB, ldz); break;
        case 16: hipLaunchKernelGGL(head_forward_kernel<16>, dim3(blocks), dim3(256), lds, s, a, NB, ldz); break;
        case 32: hipLaunchKernelGGL(head_forward_kernel<32>, dim3(blocks), dim3(256), lds, s, a, NB, ldz); break;
        case 64: hipLaunchKernelGGL(head_forward_kernel<64>, dim3(blocks), dim3(256), lds, s, a, NB, ldz); break;
        default: hipLaunchKernelGGL(head_forward_kernel<0>, dim3(blocks), dim3(256), lds, s, a, NB, ldz); break;
    }
    GAT_HIP(hipGetLastError());
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(64), 0, s, a.loss_partial, a.correct_partial, blocks,
                       a.loss_out, a.correct_out);
    GAT_HIP(hipGetLastError());
    return 0;
}
bool head_step_supported(const HeadBwdArgs& b) {
    return b.g == nullptr && b.gh_out != nullptr && b.C <= 64 && (b.DL == 4 || b.DL == 8 || b.DL == 16);
}
int launch_head_step(const HeadArgs& f, const HeadBwdArgs& b, hipStream_t s) {
    HeadBwdArgs probe = b;
    if (probe.gh_out == nullptr) probe.gh_out = reinterpret_cast<float*>(1);       // the gH output is optional here
    if (!head_step_supported(probe)) return fail(GAT_E_UNSUPPORTED, "head_step: shape outside the fused kernel");
    const int ldz = (b.C % 2 == 0) ? b.C + 1 : b.C;
    const size_t lds = ((size_t)b.C * b.DL + (size_t)128 * (ldz + b.DL)) * sizeof(float);
    int blocks = std::min(head_blocks(b.n_rows), head_bwd_blocks(b.n_rows, b.C, b.DL));      // both partial buffers
    const void* fn = b.DL == 4 ? (const void*)head_step_kernel<4> : b.DL == 8 ? (const void*)head_step_kernel<8>
                                                                                : (const void*)head_step_kernel<16>;
    const int64_t res = resident_blocks(fn, lds);
    if (res < blocks) blocks = (int)res;
    if (blocks < 1) blocks = 1;
    if (b.DL == 4) hipLaunchKernelGGL(head_step_kernel<4>, dim3(blocks), dim3(256), lds, s, f, b, ldz);
    else if (b.DL == 8) hipLaunchKernelGGL(head_step_kernel<8>, dim3(blocks), dim3(256), lds, s, f, b, ldz);
    else hipLaunchKernelGGL(head_step_kernel<16>, dim3(blocks), dim3(256), lds, s, f, b, ldz);
    GAT_HIP(hipGetLastError());
    if (!reduce_batch_finalize(f.loss_partial, f.correct_partial, blocks, f.loss_out, f.correct_out)) {
        hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(64), 0, s, f.loss_partial, f.correct_partial, blocks,
                           f.loss_out, f.correct_out);
        GAT_HIP(hipGetLastError());
    }
    return launch_reduce_partials_add(b.partial, blocks, (int64_t)b.C * b.DL, b.gradWo, s);
}
int launch_pack_result(const float* loss, const int32_t* correct, float* dst3, hipStream_t s) {
    hipLaunchKernelGGL(pack_result_kernel, dim3(1), dim3(64), 0, s, loss, correct, dst3);
    GAT_HIP(hipGetLastError());
    return 0;
}
int head_bwd_blocks(int64_t n_rows, int32_t C, int32_t DL) {
    (void)C; (void)DL;
    const int64_t b = (n_rows + 127) / 128;          // head_backward2_kernel: 128-node tiles
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
int launch_head_backward(const HeadBwdArgs& a, hipStream_t s) {
    if (a.g == nullptr && a.gh_out != nullptr && a.C <= 64 && (a.DL == 4 || a.DL == 8 || a.DL == 16)) {
        const int ldz = (a.C % 2 == 0) ? a.C + 1 : a.C;
        const size_t lds = ((size_t)a.C * a.DL + (size_t)128 * (ldz + a.DL)) * sizeof(float);
        int blocks = (int)std::min<int64_t>((a.n_rows + 127) / 128, 1024);
        const void* fn = a.DL == 4 ? (const void*)head_backward2_kernel<4> : a.DL == 8 ? (const void*)head_backward2_kernel<8>
                                                                                        : (const void*)head_backward2_kernel<16>;
        const int64_t res = resident_blocks(fn, lds);
        if (res < blocks) blocks = (int)res;
        if (blocks < 1) blocks = 1;
        if (a.DL == 4) hipLaunchKernelGGL(head_backward2_kernel<4>, dim3(blocks), dim3(256), lds, s, a, ldz);
        else if (a.DL == 8) hipLaunchKernelGGL(head_backward2_kernel<8>, dim3(blocks), dim3(256), lds, s, a, ldz);
        else hipLaunchKernelGGL(head_backward2_kernel<16>, dim3(blocks), dim3(256), lds, s, a, ldz);
        GAT_HIP(hipGetLastError());
        return launch_reduce_partials_add(a.partial, blocks, (int64_t)a.C * a.DL, a.gradWo, s);
    }
    int32_t NB, ldz;
    GAT_TRY(head_tile(a.C, a.DL, 2 * a.DL, &NB, &ldz));
    const int per_thread = (a.C * a.DL + 255) / 256;
    if (per_thread > 8) return fail(GAT_E_UNSUPPORTED, "output head: num_classes*D_last > 2048");
    int blocks = head_bwd_blocks(a.n_rows, a.C, a.DL);
    const size_t lds = ((size_t)a.C * a.DL + (size_t)NB * (ldz + 2 * a.DL)) * sizeof(float);
    {
        const int64_t res = resident_blocks((const void*)head_backward_kernel, lds);
        if (res < blocks) blocks = (int)res;
    }
    hipLaunchKernelGGL(head_backward_kernel, dim3(blocks), dim3(256), lds, s, a, NB, ldz, per_thread);
    GAT_HIP(hipGetLastError());
    return launch_reduce_partials_add(a.partial, blocks, (int64_t)a.C * a.DL, a.gradWo, s);
}

// labels_eff[i] = mask[i] ? labels[i] : ~labels[i]   (mask == nullptr: copy)
__global__ __launch_bounds__(256) void apply_mask_kernel(const int32_t* __restrict__ labels, const uint8_t* __restrict__ mask,
                                                         int32_t* __restrict__ eff, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        eff[i] = (mask == nullptr || mask[i]) ? labels[i] : ~labels[i];
}
int launch_apply_mask(const int32_t* labels, const uint8_t* mask, int32_t* eff, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, labels, mask, eff, n);
    GAT_HIP(hipGetLastError());
    return 0;
}
// Loss / accuracy of the last forward over the nodes of `mask` (validation / test split): per block a fixed-order
// partial {sum of -log(max(y[label], 1e-12)), #correct, #nodes}; the host adds the block partials in order.
__global__ __launch_bounds__(256) void eval_mask_kernel(const float* __restrict__ y, const int32_t* __restrict__ labels,
                                                        const uint8_t* __restrict__ mask, int64_t n, int32_t C,
                                                        double* __restrict__ part_loss, int32_t* __restrict__ part_cnt) {
    __shared__ double s_l[256];
    __shared__ int32_t s_c[256], s_n[256];
    double l = 0.0; int32_t cr = 0, nn = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (!mask[i]) continue;
        const float* yr = y + i * C;
        const int lab = labels[i];
        float best = -1.f; int pred = 0;
        for (int c = 0; c < C; ++c) if (c == 0 || yr[c] > best) { best = yr[c]; pred = c; }      // strict >, first max wins (E:530-535)
        l += (double)(-logf(fmaxf(yr[lab], 1e-12f)));
        cr += pred == lab; ++nn;
    }
    s_l[threadIdx.x] = l; s_c[threadIdx.x] = cr; s_n[threadIdx.x] = nn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { s_l[threadIdx.x] += s_l[threadIdx.x + off]; s_c[threadIdx.x] += s_c[threadIdx.x + off]; s_n[threadIdx.x] += s_n[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_loss[blockIdx.x] = s_l[0]; part_cnt[2 * blockIdx.x] = s_c[0]; part_cnt[2 * blockIdx.x + 1] = s_n[0]; }
}
int launch_eval_mask(const float* y, const int32_t* labels, const uint8_t* mask, int64_t n, int32_t C, double* part_loss,
                     int32_t* part_cnt, int32_t blocks, hipStream_t s) {
    hipLaunchKernelGGL(eval_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, labels, mask, n, C, part_loss, part_cnt);
    GAT_HIP(hipGetLastError());
    return 0;
}

// Xavier-uniform init on the device (E:186-248): element i of a segment = the (draw0 + i)-th draw of the
// counter-based stream  z_k = splitmix64_finalise(s0 + (k+1)*0x9E3779B97F4A7C15)  mapped to (0, 1] like
// curand_uniform (E:217) and then to (-lim, lim].
__global__ __launch_bounds__(256) void xavier_init_kernel(float* __restrict__ p, int64_t n, uint64_t s0, uint64_t draw0, float lim) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = s0 + (draw0 + (uint64_t)i + 1ull) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = ((float)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);
        p[i] = u * 2.0f * lim - lim;
    }
}
int launch_xavier_init(float* p, int64_t n, uint64_t s0, uint64_t draw0, float lim, hipStream_t s) {
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(xavier_init_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, s0, draw0, lim);
    GAT_HIP(hipGetLastError());
    return 0;
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
