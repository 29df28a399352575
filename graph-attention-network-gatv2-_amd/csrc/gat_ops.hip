// gat_ops.hip — op-level entry points, ONE PER REFERENCE KERNEL ("one per row a1-a11 (+ head/loss), taking
// device pointers in the reference layouts").  A maintainer can replace a single launch inside the reference's own main()
// — compute_max_sum_attn_score at E:1398, say — and leave the rest of that program as it is: every function takes exactly
// the argument list of the launch it replaces (sizes widened to 64 bit where they count nodes or edges) plus a stream,
// reads and writes caller-owned DEVICE memory in the reference layouts ([H][E] head-major edge tensors, [N][H][D] node
// tensors, W [H][D][2F]) and keeps the reference's accumulate-or-overwrite behaviour.
//
// Same restructuring as the fused path, one stage at a time: wherever the reference recomputes W·x per
// (head, edge) thread, the op projects PL = X·W_left^T / PR = X·W_right^T once per node on the matrix cores
// (launch_project) and the per-edge kernel reads rows of them.  Scratch (PL, PR, gPL, gPR) is allocated per call and freed
// before returning; calls synchronise their stream.  These are the unit-parity seams — the training path is the fused one
// behind gat_step (no [H][E] tensor is materialised there).  Scatters here use float atomics like the reference's own
// kernels (E:422, 783-786, 868-869), so sums over edges are order-dependent at fp32 round-off, as in the reference.
#include "gat_internal.h"

#include <algorithm>

namespace gat {
namespace {

struct Scratch {
    std::vector<void*> p;
    ~Scratch() { for (void* q : p) (void)hipFree(q); }
    int get(float** out, int64_t n) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, (size_t)std::max<int64_t>(n, 1) * sizeof(float));
        if (e != hipSuccess) return fail(GAT_E_NOMEM, std::string("op scratch: ") + hipGetErrorString(e));
        p.push_back(q); *out = (float*)q;
        return 0;
    }
};

__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }

constexpr int kBlock = 256;
static unsigned grid_for(int64_t work) { return (unsigned)std::min<int64_t>((work + kBlock - 1) / kBlock, (int64_t)1 << 20); }

// a2 (E:303-323): score[h][e] = sum_k a[h][k] * LReLU(PL[src][h][k] + PR[dst][h][k]); one thread per (e, h), h fastest:
// the H threads of an edge read its two rows contiguously.
__global__ __launch_bounds__(kBlock) void op_edge_score_kernel(const float* __restrict__ PL, const float* __restrict__ PR,
                                                              const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                              const float* __restrict__ a, float* __restrict__ score,
                                                              int64_t E, int32_t H, int32_t D, float slope) {
    const int64_t total = E * H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t e = t / H;
        const int h = (int)(t % H);
        const float* pl = PL + ((int64_t)src[e] * H + h) * D;
        const float* pr = PR + ((int64_t)dst[e] * H + h) * D;
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc += a[h * D + k] * lrelu_f(pl[k] + pr[k], slope);
        score[(int64_t)h * E + e] = acc;
    }
}

// a3 (E:326-359): one wave per destination row, heads one after the other, lanes striding the row's edges; max seeded with
// -1e9f (E:336), sum of __expf(score - max) (E:349).  max[N*h + dst], sum[N*h + dst].
__global__ __launch_bounds__(kBlock) void op_max_sum_kernel(const int32_t* __restrict__ row_ptr, const float* __restrict__ score,
                                                           int64_t N, int32_t H, int64_t E, float* __restrict__ mx,
                                                           float* __restrict__ sm) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < N; row += waves) {
        const int b = row_ptr[row], e = row_ptr[row + 1];
        for (int h = 0; h < H; ++h) {
            const float* sc = score + (int64_t)h * E;
            float m = -1e9f;
            for (int i = b + lane; i < e; i += 64) m = fmaxf(m, sc[i]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            float z = 0.f;
            for (int i = b + lane; i < e; i += 64) z += __expf(sc[i] - m);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off);
            if (lane == 0) { mx[N * h + row] = m; sm[N * h + row] = z; }
        }
    }
}

// a4 (E:362-384): alpha = __expf(score - max[dst + N*h]) / (sum[dst + N*h] + 1e-8f), thread per tid = h*E + e
__global__ __launch_bounds__(kBlock) void op_attn_coeff_kernel(const int32_t* __restrict__ dst, const float* __restrict__ score,
                                                              const float* __restrict__ mx, const float* __restrict__ sm,
                                                              float* __restrict__ alpha, int64_t E, int32_t H, int64_t N) {
    const int64_t total = E * H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t h = t / E, e = t % E;
        const int64_t off = dst[e] + N * h;
        alpha[t] = __expf(score[t] - mx[off]) / (sm[off] + 1e-8f);
    }
}

// a5 (E:386-424): out[dst][h][k] += alpha[h][e] * PL[src][h][k]  (atomicAdd like E:422: the caller zeroes);
// one thread per (e, channel), channel fastest: coalesced row reads and 4*HD-byte atomic bursts
__global__ __launch_bounds__(kBlock) void op_aggregate_kernel(const float* __restrict__ PL, const int32_t* __restrict__ src,
                                                             const int32_t* __restrict__ dst, const float* __restrict__ alpha,
                                                             float* __restrict__ out, int64_t E, int32_t H, int32_t D) {
    const int HD = H * D;
    const int64_t total = E * HD, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t e = t / HD;
        const int c = (int)(t % HD);
        atomicAdd(out + (int64_t)dst[e] * HD + c, alpha[(int64_t)(c / D) * E + e] * PL[(int64_t)src[e] * HD + c]);
    }
}

// a6 (E:426-459): LeakyReLU, then concat (hidden) or mean over heads (last); thread per output element
__global__ __launch_bounds__(kBlock) void op_post_activation_kernel(const float* __restrict__ hpre, float* __restrict__ out,
                                                                   int64_t N, int32_t H, int32_t D, int32_t is_last, float slope) {
    const int64_t total = N * (is_last ? D : H * D), stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        if (!is_last) { out[t] = lrelu_f(hpre[t], slope); continue; }
        const int64_t n = t / D;
        const int d = (int)(t % D);
        float s = 0.f;
        for (int h = 0; h < H; ++h) s += lrelu_f(hpre[(n * H + h) * D + d], slope);
        out[t] = s / (float)H;
    }
}

// C12 (E:463-512 with softmax E:132-141): z = W_o x, softmax with the DOUBLE-literal epsilon of E:140; both d_z and d_y end
// up holding the probabilities, as in the reference (softmax runs in place on node_z, then is copied).  Thread per node,
// same loop order as the reference (classes outer, features inner).
__global__ __launch_bounds__(kBlock) void op_output_head_kernel(const float* __restrict__ Wo, const float* __restrict__ HL,
                                                               float* __restrict__ z, float* __restrict__ y, int64_t N, int32_t C,
                                                               int32_t DL) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const float* x = HL + n * DL;
        float* zn = z + n * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
            for (int j = 0; j < DL; ++j) acc += Wo[c * DL + j] * x[j];
            zn[c] = acc;
            m = fmaxf(m, acc);
        }
        float s = 0.f;
        for (int c = 0; c < C; ++c) { const float v = expf(zn[c] - m); zn[c] = v; s += v; }
        for (int c = 0; c < C; ++c) {
            const float p = (float)((double)zn[c] / ((double)s + 1e-8));
            zn[c] = p;
            y[n * C + c] = p;
        }
    }
}

// C13 (E:514-537): per-node cross-entropy with the 1e-12f clamp and the strict-> arg-max
__global__ __launch_bounds__(kBlock) void op_loss_accuracy_kernel(const float* __restrict__ y, const int32_t* __restrict__ labels,
                                                                 float* __restrict__ loss, int32_t* __restrict__ correct, int64_t N,
                                                                 int32_t C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const int label = labels[n];
        loss[n] = -logf(fmaxf(y[n * C + label], 1e-12f));
        float mv = y[n * C];
        int pred = 0;
        for (int c = 1; c < C; ++c) { const float v = y[n * C + c]; if (v > mv) { mv = v; pred = c; } }
        correct[n] = pred == label;
    }
}

// C14 (E:553-608): dz = y - onehot; grad_Wo[c][d] += sum_n dz[c] H_L[n][d] (block partials in LDS, one atomicAdd per block
// and element); grad_hL[n][h][d] = (sum_c Wo[c][d] dz[c]) * LReLU'(.) / H with the exact per-head pre-activation, or the
// reference's flat index n*D + d (E:598) when flat_index is set
__global__ __launch_bounds__(kBlock) void op_output_gradients_kernel(const float* __restrict__ y, const int32_t* __restrict__ labels,
                                                                    const float* __restrict__ hL, const float* __restrict__ HL,
                                                                    const float* __restrict__ Wo, float* __restrict__ gradWo,
                                                                    float* __restrict__ gradhL, int64_t N, int32_t C, int32_t DL,
                                                                    int32_t H, float slope, int32_t flat_index) {
    extern __shared__ float sh[];                    // [C*DL] partial of grad_Wo
    for (int i = threadIdx.x; i < C * DL; i += blockDim.x) sh[i] = 0.f;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += stride) {
        const int label = labels[n];
        for (int d = 0; d < DL; ++d) {
            float s = 0.f;
            const float hv = HL[n * DL + d];
            for (int c = 0; c < C; ++c) {
                const float dz = y[n * C + c] - (c == label ? 1.0f : 0.0f);
                s += Wo[c * DL + d] * dz;
                atomicAdd(&sh[c * DL + d], dz * hv);
            }
            const float inv_heads = 1.0f / (float)H;
            for (int h = 0; h < H; ++h) {
                const float pre = flat_index ? hL[n * DL + d] : hL[(n * H + h) * DL + d];
                gradhL[(n * H + h) * DL + d] = s * (pre > 0.f ? 1.0f : slope) * inv_heads;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * DL; i += blockDim.x) atomicAdd(&gradWo[i], sh[i]);
}

// a7 (E:612-651): galpha[h][e] = sum_k g[dst][h][k] * PL[src][h][k]; thread per (e, h), h fastest
__global__ __launch_bounds__(kBlock) void op_grad_attn_coeff_kernel(const float* __restrict__ PL, const float* __restrict__ g,
                                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                                   float* __restrict__ galpha, int64_t E, int32_t H, int32_t D) {
    const int64_t total = E * H, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t e = t / H;
        const int h = (int)(t % H);
        const float* pl = PL + ((int64_t)src[e] * H + h) * D;
        const float* gj = g + ((int64_t)dst[e] * H + h) * D;
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc += gj[k] * pl[k];
        galpha[(int64_t)h * E + e] = acc;
    }
}

// a8 (E:654-696): ge_ij = sum_k galpha_kj alpha_kj (delta_ik - alpha_ij) = alpha_ij (galpha_ij - sum_k galpha_kj alpha_kj):
// one pass per (row, head) for the row's dot product, one to write — O(E) instead of the reference's O(sum deg^2).
// One wave per destination row.
__global__ __launch_bounds__(kBlock) void op_grad_attn_score_kernel(const int32_t* __restrict__ row_ptr, const float* __restrict__ alpha,
                                                                   const float* __restrict__ galpha, float* __restrict__ ge,
                                                                   int64_t N, int32_t H, int64_t E) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < N; row += waves) {
        const int b = row_ptr[row], e = row_ptr[row + 1];
        for (int h = 0; h < H; ++h) {
            const float* al = alpha + (int64_t)h * E;
            const float* ga = galpha + (int64_t)h * E;
            float dot = 0.f;
            for (int i = b + lane; i < e; i += 64) dot += ga[i] * al[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
            for (int i = b + lane; i < e; i += 64) ge[(int64_t)h * E + i] = al[i] * (ga[i] - dot);
        }
    }
}

// a9 / a10 edge part (E:746-786, 840-869): per (e, channel c = h*D + k), s = PL[src][c] + PR[dst][c],
//   gs = ge[h][e] * a[c] * LReLU'(s);  gPL[src][c] += g[dst][c] * alpha[h][e] + gs;  gPR[dst][c] += gs   (float atomics),
//   and (WITH_GA) grad_a[c] += ge * LReLU(s) through a per-block LDS partial
template <bool WITH_GA>
__global__ __launch_bounds__(kBlock) void op_messages_kernel(const float* __restrict__ PL, const float* __restrict__ PR,
                                                            const float* __restrict__ g, const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ dst, const float* __restrict__ alpha,
                                                            const float* __restrict__ ge, const float* __restrict__ a,
                                                            float* __restrict__ gPL, float* __restrict__ gPR, float* __restrict__ grad_a,
                                                            int64_t E, int32_t H, int32_t D, float slope) {
    extern __shared__ float sh_ga[];                 // [HD]
    const int HD = H * D;
    if constexpr (WITH_GA) {
        for (int i = threadIdx.x; i < HD; i += blockDim.x) sh_ga[i] = 0.f;
        __syncthreads();
    }
    const int64_t total = E * HD, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t e = t / HD;
        const int c = (int)(t % HD);
        const int64_t s_ = src[e], d_ = dst[e];
        const int64_t he = (int64_t)(c / D) * E + e;
        const float s = PL[s_ * HD + c] + PR[d_ * HD + c];
        const float gev = ge[he];
        const float gs = gev * a[c] * (s > 0.f ? 1.0f : slope);
        atomicAdd(gPL + s_ * HD + c, g[d_ * HD + c] * alpha[he] + gs);
        atomicAdd(gPR + d_ * HD + c, gs);
        if constexpr (WITH_GA) atomicAdd(&sh_ga[c], gev * lrelu_f(s, slope));
    }
    if constexpr (WITH_GA) {
        __syncthreads();
        for (int i = threadIdx.x; i < HD; i += blockDim.x) atomicAdd(&grad_a[i], sh_ga[i]);
    }
}

__global__ __launch_bounds__(kBlock) void op_add_kernel(float* __restrict__ out, const float* __restrict__ add, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] += add[i];
}

// a11 (E:879-893): gx *= LReLU'(h_pre_prev), in place
__global__ __launch_bounds__(kBlock) void op_preact_gradient_kernel(const float* __restrict__ hpre, float* __restrict__ gx, int64_t n,
                                                                   float slope) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) gx[i] *= hpre[i] > 0.f ? 1.0f : slope;
}

static int check_shape(int64_t n, int64_t e, int32_t f, int32_t h, int32_t d) {
    if (n <= 0 || e < 0 || f <= 0 || h <= 0 || d <= 0) return fail(GAT_E_INVALID, "op: bad sizes");
    if (n > 0x7fffffffLL || e > 0x7fffffffLL) return fail(GAT_E_UNSUPPORTED, "op: int32 CSR limits (E:1045-1046) exceeded");
    return 0;
}
static int project(Scratch& t, const float* x, const float* w, int64_t n, int32_t f, int32_t hd, float** PL, float** PR, hipStream_t s) {
    GAT_TRY(t.get(PL, n * hd));
    GAT_TRY(t.get(PR, n * hd));
    return launch_project(x, w, *PL, *PR, n, f, hd, kPartBoth, false, nullptr, 0, s);
}
// gPL / gPR of one layer from reference-layout inputs (shared by a9 and a10)
static int messages(Scratch& t, int64_t n, int32_t h, int64_t e, int32_t f, int32_t d, float slope, const int32_t* src, const int32_t* dst,
                    const float* alpha, const float* x, const float* w, const float* g, const float* ge, const float* a, float* grad_a,
                    float** gPL, float** gPR, hipStream_t s) {
    const int hd = h * d;
    float *PL, *PR;
    GAT_TRY(project(t, x, w, n, f, hd, &PL, &PR, s));
    GAT_TRY(t.get(gPL, n * hd));
    GAT_TRY(t.get(gPR, n * hd));
    GAT_HIP(hipMemsetAsync(*gPL, 0, (size_t)n * hd * sizeof(float), s));
    GAT_HIP(hipMemsetAsync(*gPR, 0, (size_t)n * hd * sizeof(float), s));
    if (e > 0) {
        if (grad_a) hipLaunchKernelGGL(op_messages_kernel<true>, dim3(grid_for(e * hd)), dim3(kBlock), (size_t)hd * sizeof(float), s, PL, PR, g, src, dst,
                                       alpha, ge, a, *gPL, *gPR, grad_a, e, h, d, slope);
        else hipLaunchKernelGGL(op_messages_kernel<false>, dim3(grid_for(e * hd)), dim3(kBlock), 0, s, PL, PR, g, src, dst, alpha, ge, a, *gPL, *gPR,
                                (float*)nullptr, e, h, d, slope);
        GAT_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace
}  // namespace gat

using namespace gat;

extern "C" {

int gat_op_edge_score(const float* d_x, const int32_t* d_col_idx, const int32_t* d_dst, const float* d_w, const float* d_a,
                      float* d_attn_score, int64_t n, int32_t in_dim, int32_t out_dim, int32_t h, int64_t e, float slope, void* stream) {
    if (!d_x || !d_w || !d_a || (e > 0 && (!d_col_idx || !d_dst || !d_attn_score))) return fail(GAT_E_INVALID, "gat_op_edge_score: null argument");
    GAT_TRY(check_shape(n, e, in_dim, h, out_dim));
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Scratch t;
    float *PL, *PR;
    GAT_TRY(project(t, d_x, d_w, n, in_dim, h * out_dim, &PL, &PR, s));
    hipLaunchKernelGGL(op_edge_score_kernel, dim3(grid_for(e * h)), dim3(kBlock), 0, s, PL, PR, d_col_idx, d_dst, d_a, d_attn_score, e, h, out_dim, slope);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_max_sum(const int32_t* d_row_ptr, const float* d_attn_score, int64_t n, int32_t h, int64_t e, float* d_max, float* d_sum, void* stream) {
    if (!d_row_ptr || !d_max || !d_sum || (e > 0 && !d_attn_score)) return fail(GAT_E_INVALID, "gat_op_max_sum: null argument");
    GAT_TRY(check_shape(n, e, 1, h, 1));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_max_sum_kernel, dim3(grid_for(n * 64)), dim3(kBlock), 0, s, d_row_ptr, d_attn_score, n, h, e, d_max, d_sum);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_attn_coeff(const int32_t* d_col_idx, const int32_t* d_dst, const float* d_attn_score, const float* d_max, const float* d_sum,
                      float* d_attn_coeff, int64_t e, int32_t h, int64_t n, void* stream) {
    (void)d_col_idx;                                 // carried by the reference's signature (E:362), never read there either
    if (e > 0 && (!d_dst || !d_attn_score || !d_max || !d_sum || !d_attn_coeff)) return fail(GAT_E_INVALID, "gat_op_attn_coeff: null argument");
    GAT_TRY(check_shape(n, e, 1, h, 1));
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_attn_coeff_kernel, dim3(grid_for(e * h)), dim3(kBlock), 0, s, d_dst, d_attn_score, d_max, d_sum, d_attn_coeff, e, h, n);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_aggregate(const int32_t* d_src, const int32_t* d_dst, const float* d_attn_coeff, const float* d_in_feat, const float* d_w,
                     float* d_out_feat, int64_t n, int32_t h, int64_t e, int32_t in_dim, int32_t out_dim, void* stream) {
    if (!d_in_feat || !d_w || !d_out_feat || (e > 0 && (!d_src || !d_dst || !d_attn_coeff))) return fail(GAT_E_INVALID, "gat_op_aggregate: null argument");
    GAT_TRY(check_shape(n, e, in_dim, h, out_dim));
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Scratch t;
    float* PL;
    GAT_TRY(t.get(&PL, n * h * out_dim));
    GAT_TRY(launch_project(d_in_feat, d_w, PL, nullptr, n, in_dim, h * out_dim, kPartLeft, false, nullptr, 0, s));
    hipLaunchKernelGGL(op_aggregate_kernel, dim3(grid_for(e * h * out_dim)), dim3(kBlock), 0, s, PL, d_src, d_dst, d_attn_coeff, d_out_feat, e, h, out_dim);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_post_activation(const float* d_out_feat, float* d_H, int64_t n, int32_t h, int32_t out_dim, int32_t is_last, float slope, void* stream) {
    if (!d_out_feat || !d_H) return fail(GAT_E_INVALID, "gat_op_post_activation: null argument");
    GAT_TRY(check_shape(n, 0, 1, h, out_dim));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_post_activation_kernel, dim3(grid_for(n * h * out_dim)), dim3(kBlock), 0, s, d_out_feat, d_H, n, h, out_dim, is_last, slope);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_output_head(const float* d_wo, const float* d_last_layer_output, float* d_z, float* d_y, int64_t n, int32_t c, int32_t out_dim_last, void* stream) {
    if (!d_wo || !d_last_layer_output || !d_z || !d_y) return fail(GAT_E_INVALID, "gat_op_output_head: null argument");
    if (n <= 0 || c <= 0 || out_dim_last <= 0) return fail(GAT_E_INVALID, "gat_op_output_head: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_output_head_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, d_wo, d_last_layer_output, d_z, d_y, n, c, out_dim_last);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_loss_accuracy(const float* d_y, const int32_t* d_labels, float* d_losses, int32_t* d_corrects, int64_t n, int32_t c, void* stream) {
    if (!d_y || !d_labels || !d_losses || !d_corrects) return fail(GAT_E_INVALID, "gat_op_loss_accuracy: null argument");
    if (n <= 0 || c <= 0) return fail(GAT_E_INVALID, "gat_op_loss_accuracy: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_loss_accuracy_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, d_y, d_labels, d_losses, d_corrects, n, c);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_output_gradients(const float* d_y, const int32_t* d_labels, const float* d_hL, const float* d_HL, const float* d_wo, float* grad_d_wo,
                            float* grad_d_hL, int64_t n, int32_t c, int32_t out_dim_l, int32_t num_heads, float slope, int32_t flat_lrelu_index,
                            void* stream) {
    if (!d_y || !d_labels || !d_hL || !d_HL || !d_wo || !grad_d_wo || !grad_d_hL) return fail(GAT_E_INVALID, "gat_op_output_gradients: null argument");
    if (n <= 0 || c <= 0 || out_dim_l <= 0 || num_heads <= 0) return fail(GAT_E_INVALID, "gat_op_output_gradients: bad sizes");
    if ((size_t)c * out_dim_l * sizeof(float) > 60 * 1024) return fail(GAT_E_UNSUPPORTED, "gat_op_output_gradients: C * D_L beyond the LDS partial");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 1024);
    hipLaunchKernelGGL(op_output_gradients_kernel, dim3(blocks), dim3(kBlock), (size_t)c * out_dim_l * sizeof(float), s, d_y, d_labels, d_hL, d_HL, d_wo,
                       grad_d_wo, grad_d_hL, n, c, out_dim_l, num_heads, slope, flat_lrelu_index);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_grad_attn_coeff(int64_t e, int32_t h, int32_t in_dim, int32_t out_dim, const int32_t* d_src, const int32_t* d_dst, const float* d_features,
                           const float* d_w, const float* d_grad_input, float* d_grad_attn_coeff, int64_t n, void* stream) {
    if (!d_features || !d_w || !d_grad_input || (e > 0 && (!d_src || !d_dst || !d_grad_attn_coeff))) return fail(GAT_E_INVALID, "gat_op_grad_attn_coeff: null argument");
    GAT_TRY(check_shape(n, e, in_dim, h, out_dim));
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    Scratch t;
    float* PL;
    GAT_TRY(t.get(&PL, n * h * out_dim));
    GAT_TRY(launch_project(d_features, d_w, PL, nullptr, n, in_dim, h * out_dim, kPartLeft, false, nullptr, 0, s));
    hipLaunchKernelGGL(op_grad_attn_coeff_kernel, dim3(grid_for(e * h)), dim3(kBlock), 0, s, PL, d_grad_input, d_src, d_dst, d_grad_attn_coeff, e, h, out_dim);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_grad_attn_score(const int32_t* d_row_ptr, const int32_t* d_dst, const float* d_alpha, const float* d_grad_alpha, float* d_grad_e, int64_t n,
                           int32_t h, int64_t e, void* stream) {
    (void)d_dst;                                     // the reference finds the row through d_dst (E:671); the rows are walked directly here
    if (!d_row_ptr || (e > 0 && (!d_alpha || !d_grad_alpha || !d_grad_e))) return fail(GAT_E_INVALID, "gat_op_grad_attn_score: null argument");
    GAT_TRY(check_shape(n, e, 1, h, 1));
    if (e == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_grad_attn_score_kernel, dim3(grid_for(n * 64)), dim3(kBlock), 0, s, d_row_ptr, d_alpha, d_grad_alpha, d_grad_e, n, h, e);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_grad_parameters(int64_t e, int32_t h, const int32_t* d_src, const int32_t* d_dst, const float* d_features, const float* d_input_gradients,
                           const float* d_grad_attn_score, const float* d_attn_coeff, const float* d_w, const float* d_a, float* grad_w, float* grad_a,
                           int32_t in_dim, int32_t out_dim, float slope, int64_t n, void* stream) {
    if (!d_features || !d_input_gradients || !d_w || !d_a || !grad_w || !grad_a || (e > 0 && (!d_src || !d_dst || !d_grad_attn_score || !d_attn_coeff)))
        return fail(GAT_E_INVALID, "gat_op_grad_parameters: null argument");
    GAT_TRY(check_shape(n, e, in_dim, h, out_dim));
    hipStream_t s = (hipStream_t)stream;
    Scratch t;
    float *gPL, *gPR, *scr;
    GAT_TRY(messages(t, n, h, e, in_dim, out_dim, slope, d_src, d_dst, d_attn_coeff, d_features, d_w, d_input_gradients, d_grad_attn_score, d_a, grad_a, &gPL, &gPR, s));
    GAT_TRY(t.get(&scr, grad_w_scratch_floats(n, in_dim, h * out_dim)));
    GAT_TRY(launch_grad_w(gPL, gPR, d_features, grad_w, scr, n, in_dim, h * out_dim, kPartBoth, s));
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_features_input_gradients(int64_t n, int32_t h, int64_t e, int32_t in_dim, int32_t out_dim, float slope, const int32_t* d_src,
                                    const int32_t* d_dst, const float* d_attn_coeff, const float* d_input_features, const float* d_w,
                                    const float* d_input_gradients, const float* d_grad_attn_score, const float* d_a, float* d_grad_x_features,
                                    void* stream) {
    if (!d_input_features || !d_w || !d_input_gradients || !d_a || !d_grad_x_features || (e > 0 && (!d_src || !d_dst || !d_attn_coeff || !d_grad_attn_score)))
        return fail(GAT_E_INVALID, "gat_op_features_input_gradients: null argument");
    GAT_TRY(check_shape(n, e, in_dim, h, out_dim));
    hipStream_t s = (hipStream_t)stream;
    Scratch t;
    float *gPL, *gPR, *gx;
    GAT_TRY(messages(t, n, h, e, in_dim, out_dim, slope, d_src, d_dst, d_attn_coeff, d_input_features, d_w, d_input_gradients, d_grad_attn_score, d_a, nullptr, &gPL, &gPR, s));
    GAT_TRY(t.get(&gx, n * in_dim));
    GAT_TRY(launch_grad_x(gPL, gPR, d_w, nullptr, gx, n, in_dim, h * out_dim, slope, s));
    hipLaunchKernelGGL(op_add_kernel, dim3(grid_for(n * in_dim)), dim3(kBlock), 0, s, d_grad_x_features, gx, n * in_dim);     // accumulates like E:868-869
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_preact_gradient(int64_t n, float slope, int32_t in_dim, const float* d_pre_activation, float* d_gradients, void* stream) {
    if (!d_pre_activation || !d_gradients) return fail(GAT_E_INVALID, "gat_op_preact_gradient: null argument");
    if (n <= 0 || in_dim <= 0) return fail(GAT_E_INVALID, "gat_op_preact_gradient: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(op_preact_gradient_kernel, dim3(grid_for(n * in_dim)), dim3(kBlock), 0, s, d_pre_activation, d_gradients, n * in_dim, slope);
    GAT_HIP(hipGetLastError());
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
