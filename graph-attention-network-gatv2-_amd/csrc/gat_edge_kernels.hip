// gat_edge_kernels.hip — the edge-centric GATv2 hot path for gfx950 (wave64).
//
// Replaces the reference's per-(head,edge) thread kernels (GATv2_edge_based.cu, E:<line>):
//   forward  a2 edge score E:279-324, a3 max/sum E:326-359, a4 coeff E:362-384,
//            a5 aggregate E:386-424 (atomicAdd), a6 activation E:426-459
//   backward a7 grad alpha E:612-651, a8 softmax backward E:654-696 (O(deg^2)),
//            edge parts of a9 E:698-798 and a10 E:801-874
//   a1 CSR->COO E:67-84
// with ONE destination-segmented pass per direction over projected features
//   PL = X·W_left^T, PR = X·W_right^T   (s[e,h,k] = PL[src,h,k] + PR[dst,h,k]).
// A work item = a CSR row, or a segment of a long row (power-law hubs are split: 256 edges, 64 on
// graphs below 16 M edges).  Training path (edge_fwd3 / edge_bwd3): a LANE GROUP (H*D/N lanes, N = 4
// channels per lane) owns an item, a wave carries 64/(H*D/N) items of similar length side by side;
// parity-tap path and shapes without that layout: a wave owns an item and walks it in 16-edge chunks.
// Every neighbour row PL[src] is one coalesced H*D-float read, the per-head reductions are
// DPP/shuffle all-reduces, the softmax is online (single gather of PL[src] per edge), h_pre is
// written once without atomics, and the softmax backward uses
//   sum_k galpha_k alpha_k == <g[dst,h,:], h_pre[dst,h,:]>
// which makes it O(E) and single-pass.  The backward leaves ONE 64-byte record per edge (alpha, ge and
// the LeakyReLU' decisions) in its source-major slot; gat_csc.hip's pull pass turns the records into
// gPL.  Split rows are finished by small fix-up kernels that merge the per-segment partials.
//
// HBM layouts: PL/PR/h_pre/g [rows][H*D] f32; alpha, ge [E][H] f32 (edge-major, taps only); records
// [E+1][H*D/N] words by slot; CSR int32; work items int4 {row, beg, end, slot|-1}, longest first
// inside windows of 4,096 rows.
#include "gat_internal.h"

#include <map>
#include <mutex>

namespace gat {
namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// All-reduce (sum) over aligned groups of D consecutive lanes, D a power of two <= 64.
// quad_perm xor1, xor2, then row_half_mirror / row_mirror (valid as xor steps because the
// previous steps made every quad / half-row uniform), then cross-row shuffles.
template <int D>
__device__ __forceinline__ float group_sum(float t) {
    if constexpr (D >= 2) t += dpp_mov<0xB1>(t);     // quad_perm [1,0,3,2]
    if constexpr (D >= 4) t += dpp_mov<0x4E>(t);     // quad_perm [2,3,0,1]
    if constexpr (D >= 8) t += dpp_mov<0x141>(t);    // row_half_mirror
    if constexpr (D >= 16) t += dpp_mov<0x140>(t);   // row_mirror
    if constexpr (D >= 32) t += __shfl_xor(t, 16);
    if constexpr (D >= 64) t += __shfl_xor(t, 32);
    return t;
}

// The same all-reduce for N independent values, stage by stage, with the DPP stages written as
// fused v_add_f32_dpp (hipcc emits v_mov_b32_dpp + v_add + s_nop for the builtin form).  hipcc
// pads nothing inside asm: the VALU-write -> DPP-read hazard (2 wait states) is covered by one
// `s_nop 1` per stage — inside a stage consecutive instructions touch different registers, and
// volatile asm statements keep their order.
// One stage = ONE asm statement over all N values, so every input is complete before the block and
// nothing can be scheduled between its instructions.
#define GAT_DPP_I(i, C) "v_add_f32_dpp %" #i ", %" #i ", %" #i " " C " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define GAT_STR_1(C) GAT_DPP_I(0, C)
#define GAT_STR_2(C) GAT_STR_1(C) GAT_DPP_I(1, C)
#define GAT_STR_4(C) GAT_STR_2(C) GAT_DPP_I(2, C) GAT_DPP_I(3, C)
#define GAT_STR_8(C) GAT_STR_4(C) GAT_DPP_I(4, C) GAT_DPP_I(5, C) GAT_DPP_I(6, C) GAT_DPP_I(7, C)
#define GAT_STR_16(C) GAT_STR_8(C) GAT_DPP_I(8, C) GAT_DPP_I(9, C) GAT_DPP_I(10, C) GAT_DPP_I(11, C) \
    GAT_DPP_I(12, C) GAT_DPP_I(13, C) GAT_DPP_I(14, C) GAT_DPP_I(15, C)
#define GAT_OPS_1 "+v"(t[0])
#define GAT_OPS_2 GAT_OPS_1, "+v"(t[1])
#define GAT_OPS_4 GAT_OPS_2, "+v"(t[2]), "+v"(t[3])
#define GAT_OPS_8 GAT_OPS_4, "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
#define GAT_OPS_16 GAT_OPS_8, "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15])
#define GAT_DPP_STAGE(C)                                                                     \
    if constexpr (N == 16) asm volatile("s_nop 1\n\t" GAT_STR_16(C) : GAT_OPS_16);           \
    else if constexpr (N == 8) asm volatile("s_nop 1\n\t" GAT_STR_8(C) : GAT_OPS_8);         \
    else if constexpr (N == 4) asm volatile("s_nop 1\n\t" GAT_STR_4(C) : GAT_OPS_4);         \
    else if constexpr (N == 2) asm volatile("s_nop 1\n\t" GAT_STR_2(C) : GAT_OPS_2);         \
    else asm volatile("s_nop 1\n\t" GAT_STR_1(C) : GAT_OPS_1);
template <int D, int N>
__device__ __forceinline__ void group_sum_n(float (&t)[N]) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "slot count");
    if constexpr (D >= 2) { GAT_DPP_STAGE("quad_perm:[1,0,3,2]") }
    if constexpr (D >= 4) { GAT_DPP_STAGE("quad_perm:[2,3,0,1]") }
    if constexpr (D >= 8) { GAT_DPP_STAGE("row_half_mirror") }
    if constexpr (D >= 16) { GAT_DPP_STAGE("row_mirror") }
    if constexpr (D >= 2) asm volatile("s_nop 1");     // asm result -> compiler-scheduled DPP readers
    if constexpr (D >= 32) {
#pragma unroll
        for (int u = 0; u < N; ++u) t[u] += __shfl_xor(t[u], 16);
    }
    if constexpr (D >= 64) {
#pragma unroll
        for (int u = 0; u < N; ++u) t[u] += __shfl_xor(t[u], 32);
    }
}

// ------------------------------------------------------------------------------------------------
// a1: CSR -> COO.  Reference: thread per row, serial over its edges (hub rows serialise).
// Here: thread per edge, row found by binary search in row_ptr (load-balanced, coalesced stores).
// Output identical to the reference: src[e] = col_idx[e], dst[e] = row containing e.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_to_coo_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ col_idx,
                                                         int32_t* __restrict__ src,
                                                         int32_t* __restrict__ dst, int64_t n_rows,
                                                         int64_t n_edges, int32_t dst_offset) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
        // largest row with row_ptr[row] <= e  (upper_bound - 1; skips empty rows correctly)
        int64_t lo = 0, hi = n_rows;          // invariant: row_ptr[lo] <= e < row_ptr[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)row_ptr[mid] <= e) lo = mid; else hi = mid;
        }
        src[e] = col_idx[e];
        dst[e] = (int32_t)lo + dst_offset;
    }
}

// ------------------------------------------------------------------------------------------------
// Forward edge pass, fast path: H*D in {8,16,32,64}, D a power of two.
// G = 64/HD edges are gathered per wave-instruction (lane -> (group, channel)); U gathers per
// group are issued back to back before any is consumed.  Scores are kept in the log2 domain
// (a pre-scaled by log2 e) so that every softmax term is one v_exp_f32.
// ------------------------------------------------------------------------------------------------
template <int HD, int D>
__device__ __forceinline__ void fwd_write_row(const EdgeFwdArgs& A, int64_t row, int lane, float m2, float Z,
                                              float acc) {
    constexpr int H = HD / D;
    const int c = lane % HD, gidx = lane / HD;
    const float hp = acc * __builtin_amdgcn_rcpf(Z + 1e-8f);     // E:379 epsilon
    if (gidx == 0) {
        A.hpre[row * HD + c] = hp;
        if ((c % D) == 0) {                          // softmax stats, log2 domain: the backward
            A.mstat[row * H + c / D] = m2;               // recomputes alpha = exp2(score2 - m2)/(Z+eps)
            A.zstat[row * H + c / D] = Z;
        }
    }
    const float act = lrelu(hp, A.slope);
    if (!A.is_last) {
        if (gidx == 0) A.hout[row * HD + c] = act;   // concat heads (E:452-457)
    } else {
        float t = act;                               // activate, then average heads (E:440-449)
#pragma unroll
        for (int off = D; off < HD; off <<= 1) t += __shfl_xor(t, off);
        if (lane < D) A.hout[row * D + lane] = t / (float)H;
    }
}

// alpha[b..e_end) holds raw log2-domain scores: normalise in place (contiguous [deg][H] slice).
template <int HD, int D>
__device__ __forceinline__ void fwd_normalize_slice(const EdgeFwdArgs& A, int b, int e_end, int lane, float m2,
                                                    float inv) {
    constexpr int H = HD / D;
    const int total = (e_end - b) * H;
    float* arow = A.alpha + (int64_t)b * H;
    for (int i0 = 0; i0 < total; i0 += 64) {
        const int i = i0 + lane;
        const int h = i % H;
        const float mh = __shfl(m2, h * D), ih = __shfl(inv, h * D);
        if (i < total) arow[i] = exp2_fast(arow[i] - mh) * ih;
    }
}

// A value the compiler must treat as per-lane: predicates built from it become EXEC masks instead
// of scalar branches, which keeps the chunk bodies below straight-line code (hipcc drains
// vmcnt(0) at every join point of a scalar branch that has memory operations behind it).
__device__ __forceinline__ int per_lane(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// PL[row][c] with the address as uniform base + 32-bit byte offset: one v_lshl_add_u32 per gather and
// the `global_load_dword v, voff, s[base]` form, instead of a sign-extension and two 64-bit VALU ops.
// Valid while the table is < 4 GiB (n_table * H*D * 4; checked by the launchers).
// BF: the table / message rows are stored as bf16 (cfg.storage_dtype; halves the gathered, stored and
// exchanged bytes — arithmetic stays fp32).  bf16 -> f32 is a 16-bit shift; f32 -> bf16 rounds to
// nearest even.
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <int HD, bool BF>
__device__ __forceinline__ float gather_row(const float* __restrict__ table, int row, int c) {
    if constexpr (BF) {
        const uint32_t off = (uint32_t)row * (uint32_t)(HD * 2) + (uint32_t)c * 2u;
        return bf16_to_f32(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(table) + off));
    } else {
        const uint32_t off = (uint32_t)row * (uint32_t)(HD * 4) + (uint32_t)c * 4u;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + off);
    }
}
// msg[slot][c] = v.  The message array is E*H*D*4 bytes (15.8 GB on the benchmark graph): 64-bit row
// base.  When the whole wave works on one edge the slot is wave-uniform: row base on the SALU,
// lane offset in a VGPR (`global_store_dword voff, v, s[base]`), no VALU address math.
template <int HD, bool UNIFORM, bool BF>
__device__ __forceinline__ void store_row(float* __restrict__ msg, int slot, int c, float v) {
    constexpr int ES = BF ? 2 : 4;
    char* rowb;
    uint32_t off;
    if constexpr (UNIFORM) {
        rowb = reinterpret_cast<char*>(msg) + ((int64_t)__builtin_amdgcn_readfirstlane(slot) * (HD * ES));
        // the lane offset must reach instruction selection as an opaque zext(i32) next to the scalar base: a
        // visible c*4 is re-associated into (msg + c*4) + slot*row, i.e. a 64-bit VALU add per store
        off = (uint32_t)per_lane(c * ES);
    } else {
        rowb = reinterpret_cast<char*>(msg) + (int64_t)slot * (HD * ES);
        off = (uint32_t)(c * ES);
    }
    if constexpr (BF) *reinterpret_cast<uint16_t*>(rowb + off) = f32_to_bf16(v);
    else *reinterpret_cast<float*>(rowb + off) = v;
}

// One chunk of UU slots per edge group: UU independent gathers issued back to back (indices
// clamped into the item, so loads need no predicate), then scores, then the online-softmax update.
template <int HD, int D, int UU, int USC, bool ALPHA, bool BF>
__device__ __forceinline__ void fwd_chunk(const EdgeFwdArgs& A, int e0, int e_end, int e_end_v, int c, int gidx,
                                          float pr, float ac2, bool multi, float (&sc)[USC], float& m, float& Z,
                                          float& acc) {
    constexpr int G = 64 / HD;
    constexpr int H = HD / D;
    float v[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const int j = e0 + u * G + gidx;
        const int jc = j < e_end ? j : e_end - 1;
        v[u] = gather_row<HD, BF>(A.PL, A.col_idx[jc], c);
    }
    // scores: the cross-lane stages run slot-interleaved (UU independent DPP chains), so that no
    // stage waits on the VALU->DPP hazard of its own predecessor
    float t[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) { const float s = v[u] + pr; t[u] = ac2 * fmaxf(s, s * A.slope); }
    group_sum_n<D, UU>(t);
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const int j = e0 + u * G + gidx;
        sc[u] = (j < e_end_v) ? t[u] : -INFINITY;
        cm = fmaxf(cm, sc[u]);
        if constexpr (ALPHA) {                       // tap: the reference's attn_score (natural-log domain, E:323)
            if (A.score != nullptr && (c % D) == 0 && j < e_end_v) A.score[(int64_t)j * H + c / D] = t[u] * kLn2;
        }
    }
    const float mn = fmaxf(m, cm);
    const float scale = exp2_fast(m - mn);
    Z *= scale;
    acc *= scale;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const float p = exp2_fast(sc[u] - mn);       // 0 for padded slots
        Z += p;
        acc = fmaf(p, v[u], acc);
    }
    m = mn;
    if (ALPHA && multi) {                            // park raw scores; normalised later
        float* ap = A.alpha + (int64_t)e0 * H + (gidx * H + c / D);
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const int j = e0 + u * G + gidx;
            if ((c % D) == 0 && j < e_end_v) ap[u * G * H] = sc[u];
        }
    }
}

// ALPHA: also materialise attn_coeff [E][H] (parity taps only; the training path never needs it).
template <int HD, int D, bool ALPHA, bool BF = false>
__global__ __launch_bounds__(256) void edge_fwd_kernel(EdgeFwdArgs A) {
    constexpr int G = 64 / HD;      // edges per wave-instruction
    constexpr int U = 16 / G;       // gathers in flight per group
    constexpr int CH = 16;          // edges per chunk (= U*G)
    constexpr int H = HD / D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t it = (int64_t)blockIdx.x * 4 + wave;
    if (it >= A.n_items) return;
    const int4 item = A.items[it];
    const int64_t row = item.x;
    const int b = item.y, e_end = item.z, slot = item.w;
    const int e_end_v = per_lane(e_end);
    const bool split = slot >= 0;                    // one segment of a long row
    const int c = lane % HD, gidx = lane / HD;
    const float pr = A.PR[row * HD + c];
    const float ac2 = A.a[c] * kLog2e;
    float m = -1e9f * kLog2e, Z = 0.f, acc = 0.f;    // E:336 seeds the max with -1e9f
    const bool multi = split || (e_end - b) > CH;
    float sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sc[u] = -INFINITY;

    for (int e0 = b; e0 < e_end; e0 += CH) {
        if constexpr (U >= 2) {
            if (e_end - e0 <= CH / 2) fwd_chunk<HD, D, U / 2, U, ALPHA, BF>(A, e0, e_end, e_end_v, c, gidx, pr, ac2, multi, sc, m, Z, acc);
            else fwd_chunk<HD, D, U, U, ALPHA, BF>(A, e0, e_end, e_end_v, c, gidx, pr, ac2, multi, sc, m, Z, acc);
        } else {
            fwd_chunk<HD, D, U, U, ALPHA, BF>(A, e0, e_end, e_end_v, c, gidx, pr, ac2, multi, sc, m, Z, acc);
        }
    }

    // merge the G edge groups (online-softmax combine); afterwards all groups agree
#pragma unroll
    for (int off = HD; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off), Zo = __shfl_xor(Z, off), ao = __shfl_xor(acc, off);
        const float mn = fmaxf(m, mo);
        const float s1 = exp2_fast(m - mn), s2 = exp2_fast(mo - mn);
        Z = Z * s1 + Zo * s2;
        acc = acc * s1 + ao * s2;
        m = mn;
    }

    if (split) {                                     // partial (m, Z, acc) of this segment
        if (gidx == 0) {
            A.part_acc[(int64_t)slot * HD + c] = acc;
            if ((c % D) == 0) {
                A.part_mz[(int64_t)slot * 2 * H + c / D] = m;
                A.part_mz[(int64_t)slot * 2 * H + H + c / D] = Z;
            }
        }
        return;
    }
    if constexpr (ALPHA) {
    const float inv = __builtin_amdgcn_rcpf(Z + 1e-8f);      // E:379; v_rcp_f32 (1 ulp) instead of a divide
    if (!multi) {                                    // whole row still in registers
        float* ap = A.alpha + (int64_t)b * H + (gidx * H + c / D);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = b + u * G + gidx;
            if ((c % D) == 0 && j < e_end_v) ap[u * G * H] = exp2_fast(sc[u] - m) * inv;
        }
    } else {
        // raw scores were written by other lanes of this wave: drain the stores, then sweep
        __threadfence_block();
        fwd_normalize_slice<HD, D>(A, b, e_end, lane, m, inv);
    }
    }
    fwd_write_row<HD, D>(A, row, lane, m, Z, acc);
}

// Split rows: one wave per segment merges ALL partials of its row (L2-hot, <= a few hundred
// bytes each), normalises its own alpha slice, and the row's first segment writes the outputs.
template <int HD, int D, bool ALPHA>
__global__ __launch_bounds__(256) void edge_fwd_fix_kernel(EdgeFwdArgs A) {
    constexpr int H = HD / D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // with alpha every segment normalises its own slice (wave per slot); without it only the row outputs
    // remain: wave per SPLIT ROW (the first-slot list appended to slot_info)
    const int k = blockIdx.x * 4 + wave;
    if (k >= (ALPHA ? A.n_slots : A.n_split)) return;
    const int slot = ALPHA ? k : A.slot_info[A.n_slots + k].x;
    const int4 info = A.slot_info[slot];             // {row, first_slot, nseg, item}
    const int4 item = A.items[info.w];
    const int c = lane % HD;
    float m = -1e9f * kLog2e, Z = 0.f, acc = 0.f;
    auto merge = [&](float ms, float Zs, float as) {
        const float mn = fmaxf(m, ms);
        const float s1 = exp2_fast(m - mn), s2 = exp2_fast(ms - mn);
        Z = Z * s1 + Zs * s2;
        acc = acc * s1 + as * s2;
        m = mn;
    };
    // a hub row is hundreds of segments (13 k in-edges / 64): the partials of kB segments are loaded together and merged in
    // segment order — the same arithmetic as one by one, without a memory latency per segment (83 -> us per launch at the
    // P = 8 shard shape, where the row keeps its whole in-degree and the segments are short)
    constexpr int kB = 8;
    int sg = info.y;
    const int s_end = info.y + info.z;
    for (; sg + kB <= s_end; sg += kB) {
        float ms[kB], Zs[kB], as[kB];
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            ms[i] = A.part_mz[(int64_t)(sg + i) * 2 * H + c / D];
            Zs[i] = A.part_mz[(int64_t)(sg + i) * 2 * H + H + c / D];
            as[i] = A.part_acc[(int64_t)(sg + i) * HD + c];
        }
#pragma unroll
        for (int i = 0; i < kB; ++i) merge(ms[i], Zs[i], as[i]);
    }
    for (; sg < s_end; ++sg)
        merge(A.part_mz[(int64_t)sg * 2 * H + c / D], A.part_mz[(int64_t)sg * 2 * H + H + c / D], A.part_acc[(int64_t)sg * HD + c]);
    if constexpr (ALPHA) fwd_normalize_slice<HD, D>(A, item.y, item.z, lane, m, __builtin_amdgcn_rcpf(Z + 1e-8f));
    if (slot == info.y) fwd_write_row<HD, D>(A, info.x, lane, m, Z, acc);
}

// ------------------------------------------------------------------------------------------------
// Backward edge pass, fast path.  Fixed grid, work items dealt round-robin (all items are <= kSegEdges
// edges, so the static deal is balanced) — fixed so that the per-block grad_a partials stay small.
// Per edge: one PL[src] gather, one alpha read, one message row out.
//   galpha = <g[dst,h,:], PL[src,h,:]>                 (E:632-646)
//   ge     = alpha (galpha - <g[dst,h,:], h_pre[dst,h,:]>)   (≡ E:682-693)
//   gs     = ge a LReLU'(s)                             (E:774-775)
//   grad_a += ge LReLU(s)   gPR[dst] += gs   gPL[src] += g alpha + gs   (E:769-782, 859-869)
// STORE: message row -> its CSC slot (summed per source by gpl_sum_kernel); else float atomics.
// ------------------------------------------------------------------------------------------------
template <int HD, int D, int UU, bool STORE, bool TAPS, int DBG, bool BF>
__device__ __forceinline__ void bwd_chunk(const EdgeBwdArgs& A, int e0, int e_end, int e_end_v, int c, int gidx,
                                          float g, float pr, float dot, float ac, float ac2, float m2, float inv,
                                          float& ga, float& gpr) {
    constexpr int G = 64 / HD;
    constexpr int H = HD / D;
    float v[UU];
    int sid[UU];            // gPL row (atomics path) or message slot (store path)
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const int j = e0 + u * G + gidx;
        const int jc = j < e_end ? j : e_end - 1;                // clamped: loads need no predicate
        const int src = A.col_idx[jc];
        v[u] = gather_row<HD, BF>(A.PL, src, c);
        if constexpr (STORE) sid[u] = (DBG == 2) ? jc : A.pos[jc]; else sid[u] = src;
    }
    // Compute in passes of P <= 8 slots (bounds the live registers; the first pass starts as soon as
    // its own gathers have landed).  alpha is recomputed from the forward's per-(row, head) softmax
    // stats instead of being read back from HBM:
    //   score2 = sum_k a2 LReLU(s),  alpha = exp2(score2 - m2) / (Z + eps)          (E:378-379)
    constexpr int P = UU > 8 ? 8 : UU;
#pragma unroll
    for (int p0 = 0; p0 < UU; p0 += P) {
        float al[P], ga_[P];
#pragma unroll
        for (int q = 0; q < P; ++q) { const float s = v[p0 + q] + pr; al[q] = ac2 * fmaxf(s, s * A.slope); }
        group_sum_n<D, P>(al);
#pragma unroll
        for (int q = 0; q < P; ++q) { al[q] = exp2_fast(al[q] - m2) * inv; ga_[q] = g * v[p0 + q]; }
        group_sum_n<D, P>(ga_);                                  // galpha, slot-interleaved DPP stages
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int u = p0 + q;
            const int j = e0 + u * G + gidx;
            const bool valid = j < e_end_v;
            const float ge = valid ? al[q] * (ga_[q] - dot) : 0.f;   // padded slots contribute nothing
            const float s = v[u] + pr;
            const bool pos = s > 0.f;
            const float gs = ge * ac * (pos ? 1.0f : A.slope);
            ga = fmaf(ge, fmaxf(s, s * A.slope), ga);
            gpr += gs;
            const float msg = fmaf(g, al[q], gs);                // d/dPL[src] from this edge
            if (valid && DBG != 1) {
                if constexpr (STORE) store_row<HD, G == 1, BF>(A.msg, sid[u], c, msg);
                else unsafeAtomicAdd(A.gPL + (int64_t)sid[u] * HD + c, msg);
            }
            if constexpr (TAPS) {
                if (valid && (c % D) == 0) {
                    A.ge[(int64_t)j * H + c / D] = ge;
                    if (A.galpha != nullptr) A.galpha[(int64_t)j * H + c / D] = ga_[q];
                }
            }
        }
    }
}

template <int HD, int D, bool STORE, bool TAPS, int DBG = 0, bool BF = false>
__global__ __launch_bounds__(256) void edge_bwd_kernel(EdgeBwdArgs A) {
    constexpr int G = 64 / HD;
    constexpr int U = 16 / G;
    constexpr int CH = 16;
    __shared__ float red[4][HD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % HD, gidx = lane / HD;
    const float ac = A.a[c];
    const float ac2 = ac * kLog2e;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float ga = 0.f;

    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < A.n_items; it += nwaves) {
        const int4 item = A.items[it];
        const int64_t row = item.x;
        const int b = item.y, e_end = item.z, slot = item.w;
        const int e_end_v = per_lane(e_end);
        const float hp = A.hpre[row * HD + c];
        float g;
        if (A.gh != nullptr) g = A.gh[row * A.gh_stride + (c % D)] * (hp > 0.f ? 1.0f : A.slope) * (1.0f / (float)(HD / D));   // E:598-603
        else {
            g = A.g[row * HD + c];
            if (A.g_raw) g *= hp > 0.f ? 1.0f : A.slope;         // E:888-892 applied by the consumer
        }
        const float pr = A.PR[row * HD + c];
        const float dot = group_sum<D>(g * hp);
        const float m2 = A.mstat[row * (HD / D) + c / D];
        const float inv = __builtin_amdgcn_rcpf(A.zstat[row * (HD / D) + c / D] + 1e-8f);
        float gpr = 0.f;
        for (int e0 = b; e0 < e_end; e0 += CH) {
            if constexpr (U >= 2) {
                if (e_end - e0 <= CH / 2) bwd_chunk<HD, D, U / 2, STORE, TAPS, DBG, BF>(A, e0, e_end, e_end_v, c, gidx, g, pr, dot, ac, ac2, m2, inv, ga, gpr);
                else bwd_chunk<HD, D, U, STORE, TAPS, DBG, BF>(A, e0, e_end, e_end_v, c, gidx, g, pr, dot, ac, ac2, m2, inv, ga, gpr);
            } else {
                bwd_chunk<HD, D, U, STORE, TAPS, DBG, BF>(A, e0, e_end, e_end_v, c, gidx, g, pr, dot, ac, ac2, m2, inv, ga, gpr);
            }
        }
#pragma unroll
        for (int off = HD; off < 64; off <<= 1) gpr += __shfl_xor(gpr, off);
        if (gidx == 0) {
            if (slot < 0) A.gPR[row * HD + c] = gpr;
            else A.part_acc[(int64_t)slot * HD + c] = gpr;      // segment partial, summed by the fix kernel
        }
    }
#pragma unroll
    for (int off = HD; off < 64; off <<= 1) ga += __shfl_xor(ga, off);
    if (gidx == 0) red[wave][c] = ga;
    __syncthreads();
    if (threadIdx.x < HD)
        A.ga_partial[(int64_t)blockIdx.x * HD + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Backward edge pass, PACKED lane layout (training path: store mode, no taps): a lane owns TWO adjacent
// channels, H*D/2 lanes own an edge, so a wave-instruction covers 2x the edges (H*D = 64: two 256-B rows
// per dwordx2 gather / store) and the per-channel arithmetic runs as packed fp32 (v_pk_add/mul/fma_f32).
// Per edge that is half the VALU and half the memory instructions of the layout above — the fp32 backward
// sits between its VALU floor and its HBM floor, the bf16 one is pure instruction issue.  Same math, same
// per-head summation tree up to the pairing of adjacent channels.
// ------------------------------------------------------------------------------------------------
// N adjacent channels per lane (N = 2 or 4): ext-vector arithmetic lowers to v_pk_add/mul/fma_f32 pairs
template <int N> struct VecOf;
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <int N> using vnf = typename VecOf<N>::T;

template <int N> __device__ __forceinline__ vnf<N> vzero() {
    vnf<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = 0.f;
    return r;
}
template <int N> __device__ __forceinline__ float hsum(vnf<N> v) {
    if constexpr (N == 2) return v[0] + v[1];
    else return (v[0] + v[1]) + (v[2] + v[3]);
}
template <int N> __device__ __forceinline__ vnf<N> lrelu_n(vnf<N> s, float slope) {
    const vnf<N> t = s * slope;
    vnf<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fmaxf(s[i], t[i]);
    return r;
}
// s > 0 ? a : b, elementwise
template <int N> __device__ __forceinline__ vnf<N> select_pos(vnf<N> s, vnf<N> a, vnf<N> b) {
    vnf<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = s[i] > 0.f ? a[i] : b[i];
    return r;
}
template <int N> __device__ __forceinline__ vnf<N> shfl_xor_n(vnf<N> v, int off) {
    vnf<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __shfl_xor(v[i], off);
    return r;
}

// cp = index of the lane's channel group inside the row (channels cp*N .. cp*N+N-1)
template <int HD, int N, bool BF>
__device__ __forceinline__ vnf<N> gather_row_n(const float* __restrict__ table, int row, int cp) {
    if constexpr (BF) {
        const uint32_t off = (uint32_t)row * (uint32_t)(HD * 2) + (uint32_t)cp * (uint32_t)(N * 2);
        const char* p = reinterpret_cast<const char*>(table) + off;
        vnf<N> r;
        if constexpr (N == 2) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
            r[0] = __builtin_bit_cast(float, w << 16); r[1] = __builtin_bit_cast(float, w & 0xFFFF0000u);
        } else {
            const uint2 w = *reinterpret_cast<const uint2*>(p);
            r[0] = __builtin_bit_cast(float, w.x << 16); r[1] = __builtin_bit_cast(float, w.x & 0xFFFF0000u);
            r[2] = __builtin_bit_cast(float, w.y << 16); r[3] = __builtin_bit_cast(float, w.y & 0xFFFF0000u);
        }
        return r;
    } else {
        const uint32_t off = (uint32_t)row * (uint32_t)(HD * 4) + (uint32_t)cp * (uint32_t)(N * 4);
#ifdef GAT_NT_GATHER      // experiment: gathered rows are used once by one wave — non-temporal policy
        return __builtin_nontemporal_load(reinterpret_cast<const vnf<N>*>(reinterpret_cast<const char*>(table) + off));
#else
        return *reinterpret_cast<const vnf<N>*>(reinterpret_cast<const char*>(table) + off);
#endif
    }
}
// write-once rows and records that this kernel never reads back: streaming stores keep them from displacing the gathered
// source rows in L2 (GAT_NT_STORES: experiment)
template <class T>
__device__ __forceinline__ void stream_store(T* p, const T& v) {
#ifdef GAT_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <int HD, int N, bool BF>
__device__ __forceinline__ void store_row_n(float* __restrict__ msg, int slot, int cp, vnf<N> v) {
    if constexpr (BF) {
        char* p = reinterpret_cast<char*>(msg) + (uint64_t)(uint32_t)slot * (HD * 2) + cp * (N * 2);   // slots are >= 0: zero-extend
        const uint32_t w0 = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        if constexpr (N == 2) *reinterpret_cast<uint32_t*>(p) = w0;
        else *reinterpret_cast<uint2*>(p) = make_uint2(w0, (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
    } else {
        char* p = reinterpret_cast<char*>(msg) + (uint64_t)(uint32_t)slot * (HD * 4) + cp * (N * 4);
        *reinterpret_cast<vnf<N>*>(p) = v;
    }
}

// Forward edge pass in the packed layout (training path: alpha not materialised).  Same online softmax as
// fwd_chunk; partials of split rows go to the same [slot][HD] / [slot][2H] arrays, so edge_fwd_fix_kernel
// finishes them unchanged.
template <int HD, int D, int N, int UU, bool BF>
__device__ __forceinline__ void fwd2_chunk(const EdgeFwdArgs& A, int e0, int e_end_v, int cp, int gidx, int srcv,
                                           vnf<N> pr, vnf<N> ac2, float& m, float& Z, vnf<N>& acc) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N;
    vnf<N> v[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) v[u] = gather_row_n<HD, N, BF>(A.PL, __shfl(srcv, u * G + gidx), cp);
    float t[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) t[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, A.slope));
    group_sum_n<DL, UU>(t);
    float cm = -INFINITY;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const int j = e0 + u * G + gidx;
        t[u] = (j < e_end_v) ? t[u] : -INFINITY;
        cm = fmaxf(cm, t[u]);
    }
    const float mn = fmaxf(m, cm);
    const float scale = exp2_fast(m - mn);
    Z *= scale;
    acc = acc * scale;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const float p = exp2_fast(t[u] - mn);        // 0 for padded slots
        Z += p;
        acc += p * v[u];
    }
    m = mn;
}

template <int HD, int D, int N, bool BF = false>
__global__ __launch_bounds__(256) void edge_fwd2_kernel(EdgeFwdArgs A) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int CH = 16;
    constexpr int U = CH / G;
    static_assert(D % N == 0 && U >= 1, "lane layout");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t it = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (it >= A.n_items) return;
    const int4 item = A.items[it];
    const int64_t row = item.x;
    const int b = item.y, e_end = item.z, slot = item.w;
    const int e_end_v = per_lane(e_end);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int c = N * cp;                            // first of the lane's channels
    const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(A.PR + row * HD + c);
    const vnf<N> ac2 = *reinterpret_cast<const vnf<N>*>(A.a + c) * kLog2e;
    float m = -1e9f * kLog2e, Z = 0.f;               // E:336 seeds the max with -1e9f
    vnf<N> acc = vzero<N>();
    auto load_idx = [&](int e0) {
        const int jl = e0 + (lane & (CH - 1));
        return A.col_idx[jl < e_end ? jl : e_end - 1];
    };
    int srcv = 0;
    if (b < e_end) srcv = load_idx(b);               // empty item (zero in-degree row): e_end - 1 would be b - 1, i.e. -1 for row 0
    for (int e0 = b; e0 < e_end; e0 += CH) {
        const int srcn = (e0 + CH < e_end) ? load_idx(e0 + CH) : 0;
        if constexpr (U >= 2) {
            if (e_end - e0 <= CH / 2) fwd2_chunk<HD, D, N, U / 2, BF>(A, e0, e_end_v, cp, gidx, srcv, pr, ac2, m, Z, acc);
            else fwd2_chunk<HD, D, N, U, BF>(A, e0, e_end_v, cp, gidx, srcv, pr, ac2, m, Z, acc);
        } else {
            fwd2_chunk<HD, D, N, U, BF>(A, e0, e_end_v, cp, gidx, srcv, pr, ac2, m, Z, acc);
        }
        srcv = srcn;
    }
    // merge the G edge groups (online-softmax combine); afterwards all groups agree
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off), Zo = __shfl_xor(Z, off);
        const vnf<N> ao = shfl_xor_n<N>(acc, off);
        const float mn = fmaxf(m, mo);
        const float s1 = exp2_fast(m - mn), s2 = exp2_fast(mo - mn);
        Z = Z * s1 + Zo * s2;
        acc = acc * s1 + ao * s2;
        m = mn;
    }
    if (slot >= 0) {                                 // one segment of a long row: partial (m, Z, acc)
        if (gidx == 0) {
            *reinterpret_cast<vnf<N>*>(A.part_acc + (int64_t)slot * HD + c) = acc;
            if ((c % D) == 0) {
                A.part_mz[(int64_t)slot * 2 * H + c / D] = m;
                A.part_mz[(int64_t)slot * 2 * H + H + c / D] = Z;
            }
        }
        return;
    }
    const vnf<N> hp = acc * __builtin_amdgcn_rcpf(Z + 1e-8f);    // E:379 epsilon
    if (gidx == 0) {
        *reinterpret_cast<vnf<N>*>(A.hpre + row * HD + c) = hp;
        if ((c % D) == 0) { A.mstat[row * H + c / D] = m; A.zstat[row * H + c / D] = Z; }
    }
    const vnf<N> act = lrelu_n<N>(hp, A.slope);
    if (!A.is_last) {
        if (gidx == 0) *reinterpret_cast<vnf<N>*>(A.hout + row * HD + c) = act;   // concat heads (E:452-457)
    } else {
        vnf<N> t = act;                              // activate, then average heads (E:440-449)
#pragma unroll
        for (int off = DL; off < LPE; off <<= 1) t += shfl_xor_n<N>(t, off);
        if (lane < DL) *reinterpret_cast<vnf<N>*>(A.hout + row * D + c) = t / (float)H;
    }
}

// ------------------------------------------------------------------------------------------------
// GROUP-PER-ROW kernels (training path).  The in-degree law of the benchmark graphs has median 12 / mean 25:
// with one wave per row, the per-row work (descriptor, PR / h_pre / g rows, softmax stats, the cross-group merge,
// the output rows, and three dependent memory latencies) is paid per ~25 edges and was 38 % of the forward and
// 42 % of the backward (tools/shape_probe.py: t = a*N + b*E).  Here a wave carries G = 64/(H*D/N) work items side
// by side — one per lane group (16 lanes at H*D = 64) — so that every per-row instruction serves G rows, a
// gather instruction still moves G rows of PL (one per group), and no cross-group merge exists.  The work list is
// sorted by length (build_worklist), so the G neighbours finish together; the loop runs to the longest of them.
// Same arithmetic per row as the chunked kernels with one edge group; split rows leave the same partials.
// ------------------------------------------------------------------------------------------------
template <int HD, int N>
__device__ __forceinline__ int wave_max_over_groups(int v) {
    constexpr int LPE = HD / N;
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) { const int o = __shfl_xor(v, off); v = v > o ? v : o; }
    return __builtin_amdgcn_readfirstlane(v);
}

template <int HD, int D, int N, bool BF = false>
__global__ __launch_bounds__(256) void edge_fwd3_kernel(EdgeFwdArgs A) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int U = 4;                             // edges per group and step: G*U gathers in flight per wave
    static_assert(D % N == 0 && LPE >= U, "lane layout");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int c = N * cp;                            // first of the lane's channels
    const int64_t it = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * G + gidx;
    int row = -1, b = 0, e = 0, slot = -1;           // group without an item (tail of the list): walks nothing, writes nothing
    if (it < A.n_items) { const int4 item = A.items[it]; row = item.x; b = item.y; e = item.z; slot = item.w; }
    const int64_t rowc = row < 0 ? 0 : row;          // clamped: loads need no predicate
    const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(A.PR + rowc * HD + c);
    const vnf<N> ac2 = *reinterpret_cast<const vnf<N>*>(A.a + c) * kLog2e;
    float m = -1e9f * kLog2e, Z = 0.f;               // E:336 seeds the max with -1e9f
    vnf<N> acc = vzero<N>();
    const int nst = wave_max_over_groups<HD, N>((e - b + U - 1) / U);
    // the step's U edge indices of every group: lanes 0..U-1 of the group load them (clamped into the row, and to
    // edge 0 for an empty row: col_idx always has at least one element), the owning lanes get them by shuffle
    auto load_idx = [&](int st) {
        int j = b + st * U + (cp & (U - 1));
        j = j < e ? j : e - 1;
        return A.col_idx[j > 0 ? j : 0];
    };
    int srcv = load_idx(0);
    for (int st = 0; st < nst; ++st) {
        const int srcn = load_idx(st + 1);           // next step's indices: in flight during this one
        vnf<N> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = gather_row_n<HD, N, BF>(A.PL, __shfl(srcv, gidx * LPE + u), cp);
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, A.slope));
        group_sum_n<DL, U>(t);
        float cm = -INFINITY;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            t[u] = (b + st * U + u < e) ? t[u] : -INFINITY;
            cm = fmaxf(cm, t[u]);
        }
        const float mn = fmaxf(m, cm);
        const float scale = exp2_fast(m - mn);
        Z *= scale;
        acc = acc * scale;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float p = exp2_fast(t[u] - mn);    // 0 for padded slots
            Z += p;
            acc += p * v[u];
        }
        m = mn;
        srcv = srcn;
    }
    if (row < 0) return;
    if (slot >= 0) {                                 // one segment of a long row: partial (m, Z, acc) -> edge_fwd_fix_kernel
        *reinterpret_cast<vnf<N>*>(A.part_acc + (int64_t)slot * HD + c) = acc;
        if ((c % D) == 0) {
            A.part_mz[(int64_t)slot * 2 * H + c / D] = m;
            A.part_mz[(int64_t)slot * 2 * H + H + c / D] = Z;
        }
        return;
    }
    const vnf<N> hp = acc * __builtin_amdgcn_rcpf(Z + 1e-8f);    // E:379 epsilon
    stream_store(reinterpret_cast<vnf<N>*>(A.hpre + (int64_t)row * HD + c), hp);
    if ((c % D) == 0) { A.mstat[(int64_t)row * H + c / D] = m; A.zstat[(int64_t)row * H + c / D] = Z; }
    const vnf<N> act = lrelu_n<N>(hp, A.slope);
    if (!A.is_last) {
        stream_store(reinterpret_cast<vnf<N>*>(A.hout + (int64_t)row * HD + c), act);     // concat heads (E:452-457)
    } else {
        vnf<N> t = act;                              // activate, then average heads (E:440-449)
#pragma unroll
        for (int off = DL; off < LPE; off <<= 1) t += shfl_xor_n<N>(t, off);
        if (cp < DL) *reinterpret_cast<vnf<N>*>(A.hout + (int64_t)row * D + c) = t / (float)H;
    }
}

// STASH (see edge_bwd2_kernel): instead of the H*D-float message row an edge leaves a record of H*D/N words in its
// source-major slot — per head alpha and grad_attn_score, with the N LeakyReLU' decisions of the lane's channels in
// the N low mantissa bits of the word (value rounded to nearest at that precision: relative 2^-(24-N)).
template <int HD, int D, int N, int UU, int DBG, bool BF, bool STASH>
__device__ __forceinline__ void bwd2_chunk(const EdgeBwdArgs& A, int e0, int e_end, int e_end_v, int cp, int gidx,
                                           int srcv, int posv, vnf<N> g, vnf<N> pr, float dot, vnf<N> ac, vnf<N> acs,
                                           vnf<N> ac2, float m2, float inv, vnf<N>& ga, vnf<N>& gpr) {
    constexpr int LPE = HD / N;         // lanes per edge
    constexpr int G = 64 / LPE;         // edges per wave-instruction
    constexpr int DL = D / N;           // lanes per head
    vnf<N> v[UU];
    int sid[UU];
    // srcv / posv: the chunk's <= 16 edge indices, lane k holding edge e0+k (loaded by the caller one chunk ahead,
    // with ONE coalesced load each), handed to the owning lanes through the LDS crossbar: per-lane index loads
    // would put a second memory latency in front of every gather
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        v[u] = gather_row_n<HD, N, BF>(A.PL, __shfl(srcv, u * G + gidx), cp);
        if constexpr (!STASH) sid[u] = __shfl(posv, u * G + gidx);     // STASH: fetched at the store (16 fewer live VGPRs: 4 waves/SIMD)
    }
    constexpr int PMAX = STASH ? 4 : 8;      // slots per compute pass (STASH: 4 keeps the kernel at 128 VGPRs = 4 waves/SIMD)
    constexpr int P = UU > PMAX ? PMAX : UU;
#pragma unroll
    for (int p0 = 0; p0 < UU; p0 += P) {
        float al[P], ga_[P];
        [[maybe_unused]] uint32_t slots[P];
        if constexpr (STASH) {                       // the pass's record slots, fetched together ahead of their use
#pragma unroll
            for (int q = 0; q < P; ++q) slots[q] = DBG == 2 ? (uint32_t)(e0 + (p0 + q) * G + gidx) : (uint32_t)__shfl(posv, (p0 + q) * G + gidx);
        }
#pragma unroll
        for (int q = 0; q < P; ++q) al[q] = hsum<N>(ac2 * lrelu_n<N>(v[p0 + q] + pr, A.slope));
        group_sum_n<DL, P>(al);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            al[q] = exp2_fast(al[q] - m2) * inv;
            ga_[q] = hsum<N>(g * v[p0 + q]);
        }
        group_sum_n<DL, P>(ga_);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int u = p0 + q;
            const int j = e0 + u * G + gidx;
            const bool valid = j < e_end_v;
            const float ge = valid ? al[q] * (ga_[q] - dot) : 0.f;   // padded slots contribute nothing
            const vnf<N> s = v[u] + pr;
            const vnf<N> gs = ge * select_pos<N>(s, ac, acs);        // ge * a * LReLU'(s)
            ga += ge * lrelu_n<N>(s, A.slope);
            gpr += gs;
            if constexpr (STASH) {
                static_assert(D / N == 2, "stash records: two lanes per head (one carries alpha, the other ge)");
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) bits |= (s[i] > 0.f ? 1u : 0u) << i;
                const uint32_t w = __builtin_bit_cast(uint32_t, (cp & 1) ? ge : al[q]);
                const uint32_t word = ((w + (1u << (N - 1))) & ~((1u << N) - 1u)) | bits;
                // straight-line: padded lanes write the spare record behind the last slot instead of being masked off
                // (an exec-masked store is a branch, and hipcc drains the memory counters at its join point)
                const uint32_t slot = valid ? slots[q] : A.stash_spare;
                if constexpr (DBG != 1) A.stash[(uint64_t)slot * LPE + cp] = word;
            } else {
                const vnf<N> msg = g * al[q] + gs;                   // d/dPL[src] from this edge
                if (valid && DBG != 1) store_row_n<HD, N, BF>(A.msg, sid[u], cp, msg);
            }
        }
    }
}

// STASH = true is the training path wherever D/N == 2 (D = 8 with four channels per lane, D = 4 with two): the
// per-edge record is H*D/N words (64 B at H*D = 64) instead of the H*D*4-byte message row, and gpl_pull_kernel
// (gat_csc.hip) rebuilds each message from the record and ONE gathered row of g[dst] while it sums per source.
// This kernel then also writes g[row] (dL/dh_pre with the LeakyReLU' factor applied) for that gather.
template <int HD, int D, int N, int DBG, bool BF, bool STASH>
__device__ __forceinline__ void edge_bwd2_body(const EdgeBwdArgs& A) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int U = 16 / G;
    constexpr int CH = 16;
    static_assert(D % N == 0 && U >= 1, "lane layout");
    __shared__ float red[4][HD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int c = N * cp;                                        // first of the lane's channels
    const vnf<N> ac = *reinterpret_cast<const vnf<N>*>(A.a + c);
    const vnf<N> acs = ac * A.slope;
    const vnf<N> ac2 = ac * kLog2e;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    vnf<N> ga = vzero<N>();
    vnf<N> one, slp;
#pragma unroll
    for (int i = 0; i < N; ++i) { one[i] = 1.0f; slp[i] = A.slope; }

    for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < A.n_items; it += nwaves) {
        const int4 item = A.items[it];
        const int64_t row = item.x;
        const int b = item.y, e_end = item.z, slot = item.w;
        const int e_end_v = per_lane(e_end);
        const vnf<N> hp = *reinterpret_cast<const vnf<N>*>(A.hpre + row * HD + c);
        const vnf<N> dsel = select_pos<N>(hp, one, slp);
        vnf<N> g;
        if (A.gh != nullptr) g = *reinterpret_cast<const vnf<N>*>(A.gh + row * A.gh_stride + (c % D)) * dsel * (1.0f / (float)H);   // E:598-603
        else {
            g = *reinterpret_cast<const vnf<N>*>(A.g + row * HD + c);
            if (A.g_raw) g = g * dsel;                           // E:888-892 applied by the consumer
        }
        const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(A.PR + row * HD + c);
        const float dot = group_sum<DL>(hsum<N>(g * hp));
        const float m2 = A.mstat[row * H + c / D];
        const float inv = __builtin_amdgcn_rcpf(A.zstat[row * H + c / D] + 1e-8f);
        vnf<N> gpr = vzero<N>();
        if constexpr (STASH) {                                       // one writer per row: whole rows, or a split row's first segment
            if (gidx == 0 && (slot < 0 || b == A.row_ptr[row])) {
                if (A.hbits != nullptr) {                                // last layer: decisions only (see EdgeBwdArgs::hbits)
                    uint32_t nib = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) nib |= (hp[i] > 0.f ? 1u : 0u) << i;
                    A.hbits[row * A.hb_stride + cp] = (uint8_t)nib;
                } else {
                    *reinterpret_cast<vnf<N>*>(A.gfull + row * HD + c) = g;
                }
            }
        }
        // edge indices of a chunk: lane k < 16 holds edge e0+k (clamped into the item: no predicate needed)
        auto load_idx = [&](int e0, int& srcv, int& posv) {
            const int jl = e0 + (lane & (CH - 1));
            const int jlc = jl < e_end ? jl : e_end - 1;
            srcv = A.col_idx[jlc];
            posv = (DBG == 2) ? jlc : A.pos[jlc];
        };
        int srcv = 0, posv = 0;
        if (b < e_end) load_idx(b, srcv, posv);                      // empty item: nothing to prefetch (e_end - 1 < b)
        for (int e0 = b; e0 < e_end; e0 += CH) {
            int srcn = 0, posn = 0;
            if (e0 + CH < e_end) load_idx(e0 + CH, srcn, posn);      // next chunk's indices: in flight during this one
            if constexpr (U >= 2) {
                if (e_end - e0 <= CH / 2) bwd2_chunk<HD, D, N, U / 2, DBG, BF, STASH>(A, e0, e_end, e_end_v, cp, gidx, srcv, posv, g, pr, dot, ac, acs, ac2, m2, inv, ga, gpr);
                else bwd2_chunk<HD, D, N, U, DBG, BF, STASH>(A, e0, e_end, e_end_v, cp, gidx, srcv, posv, g, pr, dot, ac, acs, ac2, m2, inv, ga, gpr);
            } else {
                bwd2_chunk<HD, D, N, U, DBG, BF, STASH>(A, e0, e_end, e_end_v, cp, gidx, srcv, posv, g, pr, dot, ac, acs, ac2, m2, inv, ga, gpr);
            }
            srcv = srcn; posv = posn;
        }
#pragma unroll
        for (int off = LPE; off < 64; off <<= 1) gpr += shfl_xor_n<N>(gpr, off);
        if (gidx == 0) {
            float* dst = slot < 0 ? A.gPR + row * HD + c : A.part_acc + (int64_t)slot * HD + c;   // segment partial -> fix kernel
            *reinterpret_cast<vnf<N>*>(dst) = gpr;
        }
    }
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) ga += shfl_xor_n<N>(ga, off);
    if (gidx == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[wave][c + i] = ga[i];
    }
    __syncthreads();
    if (threadIdx.x < HD)
        A.ga_partial[(int64_t)blockIdx.x * HD + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <int HD, int D, int N, int DBG = 0, bool BF = false, bool STASH = false>
__global__ __launch_bounds__(256) void edge_bwd2_kernel(EdgeBwdArgs A) { edge_bwd2_body<HD, D, N, DBG, BF, STASH>(A); }
// The stash variant sits a few registers above 128 VGPRs when left alone (3 waves/SIMD); it is told to fit 4 waves.
template <int HD, int D, int N, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void edge_bwd2s_kernel(EdgeBwdArgs A) {
    edge_bwd2_body<HD, D, N, DBG, false, true>(A);
}

// Group-per-row backward with per-edge records (see edge_fwd3_kernel and the STASH note at bwd2_chunk): persistent
// waves take quads of G work items (static round-robin over the length-sorted list: every wave gets the same mix).
// MSG = true: the same walk for the layers WITHOUT a record path (bf16 storage at H*D < 64: BASELINE config 5; GAT_BWD_STASH=0):
// an edge leaves its H*D-element message row g*alpha + ge*a*LReLU' in its source-major slot (summed by launch_gpl_sum) instead
// of a record, nothing is written for the pull pass (no gfull / decision bytes), any D/N lanes per head.
template <int HD, int D, int N, int DBG = 0, bool BF = false, bool MSG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void edge_bwd3_kernel(EdgeBwdArgs A) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int U = 4;
    static_assert((MSG || DL == 2) && LPE >= U, "records: two lanes per head (one carries alpha, the other ge)");
    __shared__ float red[4][HD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int c = N * cp;
    const vnf<N> ac = *reinterpret_cast<const vnf<N>*>(A.a + c);
    const vnf<N> acs = ac * A.slope;
    const vnf<N> ac2 = ac * kLog2e;
    const int64_t nquads = (A.n_items + G - 1) / G;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    vnf<N> ga = vzero<N>();

    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nquads; q += nwaves) {
        const int64_t it = q * G + gidx;
        int row = -1, b = 0, e = 0, slot = -1;       // group without an item: walks nothing, writes nothing
        if (it < A.n_items) { const int4 item = A.items[it]; row = item.x; b = item.y; e = item.z; slot = item.w; }
        const int64_t rowc = row < 0 ? 0 : row;      // clamped: loads need no predicate
        // the step's U edge indices of the group: lanes 0..U-1 of the group load them (clamped into the row; an empty
        // row reads edge 0: col_idx / pos always hold at least one element), the owning lanes get them by shuffle
        auto load_idx = [&](int st, int& srcv, int& posv) {
            int j = b + st * U + (cp & (U - 1));
            j = j < e ? j : e - 1;
            j = j > 0 ? j : 0;
            srcv = A.col_idx[j];
            posv = (DBG == 2) ? j : A.pos[j];
            // timing experiments (wrong results): K write FRONTS, each advancing sequentially with the CSR index — the store pattern of
            // records laid out by (source block, CSR index): front = slot mod K (a pseudo-random block), place = front * E/K + j / K
            if constexpr (DBG >= 15 && DBG <= 18) {
                constexpr uint32_t K = DBG == 15 ? 64u : DBG == 16 ? 512u : DBG == 17 ? 4096u : 32768u;
                const uint32_t per = A.stash_spare / K;
                posv = (int)(((uint32_t)posv % K) * per + (uint32_t)j / K);
            }
        };
        auto spread = [&](int st, int srcv, int posv, int (&src)[U], uint32_t (&sl)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                src[u] = __shfl(srcv, gidx * LPE + u);
                uint32_t p = (uint32_t)__shfl(posv, gidx * LPE + u);
                // timing experiments (wrong results): the same scattered stores folded into a window of 2^26 / 2^24 / 2^22 / 2^20 bytes
                // — is it the SPAN of the scatter (address translation reach, DRAM page locality) that costs, or the scatter as such?
                if constexpr (DBG == 10) p &= (1u << 20) - 1u;
                if constexpr (DBG == 11) p &= (1u << 18) - 1u;
                if constexpr (DBG == 12) p &= (1u << 16) - 1u;
                if constexpr (DBG == 13) p &= (1u << 14) - 1u;
                if constexpr (DBG == 14) p &= (1u << 24) - 1u;
                sl[u] = (b + st * U + u < e) ? p : A.stash_spare;    // padded lanes store to the spare record (no exec-masked store)
            }
        };
        int srcv, posv, srcn, posn;
        load_idx(0, srcv, posv);
        load_idx(1, srcn, posn);
        const vnf<N> hp = *reinterpret_cast<const vnf<N>*>(A.hpre + rowc * HD + c);
        vnf<N> dsel;
#pragma unroll
        for (int i = 0; i < N; ++i) dsel[i] = hp[i] > 0.f ? 1.0f : A.slope;
        vnf<N> g;
        if (A.gh != nullptr) g = *reinterpret_cast<const vnf<N>*>(A.gh + rowc * A.gh_stride + (c % D)) * dsel * (1.0f / (float)H);   // E:598-603
        else {
            g = *reinterpret_cast<const vnf<N>*>(A.g + rowc * HD + c);
            if (A.g_raw) g = g * dsel;               // E:888-892 applied by the consumer
        }
        const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(A.PR + rowc * HD + c);
        const float dot = group_sum<DL>(hsum<N>(g * hp));
        const float m2 = A.mstat[rowc * H + c / D];
        const float inv = __builtin_amdgcn_rcpf(A.zstat[rowc * H + c / D] + 1e-8f);
        if (!MSG && row >= 0 && (slot < 0 || b == A.row_ptr[rowc])) {    // one writer per row: whole rows, or a split row's first segment
            if (A.hbits != nullptr) {                                // last layer: the decisions only (g = gh * LReLU'(h_pre) / H is rebuilt by the pull pass)
                uint32_t nib = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) nib |= (hp[i] > 0.f ? 1u : 0u) << i;
                A.hbits[rowc * A.hb_stride + cp] = (uint8_t)nib;
            } else {
                if constexpr (BF) store_row_n<HD, N, BF>(A.gfull, (int)rowc, cp, g);   // bf16 storage: the gathered g table is 2-byte too
                else stream_store(reinterpret_cast<vnf<N>*>(A.gfull + rowc * HD + c), g);
            }
        }
        vnf<N> gpr = vzero<N>();
        const int nst = wave_max_over_groups<HD, N>((e - b + U - 1) / U);
        if constexpr (DBG == 9) {
            // Timing experiment (results unchanged): walk the row's source rows ONCE MORE before the real walk, as a forward pass
            // fused into this kernel would — does the second gather of a row, a few microseconds after the first, come out of
            // the L2 / Infinity Cache fast enough to pay for fusing the last layer's forward, head and backward per row?
            float dummy = 0.f;
            int j0 = b + (cp & (U - 1));
            j0 = j0 < e ? j0 : e - 1;
            int sv = A.col_idx[j0 > 0 ? j0 : 0];
            for (int st = 0; st < nst; ++st) {
                int jn = b + (st + 1) * U + (cp & (U - 1));
                jn = jn < e ? jn : e - 1;
                const int sn = A.col_idx[jn > 0 ? jn : 0];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const vnf<N> vv = gather_row_n<HD, N, BF>(A.PL, __shfl(sv, gidx * LPE + u), cp);
                    dummy += hsum<N>(vv);
                }
                sv = sn;
            }
            if (dummy == 12345.6789f) gpr[0] += 1.0f;          // never true in practice: keeps the walk alive
        }
        // Software pipeline.  vmcnt retires loads AND stores in issue order, so a store issued ahead of a gather makes the
        // gather's consumer wait for the store's acknowledgement as well (measured: without its record stores the kernel
        // ran 2x faster, wherever the stores went).  Per step, in this order: gathers of the step; the PREVIOUS step's
        // records; shuffles of the next step's indices (loaded two steps ago: older than everything above); index loads
        // two steps ahead; only then the wait for the gathers, with the stores and index loads still in flight.
        int src[U];
        uint32_t sl[U], pend_s[U], pend_w[U];
        [[maybe_unused]] vnf<N> pend_m[U];
        spread(0, srcv, posv, src, sl);
#pragma unroll
        for (int u = 0; u < U; ++u) { pend_w[u] = 0u; pend_s[u] = A.stash_spare; if constexpr (MSG) pend_m[u] = vzero<N>(); }
        for (int st = 0; st < nst; ++st) {
            vnf<N> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = gather_row_n<HD, N, BF>(A.PL, src[u], cp);
            if constexpr (MSG) {                     // the previous step's message rows (spare row E for padded lanes)
                if constexpr (DBG != 1) {            // (experiment library, GAT_DBG=1: the walk without its row stores — attribution of the reads)
#pragma unroll
                    for (int u = 0; u < U; ++u) store_row_n<HD, N, BF>(A.msg, (int)pend_s[u], cp, pend_m[u]);
                }
            } else if constexpr (DBG == 3) {                // timing experiment: the step's bytes as ONE 16-B-per-lane store, CSR order
                const int j0 = b + (st > 0 ? st - 1 : 0) * U;
                uint4 w4 = make_uint4(pend_w[0], pend_w[1], pend_w[2], pend_w[3]);
                *reinterpret_cast<uint4*>(A.stash + (uint64_t)(uint32_t)(j0 < e ? j0 : 0) * LPE + cp * 4) = w4;
            } else if constexpr (DBG != 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) stream_store(&A.stash[(uint64_t)pend_s[u] * LPE + cp], pend_w[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) pend_s[u] = sl[u];
            spread(st + 1, srcn, posn, src, sl);
            load_idx(st + 2, srcn, posn);
            float al[U], ga_[U];
#pragma unroll
            for (int u = 0; u < U; ++u) al[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, A.slope));
            group_sum_n<DL, U>(al);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                al[u] = exp2_fast(al[u] - m2) * inv;
                ga_[u] = hsum<N>(g * v[u]);
            }
            group_sum_n<DL, U>(ga_);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = b + st * U + u < e;
                const float ge = valid ? al[u] * (ga_[u] - dot) : 0.f;   // padded slots contribute nothing
                const vnf<N> s = v[u] + pr;
                const vnf<N> gs = ge * select_pos<N>(s, ac, acs);        // ge * a * LReLU'(s)
                ga += ge * lrelu_n<N>(s, A.slope);
                gpr += gs;
                if constexpr (MSG) {
                    pend_m[u] = g * al[u] + gs;                          // d/dPL[src] from this edge (E:859-869)
                } else {
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) bits |= (s[i] > 0.f ? 1u : 0u) << i;
                    const uint32_t w = __builtin_bit_cast(uint32_t, (cp & 1) ? ge : al[u]);
                    pend_w[u] = ((w + (1u << (N - 1))) & ~((1u << N) - 1u)) | bits;
                }
            }
        }
        if constexpr (MSG) {
            if constexpr (DBG != 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) store_row_n<HD, N, BF>(A.msg, (int)pend_s[u], cp, pend_m[u]);          // the last step's rows
            }
        } else if constexpr (DBG != 1 && DBG != 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) stream_store(&A.stash[(uint64_t)pend_s[u] * LPE + cp], pend_w[u]);      // the last step's records
        }
        if (row >= 0) {
            float* dst = slot < 0 ? A.gPR + rowc * HD + c : A.part_acc + (int64_t)slot * HD + c;   // segment partial -> fix kernel
            stream_store(reinterpret_cast<vnf<N>*>(dst), gpr);
        }
    }
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) ga += shfl_xor_n<N>(ga, off);
    if (gidx == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[wave][c + i] = ga[i];
    }
    __syncthreads();
    if (threadIdx.x < HD)
        A.ga_partial[(int64_t)blockIdx.x * HD + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// LAST LAYER, FUSED PER ROW (gat_step, H*D = 64, D = 8, fp32): forward edge pass, output head and backward edge pass of a
// destination row in ONE kernel.  For the last layer the whole chain is row-local — h_pre[d] -> H[d] (mean over heads, E:440-449)
// -> z, y, dz of node d (E:463-512, 571-573) -> gH[d] = Wo^T dz -> g[d] = gH * LReLU'(h_pre) / H (E:598-603) -> the row's
// alpha, grad_attn_coeff, grad_attn_score, records, gPR, grad_a terms — so the row's source rows PL[src] can be walked twice
// within a few microseconds: the second walk is served by the L2 / the 256 MiB Infinity Cache instead of HBM (measured with
// GAT_DBG=9: a second walk inside the backward costs 1.35 ms per launch, the stand-alone forward 2.37).  Body = edge_fwd3_kernel's
// walk, a 16-lane head (classes cp, cp+16, ...; W_o in LDS), edge_bwd3_kernel's walk; same arithmetic per row as those kernels.
// Whole rows only: the segments of split (hub) rows keep the three-kernel path.  Loss, #correct and grad_Wo come from
// head_step_kernel on the stored H (as before, without its gH output).
// MEASURED, NOT THE DEFAULT (GAT_FUSE_LAST=1 enables; DESIGN §4 "Round 3"): at the backward's 4 waves per SIMD and with the head in
// the middle, the fused launch takes as long as the separate passes for the rows it covers (5.03 ms for 76 % of the Products
// shape's edges), and the split rows' launches lose the cover of the bulk: 20.95 vs 20.78 ms per step.  Parity-tested.
template <int HD, int D, int N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void edge_last_fused_kernel(EdgeLastArgs A) {
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int U = 4, JC = 4;                         // JC * LPE >= the 64 classes W_o may have here
    static_assert(LPE == 16 && DL == 2 && D == 8 && N == 4, "16 lanes per row, two lanes per head, D = 8");
    __shared__ float red[4][HD];
    __shared__ float s_wo[64 * D];
    const EdgeFwdArgs& F = A.f;
    const EdgeBwdArgs& B = A.b;
    const int C = A.C;
    for (int i = threadIdx.x; i < C * D; i += blockDim.x) s_wo[i] = A.Wo[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cp = lane % LPE, gidx = lane / LPE;
    const int c = N * cp;
    const vnf<N> ac = *reinterpret_cast<const vnf<N>*>(F.a + c);
    const vnf<N> acs = ac * F.slope;
    const vnf<N> ac2 = ac * kLog2e;
    const int64_t nquads = (F.n_items + G - 1) / G;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    vnf<N> ga = vzero<N>();

    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nquads; q += nwaves) {
        const int64_t it = q * G + gidx;
        int row = -1, b = 0, e = 0;                   // group without an item: walks nothing, writes nothing
        if (it < F.n_items) { const int4 item = F.items[it]; row = item.x; b = item.y; e = item.z; }
        const int64_t rowc = row < 0 ? 0 : row;
        const int nst = wave_max_over_groups<HD, N>((e - b + U - 1) / U);
        const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(F.PR + rowc * HD + c);
        // ---- forward walk (edge_fwd3_kernel) ----
        float m = -1e9f * kLog2e, Z = 0.f;
        vnf<N> acc = vzero<N>();
        {
            auto load_src = [&](int st) {
                int j = b + st * U + (cp & (U - 1));
                j = j < e ? j : e - 1;
                return F.col_idx[j > 0 ? j : 0];
            };
            int srcv = load_src(0);
            for (int st = 0; st < nst; ++st) {
                const int srcn = load_src(st + 1);
                vnf<N> v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = gather_row_n<HD, N, false>(F.PL, __shfl(srcv, gidx * LPE + u), cp);
                float t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, F.slope));
                group_sum_n<DL, U>(t);
                float cm = -INFINITY;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    t[u] = (b + st * U + u < e) ? t[u] : -INFINITY;
                    cm = fmaxf(cm, t[u]);
                }
                const float mn = fmaxf(m, cm);
                const float scale = exp2_fast(m - mn);
                Z *= scale;
                acc = acc * scale;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float p = exp2_fast(t[u] - mn);
                    Z += p;
                    acc += p * v[u];
                }
                m = mn;
                srcv = srcn;
            }
        }
        const float inv = __builtin_amdgcn_rcpf(Z + 1e-8f);       // E:379 epsilon
        const vnf<N> hp = acc * inv;
        if (row >= 0) {
            stream_store(reinterpret_cast<vnf<N>*>(F.hpre + rowc * HD + c), hp);
            if ((c % D) == 0) { F.mstat[rowc * H + c / D] = m; F.zstat[rowc * H + c / D] = Z; }
        }
        // H[d] = mean over heads of LReLU(h_pre) (E:440-449): lanes of equal parity hold the same four dims afterwards
        vnf<N> hm = lrelu_n<N>(hp, F.slope);
#pragma unroll
        for (int off = DL; off < LPE; off <<= 1) hm += shfl_xor_n<N>(hm, off);
        hm = hm / (float)H;
        if (row >= 0 && cp < DL) *reinterpret_cast<vnf<N>*>(F.hout + rowc * D + c) = hm;
        const vnf<N> ho_other = shfl_xor_n<N>(hm, 1);
        float ho[D];
#pragma unroll
        for (int k = 0; k < N; ++k) { ho[k] = (cp & 1) ? ho_other[k] : hm[k]; ho[N + k] = (cp & 1) ? hm[k] : ho_other[k]; }
        // ---- output head of this node (E:463-512), its dz (E:571-573) and gH = Wo^T dz: classes cp, cp + 16, ... ----
        const int lab = A.labels[rowc];                  // < 0: outside the training mask (gat_set_train_mask): dz = 0
        float z[JC], mv = -INFINITY;
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int cls = cp + LPE * j;
            z[j] = -INFINITY;
            if (cls < C) {
                float t = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) t += s_wo[cls * D + d] * ho[d];
                z[j] = t;
                mv = fmaxf(mv, t);
            }
        }
#pragma unroll
        for (int off = 1; off < LPE; off <<= 1) mv = fmaxf(mv, __shfl_xor(mv, off));
        float ev[JC], sum = 0.f;
#pragma unroll
        for (int j = 0; j < JC; ++j) { ev[j] = (cp + LPE * j < C) ? exp2_fast((z[j] - mv) * kLog2e) : 0.f; sum += ev[j]; }
        sum = group_sum<LPE>(sum);
        const float rden = 1.0f / (sum + 1e-8f);
        float gh[D];
#pragma unroll
        for (int d = 0; d < D; ++d) gh[d] = 0.f;
#pragma unroll
        for (int j = 0; j < JC; ++j) {
            const int cls = cp + LPE * j;
            if (cls < C) {
                const float dz = lab >= 0 ? ev[j] * rden - (cls == lab ? 1.0f : 0.0f) : 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) gh[d] += s_wo[cls * D + d] * dz;
            }
        }
        group_sum_n<LPE, D>(gh);
        vnf<N> ghl;                                      // the lane's four dims of gH
#pragma unroll
        for (int k = 0; k < N; ++k) ghl[k] = (cp & 1) ? gh[N + k] : gh[k];
        if (row >= 0 && cp < DL) *reinterpret_cast<vnf<N>*>(A.gh_out + rowc * B.gh_stride + c) = ghl;   // the pull pass's node record
        vnf<N> dsel;
#pragma unroll
        for (int i = 0; i < N; ++i) dsel[i] = hp[i] > 0.f ? 1.0f : F.slope;
        const vnf<N> g = ghl * dsel * (1.0f / (float)H);                                                    // E:598-603
        if (row >= 0) {
            if (B.hbits != nullptr) {
                uint32_t nib = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) nib |= (hp[i] > 0.f ? 1u : 0u) << i;
                B.hbits[rowc * B.hb_stride + cp] = (uint8_t)nib;
            } else {
                stream_store(reinterpret_cast<vnf<N>*>(B.gfull + rowc * HD + c), g);
            }
        }
        // ---- backward walk (edge_bwd3_kernel) ----
        const float dot = group_sum<DL>(hsum<N>(g * hp));
        const float m2 = m;
        auto load_idx = [&](int st, int& srcv, int& posv) {
            int j = b + st * U + (cp & (U - 1));
            j = j < e ? j : e - 1;
            j = j > 0 ? j : 0;
            srcv = B.col_idx[j];
            posv = B.pos[j];
        };
        auto spread = [&](int st, int srcv, int posv, int (&src)[U], uint32_t (&sl)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                src[u] = __shfl(srcv, gidx * LPE + u);
                const uint32_t p = (uint32_t)__shfl(posv, gidx * LPE + u);
                sl[u] = (b + st * U + u < e) ? p : B.stash_spare;
            }
        };
        int srcv, posv, srcn, posn;
        load_idx(0, srcv, posv);
        load_idx(1, srcn, posn);
        vnf<N> gpr = vzero<N>();
        int src[U];
        uint32_t sl[U], pend_s[U], pend_w[U];
        spread(0, srcv, posv, src, sl);
#pragma unroll
        for (int u = 0; u < U; ++u) { pend_w[u] = 0u; pend_s[u] = B.stash_spare; }
        for (int st = 0; st < nst; ++st) {
            vnf<N> v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = gather_row_n<HD, N, false>(B.PL, src[u], cp);
#pragma unroll
            for (int u = 0; u < U; ++u) stream_store(&B.stash[(uint64_t)pend_s[u] * LPE + cp], pend_w[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) pend_s[u] = sl[u];
            spread(st + 1, srcn, posn, src, sl);
            load_idx(st + 2, srcn, posn);
            float al[U], ga_[U];
#pragma unroll
            for (int u = 0; u < U; ++u) al[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, F.slope));
            group_sum_n<DL, U>(al);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                al[u] = exp2_fast(al[u] - m2) * inv;
                ga_[u] = hsum<N>(g * v[u]);
            }
            group_sum_n<DL, U>(ga_);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = b + st * U + u < e;
                const float ge = valid ? al[u] * (ga_[u] - dot) : 0.f;
                const vnf<N> s = v[u] + pr;
                const vnf<N> gs = ge * select_pos<N>(s, ac, acs);
                ga += ge * lrelu_n<N>(s, F.slope);
                gpr += gs;
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) bits |= (s[i] > 0.f ? 1u : 0u) << i;
                const uint32_t w = __builtin_bit_cast(uint32_t, (cp & 1) ? ge : al[u]);
                pend_w[u] = ((w + (1u << (N - 1))) & ~((1u << N) - 1u)) | bits;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) stream_store(&B.stash[(uint64_t)pend_s[u] * LPE + cp], pend_w[u]);
        if (row >= 0) stream_store(reinterpret_cast<vnf<N>*>(B.gPR + rowc * HD + c), gpr);
    }
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) ga += shfl_xor_n<N>(ga, off);
    if (gidx == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[wave][c + i] = ga[i];
    }
    __syncthreads();
    if (threadIdx.x < HD)
        B.ga_partial[(int64_t)blockIdx.x * HD + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// gH = Wo^T dz (and nothing else) of the SPLIT rows, whose forward runs as segments + fix-up: one thread per split row, the
// head kernels' arithmetic (E:463-512, 571-573)
__global__ __launch_bounds__(256) void head_rows_kernel(const int4* __restrict__ slot_info, int32_t n_slots, int32_t n_split,
                                                       const float* __restrict__ Wo, const float* __restrict__ HL,
                                                       const int32_t* __restrict__ labels, float* __restrict__ gh_out,
                                                       int32_t gh_stride, int32_t C, int32_t DLAST) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_split) return;
    const int64_t row = slot_info[slot_info[n_slots + k].x].x;
    const float* x = HL + row * DLAST;
    const int lab = labels[row];
    float mv = -INFINITY;
    for (int c = 0; c < C; ++c) {
        float t = 0.f;
        for (int d = 0; d < DLAST; ++d) t += Wo[c * DLAST + d] * x[d];
        mv = fmaxf(mv, t);
    }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
        float t = 0.f;
        for (int d = 0; d < DLAST; ++d) t += Wo[c * DLAST + d] * x[d];
        sum += exp2_fast((t - mv) * kLog2e);
    }
    const double rden = 1.0 / ((double)sum + 1e-8);
    float acc[16];
    for (int d = 0; d < DLAST; ++d) acc[d] = 0.f;
    for (int c = 0; c < C; ++c) {
        float t = 0.f;
        for (int d = 0; d < DLAST; ++d) t += Wo[c * DLAST + d] * x[d];
        const float y = (float)((double)exp2_fast((t - mv) * kLog2e) * rden);
        const float dz = lab >= 0 ? y - (c == lab ? 1.0f : 0.0f) : 0.0f;
        for (int d = 0; d < DLAST; ++d) acc[d] += Wo[c * DLAST + d] * dz;
    }
    for (int d = 0; d < DLAST; ++d) gh_out[row * gh_stride + d] = acc[d];
}

#ifdef GAT_EXPERIMENTS      // measured, not ahead (DESIGN §4 "Round 3"): in the experiment library only (make experiments)
// ------------------------------------------------------------------------------------------------
// EXPERIMENT (GAT_DBG=4, H*D = 64, D = 8, fp32): wave-specialised record stores.
// Blocks of NCW compute waves + ONE storing wave.  A compute wave runs the step body of edge_bwd3_kernel but hands the
// step's 16 records (+ their slots) to the storing wave through a ring in LDS instead of issuing global stores, so that its
// vmcnt queue holds loads only; the storing wave drains the rings with ONE global_store_dwordx4 per step (4 lanes per
// 64-byte record, 16 records per instruction, instead of four dword stores of 4 records each).  Hand-off: per compute wave
// a head word (steps published) and a tail word (steps drained) in LDS; LDS executes a CU's instructions in order, so a
// reader that sees the new head sees the data written before it, and a tail written after the data reads were issued
// cannot be overtaken by the producer's next writes.  Every poll loop is bounded (no grid can hang on a lost update).
// Results are bitwise those of edge_bwd3_kernel (same arithmetic, same records).
template <int NCW, int R>
struct RecRing {
    uint32_t rec[NCW][R][16 * 16];      // 16 records of 16 words per step
    uint32_t slot[NCW][R][16];
    int head[NCW], tail[NCW], done[NCW];
};
// POLICY: cache policy of the record stores — 0 plain; 1 `sc1` (write-through, the line is not kept in the XCD's L2:
// MI355X_MICROARCH.md, stores of each flavour); 2 `sc0 sc1`; 3 `nt`.  A/B of whether partial-line (64 of 128 bytes) record
// writes cost fills / write-allocation in L2.
template <int NCW, int R, int POLICY = 0>
__global__ __launch_bounds__((NCW + 1) * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void edge_bwd4_kernel(EdgeBwdArgs A) {
    constexpr int HD = 64, D = 8, N = 4;
    constexpr int LPE = HD / N, G = 64 / LPE, DL = D / N, H = HD / D;
    constexpr int U = 4;
    __shared__ float red[NCW][HD];
    __shared__ RecRing<NCW, R> ring;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NCW) { ring.head[threadIdx.x] = 0; ring.tail[threadIdx.x] = 0; ring.done[threadIdx.x] = 0; }
    __syncthreads();
    constexpr int kSpinMax = 1 << 20;
    if (wave == NCW) {                               // ---- the storing wave ----
        int drained[NCW];
#pragma unroll
        for (int w = 0; w < NCW; ++w) drained[w] = 0;
        int idle = 0;
        for (;;) {
            bool any = false, all_done = true;
#pragma unroll
            for (int w = 0; w < NCW; ++w) {
                const int fin = __hip_atomic_load(&ring.done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int hd = __hip_atomic_load(&ring.head[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                if (hd > drained[w]) {
                    const int rs = drained[w] % R;
                    const uint4 v = *reinterpret_cast<const uint4*>(&ring.rec[w][rs][lane * 4]);    // record lane/4, quarter lane%4
                    const uint32_t sl = ring.slot[w][rs][lane >> 2];
                    asm volatile("" ::: "memory");
                    ++drained[w];
                    __hip_atomic_store(&ring.tail[w], drained[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    uint32_t* dstp = A.stash + (uint64_t)sl * LPE + (lane & 3) * 4;
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const u4 vv = {v.x, v.y, v.z, v.w};
                    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dstp), "v"(vv) : "memory");
                    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dstp), "v"(vv) : "memory");
                    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dstp), "v"(vv) : "memory");
                    else *reinterpret_cast<uint4*>(dstp) = v;
                    any = true;
                }
                if (!fin || hd > drained[w]) all_done = false;
            }
            if (all_done) break;
            if (!any) { __builtin_amdgcn_s_sleep(2); if (++idle > kSpinMax) break; } else idle = 0;
        }
    } else {                                         // ---- compute waves ----
        const int cp = lane % LPE, gidx = lane / LPE;
        const int c = N * cp;
        const vnf<N> ac = *reinterpret_cast<const vnf<N>*>(A.a + c);
        const vnf<N> acs = ac * A.slope;
        const vnf<N> ac2 = ac * kLog2e;
        const int64_t nquads = (A.n_items + G - 1) / G;
        const int64_t nwaves = (int64_t)gridDim.x * NCW;
        vnf<N> ga = vzero<N>();
        int published = 0, tail_seen = 0;
        for (int64_t q = (int64_t)blockIdx.x * NCW + wave; q < nquads; q += nwaves) {
            const int64_t it = q * G + gidx;
            int row = -1, b = 0, e = 0, slot = -1;
            if (it < A.n_items) { const int4 item = A.items[it]; row = item.x; b = item.y; e = item.z; slot = item.w; }
            const int64_t rowc = row < 0 ? 0 : row;
            auto load_idx = [&](int st, int& srcv, int& posv) {
                int j = b + st * U + (cp & (U - 1));
                j = j < e ? j : e - 1;
                j = j > 0 ? j : 0;
                srcv = A.col_idx[j];
                posv = A.pos[j];
            };
            auto spread = [&](int st, int srcv, int posv, int (&src)[U], uint32_t (&sl)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    src[u] = __shfl(srcv, gidx * LPE + u);
                    const uint32_t p = (uint32_t)__shfl(posv, gidx * LPE + u);
                    sl[u] = (b + st * U + u < e) ? p : A.stash_spare;
                }
            };
            int srcv, posv, srcn, posn;
            load_idx(0, srcv, posv);
            load_idx(1, srcn, posn);
            const vnf<N> hp = *reinterpret_cast<const vnf<N>*>(A.hpre + rowc * HD + c);
            vnf<N> dsel;
#pragma unroll
            for (int i = 0; i < N; ++i) dsel[i] = hp[i] > 0.f ? 1.0f : A.slope;
            vnf<N> g;
            if (A.gh != nullptr) g = *reinterpret_cast<const vnf<N>*>(A.gh + rowc * A.gh_stride + (c % D)) * dsel * (1.0f / (float)H);
            else {
                g = *reinterpret_cast<const vnf<N>*>(A.g + rowc * HD + c);
                if (A.g_raw) g = g * dsel;
            }
            const vnf<N> pr = *reinterpret_cast<const vnf<N>*>(A.PR + rowc * HD + c);
            const float dot = group_sum<DL>(hsum<N>(g * hp));
            const float m2 = A.mstat[rowc * H + c / D];
            const float inv = __builtin_amdgcn_rcpf(A.zstat[rowc * H + c / D] + 1e-8f);
            if (row >= 0 && (slot < 0 || b == A.row_ptr[rowc])) {
                if (A.hbits != nullptr) {
                    uint32_t nib = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) nib |= (hp[i] > 0.f ? 1u : 0u) << i;
                    A.hbits[rowc * A.hb_stride + cp] = (uint8_t)nib;
                } else {
                    stream_store(reinterpret_cast<vnf<N>*>(A.gfull + rowc * HD + c), g);
                }
            }
            vnf<N> gpr = vzero<N>();
            const int nst = wave_max_over_groups<HD, N>((e - b + U - 1) / U);
            int src[U];
            uint32_t sl[U];
            spread(0, srcv, posv, src, sl);
            for (int st = 0; st < nst; ++st) {
                vnf<N> v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = gather_row_n<HD, N, false>(A.PL, src[u], cp);
                uint32_t cur_s[U];
#pragma unroll
                for (int u = 0; u < U; ++u) cur_s[u] = sl[u];
                spread(st + 1, srcn, posn, src, sl);
                load_idx(st + 2, srcn, posn);
                float al[U], ga_[U];
#pragma unroll
                for (int u = 0; u < U; ++u) al[u] = hsum<N>(ac2 * lrelu_n<N>(v[u] + pr, A.slope));
                group_sum_n<DL, U>(al);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    al[u] = exp2_fast(al[u] - m2) * inv;
                    ga_[u] = hsum<N>(g * v[u]);
                }
                group_sum_n<DL, U>(ga_);
                uint32_t wrd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool valid = b + st * U + u < e;
                    const float ge = valid ? al[u] * (ga_[u] - dot) : 0.f;
                    const vnf<N> s = v[u] + pr;
                    const vnf<N> gs = ge * select_pos<N>(s, ac, acs);
                    ga += ge * lrelu_n<N>(s, A.slope);
                    gpr += gs;
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) bits |= (s[i] > 0.f ? 1u : 0u) << i;
                    const uint32_t w = __builtin_bit_cast(uint32_t, (cp & 1) ? ge : al[u]);
                    wrd[u] = ((w + (1u << (N - 1))) & ~((1u << N) - 1u)) | bits;
                }
                // hand the step's records to the storing wave: wait for a free ring slot (bounded), write, publish
                for (int spin = 0; published - tail_seen >= R && spin < kSpinMax; ++spin) {
                    tail_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ring.tail[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (published - tail_seen >= R) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                const int rs = published % R;
#pragma unroll
                for (int u = 0; u < U; ++u) ring.rec[wave][rs][(u * G + gidx) * 16 + cp] = wrd[u];
                ring.slot[wave][rs][(cp & 3) * G + gidx] = (cp & 2) ? ((cp & 1) ? cur_s[3] : cur_s[2]) : ((cp & 1) ? cur_s[1] : cur_s[0]);
                asm volatile("" ::: "memory");
                ++published;
                if (lane == 0) __hip_atomic_store(&ring.head[wave], published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (row >= 0) {
                float* dst = slot < 0 ? A.gPR + rowc * HD + c : A.part_acc + (int64_t)slot * HD + c;
                stream_store(reinterpret_cast<vnf<N>*>(dst), gpr);
            }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ring.done[wave], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int off = LPE; off < 64; off <<= 1) ga += shfl_xor_n<N>(ga, off);
        if (gidx == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) red[wave][c + i] = ga[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < HD) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NCW; ++w) t += red[w][threadIdx.x];
        A.ga_partial[(int64_t)blockIdx.x * HD + threadIdx.x] = t;
    }
}

#endif  // GAT_EXPERIMENTS

// gPR of split rows: sum of the row's segment partials in segment order (one thread per channel).
__global__ __launch_bounds__(256) void edge_bwd_fix_kernel(const int4* __restrict__ slot_info, int32_t n_slots,
                                                          int32_t n_split, const float* __restrict__ part,
                                                          float* __restrict__ gPR, int32_t HD) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = t / HD;                        // k-th split row
    const int c = (int)(t % HD);
    if (k >= n_split) return;
    const int4 info = slot_info[slot_info[n_slots + k].x];
    float s = 0.f;
    int sg = info.y;
    const int s_end = info.y + info.z;
    for (; sg + 8 <= s_end; sg += 8) {               // eight loads in flight, added in segment order (see edge_fwd_fix_kernel)
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = part[(int64_t)(sg + i) * HD + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; sg < s_end; ++sg) s += part[(int64_t)sg * HD + c];
    gPR[(int64_t)info.x * HD + c] = s;
}

// ------------------------------------------------------------------------------------------------
// Generic (any H, D) forward / backward: one wave per row, head statistics by "lane = head",
// aggregation by "lane = channel" with a stride-64 loop.  Correctness path for shapes outside the
// fast path; same math, literal alpha-weighted sums in ascending edge order.
// dynamic LDS: forward  act[HD];  backward dot[H] ge[H] al[H] ga[HD] gpr[HD]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void edge_fwd_generic(EdgeFwdArgs A) {
    extern __shared__ float lds[];
    const int H = A.H, D = A.D, HD = H * D;
    const int lane = threadIdx.x;
    const float slope = A.slope;
    for (int64_t row = blockIdx.x; row < A.n_rows; row += gridDim.x) {
        const int b = A.row_ptr[row], e_end = A.row_ptr[row + 1];
        for (int h = lane; h < H; h += 64) {
            float m = -1e9f;
            for (int e = b; e < e_end; ++e) {
                const int64_t sid = A.col_idx[e];
                float s = 0.f;
                for (int k = 0; k < D; ++k) {
                    const int ch = h * D + k;
                    s += A.a[ch] * lrelu(A.PL[sid * HD + ch] + A.PR[row * HD + ch], slope);
                }
                A.alpha[(int64_t)e * H + h] = s;
                if (A.score != nullptr) A.score[(int64_t)e * H + h] = s;
                m = fmaxf(m, s);
            }
            float Z = 0.f;
            for (int e = b; e < e_end; ++e) Z += __expf(A.alpha[(int64_t)e * H + h] - m);
            for (int e = b; e < e_end; ++e) {
                const int64_t i = (int64_t)e * H + h;
                A.alpha[i] = __expf(A.alpha[i] - m) / (Z + 1e-8f);
            }
            if (A.mstat != nullptr) { A.mstat[row * H + h] = m; A.zstat[row * H + h] = Z; }
        }
        __threadfence_block();
        __syncthreads();
        for (int ch = lane; ch < HD; ch += 64) {
            float acc = 0.f;
            for (int e = b; e < e_end; ++e) {
                const int64_t sid = A.col_idx[e];
                acc += A.alpha[(int64_t)e * H + ch / D] * A.PL[sid * HD + ch];
            }
            A.hpre[row * HD + ch] = acc;
            const float act = lrelu(acc, slope);
            if (!A.is_last) A.hout[row * HD + ch] = act; else lds[ch] = act;
        }
        if (A.is_last) {
            __syncthreads();
            for (int k = lane; k < D; k += 64) {
                float t = 0.f;
                for (int h = 0; h < H; ++h) t += lds[h * D + k];
                A.hout[row * D + k] = t / (float)H;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void edge_bwd_generic(EdgeBwdArgs A) {
    extern __shared__ float lds[];
    const int H = A.H, D = A.D, HD = H * D;
    float* s_dot = lds;
    float* s_ge = lds + H;
    float* s_al = lds + 2 * H;
    float* s_ga = lds + 3 * H;
    float* s_gpr = s_ga + HD;
    const int lane = threadIdx.x;
    const float slope = A.slope;
    for (int ch = lane; ch < HD; ch += 64) s_ga[ch] = 0.f;
    auto gval = [&](int64_t i) {                                  // dL/dh_pre (see EdgeBwdArgs::g_raw)
        if (A.gh != nullptr) return A.gh[(i / HD) * A.gh_stride + (i % D)] * (A.hpre[i] > 0.f ? 1.0f : slope) * (1.0f / (float)H);
        const float gv = A.g[i];
        return A.g_raw ? gv * (A.hpre[i] > 0.f ? 1.0f : slope) : gv;
    };
    for (int64_t row = blockIdx.x; row < A.n_rows; row += gridDim.x) {
        const int b = A.row_ptr[row], e_end = A.row_ptr[row + 1];
        for (int h = lane; h < H; h += 64) {
            float t = 0.f;
            for (int k = 0; k < D; ++k) t += gval(row * HD + h * D + k) * A.hpre[row * HD + h * D + k];
            s_dot[h] = t;
        }
        for (int ch = lane; ch < HD; ch += 64) s_gpr[ch] = 0.f;
        __syncthreads();
        for (int e = b; e < e_end; ++e) {
            const int64_t sid = A.col_idx[e];
            for (int h = lane; h < H; h += 64) {
                float t = 0.f;
                for (int k = 0; k < D; ++k) t += gval(row * HD + h * D + k) * A.PL[sid * HD + h * D + k];
                const float al = A.alpha[(int64_t)e * H + h];
                const float ge = al * (t - s_dot[h]);
                s_ge[h] = ge;
                s_al[h] = al;
                if (A.ge != nullptr) A.ge[(int64_t)e * H + h] = ge;
                if (A.galpha != nullptr) A.galpha[(int64_t)e * H + h] = t;
            }
            __syncthreads();
            for (int ch = lane; ch < HD; ch += 64) {
                const float v = A.PL[sid * HD + ch];
                const float s = v + A.PR[row * HD + ch];
                const bool pos = s > 0.f;
                const float ge = s_ge[ch / D];
                const float gs = ge * A.a[ch] * (pos ? 1.0f : slope);
                s_ga[ch] += ge * (pos ? s : s * slope);
                s_gpr[ch] += gs;
                unsafeAtomicAdd(A.gPL + sid * HD + ch, gval(row * HD + ch) * s_al[ch / D] + gs);
            }
            __syncthreads();
        }
        for (int ch = lane; ch < HD; ch += 64) A.gPR[row * HD + ch] = s_gpr[ch];
        __syncthreads();
    }
    for (int ch = lane; ch < HD; ch += 64) A.ga_partial[(int64_t)blockIdx.x * HD + ch] = s_ga[ch];
}

// [A][B] -> [B][A] elementwise transposes for taps (small / test-only traffic)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src,
                                                        float* __restrict__ dst, int64_t A_, int64_t B_) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = A_ * B_;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t a = i / B_, b = i % B_;     // src[a][b]
        dst[b * A_ + a] = src[i];
    }
}

// channels per lane of the two-lanes-per-head kernels (stash backward, group-per-row kernels) for this shape, 0 = none
template <int HD, int D> constexpr int stash_n() { return (D == 8 && HD >= 32) ? 4 : (D == 4 ? 2 : 0); }
// GAT_ROWGROUP=0: one wave per row (chunked kernels) instead of one lane group per row (A/B)
static bool row_groups() {
    static const bool v = [] { const char* e = choice_env("GAT_ROWGROUP"); return !(e && e[0] == '0'); }();
    return v;
}
// GAT_PACKED=0 keeps the one-channel-per-lane kernels on the training path too (A/B)
static bool packed_layout() {
    static const bool v = [] { const char* e = choice_env("GAT_PACKED"); return !(e && e[0] == '0'); }();
    return v;
}
// channels per lane of the packed kernels where the shape allows 4 (H*D >= 32, D % 4 == 0): GAT_CPL=2|4 (A/B)
static int lane_channels() {
    static const int v = [] { const char* e = choice_env("GAT_CPL"); return (e && e[0] == '2') ? 2 : 4; }();
    return v;
}

template <int HD, int D>
int run_fwd(const EdgeFwdArgs& a, hipStream_t s) {
    if (a.items == nullptr) return fail(GAT_E_INVALID, "edge_forward: work-item list missing");
    const int64_t blocks = (a.n_items + 3) / 4;
    if (a.mstat == nullptr || a.zstat == nullptr) return fail(GAT_E_INVALID, "edge_forward: stats buffers missing");
    const dim3 grid((unsigned)blocks), block(256);
    const dim3 fgrid((unsigned)(((a.alpha != nullptr ? a.n_slots : a.n_split) + 3) / 4));
    if (a.alpha != nullptr) {
        if (a.bf16) hipLaunchKernelGGL((edge_fwd_kernel<HD, D, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((edge_fwd_kernel<HD, D, true, false>), grid, block, 0, s, a);
        if (a.n_slots > 0) hipLaunchKernelGGL((edge_fwd_fix_kernel<HD, D, true>), fgrid, block, 0, s, a);
    } else {
        bool launched = false;
        if constexpr (D % 2 == 0) {
            if (packed_layout()) {                   // two or four channels per lane (see edge_fwd2_kernel)
                // one wave per block: a 4-wave block lives as long as its longest item (a 256-edge segment next to
                // 10-edge rows) and pins the other three wave slots meanwhile — 5.51 -> 5.02 ms per step
                static const int wpb = [] { const char* e = choice_env("GAT_FWD_WAVES"); const int v = e ? atoi(e) : 1; return (v == 1 || v == 2 || v == 4) ? v : 1; }();
                const dim3 grid((unsigned)((a.n_items + wpb - 1) / wpb)), block(64 * wpb);
                if constexpr (stash_n<HD, D>() != 0) {        // group-per-row kernel (GAT_ROWGROUP=0: the chunked one, A/B)
                    if (row_groups()) {
                        constexpr int NN = stash_n<HD, D>(), GG = 64 / (HD / NN);
                        const dim3 g3((unsigned)((a.n_items + (int64_t)GG * wpb - 1) / ((int64_t)GG * wpb)));
                        if (a.bf16) hipLaunchKernelGGL((edge_fwd3_kernel<HD, D, NN, true>), g3, block, 0, s, a);
                        else hipLaunchKernelGGL((edge_fwd3_kernel<HD, D, NN, false>), g3, block, 0, s, a);
                        if (a.n_slots > 0) hipLaunchKernelGGL((edge_fwd_fix_kernel<HD, D, false>), fgrid, dim3(256), 0, s, a);
                        GAT_HIP(hipGetLastError());
                        return 0;
                    }
                }
                bool four = false;
                if constexpr (HD >= 32 && D % 4 == 0) {
                    if (lane_channels() == 4) {
                        if (a.bf16) hipLaunchKernelGGL((edge_fwd2_kernel<HD, D, 4, true>), grid, block, 0, s, a);
                        else hipLaunchKernelGGL((edge_fwd2_kernel<HD, D, 4, false>), grid, block, 0, s, a);
                        four = true;
                    }
                }
                if (!four) {
                    if (a.bf16) hipLaunchKernelGGL((edge_fwd2_kernel<HD, D, 2, true>), grid, block, 0, s, a);
                    else hipLaunchKernelGGL((edge_fwd2_kernel<HD, D, 2, false>), grid, block, 0, s, a);
                }
                launched = true;
            }
        }
        if (!launched) {
            if (a.bf16) hipLaunchKernelGGL((edge_fwd_kernel<HD, D, false, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((edge_fwd_kernel<HD, D, false, false>), grid, block, 0, s, a);
        }
        if (a.n_slots > 0) hipLaunchKernelGGL((edge_fwd_fix_kernel<HD, D, false>), fgrid, block, 0, s, a);
    }
    GAT_HIP(hipGetLastError());
    return 0;
}
// Blocks of `fn` (256 threads, no dynamic LDS) that are resident on the whole chip at once, from the
// occupancy API (cached per kernel).  The backward grid is exactly this size: its items are dealt
// statically, so a second, partially filled round of blocks would be pure tail.
static int resident_blocks(const void* fn) {
    static std::mutex mu;
    static std::map<const void*, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(fn);
    if (it != cache.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    return cache[fn] = per_cu * (cus > 0 ? cus : 256);
}
static bool packed_backward() { return packed_layout(); }
// GAT_GROUP_MSG=0: message-row layers on the wave-per-row kernel (edge_bwd2_kernel) instead of the group-per-row one (A/B)
static bool group_msg() {
    static const bool v = [] { const char* e = choice_env("GAT_GROUP_MSG"); return !(e && e[0] == '0'); }();
    return v;
}
struct BwdSel { bool store, taps, bf16, stash; };
template <int HD, int D, bool BF>
const void* bwd_variant(bool store, bool taps, bool stash = false) {
    if constexpr (stash_n<HD, D>() != 0) {
        if (stash && !taps) {
            if (BF) return (const void*)edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 0, true>;
            return row_groups() ? (const void*)edge_bwd3_kernel<HD, D, stash_n<HD, D>()> : (const void*)edge_bwd2s_kernel<HD, D, stash_n<HD, D>()>;
        }
    }
    if constexpr (stash_n<HD, D>() != 0) {        // message rows from the group-per-row kernel
        if (store && !taps && packed_backward() && row_groups() && group_msg()) return (const void*)edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 0, BF, true>;
    }
    if constexpr (D % 2 == 0) {
        if (store && !taps && packed_backward()) {
            if constexpr (HD >= 32 && D % 4 == 0) {
                if (lane_channels() == 4) return (const void*)edge_bwd2_kernel<HD, D, 4, 0, BF>;
            }
            return (const void*)edge_bwd2_kernel<HD, D, 2, 0, BF>;
        }
    }
    return store ? (taps ? (const void*)edge_bwd_kernel<HD, D, true, true, 0, BF> : (const void*)edge_bwd_kernel<HD, D, true, false, 0, BF>)
                 : (taps ? (const void*)edge_bwd_kernel<HD, D, false, true, 0, BF> : (const void*)edge_bwd_kernel<HD, D, false, false, 0, BF>);
}
template <int HD, int D>
int bwd_resident(const BwdSel& sel, hipStream_t) {
    return resident_blocks(sel.bf16 ? bwd_variant<HD, D, true>(sel.store, sel.taps, sel.stash) : bwd_variant<HD, D, false>(sel.store, sel.taps, sel.stash));
}

template <int HD, int D>
int run_bwd(const EdgeBwdArgs& a, hipStream_t s) {
    if (a.items == nullptr) return fail(GAT_E_INVALID, "edge_backward: work-item list missing");
    const dim3 grid((unsigned)a.ga_blocks), block(256);
    const bool store = a.pos != nullptr && a.msg != nullptr, taps = a.ge != nullptr;
#ifdef GAT_EXPERIMENTS
    if constexpr (HD == 64 && D == 8) {          // timing experiments (GAT_DBG=1: no message store, 2: sequential slots) — WRONG RESULTS
        if (store && !taps && a.stash == nullptr && a.dbg == 1) { hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, false, 1>), grid, block, 0, s, a); return 0; }
        if (store && !taps && a.stash == nullptr && a.dbg == 2) { hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, false, 2>), grid, block, 0, s, a); return 0; }
    }
#endif
    bool launched = false;
    if constexpr (stash_n<HD, D>() != 0) {
        if (a.stash != nullptr && !taps && a.bf16) {                // bf16 storage: always the group-per-row kernel
            if (a.gfull == nullptr || a.pos == nullptr) return fail(GAT_E_INVALID, "edge_backward: stash path needs gfull and pos");
            hipLaunchKernelGGL((edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 0, true>), grid, block, 0, s, a);
            launched = true;
        }
        if (a.stash != nullptr && !taps && !a.bf16) {
            if (a.gfull == nullptr || a.pos == nullptr) return fail(GAT_E_INVALID, "edge_backward: stash path needs gfull and pos");
            bool dbg_done = false;
#ifdef GAT_EXPERIMENTS                               // GAT_DBG exists in the experiment library only (make experiments): most values give WRONG results
            if constexpr (HD == 64 && D == 8) {      // timing experiments (GAT_DBG=1: no record store, 2: records in CSR order)
                if (a.dbg == 9 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 9>), grid, block, 0, s, a); dbg_done = true; }   // double walk
                else if (a.dbg == 10 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 10>), grid, block, 0, s, a); dbg_done = true; }   // scatter folded into 64 MiB
                else if (a.dbg == 11 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 11>), grid, block, 0, s, a); dbg_done = true; }   // 16 MiB
                else if (a.dbg == 12 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 12>), grid, block, 0, s, a); dbg_done = true; }   // 4 MiB
                else if (a.dbg == 13 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 13>), grid, block, 0, s, a); dbg_done = true; }   // 1 MiB
                else if (a.dbg == 15 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 15>), grid, block, 0, s, a); dbg_done = true; }   // 64 sequential write fronts
                else if (a.dbg == 16 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 16>), grid, block, 0, s, a); dbg_done = true; }   // 512
                else if (a.dbg == 17 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 17>), grid, block, 0, s, a); dbg_done = true; }   // 4,096
                else if (a.dbg == 18 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 18>), grid, block, 0, s, a); dbg_done = true; }   // 32,768
                else if (a.dbg == 14 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 14>), grid, block, 0, s, a); dbg_done = true; }   // 1 GiB
                else if (a.dbg == 4 && row_groups()) { hipLaunchKernelGGL((edge_bwd4_kernel<7, 4>), grid, dim3(512), 0, s, a); dbg_done = true; }   // wave-specialised stores: 7 + 1 waves
                else if (a.dbg == 6 && row_groups()) { hipLaunchKernelGGL((edge_bwd4_kernel<3, 4, 1>), grid, dim3(256), 0, s, a); dbg_done = true; }   // 3 + 1 waves, sc1 stores
                else if (a.dbg == 7 && row_groups()) { hipLaunchKernelGGL((edge_bwd4_kernel<3, 4, 2>), grid, dim3(256), 0, s, a); dbg_done = true; }   // sc0 sc1
                else if (a.dbg == 8 && row_groups()) { hipLaunchKernelGGL((edge_bwd4_kernel<3, 4, 3>), grid, dim3(256), 0, s, a); dbg_done = true; }   // nt
                else if (a.dbg == 5 && row_groups()) { hipLaunchKernelGGL((edge_bwd4_kernel<3, 4>), grid, dim3(256), 0, s, a); dbg_done = true; }   // 3 + 1 waves
                else if (a.dbg == 3 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 3>), grid, block, 0, s, a); dbg_done = true; }
                else if (a.dbg == 1 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 1>), grid, block, 0, s, a); dbg_done = true; }
                else if (a.dbg == 2 && row_groups()) { hipLaunchKernelGGL((edge_bwd3_kernel<64, 8, 4, 2>), grid, block, 0, s, a); dbg_done = true; }
                else if (a.dbg == 1) { hipLaunchKernelGGL((edge_bwd2s_kernel<64, 8, 4, 1>), grid, block, 0, s, a); dbg_done = true; }
                else if (a.dbg == 2) { hipLaunchKernelGGL((edge_bwd2s_kernel<64, 8, 4, 2>), grid, block, 0, s, a); dbg_done = true; }
            }
#endif
            if (!dbg_done) {
                if (row_groups()) hipLaunchKernelGGL((edge_bwd3_kernel<HD, D, stash_n<HD, D>()>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((edge_bwd2s_kernel<HD, D, stash_n<HD, D>()>), grid, block, 0, s, a);
            }
            launched = true;
        }
    }
    if constexpr (stash_n<HD, D>() != 0) {
#ifdef GAT_EXPERIMENTS       // GAT_DBG=1 on a message-row layer (BASELINE config 5): the walk without its message-row stores — WRONG RESULTS, timing / PMC
        if constexpr (HD == 32 && D == 8) {           // attribution only (DESIGN §4 "Round 4": how much of the backward's reads is fill for its 64-byte row writes)
            if (!launched && store && !taps && packed_backward() && row_groups() && group_msg() && a.bf16 && a.dbg == 1) {
                hipLaunchKernelGGL((edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 1, true, true>), grid, block, 0, s, a);
                launched = true;
            }
        }
#endif
        if (!launched && store && !taps && packed_backward() && row_groups() && group_msg()) {
            if (a.bf16) hipLaunchKernelGGL((edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 0, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((edge_bwd3_kernel<HD, D, stash_n<HD, D>(), 0, false, true>), grid, block, 0, s, a);
            launched = true;
        }
    }
    if constexpr (D % 2 == 0) {
        if (!launched && store && !taps && packed_backward()) {
            bool four = false;
            if constexpr (HD >= 32 && D % 4 == 0) {
                if (lane_channels() == 4) {
                    if (a.bf16) hipLaunchKernelGGL((edge_bwd2_kernel<HD, D, 4, 0, true>), grid, block, 0, s, a);
                    else hipLaunchKernelGGL((edge_bwd2_kernel<HD, D, 4, 0, false>), grid, block, 0, s, a);
                    four = true;
                }
            }
            if (!four) {
                if (a.bf16) hipLaunchKernelGGL((edge_bwd2_kernel<HD, D, 2, 0, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((edge_bwd2_kernel<HD, D, 2, 0, false>), grid, block, 0, s, a);
            }
            launched = true;
        }
    }
    if (launched) {
    } else if (a.bf16) {
        if (store && taps) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, true, 0, true>), grid, block, 0, s, a);
        else if (store) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, false, 0, true>), grid, block, 0, s, a);
        else if (taps) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, false, true, 0, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((edge_bwd_kernel<HD, D, false, false, 0, true>), grid, block, 0, s, a);
    } else {
        if (store && taps) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, true>), grid, block, 0, s, a);
        else if (store) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, true, false>), grid, block, 0, s, a);
        else if (taps) hipLaunchKernelGGL((edge_bwd_kernel<HD, D, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((edge_bwd_kernel<HD, D, false, false>), grid, block, 0, s, a);
    }
    GAT_HIP(hipGetLastError());
    if (a.n_slots > 0) {
        const int64_t threads = (int64_t)a.n_split * HD;
        hipLaunchKernelGGL(edge_bwd_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                           a.slot_info, a.n_slots, a.n_split, a.part_acc, a.gPR, HD);
        GAT_HIP(hipGetLastError());
    }
    return 0;
}

constexpr int kGenericBlocks = 4096;

}  // namespace

#define GAT_DISPATCH_HD_D(FN, ARGS, S)                                            \
    switch (HD * 1000 + D) {                                                      \
        case 64008: return FN<64, 8>(ARGS, S);                                    \
        case 64004: return FN<64, 4>(ARGS, S);                                    \
        case 64016: return FN<64, 16>(ARGS, S);                                   \
        case 64032: return FN<64, 32>(ARGS, S);                                   \
        case 64064: return FN<64, 64>(ARGS, S);                                   \
        case 32008: return FN<32, 8>(ARGS, S);                                    \
        case 32004: return FN<32, 4>(ARGS, S);                                    \
        case 32016: return FN<32, 16>(ARGS, S);                                   \
        case 32032: return FN<32, 32>(ARGS, S);                                   \
        case 16004: return FN<16, 4>(ARGS, S);                                    \
        case 16008: return FN<16, 8>(ARGS, S);                                    \
        case 16016: return FN<16, 16>(ARGS, S);                                   \
        case 8004: return FN<8, 4>(ARGS, S);                                      \
        case 8008: return FN<8, 8>(ARGS, S);                                      \
        default: break;                                                           \
    }

int launch_edge_forward(const EdgeFwdArgs& a, hipStream_t s) {
    if (a.n_rows <= 0) return 0;
    const int HD = a.H * a.D, D = a.D;
    if (edge_fast_path(a.H, a.D, a.n_table)) { GAT_DISPATCH_HD_D(run_fwd, a, s) }
    const int64_t blocks = a.n_rows < kGenericBlocks * 8 ? a.n_rows : kGenericBlocks * 8;
    hipLaunchKernelGGL(edge_fwd_generic, dim3((unsigned)blocks), dim3(64), (size_t)HD * sizeof(float), s, a);
    GAT_HIP(hipGetLastError());
    return 0;
}

template <int HD, int D>
int stash_words_t(const int&, hipStream_t) { return stash_n<HD, D>() ? HD / stash_n<HD, D>() : 0; }
int edge_stash_words(int32_t H, int32_t D_) {
    const int HD = H * D_, D = D_;
    const int dummy = 0;
    auto probe = [&]() -> int {
        GAT_DISPATCH_HD_D(stash_words_t, dummy, nullptr)
        return 0;
    };
    return probe();
}

int edge_backward_blocks(int64_t n_items, int32_t H, int32_t D_, bool store, bool taps, bool bf16, bool stash) {
    int64_t want = (n_items + 3) / 4;
    if (want < 1) want = 1;
    const int HD = H * D_, D = D_;
    const BwdSel sel{store, taps, bf16, stash};
    auto cap = [&]() -> int {
        GAT_DISPATCH_HD_D(bwd_resident, sel, nullptr)
        return 2048;                                  // generic path: one wave per block
    };
    int c = cap();
    // experiment library, GAT_DBG=4: 512-thread blocks of edge_bwd4_kernel — two resident per CU instead of four
#ifdef GAT_EXPERIMENTS
    static const int dbg = [] { const char* e = getenv("GAT_DBG"); return e ? atoi(e) : 0; }();
    if (dbg == 4 && HD == 64 && D == 8 && stash && !bf16 && !taps) c = c / 2;
#endif
    return (int)(want < c ? want : c);
}

static int fast_probe(const int&, hipStream_t) { return 1; }
template <int HD, int D>
int fast_probe_t(const int& x, hipStream_t s) { return fast_probe(x, s); }

bool edge_fast_path(int32_t H, int32_t D_, int64_t n_table) {
    const int HD = H * D_, D = D_;
    if (n_table * HD * (int64_t)sizeof(float) >= ((int64_t)1 << 32)) return false;
    const int dummy = 0;
    auto probe = [&]() -> int {
        GAT_DISPATCH_HD_D(fast_probe_t, dummy, nullptr)
        return 0;
    };
    return probe() == 1;
}

int launch_edge_backward(const EdgeBwdArgs& a, hipStream_t s) {
    const int HD = a.H * a.D, D = a.D;
    if (a.ga_blocks < 1) return fail(GAT_E_INVALID, "edge_backward: ga_blocks must come from edge_backward_blocks()");
    if (edge_fast_path(a.H, a.D, a.n_table)) { GAT_DISPATCH_HD_D(run_bwd, a, s) }
    if (a.pos != nullptr) return fail(GAT_E_INVALID, "edge_backward: the generic path has no store mode");
    const size_t lds = (size_t)(3 * a.H + 2 * HD) * sizeof(float);
    if (lds > 64 * 1024) return fail(GAT_E_UNSUPPORTED, "edge_backward: H*D too large for the generic path");
    hipLaunchKernelGGL(edge_bwd_generic, dim3((unsigned)a.ga_blocks), dim3(64), lds, s, a);
    GAT_HIP(hipGetLastError());
    return 0;
}

bool edge_last_fused_supported(int32_t H, int32_t D, int32_t C) { return H * D == 64 && D == 8 && C >= 1 && C <= 64; }
int edge_last_fused_blocks(int64_t n_items) {
    const int64_t want = std::max<int64_t>(1, ((n_items + 3) / 4 + 3) / 4);        // 4 rows per wave, 4 waves per block
    const int cap = resident_blocks((const void*)edge_last_fused_kernel<64, 8, 4>);
    return (int)std::min<int64_t>(want, cap);
}
int launch_edge_last_fused(const EdgeLastArgs& a, hipStream_t s) {
    if (!edge_last_fused_supported(a.f.H, a.f.D, a.C)) return fail(GAT_E_UNSUPPORTED, "edge_last_fused: shape outside the kernel");
    if (a.f.n_items <= 0) return 0;
    if (a.f.items == nullptr || a.b.pos == nullptr || a.b.stash == nullptr || a.gh_out == nullptr || a.b.ga_blocks < 1 ||
        (a.b.hbits == nullptr && a.b.gfull == nullptr))
        return fail(GAT_E_INVALID, "edge_last_fused: missing buffers");
    hipLaunchKernelGGL((edge_last_fused_kernel<64, 8, 4>), dim3((unsigned)a.b.ga_blocks), dim3(256), 0, s, a);
    GAT_HIP(hipGetLastError());
    return 0;
}
int launch_head_rows(const int4* slot_info, int32_t n_slots, int32_t n_split, const float* Wo, const float* HL, const int32_t* labels,
                     float* gh_out, int32_t gh_stride, int32_t C, int32_t DLAST, hipStream_t s) {
    if (n_split <= 0) return 0;
    if (DLAST > 16) return fail(GAT_E_UNSUPPORTED, "head_rows: D_last > 16");
    hipLaunchKernelGGL(head_rows_kernel, dim3((unsigned)((n_split + 255) / 256)), dim3(256), 0, s, slot_info, n_slots, n_split, Wo, HL,
                       labels, gh_out, gh_stride, C, DLAST);
    GAT_HIP(hipGetLastError());
    return 0;
}

int launch_csr_to_coo(const int32_t* row_ptr, const int32_t* col_idx, int32_t* src, int32_t* dst,
                      int64_t n_rows, int64_t n_edges, int64_t dst_offset, hipStream_t s) {
    if (n_edges <= 0) return 0;
    int64_t blocks = (n_edges + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(csr_to_coo_kernel, dim3((unsigned)blocks), dim3(256), 0, s, row_ptr, col_idx, src,
                       dst, n_rows, n_edges, (int32_t)dst_offset);
    GAT_HIP(hipGetLastError());
    return 0;
}

static int launch_transpose(const float* src, float* dst, int64_t A_, int64_t B_, hipStream_t s) {
    const int64_t total = A_ * B_;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, A_, B_);
    GAT_HIP(hipGetLastError());
    return 0;
}
int launch_transpose_eh_to_he(const float* src_eh, float* dst_he, int64_t E, int32_t H, hipStream_t s) {
    return launch_transpose(src_eh, dst_he, E, H, s);
}
int launch_transpose_he_to_eh(const float* src_he, float* dst_eh, int64_t E, int32_t H, hipStream_t s) {
    return launch_transpose(src_he, dst_eh, H, E, s);
}
int launch_transpose_nh_to_hn(const float* src_nh, float* dst_hn, int64_t N, int32_t H, hipStream_t s) {
    return launch_transpose(src_nh, dst_hn, N, H, s);
}

}  // namespace gat
