// gat_halo.hip — halo form of the two table exchanges of a destination-range shard (gat_comm_option GAT_COMM_HALO).
//
// The reference reads, per edge, the feature row of the edge's SOURCE (E:287-290 score, E:407-409 aggregation) and adds
// the edge's gradient contribution into that source's row (E:868-869).  A shard owns a destination range, so the only
// rows that ever cross shards are the source rows its edges name.  The full exchange (gat_comm.hip) moves every row of
// every slice; here each rank receives exactly the rows of each peer's slice that its own col_idx references, and in the
// backward sends exactly those rows' partial sums back:
//   set-up  (collective, once per graph): mark the referenced table rows on the device; per peer q the sorted list
//           need[q] of rows of q's slice that are marked; the lists are all-gathered (setup only) so that every rank
//           also knows send[p] = need_p[rank];
//   forward : pack PL[own rows in send order] -> pairwise send / receive (one pair per peer: xGMI is a full mesh) ->
//             scatter the arrivals into the table rows need[.];
//   backward: pack gPL[need[.]] (the only rows of other slices that can be non-zero) -> pairwise exchange (the mirror) ->
//             own[r] = sum over ranks in ascending order, the own partial at its rank's position — the order of the
//             host transport's reduce-scatter, so the two forms agree value for value (a rank that does not reference
//             a row contributes +0 there).
// Wire bytes: (rows referenced) / (rows) of the full exchange — DESIGN §7 has the table.  The price is one pack and one
// unpack pass over the exchanged rows on each side.
#include "gat_internal.h"

#include <algorithm>
#include <cstring>

namespace gat {
namespace {

__global__ __launch_bounds__(256) void halo_mark_kernel(const int32_t* __restrict__ col_idx, int64_t n_edges, uint8_t* __restrict__ mark) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += stride) mark[col_idx[e]] = 1;
}
// dst[i][:] = src[rows[i] + row0][:]   (rows of rf floats; VEC = float4 when rf % 4 == 0)
template <class VEC>
__global__ __launch_bounds__(256) void halo_pack_kernel(const VEC* __restrict__ src, const int32_t* __restrict__ rows, int64_t row0,
                                                        int64_t n, int32_t rv, VEC* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * rv; t += stride) {
        const int64_t i = t / rv;
        const int k = (int)(t % rv);
        dst[t] = src[((int64_t)rows[i] + row0) * rv + k];
    }
}
// dst[rows[i]][:] = src[i][:]
template <class VEC>
__global__ __launch_bounds__(256) void halo_unpack_kernel(const VEC* __restrict__ src, const int32_t* __restrict__ rows, int64_t n, int32_t rv,
                                                          VEC* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * rv; t += stride) {
        const int64_t i = t / rv;
        const int k = (int)(t % rv);
        dst[(int64_t)rows[i] * rv + k] = src[t];
    }
}
// own[r][k] = (sum of the arrivals from lower ranks, ascending) + own[r][k] + (arrivals from higher ranks, ascending)
__global__ __launch_bounds__(256) void halo_sum_kernel(float* __restrict__ own, const float* __restrict__ recv, const int32_t* __restrict__ arr_ptr,
                                                       const int32_t* __restrict__ arr_split, const int32_t* __restrict__ arr_pos, int64_t n_rows,
                                                       int32_t rf) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * rf; t += stride) {
        const int64_t r = t / rf;
        const int k = (int)(t % rf);
        const int b = arr_ptr[r], e = arr_ptr[r + 1];
        if (b == e) continue;
        const int sp = b + arr_split[r];
        float acc = 0.f;
        for (int i = b; i < sp; ++i) acc += recv[(int64_t)arr_pos[i] * rf + k];
        acc += own[t];
        for (int i = sp; i < e; ++i) acc += recv[(int64_t)arr_pos[i] * rf + k];
        own[t] = acc;
    }
}

static unsigned grid_for(int64_t threads) { return (unsigned)std::min<int64_t>(std::max<int64_t>((threads + 255) / 256, 1), 16384); }

static int pack_rows(const float* src, const int32_t* rows, int64_t row0, int64_t n, int64_t rf, float* dst, hipStream_t s) {
    if (n <= 0) return 0;
    if (rf % 4 == 0)
        hipLaunchKernelGGL(halo_pack_kernel<float4>, dim3(grid_for(n * (rf / 4))), dim3(256), 0, s, reinterpret_cast<const float4*>(src), rows, row0, n,
                           (int32_t)(rf / 4), reinterpret_cast<float4*>(dst));
    else
        hipLaunchKernelGGL(halo_pack_kernel<float>, dim3(grid_for(n * rf)), dim3(256), 0, s, src, rows, row0, n, (int32_t)rf, dst);
    GAT_HIP(hipGetLastError());
    return 0;
}
static int unpack_rows(const float* src, const int32_t* rows, int64_t n, int64_t rf, float* dst, hipStream_t s) {
    if (n <= 0) return 0;
    if (rf % 4 == 0)
        hipLaunchKernelGGL(halo_unpack_kernel<float4>, dim3(grid_for(n * (rf / 4))), dim3(256), 0, s, reinterpret_cast<const float4*>(src), rows, n,
                           (int32_t)(rf / 4), reinterpret_cast<float4*>(dst));
    else
        hipLaunchKernelGGL(halo_unpack_kernel<float>, dim3(grid_for(n * rf)), dim3(256), 0, s, src, rows, n, (int32_t)rf, dst);
    GAT_HIP(hipGetLastError());
    return 0;
}

}  // namespace

void halo_free(HaloPlan* h) {
    if (!h) return;
    for (void* p : {(void*)h->need_rows, (void*)h->send_rows, (void*)h->arr_ptr, (void*)h->arr_split, (void*)h->arr_pos, (void*)h->sendbuf, (void*)h->recvbuf})
        if (p) (void)hipFree(p);
    *h = HaloPlan();
}

// Collective: every rank of the transport calls it with its own shard.
int halo_build(Comm* comm, const int32_t* d_col_idx, int64_t n_edges, int64_t n_table, int32_t hd_max, HaloPlan* out, hipStream_t s) {
    if (!comm || !out) return fail(GAT_E_INVALID, "halo_build: null argument");
    const int P = comm->world, me = comm->rank;
    if (n_table % P != 0) return fail(GAT_E_INVALID, "halo_build: table is not [world][max_rows]");
    const int64_t max_rows = n_table / P;
    halo_free(out);
    HaloPlan h;
    h.world = P; h.rank = me; h.max_rows = max_rows;
    // 1. which table rows do this shard's edges name?
    std::vector<uint8_t> mark((size_t)n_table, 0);
    {
        uint8_t* d_mark = nullptr;
        GAT_HIP(hipMalloc((void**)&d_mark, (size_t)std::max<int64_t>(n_table, 1)));
        hipError_t e = hipMemsetAsync(d_mark, 0, (size_t)n_table, s);
        if (e == hipSuccess && n_edges > 0) {
            hipLaunchKernelGGL(halo_mark_kernel, dim3(grid_for(n_edges)), dim3(256), 0, s, d_col_idx, n_edges, d_mark);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(mark.data(), d_mark, (size_t)n_table, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_mark);
        if (e != hipSuccess) return fail((int)e, std::string("halo_build (mark): ") + hipGetErrorString(e));
    }
    // need lists: table row ids grouped by peer (ascending), the own slice left out (at world 1 it is kept: a self exchange,
    // so that the pairwise path is exercised on a one-GPU box — the arrival is then not summed, see below)
    std::vector<int32_t> need;            // table row ids
    h.need_cnt.assign((size_t)P, 0);
    for (int q = 0; q < P; ++q) {
        if (q == me && P > 1) continue;
        for (int64_t r = 0; r < max_rows; ++r)
            if (mark[(size_t)(q * max_rows + r)]) { need.push_back((int32_t)(q * max_rows + r)); ++h.need_cnt[(size_t)q]; }
    }
    h.n_need = (int64_t)need.size();
    // 2. counts of every rank (all-gather of a [world][world] int table, bytes travel as floats)
    std::vector<int32_t> cnt((size_t)P * P, 0);          // cnt[p][q] = rows p needs from q
    {
        int32_t* d = nullptr;
        GAT_HIP(hipMalloc((void**)&d, (size_t)P * P * sizeof(int32_t)));
        std::vector<int32_t> mine((size_t)P);
        for (int q = 0; q < P; ++q) mine[(size_t)q] = (int32_t)h.need_cnt[(size_t)q];
        int rc = 0;
        hipError_t e = hipMemcpyAsync(d + (size_t)me * P, mine.data(), (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) rc = comm->all_gather(reinterpret_cast<float*>(d), P, s);
        if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(cnt.data(), d, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(s);
        (void)hipFree(d);
        if (rc != 0) return rc;
        if (e != hipSuccess) return fail((int)e, std::string("halo_build (counts): ") + hipGetErrorString(e));
    }
    int64_t maxn = 1, total = 0;
    for (int p = 0; p < P; ++p) {
        int64_t t = 0;
        for (int q = 0; q < P; ++q) t += cnt[(size_t)p * P + q];
        maxn = std::max(maxn, t); total += t;
    }
    h.referenced_fraction = P > 1 ? (double)total / ((double)P * (double)(P - 1) * (double)max_rows) : (double)total / (double)std::max<int64_t>(max_rows, 1);
    // 3. the lists of every rank (local row ids inside the peer's slice), to find out what each peer wants from this rank
    std::vector<int32_t> all((size_t)P * maxn, 0);
    {
        int32_t* d = nullptr;
        GAT_HIP(hipMalloc((void**)&d, (size_t)P * maxn * sizeof(int32_t)));
        std::vector<int32_t> mine((size_t)maxn, 0);
        for (size_t i = 0; i < need.size(); ++i) mine[i] = (int32_t)(need[i] % max_rows);
        int rc = 0;
        hipError_t e = hipMemcpyAsync(d + (size_t)me * maxn, mine.data(), (size_t)maxn * sizeof(int32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) rc = comm->all_gather(reinterpret_cast<float*>(d), maxn, s);
        if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(all.data(), d, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(s);
        (void)hipFree(d);
        if (rc != 0) return rc;
        if (e != hipSuccess) return fail((int)e, std::string("halo_build (lists): ") + hipGetErrorString(e));
    }
    std::vector<int32_t> send;            // own-slice local row ids, grouped by requesting peer (ascending)
    h.send_cnt.assign((size_t)P, 0);
    for (int p = 0; p < P; ++p) {
        if (p == me && P > 1) continue;
        int64_t off = 0;
        for (int q = 0; q < me; ++q) off += cnt[(size_t)p * P + q];
        const int64_t n = cnt[(size_t)p * P + me];
        h.send_cnt[(size_t)p] = n;
        for (int64_t i = 0; i < n; ++i) {
            const int32_t r = all[(size_t)p * maxn + off + i];
            if (r < 0 || r >= max_rows) return fail(GAT_E_COMM, "halo_build: a peer's request list names a row outside this rank's slice");
            send.push_back(r);
        }
    }
    h.n_send = (int64_t)send.size();
    // 4. arrivals per own row for the backward sum: positions in the receive buffer (= send order), lower ranks first.
    //    A self exchange (world 1) is not summed: the own partial already holds those contributions.
    std::vector<int32_t> arr_ptr((size_t)max_rows + 1, 0), arr_split((size_t)max_rows, 0), arr_pos;
    if (P > 1) {
        for (int32_t r : send) ++arr_ptr[(size_t)r + 1];
        for (int64_t r = 0; r < max_rows; ++r) arr_ptr[(size_t)r + 1] += arr_ptr[(size_t)r];
        arr_pos.resize(send.size());
        std::vector<int32_t> fill(arr_ptr.begin(), arr_ptr.end() - 1);
        int64_t pos = 0;
        for (int p = 0; p < P; ++p) {
            if (p == me) continue;
            for (int64_t i = 0; i < h.send_cnt[(size_t)p]; ++i, ++pos) {
                const int32_t r = send[(size_t)pos];
                arr_pos[(size_t)fill[(size_t)r]++] = (int32_t)pos;
                if (p < me) ++arr_split[(size_t)r];
            }
        }
    }
    auto upload = [&](int32_t** d, const std::vector<int32_t>& v) -> int {
        GAT_HIP(hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
        if (!v.empty()) GAT_HIP(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        return 0;
    };
    int rc = upload(&h.need_rows, need);
    if (rc == 0) rc = upload(&h.send_rows, send);
    if (rc == 0) rc = upload(&h.arr_ptr, arr_ptr);
    if (rc == 0) rc = upload(&h.arr_split, arr_split);
    if (rc == 0) rc = upload(&h.arr_pos, arr_pos);
    h.buf_rows = std::max<int64_t>(std::max(h.n_need, h.n_send), 1);
    if (rc == 0 && hipMalloc((void**)&h.sendbuf, (size_t)h.buf_rows * hd_max * sizeof(float)) != hipSuccess) rc = fail(GAT_E_NOMEM, "halo_build: send buffer");
    if (rc == 0 && hipMalloc((void**)&h.recvbuf, (size_t)h.buf_rows * hd_max * sizeof(float)) != hipSuccess) rc = fail(GAT_E_NOMEM, "halo_build: receive buffer");
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = fail(GAT_E_STATE, "halo_build: upload failed");
    if (rc != 0) { halo_free(&h); return rc; }
    *out = h;
    return 0;
}

// table: [world][max_rows] rows of rf floats (PL rows of the storage dtype counted in floats); the own slice is written
int halo_forward(Comm* comm, const HaloPlan& h, float* table, int64_t rf, hipStream_t s) {
    GAT_TRY(pack_rows(table, h.send_rows, (int64_t)h.rank * h.max_rows, h.n_send, rf, h.sendbuf, s));
    GAT_TRY(comm->exchange_rows(h.sendbuf, h.send_cnt.data(), h.recvbuf, h.need_cnt.data(), rf, s));
    return unpack_rows(h.recvbuf, h.need_rows, h.n_need, rf, table, s);
}
// gPL: [world][max_rows] rows of rf floats holding this rank's partial sums; afterwards the own slice holds the sums over ranks
int halo_backward(Comm* comm, const HaloPlan& h, float* gPL, int64_t rf, hipStream_t s) {
    GAT_TRY(pack_rows(gPL, h.need_rows, 0, h.n_need, rf, h.sendbuf, s));
    GAT_TRY(comm->exchange_rows(h.sendbuf, h.need_cnt.data(), h.recvbuf, h.send_cnt.data(), rf, s));
    if (h.world == 1 || h.n_send == 0) return 0;
    float* own = gPL + (int64_t)h.rank * h.max_rows * rf;
    hipLaunchKernelGGL(halo_sum_kernel, dim3(grid_for(h.max_rows * rf)), dim3(256), 0, s, own, h.recvbuf, h.arr_ptr, h.arr_split, h.arr_pos, h.max_rows, (int32_t)rf);
    GAT_HIP(hipGetLastError());
    return 0;
}

}  // namespace gat
