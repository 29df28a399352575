// gat_comm.hip — the three exchanges of a destination-range shard, behind one interface.
//
// The reference is a single process on one GPU; sharding is this build's scaling axis.
// Per layer and direction one table crosses shards — PL rows forward (all-gather), gPL partial sums
// backward (reduce-scatter) — plus one all-reduce of the packed parameter gradients.  Tables are
// [world][slice] arrays (shard.py / host/shard_plan.h pad every rank to max_rows), so all three are
// the fixed-count, in-place forms.
//
//   RcclComm : RCCL over xGMI, enqueued on the context's stream.  librccl is dlopen'ed at
//              gat_comm_init_rccl — single-GPU users of libgatv2_hip.so never load it, and inside a
//              torch process the copy torch already loaded (same soname) is the one that binds.
//   HostComm : staged through a POSIX shared-memory segment, ranks meeting at a generation barrier with a
//              deadline (a dead peer makes the others return GAT_E_COMM, not hang); sums in ascending rank
//              order (bitwise identical on every rank).  For tests (several ranks sharing one GPU, which
//              RCCL refuses) and boxes without peer links.
#include "gat_internal.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <memory>
#include <ctime>

namespace gat {

Comm::~Comm() {}

namespace {

// ---- bf16 travel format of the gPL reduce-scatter ---------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t u = __builtin_bit_cast(uint32_t, src[i]);
        dst[i] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
}
// own[i] = sum over ranks q in ascending order of (q == rank ? own[i] (fp32) : bf16 recv[q][i])
__global__ __launch_bounds__(256) void sum_arrivals_kernel(float* __restrict__ own, const uint16_t* __restrict__ recv, int world, int rank,
                                                           int64_t slice) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slice; i += stride) {
        float acc = 0.f;
        for (int q = 0; q < world; ++q)
            acc += q == rank ? own[i] : __builtin_bit_cast(float, (uint32_t)recv[(int64_t)q * slice + i] << 16);
        own[i] = acc;
    }
}
static int launch_pack_bf16(const float* src, uint16_t* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, src, dst, n);
    GAT_HIP(hipGetLastError());
    return 0;
}
static int launch_sum_arrivals(float* own, const uint16_t* recv, int world, int rank, int64_t slice, hipStream_t s) {
    if (slice <= 0) return 0;
    hipLaunchKernelGGL(sum_arrivals_kernel, dim3((unsigned)std::min<int64_t>((slice + 255) / 256, 8192)), dim3(256), 0, s, own, recv, world, rank, slice);
    GAT_HIP(hipGetLastError());
    return 0;
}

// ---- RCCL (types restated from the public NCCL API; resolved at run time) ------------------------------
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
static_assert(sizeof(ncclUniqueId) == GAT_COMM_ID_BYTES, "unique id size");
enum { kNcclSuccess = 0, kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0 };

struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

int load_rccl(RcclApi** out) {
    static RcclApi api;
    static int state = 0;                                    // 0 = not tried, 1 = ok, -1 = failed
    static std::string why;
    if (state == 0) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.h) break;
        }
        if (!api.h) { state = -1; why = std::string("dlopen(librccl): ") + dlerror(); }
        else {
            bool ok = true;
            auto sym = [&](const char* s) { void* p = dlsym(api.h, s); if (!p) { ok = false; why = std::string("librccl lacks ") + s; } return p; };
            api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
            api.ReduceScatter = (decltype(api.ReduceScatter))sym("ncclReduceScatter");
            api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
            api.Send = (decltype(api.Send))sym("ncclSend");
            api.Recv = (decltype(api.Recv))sym("ncclRecv");
            api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
            state = ok ? 1 : -1;
        }
    }
    if (state != 1) return fail(GAT_E_UNSUPPORTED, why);
    *out = &api;
    return 0;
}

#define GAT_NCCL(api, x)                                                                                   \
    do {                                                                                                   \
        const int r__ = (x);                                                                               \
        if (r__ != kNcclSuccess) return fail(GAT_E_COMM, std::string(#x) + ": " + (api)->GetErrorString(r__)); \
    } while (0)

struct RcclComm final : Comm {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int all_gather(float* table, int64_t slice, hipStream_t s) override {
        // in place: sendbuff == recvbuff + rank * sendcount
        GAT_NCCL(api, api->AllGather(table + (int64_t)rank * slice, table, (size_t)slice, kNcclFloat32, comm, s));
        return 0;
    }
    // part of every slice: one send + one receive per peer, grouped (the direct exchange over the xGMI mesh); in place
    int all_gather_part(float* table, int64_t slice, int64_t off, int64_t cnt, hipStream_t s) override {
        if (cnt <= 0 || world == 1) return 0;
        GAT_NCCL(api, api->GroupStart());
        for (int q = 0; q < world; ++q) {
            if (q == rank) continue;
            GAT_NCCL(api, api->Send(table + (int64_t)rank * slice + off, (size_t)cnt, kNcclFloat32, q, comm, s));
            GAT_NCCL(api, api->Recv(table + (int64_t)q * slice + off, (size_t)cnt, kNcclFloat32, q, comm, s));
        }
        GAT_NCCL(api, api->GroupEnd());
        return 0;
    }
    int reduce_scatter(float* table, int64_t slice, hipStream_t s) override {
        // in place: recvbuff == sendbuff + rank * recvcount
        GAT_NCCL(api, api->ReduceScatter(table, table + (int64_t)rank * slice, (size_t)slice, kNcclFloat32, kNcclSum, comm, s));
        return 0;
    }
    int all_reduce(float* buf, int64_t n, hipStream_t s) override {
        GAT_NCCL(api, api->AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, s));
        return 0;
    }
    int exchange_rows(const float* sendbuf, const int64_t* cnt_send, float* recvbuf, const int64_t* cnt_recv, int64_t rf, hipStream_t s) override {
        int64_t off_s = 0, off_r = 0;
        GAT_NCCL(api, api->GroupStart());
        for (int q = 0; q < world; ++q) {          // (world 1: the self pair, so that the path runs on a one-GPU box)
            if (cnt_send[q] > 0) GAT_NCCL(api, api->Send(sendbuf + off_s * rf, (size_t)(cnt_send[q] * rf), kNcclFloat32, q, comm, s));
            if (cnt_recv[q] > 0) GAT_NCCL(api, api->Recv(recvbuf + off_r * rf, (size_t)(cnt_recv[q] * rf), kNcclFloat32, q, comm, s));
            off_s += cnt_send[q]; off_r += cnt_recv[q];
        }
        GAT_NCCL(api, api->GroupEnd());
        return 0;
    }
    // bf16 travel format: one point-to-point send + receive per peer (xGMI is a full mesh: every pair has its own link),
    // bytes typed as int8 so that no RCCL reduction is involved; the fp32 sum happens on arrival (sum_arrivals_kernel)
    uint16_t* stage = nullptr; int64_t stage_elems = 0;           // [2][world][slice]: packed outgoing table | arrivals
    ~RcclComm() override { if (comm) (void)api->CommDestroy(comm); if (stage) (void)hipFree(stage); }
    int reduce_scatter_bf16(float* table, int64_t slice, hipStream_t s) override {
        const int64_t need = 2 * (int64_t)world * slice;
        if (need > stage_elems) {
            GAT_HIP(hipStreamSynchronize(s));
            if (stage) (void)hipFree(stage);
            stage = nullptr; stage_elems = 0;
            GAT_HIP(hipMalloc((void**)&stage, (size_t)need * sizeof(uint16_t)));
            stage_elems = need;
        }
        uint16_t* out = stage; uint16_t* in = stage + (int64_t)world * slice;
        GAT_TRY(launch_pack_bf16(table, out, (int64_t)world * slice, s));
        GAT_NCCL(api, api->GroupStart());
        for (int q = 0; q < world; ++q) {
            // world 1: a self send/recv pair through the staging buffers, so that the point-to-point symbols, the group and
            // the launch are exercised on a one-GPU box too (the arrival is not summed: q == rank keeps its fp32 partial)
            if (q == rank && world > 1) continue;
            GAT_NCCL(api, api->Send(out + (int64_t)q * slice, (size_t)slice * 2, kNcclInt8, q, comm, s));
            GAT_NCCL(api, api->Recv(in + (int64_t)q * slice, (size_t)slice * 2, kNcclInt8, q, comm, s));
        }
        GAT_NCCL(api, api->GroupEnd());
        return launch_sum_arrivals(table + (int64_t)rank * slice, in, world, rank, slice, s);
    }
};

// ---- host-staged transport ------------------------------------------------------------------------------
// The meeting point is a generation barrier on two words of the segment, polled with a deadline — not a
// pthread_barrier, which has no timeout: a rank whose peer died (or never reached the exchange) would wait for ever.
// On a timeout the rank also raises `failed`, which every other waiter checks, so that the whole group returns
// GAT_E_COMM within one polling interval instead of each rank running into its own deadline.
struct ShmHeader {
    uint32_t arrived;              // ranks that reached the current meeting
    uint32_t generation;           // bumped by the last arriver
    int32_t failed;                // set by the first rank that gave up; sticky
    int32_t world;
    int32_t ready;                 // set by rank 0 once the header is initialised
    int64_t bytes_per_rank;
};
static double comm_timeout_s() {   // GAT_COMM_TIMEOUT_S (default 120 s): how long a rank waits for its peers at an exchange
    static const double v = [] { const char* e = choice_env("GAT_COMM_TIMEOUT_S"); const double x = e ? atof(e) : 0.0; return x > 0.0 ? x : 120.0; }();
    return v;
}
constexpr size_t kShmHeader = 4096;

struct HostComm final : Comm {
    std::string name;
    char* base = nullptr;
    size_t total = 0;
    int64_t bytes_per_rank = 0;
    std::vector<float> tmp;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    float* area(int p) const { return reinterpret_cast<float*>(base + kShmHeader + (size_t)p * bytes_per_rank); }
    ~HostComm() override {
        if (dstage) (void)hipFree(dstage);
        if (base) munmap(base, total);
        if (rank == 0 && !name.empty()) shm_unlink(name.c_str());
    }
    int meet() {
        ShmHeader* h = hdr();
        if (__atomic_load_n(&h->failed, __ATOMIC_ACQUIRE)) return fail(GAT_E_COMM, "host transport: a peer gave up at an earlier exchange");
        const uint32_t gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
        if (__atomic_add_fetch(&h->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)world) {
            __atomic_store_n(&h->arrived, 0u, __ATOMIC_RELAXED);
            __atomic_store_n(&h->generation, gen + 1u, __ATOMIC_RELEASE);
            return 0;
        }
        struct timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int spin = 0;; ++spin) {
            if (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) != gen) return 0;
            if (__atomic_load_n(&h->failed, __ATOMIC_ACQUIRE)) return fail(GAT_E_COMM, "host transport: a peer gave up waiting (see its error)");
            if (spin > 2000) { struct timespec ts = {0, 200 * 1000}; nanosleep(&ts, nullptr); }       // 0.2 ms once the peers are clearly late
            if ((spin & 255) == 255) {
                struct timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double waited = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
                if (waited > comm_timeout_s()) {
                    __atomic_store_n(&h->failed, 1, __ATOMIC_RELEASE);
                    return fail(GAT_E_COMM, "host transport: rank " + std::to_string(rank) + " waited " + std::to_string((int)waited) +
                                                " s at an exchange for peers that did not arrive (dead or stuck rank; GAT_COMM_TIMEOUT_S)");
                }
            }
        }
    }
    int fits(int64_t floats) const {
        if (floats * (int64_t)sizeof(float) > bytes_per_rank)
            return fail(GAT_E_COMM, "host transport: exchange larger than bytes_per_rank given to gat_comm_init_host");
        return 0;
    }
    int all_gather(float* table, int64_t slice, hipStream_t s) override {
        GAT_TRY(fits(slice));
        GAT_HIP(hipMemcpyAsync(area(rank), table + (int64_t)rank * slice, slice * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        for (int p = 0; p < world; ++p)
            if (p != rank)
                GAT_HIP(hipMemcpyAsync(table + (int64_t)p * slice, area(p), slice * sizeof(float), hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
    int all_gather_part(float* table, int64_t slice, int64_t off, int64_t cnt, hipStream_t s) override {
        if (cnt <= 0) return 0;
        GAT_TRY(fits(slice));
        GAT_HIP(hipMemcpyAsync(area(rank) + off, table + (int64_t)rank * slice + off, cnt * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        for (int p = 0; p < world; ++p)
            if (p != rank)
                GAT_HIP(hipMemcpyAsync(table + (int64_t)p * slice + off, area(p) + off, cnt * sizeof(float), hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
    int sum_into_tmp(int64_t offset, int64_t n) {
        tmp.resize((size_t)n);
        const float* a0 = area(0) + offset;
        for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = a0[i];
        for (int p = 1; p < world; ++p) {
            const float* ap = area(p) + offset;
            for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] += ap[i];
        }
        return 0;
    }
    int reduce_scatter(float* table, int64_t slice, hipStream_t s) override {
        GAT_TRY(fits(slice * world));
        GAT_HIP(hipMemcpyAsync(area(rank), table, slice * world * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        GAT_TRY(sum_into_tmp((int64_t)rank * slice, slice));
        GAT_HIP(hipMemcpyAsync(table + (int64_t)rank * slice, tmp.data(), slice * sizeof(float), hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
    // the same travel format through the segment: every rank publishes its whole table rounded to bf16, then adds the
    // other ranks' versions of ITS slice to its own fp32 partial, in ascending rank order (what sum_arrivals_kernel does)
    uint16_t* dstage = nullptr; int64_t dstage_elems = 0;
    std::vector<uint16_t> hstage;
    int reduce_scatter_bf16(float* table, int64_t slice, hipStream_t s) override {
        const int64_t n = slice * world;
        GAT_TRY(fits((n + 1) / 2));
        if (n > dstage_elems) {
            GAT_HIP(hipStreamSynchronize(s));
            if (dstage) (void)hipFree(dstage);
            dstage = nullptr; dstage_elems = 0;
            GAT_HIP(hipMalloc((void**)&dstage, (size_t)n * sizeof(uint16_t)));
            dstage_elems = n;
        }
        GAT_TRY(launch_pack_bf16(table, dstage, n, s));
        GAT_HIP(hipMemcpyAsync(area(rank), dstage, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
        tmp.resize((size_t)slice);
        GAT_HIP(hipMemcpyAsync(tmp.data(), table + (int64_t)rank * slice, (size_t)slice * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        std::vector<float> own(tmp);
        for (int64_t i = 0; i < slice; ++i) {
            float acc = 0.f;
            for (int q = 0; q < world; ++q) {
                if (q == rank) { acc += own[(size_t)i]; continue; }
                const uint32_t u = (uint32_t)reinterpret_cast<const uint16_t*>(area(q))[(int64_t)rank * slice + i] << 16;
                float f; memcpy(&f, &u, sizeof(f));
                acc += f;
            }
            tmp[(size_t)i] = acc;
        }
        GAT_HIP(hipMemcpyAsync(table + (int64_t)rank * slice, tmp.data(), (size_t)slice * sizeof(float), hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
    // packed rows: every rank publishes [world] block offsets (in floats) followed by its send buffer; a receiver copies
    // the block addressed to it out of each peer's area
    int exchange_rows(const float* sendbuf, const int64_t* cnt_send, float* recvbuf, const int64_t* cnt_recv, int64_t rf, hipStream_t s) override {
        const int64_t hdr = ((int64_t)world * 2 + 15) / 16 * 16;           // header floats (int64 offsets), 64-byte aligned
        int64_t total = 0;
        for (int q = 0; q < world; ++q) total += cnt_send[q] * rf;
        GAT_TRY(fits(hdr + total));
        int64_t* offs = reinterpret_cast<int64_t*>(area(rank));
        int64_t o = 0;
        for (int q = 0; q < world; ++q) { offs[q] = o; o += cnt_send[q] * rf; }
        if (total > 0) GAT_HIP(hipMemcpyAsync(area(rank) + hdr, sendbuf, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        int64_t off_r = 0;
        for (int p = 0; p < world; ++p) {
            if (cnt_recv[p] > 0) {
                const int64_t po = reinterpret_cast<const int64_t*>(area(p))[rank];
                if (po < 0 || (hdr + po + cnt_recv[p] * rf) * (int64_t)sizeof(float) > bytes_per_rank)
                    return fail(GAT_E_COMM, "host transport: a peer's block offset is outside its area (mismatched halo plans)");
                GAT_HIP(hipMemcpyAsync(recvbuf + off_r * rf, area(p) + hdr + po, (size_t)(cnt_recv[p] * rf) * sizeof(float), hipMemcpyHostToDevice, s));
            }
            off_r += cnt_recv[p];
        }
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
    int all_reduce(float* buf, int64_t n, hipStream_t s) override {
        GAT_TRY(fits(n));
        GAT_HIP(hipMemcpyAsync(area(rank), buf, n * sizeof(float), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        GAT_TRY(meet());
        GAT_TRY(sum_into_tmp(0, n));
        GAT_HIP(hipMemcpyAsync(buf, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return meet();
    }
};

}  // namespace

int comm_unique_id(void* id_out) {
    if (!id_out) return fail(GAT_E_INVALID, "null id buffer");
    RcclApi* api = nullptr;
    GAT_TRY(load_rccl(&api));
    ncclUniqueId id;
    GAT_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int comm_create_rccl(int world, int rank, const void* id_bytes, Comm** out) {
    if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) return fail(GAT_E_INVALID, "gat_comm_init_rccl: bad arguments");
    RcclApi* api = nullptr;
    GAT_TRY(load_rccl(&api));
    auto c = std::make_unique<RcclComm>();
    c->api = api; c->world = world; c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    GAT_NCCL(api, api->CommInitRank(&c->comm, world, id, rank));
    *out = c.release();
    return 0;
}

int comm_create_host(int world, int rank, const char* shm_name, int64_t bytes_per_rank, Comm** out) {
    if (!shm_name || !out || world < 1 || rank < 0 || rank >= world || bytes_per_rank < 16)
        return fail(GAT_E_INVALID, "gat_comm_init_host: bad arguments");
    auto c = std::make_unique<HostComm>();
    c->world = world; c->rank = rank; c->name = shm_name;
    c->bytes_per_rank = (bytes_per_rank + 4095) & ~(int64_t)4095;
    c->total = kShmHeader + (size_t)world * c->bytes_per_rank;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(shm_name);
        fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(GAT_E_COMM, std::string("shm_open(") + shm_name + "): " + strerror(errno));
        if (ftruncate(fd, (off_t)c->total) != 0) { close(fd); return fail(GAT_E_COMM, std::string("ftruncate: ") + strerror(errno)); }
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; ++tries) {            // rank 0 creates it: wait up to 30 s
            fd = shm_open(shm_name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->total)) { close(fd); fd = -1; }
            if (fd < 0) { struct timespec ts = {0, 10 * 1000 * 1000}; nanosleep(&ts, nullptr); }
        }
        if (fd < 0) return fail(GAT_E_COMM, std::string("host transport: segment ") + shm_name + " did not appear");
    }
    void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(GAT_E_COMM, std::string("mmap: ") + strerror(errno));
    c->base = (char*)p;
    ShmHeader* h = c->hdr();
    if (rank == 0) {
        h->arrived = 0; h->generation = 0; h->failed = 0;       // (a fresh segment is zero-filled anyway)
        h->world = world; h->bytes_per_rank = c->bytes_per_rank;
        __atomic_store_n(&h->ready, 1, __ATOMIC_RELEASE);
    } else {
        for (int tries = 0; tries < 3000 && !__atomic_load_n(&h->ready, __ATOMIC_ACQUIRE); ++tries) {
            struct timespec ts = {0, 10 * 1000 * 1000}; nanosleep(&ts, nullptr);
        }
        if (!__atomic_load_n(&h->ready, __ATOMIC_ACQUIRE)) return fail(GAT_E_COMM, "host transport: rank 0 never initialised the segment");
        if (h->world != world || h->bytes_per_rank != c->bytes_per_rank) return fail(GAT_E_COMM, "host transport: ranks disagree on world / bytes_per_rank");
    }
    GAT_TRY(c->meet());
    *out = c.release();
    return 0;
}

}  // namespace gat
