// gat_internal.h — declarations shared by the HIP translation units behind include/gatv2_abi.h.
// gfx950 (MI355X) only: wave64, fp32 MFMA, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "gatv2_abi.h"

namespace gat {

constexpr int kWave = 64;

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// getenv for CHOICE switches (same results, other kernel / launch shape): records what was set for gat_switches().  Timing-only and
// wrong-result experiments (GAT_DBG) are not choice switches: they exist only in a -DGAT_EXPERIMENTS build (make experiments).
const char* choice_env(const char* name);

#define GAT_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::gat::fail((int)e__, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define GAT_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != 0) return rc__;   \
    } while (0)

// ---- edge-centric kernels (gat_edge_kernels.hip) ------------------------------------------------
struct EdgeFwdArgs {
    const int32_t* row_ptr;   // [n_rows+1]
    const int32_t* col_idx;   // [E] table row ids
    const float* PL;          // [n_table][HD]
    const float* PR;          // [n_rows][HD]
    const float* a;           // [HD]
    float* alpha;             // [E][H]; fast path: null = do not materialise (training path)
    float* score;             // [E][H] or null: raw attention scores (tap, E:323); only with alpha
    float* hpre;              // [n_rows][HD]
    float* hout;              // hidden [n_rows][HD]; last [n_rows][D]
    float* mstat;             // [n_rows][H] softmax max per (row, head); fast path: log2 domain
    float* zstat;             // [n_rows][H] softmax sum
    int64_t n_rows;
    int64_t n_table;          // rows of the gathered PL table (fast path: < 4 GiB, see edge_fast_path)
    int32_t bf16;             // PL (and msg) rows stored as bf16 (cfg.storage_dtype)
    int32_t H, D;
    int32_t is_last;
    float slope;
    // work items of the wave-per-item fast path (see WorkList); unused by the generic path
    const int4* items;        // [n_items] {row, beg, end, slot|-1}
    int64_t n_items;
    const int4* slot_info;    // [n_slots] {row, first_slot, nseg, item}
    int32_t n_slots;
    int32_t n_split;          // split rows: slot_info[n_slots + k].x = first slot of the k-th
    float* part_acc;          // [n_slots][HD]  per-segment partial sums of split rows
    float* part_mz;           // [n_slots][2H]  per-segment (max, sum)
};

// Work decomposition of a CSR graph for the wave-per-item kernels: every row with <= kSegEdges
// in-edges is one item, longer rows (power-law hubs) are cut into segments of kSegEdges edges
// (slot >= 0) that are listed first; slot_info describes the split rows for the fix-up kernels.
constexpr int kSegEdges = 256;      // swept 64..512 on the Products shape (GAT_SEG_EDGES): 256 best by ~0.5 %
struct WorkList {
    std::vector<int32_t> items;       // 4 per item
    std::vector<int32_t> slot_info;   // 4 per slot {row, first slot, segments, item}; then 4 per SPLIT ROW {first slot,-,-,-}
    int64_t n_items = 0;
    int32_t n_slots = 0;
    int32_t n_split = 0;              // rows cut into segments (entries n_slots .. n_slots+n_split-1 of slot_info)
};
void build_worklist(const int32_t* row_ptr, int64_t n_rows, WorkList& w);
int launch_edge_forward(const EdgeFwdArgs& a, hipStream_t s);

struct EdgeBwdArgs {
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const float* PL;          // [n_table][HD]
    const float* PR;          // [n_rows][HD]
    const float* a;           // [HD]
    const float* alpha;       // [E][H]  generic path only (the fast path recomputes it from mstat/zstat)
    const float* mstat;       // [n_rows][H] forward softmax max (fast path: log2 domain)
    const float* zstat;       // [n_rows][H] forward softmax sum
    const float* hpre;        // [n_rows][HD]
    const float* g;           // [n_rows][HD]  dL/dh_pre — or, with g_raw, dL/d(layer output): the edge kernels then
                              // apply LReLU'(h_pre) on load (E:888-892 folded into the consumer, which reads h_pre anyway)
    int32_t g_raw;
    const float* gh;          // last layer, or null: [n_rows][D] head-independent output gradient; the kernels form
                              // g[n,h,d] = gh[n,d] * LReLU'(h_pre[n,h,d]) / H themselves (E:598-603) and ignore `g`
    float* gPL;               // [n_table][HD]  atomics path only: zeroed by the caller, added into
    const int32_t* pos;       // [E] CSC slot of every edge, or null = atomics path
    float* msg;               // [E][HD] message rows by slot (store path; summed by launch_gpl_sum)
    uint32_t* stash;          // or: [E + 1][HD/N] per-edge records by slot (stash path, see edge_stash_words; launch_gpl_pull)
    uint32_t stash_spare;     // index of the spare record behind the last slot (= E): where padded lanes store
    float* gfull;             // [n_rows][HD] written with the stash path: dL/dh_pre incl. the LReLU' factor (gathered by launch_gpl_pull)
    int32_t gh_stride;        // floats between consecutive rows of gh (D, or 16 when gh rows and their decision bytes share 64-B records)
    int32_t hb_stride;        // bytes between consecutive rows of hbits
    uint8_t* hbits;           // last layer (gh != null), or null: [n_rows][HD/N] one byte per lane = the LReLU'(h_pre) decisions of its
                              // N channels; written INSTEAD of gfull — the pull pass rebuilds g = gh * LReLU'(h_pre) / H from gh
                              // (32 B per node at D = 8) and these 16 B instead of gathering a 256-B row of g
    float* gPR;               // [n_rows][HD]   written
    float* ge;                // [E][H] or null (tap)
    float* galpha;            // [E][H] or null (tap, E:646); only with ge
    float* ga_partial;        // [ga_blocks][HD] written
    int32_t ga_blocks;        // grid size the launcher must use (== rows of ga_partial)
    int64_t n_rows;
    int64_t n_table;          // rows of the gathered PL table (fast path: < 4 GiB, see edge_fast_path)
    int32_t bf16;             // PL (and msg) rows stored as bf16 (cfg.storage_dtype)
    int32_t H, D;
    float slope;
    const int4* items;        // as in EdgeFwdArgs
    int64_t n_items;
    const int4* slot_info;
    int32_t n_slots;
    int32_t n_split;          // split rows: slot_info[n_slots + k].x = first slot of the k-th
    float* part_acc;          // [n_slots][HD]  per-segment gPR partials of split rows
    int32_t dbg;
};
int launch_edge_backward(const EdgeBwdArgs& a, hipStream_t s);
// Last layer, fused per row (gat_step): forward edge pass + output head + backward edge pass of every WHOLE row in one kernel
// (edge_last_fused_kernel); f / b as for the separate passes (f.items / f.n_items: the whole-row items only; b.ga_partial /
// b.ga_blocks from edge_last_fused_blocks), gh_out = the [n_rows][gh_stride] node records the pull pass reads.
struct EdgeLastArgs {
    EdgeFwdArgs f;
    EdgeBwdArgs b;
    const float* Wo;          // [C][D]
    const int32_t* labels;    // [n_rows]; negative = outside the training mask
    float* gh_out;            // [n_rows][b.gh_stride]
    int32_t C;
};
bool edge_last_fused_supported(int32_t H, int32_t D, int32_t C);
int edge_last_fused_blocks(int64_t n_items);
int launch_edge_last_fused(const EdgeLastArgs& a, hipStream_t s);
// gH = Wo^T dz of the split rows only (their forward runs as segments): slot_info as in EdgeFwdArgs
int launch_head_rows(const int4* slot_info, int32_t n_slots, int32_t n_split, const float* Wo, const float* HL, const int32_t* labels,
                     float* gh_out, int32_t gh_stride, int32_t C, int32_t DLAST, hipStream_t s);
// Words per edge record of the stash path for an (H, D) layer, 0 = that shape has no stash path (the message-row
// path is used): two lanes per head are needed (D = 8 with four channels per lane, D = 4 with two).
int edge_stash_words(int32_t H, int32_t D);
// Grid size (== rows of ga_partial) for the backward of an (H, D) layer over n_items work items.
int edge_backward_blocks(int64_t n_items, int32_t H, int32_t D, bool store, bool taps, bool bf16, bool stash = false);
// wave-per-row templates cover this (H, D), and the gathered table is < 4 GiB (they address it as
// uniform base + 32-bit byte offset); anything else runs the generic kernels
bool edge_fast_path(int32_t H, int32_t D, int64_t n_table);

// ---- source-major slot index + segmented sum (gat_csc.hip) ----------------------------------------
// csrc (optional): [n_edges + kPullPad] table row of every slot (the sorted keys), -1 in the padding
int build_csc(const int32_t* col_idx, int64_t n_edges, int64_t n_table, int32_t* pos, int32_t* src_ptr,
              int32_t* csrc, hipStream_t s);
// sources with long slot lists, cut into chunks (gat_csc.hip kHeavySlots): built once per graph from the CSC pointers
struct HeavyList {
    std::vector<int32_t> chunks;    // int4 {first slot, end slot, partial row, -}
    std::vector<int32_t> heavy;     // int4 {source, first partial, partial count, -}
    int32_t threshold = 0;          // slots above which a list is chunked (depends on the graph size)
    std::vector<int32_t> items;     // int4 {source, first slot, end slot, partial row | -1}: chunks first, then the other sources by length
    // slot-parallel form (SlotRuns): run length (0 = not built), number of runs, lists crossing a run boundary, rows without slots
    int32_t run = 0; int64_t n_runs = 0;
    std::vector<int32_t> open;      // int4 {source, first run, last run, -}
    std::vector<int32_t> empty;     // table rows without slots
};
int build_heavy_list(const int32_t* d_src_ptr, int64_t n_table, int64_t n_edges, HeavyList* out, hipStream_t s, int32_t run = 0);
// Slot-parallel ("runs") form of the source-major pass (gat_csc.hip gpl_pull_runs_kernel): the flat slot stream cut into runs of
// `run` slots, one per lane group, segmented by a streamed source id per slot.  Device-side index, built once per graph.
struct SlotRuns {
    const int32_t* csrc = nullptr;      // [n_slots + kPullPad] table row of every slot, -1 behind the last
    int32_t run = 0;                    // slots per run (a multiple of 32)
    int64_t n_runs = 0;
    const int4* open = nullptr; int32_t n_open = 0;       // lists crossing a run boundary {source, first run, last run, -}
    const int32_t* empty = nullptr; int64_t n_empty = 0;  // table rows without slots
    float* part = nullptr;              // [2 n_runs][HDmax] partial rows of the runs' open segments
};
int launch_gpl_sum(const int32_t* src_ptr, const float* msg, float* gPL, int64_t n_table, int64_t n_slots,
                   int32_t HD, bool msg_bf16, const int4* chunks, int32_t n_chunks, const int4* heavy,
                   int32_t n_heavy, float* part, hipStream_t s, const SlotRuns* runs = nullptr);
// slots of padding behind the record buffer and the destination list: the pull pass reads whole 16-slot chunks (and one chunk
// of destinations ahead) without clamping its indices
// REQUIREMENT on callers of launch_gpl_pull: pass the allocated slot count as slot_capacity; below n_slots + kPullPad the clamped
// first form is used.  The padding's CONTENT is not relied on (slots beyond a list are zeroed after their load), only its presence;
// the gathered tables are addressed with 32-bit byte offsets (n_table * H*D * 4 < 4 GiB: edge_fast_path; node records: n_rows < 2^26).
constexpr int64_t kPullPad = 32;
// cdst[pos[e]] = destination row of CSR edge e (the source-major twin of a1's dst array; built once per graph)
int build_csc_dst(const int32_t* row_ptr, const int32_t* pos, int32_t* cdst, int64_t n_rows, int64_t n_edges, hipStream_t s);
// Stash path: gPL[s][:] = sum over the slots of s of  g[cdst][:] * alpha + ge * a (.) LReLU'  rebuilt from the records
// gh / hbits non-null (last layer): g rows are rebuilt from gh [n_rows][D] and the per-lane decision bytes instead of read from gfull
int launch_gpl_pull(const int32_t* src_ptr, const uint32_t* stash, const int32_t* cdst, const float* gfull, bool g_bf16,
                    const float* gh, const uint8_t* hbits, int32_t gh_stride, int32_t hb_stride, const float* a,
                    float slope, float* gPL, int64_t n_table, int64_t n_slots, int32_t H, int32_t D, const int4* chunks,
                    int32_t n_chunks, const int4* heavy, int32_t n_heavy, float* part, const int4* items, int64_t n_items, const SlotRuns* runs,
                    int64_t slot_capacity /* slots allocated in stash AND cdst: n_slots + kPullPad enables the unclamped forms */, hipStream_t s);

int launch_csr_to_coo(const int32_t* row_ptr, const int32_t* col_idx, int32_t* src, int32_t* dst,
                      int64_t n_rows, int64_t n_edges, int64_t table_row0, hipStream_t s);

// out[c] += sum_b partial[b][c]   (deterministic: one thread per c, ascending b)
// ---- exchange transports (gat_comm.hip) ---------------------------------------------------------------
struct Comm {
    int world = 1, rank = 0;
    virtual ~Comm();
    // table: [world][slice] floats; the rank's slice already written / afterwards holding the sum
    virtual int all_gather(float* table, int64_t slice, hipStream_t s) = 0;
    // the same for floats [off, off+cnt) of EVERY rank's slice only (chunk-pipelined forward exchange)
    virtual int all_gather_part(float* table, int64_t slice, int64_t off, int64_t cnt, hipStream_t s) = 0;
    virtual int reduce_scatter(float* table, int64_t slice, hipStream_t s) = 0;
    // The same exchange with the REMOTE partial sums travelling as bf16 (half the xGMI volume): every rank rounds the
    // slices it sends to bf16 (nearest even), keeps its own partial in fp32, and adds what arrives in fp32 in ascending
    // rank order.  Option GAT_COMM_GPL_BF16; relative error of the summed rows ~ 2^-9 per remote term.
    virtual int reduce_scatter_bf16(float* table, int64_t slice, hipStream_t s) = 0;
    virtual int all_reduce(float* buf, int64_t n, hipStream_t s) = 0;
    // Pairwise exchange of packed rows (halo form, gat_halo.hip): this rank sends cnt_send[q] rows of rf floats to every rank q
    // (sendbuf grouped by q, ascending) and receives cnt_recv[q] rows from it (recvbuf grouped the same way); the counts (host
    // arrays of `world` entries) are fixed at set-up and consistent between the two ends.  One send / receive pair per peer.
    virtual int exchange_rows(const float* sendbuf, const int64_t* cnt_send, float* recvbuf, const int64_t* cnt_recv, int64_t rf, hipStream_t s) = 0;
};
// Halo form of the table exchanges (gat_halo.hip): per peer, the rows of its slice this shard's edges reference.
struct HaloPlan {
    int world = 0, rank = 0;
    int64_t max_rows = 0;
    std::vector<int64_t> need_cnt, send_cnt;      // [world] rows received from / sent to each rank in the forward exchange
    int64_t n_need = 0, n_send = 0, buf_rows = 0;
    int32_t* need_rows = nullptr;                 // device [n_need] table rows, grouped by owner (ascending)
    int32_t* send_rows = nullptr;                 // device [n_send] rows of the own slice (local ids), grouped by requester (ascending)
    int32_t *arr_ptr = nullptr, *arr_split = nullptr, *arr_pos = nullptr;      // backward sum: arrivals per own row, lower ranks first
    float *sendbuf = nullptr, *recvbuf = nullptr; // [buf_rows][max H*D]
    double referenced_fraction = 1.0;             // rows on the wire / rows of the full exchange, over all ranks
};
int halo_build(Comm* comm, const int32_t* d_col_idx, int64_t n_edges, int64_t n_table, int32_t hd_max, HaloPlan* out, hipStream_t s);
int halo_forward(Comm* comm, const HaloPlan& h, float* table, int64_t rf, hipStream_t s);
int halo_backward(Comm* comm, const HaloPlan& h, float* gPL, int64_t rf, hipStream_t s);
void halo_free(HaloPlan* h);
int comm_unique_id(void* id_out);
int comm_create_rccl(int world, int rank, const void* id_bytes, Comm** out);
int comm_create_host(int world, int rank, const char* shm_name, int64_t bytes_per_rank, Comm** out);

int launch_pack_result(const float* loss, const int32_t* correct, float* dst3, hipStream_t s);
// Deferred slab reductions.  Between reduce_batch_begin() and reduce_batch_flush() (same host thread) launch_reduce_gradw /
// launch_reduce_partials_add queue their job instead of launching it, and the flush runs all of them — plus the result
// pack, if asked for — as ONE kernel: on Cora / Pubmed-size graphs every launch is ~5 us of a ~250 us step.  The slabs
// of a queued job must stay untouched until the flush (per-layer scratch regions: gat_abi.hip).
struct ReduceJob { const float* slabs; float* out; int64_t width; int32_t nslabs, HD, F, c_base, block0; };   // HD == 0: out[idx]
struct ReduceBatch {
    ReduceJob jobs[8]; int32_t n, blocks;
    const float* loss; const int32_t* correct; float* dst3; float* dst3_host;      // result pack (dst3_host: optional pinned copy)
    const double* fin_loss; const int32_t* fin_corr; int32_t fin_n; float* fin_loss_out; int32_t* fin_corr_out;   // head finalize
};
void reduce_batch_begin();
void reduce_batch_abort();
void reduce_batch_pack(const float* loss, const int32_t* correct, float* dst3, float* dst3_host);
// the output head's block partials -> {loss, #correct} (head_finalize_kernel) inside the batch kernel's pack block; false = not queued
bool reduce_batch_finalize(const double* loss_partial, const int32_t* correct_partial, int32_t nblocks, float* loss_out, int32_t* correct_out);
int reduce_batch_flush(hipStream_t s);
int launch_reduce_partials_add(const float* partial, int32_t nblocks, int64_t width, float* out,
                               hipStream_t s);

// gradW[(c%HD)][(c/HD)*F + f] += sum_z slabs[z][c - c_base][f]   (c over c_base .. c_base+M-1, fixed order)
int launch_reduce_gradw(const float* slabs, int32_t ksplit, int32_t HD, int32_t F, int32_t c_base, int32_t M,
                        float* gradW, hipStream_t s);

// Blocks of a 256-thread kernel that are resident on the whole chip at once (occupancy API incl.
// dynamic LDS; cached).  Persistent kernels use exactly this grid: a larger one runs a second,
// partly filled round of blocks.
int64_t resident_blocks(const void* fn, size_t dyn_lds, int threads = 256);

// layout converters for taps / op-level entry points
int launch_transpose_eh_to_he(const float* src_eh, float* dst_he, int64_t E, int32_t H, hipStream_t s);
int launch_transpose_he_to_eh(const float* src_he, float* dst_eh, int64_t E, int32_t H, hipStream_t s);
int launch_transpose_nh_to_hn(const float* src_nh, float* dst_hn, int64_t N, int32_t H, hipStream_t s);

// ---- dense kernels (gat_dense_kernels.hip) -------------------------------------------------------
// PL[table_row0 + i][j] = sum_f X[i][f] * W[j][f],  PR[i][j] = sum_f X[i][f] * W[j][F+f]
// W in reference layout [HD][2F].
// part: both halves in one pass over X, or only the W_left (PL) / W_right (PR) half — the two halves run over
// different row sets when the layer input is replicated on every shard (gat_set_source_features).
enum : int32_t { kPartBoth = 0, kPartLeft = 1, kPartRight = 2 };
// pl_bf16: PL_rows points at bf16 rows ([.][HD] of 2 bytes) — cfg.storage_dtype
// scratch / scratch_floats: a buffer of at least project_scratch_floats(n_rows, F, HD, part) floats (0 = this launch never
// needs one), or null; few rows with a long K (Cora / Pubmed shapes) take a split-K path through it.  A scratch smaller than
// this launch needs (or null) selects the streaming kernel — never an overrun.
int64_t project_scratch_floats(int64_t n_rows, int32_t F, int32_t HD, int32_t part);
// ldx: floats between consecutive rows of X (0 = F).  A pitch that is a multiple of 4 with ZEROS behind column F lets the kernels use
// 16-byte loads for an odd F (a float4 that starts below F may read up to three padding floats: they must be finite).
int launch_project(const float* X, const float* W, float* PL_rows, float* PR, int64_t n_rows,
                   int32_t F, int32_t HD, int32_t part, bool pl_bf16, float* scratch, int64_t scratch_floats, hipStream_t s, int32_t ldx = 0);
// gradW[j][0:F] += sum_n gPL[n][j] X[n][:],  gradW[j][F:2F] += sum_n gPR[n][j] X[n][:]
// scratch: at least grad_w_scratch_floats(n_rows, F, HD) floats.
int64_t grad_w_scratch_floats(int64_t n_rows, int32_t F, int32_t HD);
int launch_grad_w(const float* gPL_rows, const float* gPR, const float* X, float* gradW,
                  float* scratch, int64_t n_rows, int32_t F, int32_t HD, int32_t part, hipStream_t s, int32_t ldx = 0);
// gprev[n][f] = (sum_j gPL[n][j] W[j][f] + gPR[n][j] W[j][F+f]) * LReLU'(hpre_prev[n][f])
// hpre_prev == nullptr: the plain sum is stored (the consumer applies LReLU', see EdgeBwdArgs::g_raw)
int launch_grad_x(const float* gPL_rows, const float* gPR, const float* W, const float* hpre_prev,
                  float* gprev, int64_t n_rows, int32_t F, int32_t HD, float slope, hipStream_t s);

struct HeadArgs {
    const float* Wo;          // [C][DL]
    const float* HL;          // [n_rows][DL]  post-activation last layer output
    const int32_t* labels;    // [n_rows]
    float* y;                 // [n_rows][C]
    double* loss_partial;     // [blocks]
    int32_t* correct_partial; // [blocks]
    float* loss_out;          // [1] device
    int32_t* correct_out;     // [1] device
    int64_t n_rows;
    int32_t C, DL;
};
int head_blocks(int64_t n_rows);
int launch_head_forward(const HeadArgs& a, hipStream_t s);

struct HeadBwdArgs {
    const float* Wo;          // [C][DL]
    const float* HL;          // [n_rows][DL]
    const float* y;           // [n_rows][C]
    const int32_t* labels;
    const float* hpre;        // last layer pre-activation [n_rows][H][DL]
    float* g;                 // [n_rows][H][DL], or null: only gh_out is written and the edge backward of the last
                              // layer expands it (EdgeBwdArgs::gh) — saves writing 4*N*H*D and reading h_pre here
    float* gh_out;            // [n_rows][DL]  Wo^T dz  (the head-independent part of g), or null
    int32_t gh_stride;        // floats between consecutive rows of gh_out (0 = DL)
    float* gradWo;            // [C][DL] added into
    float* partial;           // [head_bwd_blocks][C*DL]
    int64_t n_rows;
    int32_t C, DL, H;
    float slope;
    int32_t flat_index;
};
int head_bwd_blocks(int64_t n_rows, int32_t C, int32_t DL);
int launch_head_backward(const HeadBwdArgs& a, hipStream_t s);
// forward + backward of the output head in one kernel (gat_step; y is not stored)
bool head_step_supported(const HeadBwdArgs& b);
int launch_head_step(const HeadArgs& f, const HeadBwdArgs& b, hipStream_t s);

// train / validation masks (README R:134 "later"): labels_eff = mask ? label : ~label — the head kernels skip negative
// labels (no loss, no count, dz = 0); eval_mask: fixed-order block partials of loss / #correct / #nodes over a mask
int launch_apply_mask(const int32_t* labels, const uint8_t* mask, int32_t* eff, int64_t n, hipStream_t s);
int launch_eval_mask(const float* y, const int32_t* labels, const uint8_t* mask, int64_t n, int32_t C, double* part_loss,
                     int32_t* part_cnt, int32_t blocks, hipStream_t s);
// p[i] = the (draw0+i)-th draw of the seeded counter stream mapped to (-lim, lim]  (Xavier-uniform, E:186-248)
int launch_xavier_init(float* p, int64_t n, uint64_t s0, uint64_t draw0, float lim, hipStream_t s);
int launch_sgd(float* p, const float* g, float lr, int64_t n, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, float lr, int64_t n, float b1, float b2,
                float eps, int32_t t, hipStream_t s);
// per-group clip (E:250-278) with the norm kept on the device; scratch >= 1 float
int launch_clip(float* g, int64_t n, float thresh, float* scratch, hipStream_t s);

}  // namespace gat
