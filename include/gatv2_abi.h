/* gatv2_abi.h — C ABI of the MI355X-native GATv2 edge-centric hot path (libgatv2_hip.so).
 *
 * The reference (GATv2_edge_based.cu, cited E:<line>) has no FFI/plugin seam: `main` launches
 * its kernels inline (E:1370-1642).  This header defines the seam at exactly those launch
 * sites; every entry point names the reference launch(es) it replaces.  Plain C: pointers and
 * sizes only, no C++/torch types.  All functions return 0 on success, otherwise a non-zero
 * status (a hipError_t value, or one of GAT_E_*); gat_last_error() gives the message.
 *
 * Layouts at the boundary are the REFERENCE layouts:
 *   W   flat [l][H_l][D_l][2*F_l]  (cols 0..F-1 multiply x_src, F..2F-1 x_dst; E:294-316)
 *   a   flat [l][H_l][D_l]                                                    (E:296)
 *   Wo  [C][D_last]                                                           (E:464)
 *   edge tensors [H][E] head-major (E:297), node tensors [N][H][D] (E:410)
 * Inside the context tensors live in MI355X-friendly layouts (edge tensors [E][H], projected
 * features PL/PR [N][H*D]); gat_tap() converts back.
 *
 * Threading: a context is not thread-safe; one context per GPU / per rank.
 * Ownership: the caller owns host arrays and the gat_ctx*; the context owns its device memory
 * except buffers handed in through gat_bind_table() (borrowed, never freed).
 */
#ifndef GATV2_ABI_H
#define GATV2_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAT_ABI_VERSION 6

enum {
    GAT_OK = 0,
    GAT_E_INVALID = 10001,   /* bad argument / shape */
    GAT_E_STATE = 10002,     /* call order (e.g. forward before set_graph) */
    GAT_E_NOMEM = 10003,
    GAT_E_UNSUPPORTED = 10004,
    GAT_E_COMM = 10005       /* exchange transport (RCCL / host-staged) */
};

typedef struct gat_ctx gat_ctx;

/* Mirrors the reference's CLI model config (E:934-1010, 1115-1118, 1143). */
typedef struct gat_config {
    int32_t num_layers;        /* --num-layers */
    const int32_t* heads;      /* --heads,   [num_layers] */
    const int32_t* outdims;    /* --outdims, [num_layers] */
    int32_t in_dim;            /* input_feature_vector_dim (E:1079) */
    int32_t num_classes;       /* max(label)+1 (E:1107) */
    float negative_slope;      /* 0.01f (E:1143) */
    int32_t device;            /* HIP device ordinal */
    void* stream;              /* hipStream_t to run on; NULL = context creates its own */
    int32_t flat_lrelu_index;  /* 0 = exact per-head LReLU' in the output gradient (default);
                                  1 = the reference's flat index n*D+d (E:598) */
    int32_t collect_timing;    /* 1 = bracket every kernel with hipEvents (gat_kernel_stats) */
    int32_t keep_taps;         /* 1 = also keep tensors only parity tests read (ge, max, sum) */
    int32_t storage_dtype;     /* GAT_DTYPE_F32 (default) | GAT_DTYPE_BF16: the projected source table PL (gathered
                                  per edge, exchanged between shards) and the per-edge message rows are STORED as
                                  bf16; every sum, the softmax and all other tensors stay fp32.  BASELINE config 5
                                  ("bf16"); parity bar vs the fp32 oracle: alpha, loss 1e-2. */
} gat_config;
enum { GAT_DTYPE_F32 = 0, GAT_DTYPE_BF16 = 1 };

const char* gat_last_error(void);
int gat_abi_version(void);
int gat_device_count(int* count);
/* The environment switches this process has read and found set, as "NAME=VALUE NAME=VALUE" (empty string: none).  They select among
 * kernels / launch shapes that compute the same results (A/B measurements and the test matrices use them) — never a
 * different result: the timing-only experiment variants (GAT_DBG) exist only in the experiment library (`make experiments`,
 * libgatv2_hip_exp.so, whose text starts with "[experiment library]").  buf: at least a few hundred bytes. */
int gat_switches(char* buf, int64_t cap);

/* ---- lifecycle (replaces the inline cudaMalloc plan, E:1151-1357) ------------------------- */
int gat_create(const gat_config* cfg, gat_ctx** out);
int gat_destroy(gat_ctx* ctx);
int gat_sync(gat_ctx* ctx);                       /* wait for the context's stream */
int gat_mem_info(size_t* free_bytes, size_t* total_bytes);   /* cudaMemGetInfo, E:930, 1362 */

/* ---- data (H2D copies E:1151-1172; CSR->COO E:1186) --------------------------------------
 * Rows are destinations, columns are sources (E:74-82).  Single GPU: n_table == n_rows,
 * table_row0 == 0.  Destination-range shard: the context owns rows
 * [table_row0, table_row0+n_rows) of an n_table-row source table and col_idx holds table row
 * ids (see INTEGRATION.md "sharding"); host arrays are copied. */
/* Limits: n_edges, n_rows and n_table must each fit int32 (the reference's CSR is int32 too, E:1045-1046; the
 * one-time source-major index is built with a 32-bit-count radix sort) — larger graphs are refused with
 * GAT_E_UNSUPPORTED, never truncated.  Offsets INTO tensors are 64-bit everywhere.  The wave-per-row kernels
 * address the gathered PL table as base + 32-bit byte offset: a table of 4 GiB or more (n_table * H*D * 4 bytes, i.e.
 * > 16.7 M rows at H*D = 64) runs on the generic kernels instead — same results, float atomics, several times slower
 * (no BASELINE config is that large per GPU; bf16 storage is refused there). */
int gat_set_graph(gat_ctx* ctx, const int32_t* row_ptr, const int32_t* col_idx, int64_t n_rows,
                  int64_t n_edges, int64_t n_table, int64_t table_row0);
/* Which edge kernels layer `layer` runs on (known once the graph is set).  The size cliff above is never silent: the
 * gat_set_* call that completes the context (graph + features + labels) still returns 0, but leaves a text starting with
 * "warning:" in gat_last_error() when a layer is GAT_PATH_GENERIC_SIZE (one text naming every affected layer; any later
 * failing call replaces it).  gat_last_error() is a per-thread last-message slot, so the durable way to find out is this query:
 * call it for every layer after the context is complete (train_edge does, and prints the warning once to stderr). */
enum {
    GAT_PATH_GENERIC_SHAPE = 0,   /* (H, D) outside the wave-per-row templates (H*D not in {8,16,32,64} or D not a power of two) */
    GAT_PATH_FAST = 1,            /* wave-per-row / group-per-row kernels, no atomics */
    GAT_PATH_GENERIC_SIZE = 2     /* the shape has fast kernels, but the gathered table is >= 4 GiB */
};
int gat_layer_path(gat_ctx* ctx, int32_t layer, int32_t* path);
/* Non-finite inputs: the reference's plain fp32 loops (E:303-316) turn a +-inf feature into +-inf or NaN sums; every row
 * that aggregates the node within the model's layers ends with NaN class probabilities, and the loss stays finite because
 * E:527 clamps with fmaxf(prob, 1e-12f), which drops a NaN.  Here the dense products cut every fp32 operand into three bf16
 * pieces (x - bf16(x) is inf - inf for an infinite x), so a non-finite — or within half a bf16 ulp of FLT_MAX — feature or
 * weight yields NaN at once where the reference may still hold +-inf: a row the reference rescues (an edge whose score is
 * -inf gets attention 0 there) can be NaN here.  The poisoned rows are a superset of the reference's and a subset of the
 * node's L-hop neighbourhood; all other rows are unaffected; nothing hangs or faults (tests/test_nonfinite.py). */
int gat_set_features(gat_ctx* ctx, const float* x, int64_t n_rows, int32_t in_dim);   /* [n_rows][F0] */
int gat_set_labels(gat_ctx* ctx, const int32_t* labels, int64_t n_rows);
/* Train / validation / test splits — beyond the reference, which trains and evaluates on ALL nodes (E:514-537,
 * README R:134 "later").  mask[n_rows]: 1 = the node belongs to the split.  With a training mask, loss, #correct and
 * the output gradient dz = y - onehot (E:571-573) are taken over the masked nodes only (every node still takes part in
 * the message passing); NULL restores the reference's behaviour.  gat_eval_mask: loss sum / #correct / #nodes of the
 * LAST forward over another split (call after gat_forward or gat_step). */
int gat_set_train_mask(gat_ctx* ctx, const uint8_t* mask, int64_t n_rows);
int gat_eval_mask(gat_ctx* ctx, const uint8_t* mask, int64_t n_rows, double* loss_sum, int32_t* n_correct, int32_t* n_nodes);
/* Device-resident variants (pointers on ctx's device; copied D2D). */
int gat_set_graph_device(gat_ctx* ctx, const int32_t* d_row_ptr, const int32_t* d_col_idx,
                         int64_t n_rows, int64_t n_edges, int64_t n_table, int64_t table_row0);
int gat_set_features_device(gat_ctx* ctx, const float* d_x, int64_t n_rows, int32_t in_dim);
int gat_set_labels_device(gat_ctx* ctx, const int32_t* d_labels, int64_t n_rows);
/* Replicated layer-0 input for a shard (call INSTEAD of gat_set_features, after gat_set_graph):
 * the features of every row of the source table, [n_table][F0] (padding rows: zeros).  The input
 * features are static (read-only in the reference too, E:1151), so each shard can project the
 * whole layer-0 table itself and accumulate gradW_left of layer 0 from its partial gPL table;
 * layer 0 then needs no PL all-gather and no gPL reduce-scatter (gat_layer_exchange reports 0) —
 * the packed parameter-gradient all-reduce completes the sum over shards. */
int gat_set_source_features(gat_ctx* ctx, const float* x_table, int64_t n_table, int32_t in_dim);
int gat_set_source_features_device(gat_ctx* ctx, const float* d_x_table, int64_t n_table, int32_t in_dim);

/* ---- parameters (Xavier init E:186-248; flat layouts E:1242-1258) --------------------------- */
enum { GAT_PARAM_W = 0, GAT_PARAM_A = 1, GAT_PARAM_WO = 2 };
int gat_param_count(gat_ctx* ctx, int group, int64_t* count);
int gat_params_init(gat_ctx* ctx, uint64_t seed);     /* U(-lim,lim], lim as E:208, 236 */
int gat_params_set(gat_ctx* ctx, int group, const float* host, int64_t count);
int gat_params_get(gat_ctx* ctx, int group, float* host, int64_t count);
int gat_grads_get(gat_ctx* ctx, int group, float* host, int64_t count);
int gat_grads_set(gat_ctx* ctx, int group, const float* host, int64_t count);
/* Device address of the packed gradient buffer [gradW | grada | gradWo] (for the all-reduce). */
int gat_grads_device(gat_ctx* ctx, void** d_ptr, int64_t* count);
/* Async D2D copies of the packed gradients on the context's stream, to / from a caller-owned
 * device buffer of `count` floats (the buffer the host all-reduces). */
int gat_grads_export(gat_ctx* ctx, void* d_dst, int64_t count);
int gat_grads_import(gat_ctx* ctx, const void* d_src, int64_t count);
/* Async: the last gat_head_forward's results as three floats at d_dst3 — [loss_sum (E:542),
 * n_correct & 4095, n_correct >> 12] — so the host can append them to the packed gradients and
 * sum everything over shards in ONE float all-reduce, with no mid-step device sync
 * (n_correct = lo + 4096*hi after the sum; exact). */
int gat_result_export(gat_ctx* ctx, void* d_dst3);

/* ---- the step (world == 1): epoch body E:1374-1557 ------------------------------------------- */
/* forward over all layers + output head + loss; returns sum loss (E:542) and #correct (E:543). */
int gat_forward(gat_ctx* ctx, float* loss_sum, int32_t* n_correct);
int gat_backward(gat_ctx* ctx);                   /* E:1463-1557; adds into the grad buffers */
int gat_zero_grad(gat_ctx* ctx);                  /* E:1631-1637 */
int gat_clip(gat_ctx* ctx, float threshold);      /* clip_grad_norm x3, E:1561-1567 */
int gat_step_sgd(gat_ctx* ctx, float lr);         /* E:1601-1624 */
int gat_step_adam(gat_ctx* ctx, float lr, float beta1, float beta2, float eps, int32_t t);  /* E:1571-1596 */

/* ---- the step in phases (any world size; the host runs the exchange between phases) ---------
 * forward, layer l:  project -> [all-gather PL table] -> forward_edges
 * backward, layer l: backward_edges -> [reduce-scatter gPL table] -> backward_dense
 * after the last backward phase: [all-reduce packed grads, loss scalars]. */
int gat_layer_project(gat_ctx* ctx, int32_t layer);         /* W_l·x / W_r·x parts of E:1386, 1416 */
int gat_layer_forward_edges(gat_ctx* ctx, int32_t layer);   /* E:1386, 1398, 1407, 1416, 1428 */
int gat_head_forward(gat_ctx* ctx, float* loss_sum, int32_t* n_correct);   /* E:1446, 1457, 1460 */
int gat_head_backward(gat_ctx* ctx);                        /* E:1468 */
int gat_layer_backward_edges(gat_ctx* ctx, int32_t layer);  /* E:1489, 1503 + edge parts of 1517, 1533 */
int gat_layer_backward_dense(gat_ctx* ctx, int32_t layer);  /* dense parts of E:1517, 1533; E:1546 */
/* *needed = 1 if the host must run the PL all-gather / gPL reduce-scatter for this layer
 * (0 for a single shard, and for layer 0 of a shard with replicated input). */
int gat_layer_exchange(gat_ctx* ctx, int32_t layer, int32_t* needed);

/* ---- exchanges inside the library (optional; the alternative to driving the phases yourself) ----
 * With a transport attached, gat_forward / gat_backward / gat_step accept a sharded context and run
 * the exchanges on the context's stream: all-gather of PL / reduce-scatter of gPL for every layer
 * with gat_layer_exchange() == 1, and one all-reduce of the packed gradients at the end of the
 * backward.  gat_forward then returns the GLOBAL loss sum and #correct (a 3-float all-reduce), so
 * every rank prints the same numbers and takes the same optimizer step.
 *   rccl : RCCL over xGMI; librccl is dlopen'ed here, not at library load.  Rank 0 obtains the id
 *          (gat_comm_unique_id) and the HOST ships its GAT_COMM_ID_BYTES to the other ranks.
 *   host : staged through POSIX shared memory `shm_name` (rank 0 creates it); bytes_per_rank >=
 *          n_table*max(H*D)*4 and >= (n_params+3)*4.  For tests (ranks sharing one GPU) and boxes
 *          without peer links; sums in ascending rank order. */
#define GAT_COMM_ID_BYTES 128
int gat_comm_unique_id(void* id_out);
int gat_comm_init_rccl(gat_ctx* ctx, int32_t world, int32_t rank, const void* id);
int gat_comm_init_host(gat_ctx* ctx, int32_t world, int32_t rank, const char* shm_name, int64_t bytes_per_rank);
/* Options of the exchanges run by the library (any transport).
 *   GAT_COMM_GPL_BF16 = 1: in the gPL reduce-scatter the REMOTE partial sums travel as bf16 (half the xGMI volume of the
 *   backward exchange); each rank keeps its own partial in fp32 and adds the arrivals in fp32, in ascending rank order.
 *   Default 0 (fp32 on the wire).  Gradients then differ from the fp32 exchange by ~2^-9 relative per remote term: the
 *   parity bar of this mode is 1e-2 (like bf16 storage).
 *   GAT_COMM_PIPELINE = K (1..64, default 1): the forward exchange of a layer runs in K row chunks on a second stream,
 *   chunk k travelling while chunk k+1 is projected; results are bitwise those of K = 1.
 *   GAT_COMM_HALO = 0 | 1 | 2 (default 0): the table exchanges move only the rows the receiving shard's edges reference
 *   (the rows the reference reads per edge, E:287-290 / E:407-409, and adds into, E:868-869) instead of whole slices: per
 *   peer one packed send / receive pair forward, its mirror backward, the own slice summed in ascending rank order.  1 = on,
 *   2 = on if fewer than half of the rows of a full exchange would travel (a halo costs a pack and an unpack pass on each side).
 *   COLLECTIVE: every rank of the transport makes the call (the request lists are exchanged once, here).  Results equal those of
 *   the full exchange value for value on the host transport; excludes GAT_COMM_GPL_BF16; GAT_COMM_PIPELINE is ignored while on. */
enum { GAT_COMM_GPL_BF16 = 1, GAT_COMM_PIPELINE = 2, GAT_COMM_HALO = 3 };
int gat_comm_option(gat_ctx* ctx, int32_t option, int32_t value);
/* What GAT_COMM_HALO set up: active (0/1), rows this rank receives / sends per forward table exchange (the backward is the mirror),
 * and rows on the wire / rows of the full exchange over all ranks (1.0 before the option was ever set).  Any pointer may be NULL. */
int gat_comm_halo_info(gat_ctx* ctx, int32_t* active, int64_t* rows_received, int64_t* rows_sent, double* referenced_fraction);
/* forward + backward without a host round-trip in between; with a transport the loss and #correct
 * ride in the tail of the gradient all-reduce.  Returns the global loss sum / #correct. */
int gat_step(gat_ctx* ctx, float* loss_sum, int32_t* n_correct);
/* enable = 1: gat_step captures its kernel sequence into a hipGraph (after one eager step) and replays it —
 * one launch per step instead of ~25; for graphs small enough to be launch-bound.  Single shard, no
 * transport, collect_timing = 0.  Results are those of the eager step, bit for bit. */
int gat_step_graph(gat_ctx* ctx, int32_t enable);

/* Exchange buffers.  GAT_TABLE_PL: projected source features, [n_table][H_l*D_l] f32, the
 * context writes rows [table_row0, +n_rows) in gat_layer_project and reads all rows in the edge
 * phases.  GAT_TABLE_GPL: gradient wrt PL, same shape, the edge backward adds into ALL rows; after
 * the host's reduce-scatter rows [table_row0, +n_rows) must hold the summed values.
 * gat_table() returns the context's own buffer; gat_bind_table() substitutes a caller-owned one
 * (e.g. memory registered with the collective library). */
enum { GAT_TABLE_PL = 0, GAT_TABLE_GPL = 1 };
int gat_table(gat_ctx* ctx, int which, int32_t layer, void** d_ptr, int64_t* n_rows, int64_t* row_floats);
int gat_bind_table(gat_ctx* ctx, int which, int32_t layer, void* d_ptr, int64_t bytes);

/* ---- taps: intermediates converted to the reference layout, for parity tests -----------------
 * host_dst receives `count` floats (int32 for SRC/DST). */
enum {
    GAT_TAP_SRC = 0,        /* int32 [E]             d_src  (E:1186) */
    GAT_TAP_DST = 1,        /* int32 [E]             d_dst */
    GAT_TAP_ALPHA = 2,      /* [H][E]                attn_coeff[l] (E:381) */
    GAT_TAP_HPRE = 3,       /* [N][H][D]             d_h[l] (E:422) */
    GAT_TAP_HOUT = 4,       /* [N][H*D] | [N][D]     d_layer_outputs[l] (E:449, 456) */
    GAT_TAP_Y = 5,          /* [N][C]                d_y (E:508) */
    GAT_TAP_G = 6,          /* [N][H][D]             input_gradients[l] (E:601, 891).  The device keeps the factors of
                                                      E:598-603 / E:888-892 unapplied (the edge backward applies them on
                                                      load); the tap returns the reference's tensor */
    GAT_TAP_GE = 7,         /* [H][E]                grad_attn_score (E:693) */
    GAT_TAP_MAX = 8,        /* [H][N]                d_max_attn_score (E:356) */
    GAT_TAP_SUM = 9,        /* [H][N]                d_sum_score_exp (E:357) */
    GAT_TAP_PL = 10,        /* [n_table][H*D]        W_left·x  (the product's own intermediate) */
    GAT_TAP_PR = 11,        /* [N][H*D]              W_right·x */
    GAT_TAP_SCORE = 12,     /* [H][E]                attn_score[l] (E:323), keep_taps */
    GAT_TAP_GALPHA = 13,    /* [H][E]                grad_attn_coeff (E:646), keep_taps */
    GAT_TAP_GX = 14         /* [N][F_l], l >= 1      input_gradients[l-1] as compute_features_input_gradients leaves
                                                      it (E:868-869), BEFORE the LReLU'(h_pre_{l-1}) factor of E:888-892 */
};
int gat_tap(gat_ctx* ctx, int tensor, int32_t layer, void* host_dst, int64_t count);

/* ---- op-level entry points, whole layers: caller-provided DEVICE pointers in the reference layouts
 *      (unit parity).  `stream` may be NULL (default stream).  One entry point per reference KERNEL: below. ---- */
/* a1  csr_to_coo_kernel E:67-84 */
int gat_op_csr_to_coo(const int32_t* d_row_ptr, const int32_t* d_col_idx, int32_t* d_src,
                      int32_t* d_dst, int64_t n_rows, int64_t n_edges, void* stream);
/* a2-a6 forward of one layer (E:279-459) from reference-layout W,a: writes attn_coeff [H][E],
 * h_pre [N][H][D], H_out. */
int gat_op_layer_forward(const int32_t* d_row_ptr, const int32_t* d_col_idx, const float* d_x,
                         const float* d_w, const float* d_a, float* d_attn_coeff, float* d_hpre,
                         float* d_hout, int64_t n, int64_t e, int32_t f, int32_t h, int32_t d,
                         int32_t is_last, float slope, void* stream);
/* a7-a11 backward of one layer (E:612-893): adds into d_grad_w/d_grad_a; writes g_prev
 * (= gx ⊙ LReLU'(hpre_prev)) when d_hpre_prev != NULL. */
int gat_op_layer_backward(const int32_t* d_row_ptr, const int32_t* d_col_idx, const float* d_x,
                          const float* d_w, const float* d_a, const float* d_attn_coeff,
                          const float* d_hpre, const float* d_g, float* d_grad_w, float* d_grad_a,
                          const float* d_hpre_prev, float* d_g_prev, int64_t n, int64_t e,
                          int32_t f, int32_t h, int32_t d, float slope, void* stream);

/* ---- per-kernel entry points: ONE per reference kernel of the hot path, each with the argument list of the launch it
 *      replaces (node / edge counts widened to 64 bit, a stream appended; `n` added where the reference kernel reads the
 *      node count only through its indices).  Caller-owned DEVICE pointers in the reference layouts: edge tensors [H][E]
 *      head-major, node tensors [N][H][D], W [H][D][2F], a [H][D].  Accumulate-or-overwrite behaviour is the reference's.
 *      A maintainer can swap a single launch of the reference's main() for the matching call and keep everything else
 *      (INTEGRATION.md §2b).  Scratch (projected features, message tables) is allocated per call; calls synchronise their
 *      stream.  Scatters use float atomics like the reference's own kernels (sums over edges order-dependent at fp32
 *      round-off).  csrc/gat_ops.hip.
 *      THESE ARE PARITY SEAMS, NOT THE FAST PATH: every call hipMalloc's and frees its scratch, re-projects X.W on the matrix cores,
 *      synchronises the stream, and scatters with global float atomics (~1.3 TB/s on this chip).  A maintainer who wants the speed
 *      swaps the epoch loop for gat_step / the phase API on a context (which owns its buffers and runs without atomics or
 *      per-call allocation), not kernel by kernel. ---------------------------------------------------------------------- */
/* a2  gatv2_edge_score_kernel E:279-324, launch E:1386: attn_score[h][e] (overwritten) */
int gat_op_edge_score(const float* d_input_features, const int32_t* d_col_idx, const int32_t* d_dst, const float* d_w,
                      const float* d_a, float* d_attn_score, int64_t n, int32_t in_dim, int32_t out_dim, int32_t h,
                      int64_t e, float negative_slope, void* stream);
/* a3  compute_max_sum_attn_score E:326-359, launch E:1394-1398: max_score / score_sum at index n*h + dst (overwritten);
 *     zero in-degree rows give max = -1e9f, sum = 0 (E:336) */
int gat_op_max_sum(const int32_t* d_row_ptr, const float* d_attn_score, int64_t n, int32_t h, int64_t e,
                   float* d_max_score, float* d_score_sum, void* stream);
/* a4  compute_attn_coeff E:362-384, launch E:1407 (d_col_idx is carried by the reference's signature and unused there too) */
int gat_op_attn_coeff(const int32_t* d_col_idx, const int32_t* d_dst, const float* d_attn_score, const float* d_max_attn_score,
                      const float* d_sum_score_exp, float* d_attn_coeff, int64_t e, int32_t h, int64_t n, void* stream);
/* a5  aggregate_kernel E:386-424, launch E:1416: ADDS into d_out_feat [N][H][D] (the caller zeroes it) */
int gat_op_aggregate(const int32_t* d_src, const int32_t* d_dst, const float* d_attn_coeff, const float* d_in_feat,
                     const float* d_w, float* d_out_feat, int64_t n, int32_t h, int64_t e, int32_t in_dim, int32_t out_dim,
                     void* stream);
/* a6  postActivationLayerOutput E:426-459, launch E:1428: d_H [N][H*D], or [N][D] (mean over heads) when is_last_layer */
int gat_op_post_activation(const float* d_out_feat, float* d_H, int64_t n, int32_t h, int32_t out_dim, int32_t is_last_layer,
                           float negative_slope, void* stream);
/* C12 gatv2_output_kernel E:463-512 (+ softmax E:132-141), launch E:1446: d_z and d_y [N][C] both receive the
 *     probabilities, as in the reference */
int gat_op_output_head(const float* d_wo, const float* d_last_layer_output, float* d_z, float* d_y, int64_t num_nodes, int32_t c,
                       int32_t out_dim_last_layer, void* stream);
/* C13 compute_loss_accuracy_kernel E:514-537, launch E:1457: per-node loss / correct arrays (summed by the caller, E:542) */
int gat_op_loss_accuracy(const float* d_y, const int32_t* d_labels, float* d_losses, int32_t* d_corrects, int64_t n, int32_t c,
                         void* stream);
/* C14 compute_output_gradients E:553-608, launch E:1468: ADDS into grad_d_wo [C][D_L]; writes grad_d_hL [N][H][D_L].
 *     flat_lrelu_index: 0 = the exact per-head LReLU' index, 1 = the reference's n*D_L + d (E:598) */
int gat_op_output_gradients(const float* d_y, const int32_t* d_labels, const float* d_hL, const float* d_HL, const float* d_wo,
                            float* grad_d_wo, float* grad_d_hL, int64_t n, int32_t c, int32_t out_dim_l, int32_t num_heads,
                            float negative_slope, int32_t flat_lrelu_index, void* stream);
/* a7  kernel_grad_atten_coeff E:612-651, launch E:1489: grad_attn_coeff[h][e] (overwritten) */
int gat_op_grad_attn_coeff(int64_t num_edges, int32_t num_heads, int32_t in_dim, int32_t out_dim, const int32_t* d_src,
                           const int32_t* d_dst, const float* d_features, const float* d_w, const float* d_grad_input,
                           float* d_grad_attn_coeff, int64_t n, void* stream);
/* a8  compute_grad_attn_score_kernel E:654-696, launch E:1499-1506: grad_attn_score[h][e] (overwritten); one pass per row
 *     instead of the reference's O(deg) loop per edge (same sums up to fp32 order) */
int gat_op_grad_attn_score(const int32_t* d_row_ptr, const int32_t* d_dst, const float* d_alpha, const float* d_grad_alpha,
                           float* d_grad_e, int64_t n, int32_t h, int64_t e, void* stream);
/* a9  compute_grad_parameters_kernel E:698-798, launch E:1517: ADDS into grad_w [H][D][2F] and grad_a [H][D] */
int gat_op_grad_parameters(int64_t e, int32_t h, const int32_t* d_src, const int32_t* d_dst, const float* d_features,
                           const float* d_input_gradients, const float* d_grad_attn_score, const float* d_attn_coeff,
                           const float* d_w, const float* d_a, float* grad_w, float* grad_a, int32_t in_dim, int32_t out_dim,
                           float negative_slope, int64_t n, void* stream);
/* a10 compute_features_input_gradients E:801-874, launch E:1533: ADDS into grad_x_features [N][F] (zeroed per epoch, E:1636) */
int gat_op_features_input_gradients(int64_t n, int32_t h, int64_t e, int32_t in_dim, int32_t out_dim, float negative_slope,
                                    const int32_t* d_src, const int32_t* d_dst, const float* d_attn_coeff,
                                    const float* d_input_features, const float* d_w, const float* d_input_gradients,
                                    const float* d_grad_attn_score, const float* d_attn_vector, float* d_grad_x_features,
                                    void* stream);
/* a11 compute_preActivation_inputFeatures_gradient E:879-893, launch E:1546: in place on input_features_gradients [N][F] */
int gat_op_preact_gradient(int64_t n, float negative_slope, int32_t in_dim, const float* d_pre_activation_input_features,
                           float* d_input_features_gradients, void* stream);

/* ---- synthetic workloads on the device --------------------------------------------
 * The reference's datasets are a download link (README R:21); every BASELINE workload here is a deterministic
 * synthetic graph of the stated shape (SURVEY 8d law; host implementation: synth.py).  These three calls do the
 * parts proportional to E and N*F on the GPU, bit-for-bit equal to the host generator.  Outputs are DEVICE
 * pointers (feed them to gat_set_*_device).  `stream` may be NULL.
 *   sources : per edge e, u = hash(seed, 3, e) -> rank = first r with h_cdf[r] > u -> h_node_of_rank[rank]; the
 *             sources of every row then sorted ascending (one radix sort of dst*n + src).  h_cdf / h_node_of_rank /
 *             h_row_ptr are the host's N-sized tables (synth.graph_tables).
 *   features: kind 0 = U[-1,1) fp32, kind 1 = sparse binary rows normalised by their count (Cora-like); rows
 *             [row0, row0+rows) of the [n][f] matrix.
 *   labels  : hash % num_classes, node 0 forced to num_classes-1.
 *   argsort : the three N-sized stable sorts of the host tables (two node permutations, the largest-remainder order). */
int gat_synth_sources_device(const double* h_cdf, const int32_t* h_node_of_rank, const int32_t* h_row_ptr, int64_t n,
                             int64_t n_edges, uint64_t seed, int32_t* d_col_idx, void* stream);
/* host keys -> host order, sorted on the device: order[i] = index of the i-th smallest key, ties in index order */
int gat_synth_argsort_u64(const uint64_t* h_keys, int64_t n, int32_t* h_order, void* stream);
int gat_synth_features_device(uint64_t seed, int64_t row0, int64_t rows, int32_t f, int32_t kind, float* d_x, void* stream);
int gat_synth_labels_device(uint64_t seed, int64_t row0, int64_t rows, int32_t num_classes, int32_t* d_labels, void* stream);

/* ---- measurement ------------------------------------------------------------------------------ */
enum {
    GAT_K_PROJECT = 0, GAT_K_EDGE_FWD = 1, GAT_K_HEAD_FWD = 2, GAT_K_HEAD_BWD = 3,
    GAT_K_EDGE_BWD = 4, GAT_K_GPL_SUM = 5, GAT_K_GRAD_W = 6, GAT_K_GRAD_X = 7, GAT_K_MISC = 8,
    GAT_K_EXCHANGE = 9,     /* transport calls of gat_forward / gat_backward / gat_step (event-timed like kernels) */
    GAT_K_EDGE_FUSED = 10,  /* experiment, off unless GAT_FUSE_LAST=1 (measured not ahead: DESIGN §4): gat_step with the last layer's
                               forward edge pass, the gH part of the head and its backward edge pass fused per destination row
                               (E:1386-1428 + 1468 + 1489-1533 of that layer in one launch; gat_algorithmic_bytes then moves
                               that layer's edge bytes to this class) */
    GAT_K_COUNT = 11
};
/* Accumulated HIP-event time of kernel class `k` since the last gat_kernel_stats_reset (needs
 * collect_timing=1).  Synchronises the stream. */
int gat_kernel_stats(gat_ctx* ctx, int k, int64_t* launches, double* total_ms);
int gat_kernel_stats_reset(gat_ctx* ctx);
const char* gat_kernel_name(int k);
/* Algorithmic HBM bytes of one forward+backward step on this context's shard. */
int gat_algorithmic_bytes(gat_ctx* ctx, double* bytes_step, double* bytes_per_kernel /* [GAT_K_COUNT] or NULL */);
/* The same figure from the shape alone (host arithmetic, no device, no context): SURVEY §8d's formula with
 * b = 4 (GAT_DTYPE_F32) or 2 (GAT_DTYPE_BF16) bytes on every float term.  n_table / replicated_input describe a
 * destination-range shard (single GPU: n_table = n_rows, replicated_input = 0).  Only cfg->num_layers, heads,
 * outdims, in_dim, num_classes and storage_dtype are read. */
int gat_algorithmic_bytes_shape(const gat_config* cfg, int64_t n_rows, int64_t n_edges, int64_t n_table,
                                int32_t replicated_input, double* bytes_step, double* bytes_per_kernel);
/* The same model at the granularity the memory fabric serves: every per-edge gathered / scattered row (PL[src] in both edge
 * passes, the gPL scatter) rounded up to whole 128-byte requests.  Equal to the figure above when a row is a multiple of 128 B
 * (fp32, H*D = 64); larger for bf16 rows at H*D < 64 (BASELINE config 5: 64-byte rows cost a line each) — the roofline such a
 * shape can actually be served at.  bench.py reports it as `frac_request_granular`. */
int gat_request_bytes_shape(const gat_config* cfg, int64_t n_rows, int64_t n_edges, int64_t n_table,
                            int32_t replicated_input, double* bytes_step, double* bytes_per_kernel);

#ifdef __cplusplus
}
#endif
#endif /* GATV2_ABI_H */
