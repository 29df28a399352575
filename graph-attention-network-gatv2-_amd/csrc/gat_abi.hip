// gat_abi.hip — implementation of include/gatv2_abi.h: the context (device memory plan, stream,
// event timing) and the C entry points that sequence the kernels of gat_edge_kernels.hip and
// gat_dense_kernels.hip.  Replaces the inline allocation + launch code of the reference's main()
// (GATv2_edge_based.cu E:1151-1357 memory plan, E:1370-1642 epoch loop).
#include "gat_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

namespace gat {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code ? code : GAT_E_INVALID;
}
// Choice switches: environment variables that select among kernels / launch shapes computing the SAME results (the test
// matrices force every one).  Each is read once per process through here, and what was set is reported by gat_switches().
static std::mutex g_choice_mu;
static std::vector<std::pair<std::string, std::string>> g_choices;
const char* choice_env(const char* name) {
    const char* e = getenv(name);
    if (e) {
        std::lock_guard<std::mutex> lock(g_choice_mu);
        bool seen = false;
        for (auto& kv : g_choices) if (kv.first == name) { kv.second = e; seen = true; }
        if (!seen) g_choices.emplace_back(name, e);
    }
    return e;
}
static std::string choices_text() {
    std::lock_guard<std::mutex> lock(g_choice_mu);
    std::string t;
    for (auto& kv : g_choices) { if (!t.empty()) t += ' '; t += kv.first + "=" + kv.second; }
    return t;
}

struct Layer {
    int32_t H = 0, D = 0, F = 0, HD = 0;
    int64_t w_off = 0, a_off = 0;
    float* PL = nullptr;      // [n_table][HD]  (table; may be caller-owned)
    bool PL_bound = false;
    float* PR = nullptr;      // [n_rows][HD]
    float* alpha = nullptr;   // [E][H]
    float* hpre = nullptr;    // [n_rows][HD]
    float* hout = nullptr;    // [n_rows][HD] / last: [n_rows][D]
    float* g = nullptr;       // [n_rows][HD]   dL/dh_pre (input_gradients[l], E:1339)
    float* ge = nullptr;      // [E][H]  (keep_taps)
    float* score = nullptr;   // [E][H]  (keep_taps) raw attention scores, E:323
    float* galpha = nullptr;  // [E][H]  (keep_taps) grad wrt attn_coeff, E:646
    float* mstat = nullptr;   // [n_rows][H] (keep_taps)
    bool stash = false;       // backward scatter through per-edge records + pull pass (edge_stash_words) instead of message rows
    float* zstat = nullptr;
};

struct Pending { int k; hipEvent_t e0, e1; };

void build_worklist(const int32_t* rp, int64_t n_rows, WorkList& w) {
    static const int seg_env = [] { const char* e = choice_env("GAT_SEG_EDGES"); const int v = e ? atoi(e) : 0; return v >= 16 ? v : 0; }();
    // GAT_SEG_EDGES=<n>: sweep of the hub-row segment length.  Default 256 (swept on the Products shape); graphs with
    // few edges get shorter segments: the persistent backward's critical path is its longest item (a 256-edge segment
    // is 64 dependent gather steps), which at Arxiv size (1.17 M edges) was longer than everything else together
    // (Products shape 256; 64 below 16 M edges; Arxiv shape 32: 1.35 -> 1.29 ms; Pubmed shape 16: 0.34 -> 0.31 ms)
    const int64_t ne = rp[n_rows];
    const int kSegEdges = seg_env ? seg_env : (ne < (512 << 10) ? 16 : ne < (4 << 20) ? 32 : ne < (16 << 20) ? 64 : gat::kSegEdges);
    w.items.clear(); w.slot_info.clear(); w.n_slots = 0; w.n_split = 0;
    std::vector<int32_t> firsts;
    // pass 1: segments of split rows first (the longest items start earliest)
    for (int64_t r = 0; r < n_rows; ++r) {
        const int32_t b = rp[r], e = rp[r + 1];
        if (e - b <= kSegEdges) continue;
        const int32_t nseg = (e - b + kSegEdges - 1) / kSegEdges, first = w.n_slots;
        firsts.push_back(first);
        for (int32_t sgm = 0; sgm < nseg; ++sgm) {
            const int32_t sb = b + sgm * kSegEdges, se = std::min(e, sb + kSegEdges);
            const int32_t item = (int32_t)(w.items.size() / 4);
            w.items.insert(w.items.end(), {(int32_t)r, sb, se, w.n_slots});
            w.slot_info.insert(w.slot_info.end(), {(int32_t)r, first, nseg, item});
            ++w.n_slots;
        }
    }
    // pass 2: every other row is one item.  The group-per-row kernels put 64/(H*D/N) consecutive items side by side in
    // one wave, each lane group walking its own row, so neighbours should be of (nearly) equal length: rows are sorted
    // by in-degree, longest first (counting sort, row order kept inside a degree) — inside WINDOWS of consecutive
    // rows, so that the list still walks the graph roughly in row order (the per-row reads and writes of PR / h_pre / g
    // stay near each other).  GAT_SORT_WINDOW rows per window, 0 = one window over all rows; swept on the Products
    // shape: 4096 and one window are within noise for the backward, 4096 is ~3 % better for the forward.
    {
        static const int64_t win_env = [] { const char* e = choice_env("GAT_SORT_WINDOW"); return e ? atoll(e) : (int64_t)-1; }();
        const int64_t win = win_env < 0 ? 4096 : (win_env == 0 ? n_rows : win_env);
        std::vector<int64_t> cnt((size_t)kSegEdges + 2);
        for (int64_t r0 = 0; r0 < n_rows; r0 += win) {
            const int64_t r1 = std::min(n_rows, r0 + win);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t r = r0; r < r1; ++r) { const int32_t d = rp[r + 1] - rp[r]; if (d <= kSegEdges) ++cnt[(size_t)(kSegEdges - d) + 1]; }
            for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
            const size_t base = w.items.size();
            w.items.resize(base + 4 * (size_t)cnt.back());
            for (int64_t r = r0; r < r1; ++r) {
                const int32_t b = rp[r], e = rp[r + 1];
                if (e - b > kSegEdges) continue;
                int32_t* it = &w.items[base + 4 * (size_t)cnt[(size_t)(kSegEdges - (e - b))]++];
                it[0] = (int32_t)r; it[1] = b; it[2] = e; it[3] = -1;
            }
        }
    }
    w.n_items = (int64_t)(w.items.size() / 4);
    w.n_split = (int32_t)firsts.size();                 // appended after the per-slot entries: one int4 per split row
    for (int32_t f : firsts) w.slot_info.insert(w.slot_info.end(), {f, 0, 0, 0});
}

}  // namespace gat

struct gat_ctx {
    gat_config cfg{};
    std::vector<int32_t> heads, outdims;
    std::vector<gat::Layer> layers;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n_rows = 0, n_edges = 0, n_table = 0, table_row0 = 0;
    bool have_graph = false, have_x = false, have_labels = false, buffers_ready = false;
    int32_t* row_ptr = nullptr; int32_t* col_idx = nullptr; int32_t* labels = nullptr;
    int32_t* labels_eff = nullptr;                  // with a training mask: label, or ~label outside the mask (gat_set_train_mask)
    int32_t* labels_eff_buf = nullptr;              // its storage (labels_eff is null while no mask is set)
    uint8_t* mask_tmp = nullptr;                    // [n_rows] device copy of the mask of the last gat_eval_mask
    uint8_t* train_mask = nullptr;                  // [n_rows] device copy of the active training mask (re-applied when the labels change)
    int64_t n_labels = 0;                           // length the labels were set with
    double* eval_loss = nullptr; int32_t* eval_cnt = nullptr;      // block partials of gat_eval_mask
    float* X0 = nullptr;
    int32_t ld0 = 0;                                // floats between rows of X0 / Xtab: in_dim rounded up to a multiple of 4 (zeros behind
                                                    // column in_dim), so that an odd in_dim — Cora's 1,433 — still gets 16-byte loads
    float* Xtab = nullptr;                          // [n_table][in_dim] replicated layer-0 input (gat_set_source_features)
    int64_t nW = 0, nA = 0, nWo = 0;
    float* params = nullptr;   // [W | a | Wo]
    float* grads = nullptr;    // [gradW | grada | gradWo] + 4 floats of tail: [loss, correct lo, correct hi, -]
    std::unique_ptr<gat::Comm> comm;                // exchange transport of a shard (gat_comm_init_*)
    int32_t comm_chunks = 1;                        // gat_comm_option(GAT_COMM_PIPELINE): row chunks of the pipelined forward exchange
    hipStream_t comm_stream = nullptr;              // second stream of the pipelined exchange (created on first use)
    std::vector<hipEvent_t> comm_events;
    bool comm_gpl_bf16 = false;                     // gat_comm_option(GAT_COMM_GPL_BF16): remote gPL partials travel as bf16
    gat::HaloPlan halo;                             // gat_comm_option(GAT_COMM_HALO): rows each peer's edges reference (built collectively)
    bool halo_on = false;                           // the table exchanges use it
    float* grads_prev = nullptr;                    // [n_params] with a transport: what the buffer held before this step (see reduce_begin)
    // gat_step as a replayed hipGraph (gat_step_graph): 0 off, 1 armed (next step runs eagerly, then captures), 2 ready
    int graph_state = 0, graph_warm = 0;
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
    float* pinned_tail = nullptr;                   // [4] host-pinned landing zone of {loss, correct lo, correct hi}
    float* adam_m = nullptr; float* adam_v = nullptr;
    int32_t HDmax = 0, Hmax = 0;
    float* gPL = nullptr; bool gPL_bound = false;   // [n_table][HDmax]
    float* gPR = nullptr;                           // [n_rows][HDmax]
    // grad_w off the edge passes' stream (gat_backward / gat_step): a second gPL / gPR pair for the odd layers, so that layer l's
    // grad_w may still read its operands while layer l-1's edge passes write theirs; side stream + fork / join events per layer
    float* gPL_alt = nullptr; float* gPR_alt = nullptr;
    bool ov_active = false;                         // inside backward_phases with the overlap on: odd layers use the second pair
    hipStream_t side_stream = nullptr;
    std::vector<hipEvent_t> ev_fork, ev_join;
    float* gH = nullptr;                            // [n_rows][gh_stride] head-independent output gradient (HeadBwdArgs::gh_out)
    int32_t gh_stride = 0;                          // D_last, or 16: 64-B records {gH row | the row's LReLU'(h_pre) decision bytes at +32}
    bool y_valid = false;                           // c->y holds the last forward's probabilities (not after a fused head step)
    int32_t* csc_pos = nullptr;                     // [E] slot of each CSR edge in source-major order
    int32_t* csc_ptr = nullptr;                     // [n_table+1]
    int4* gpl_chunks = nullptr; int4* gpl_heavy = nullptr; float* gpl_part = nullptr;   // long source lists (HeavyList)
    int4* pull_items = nullptr; int64_t n_pull_items = 0;                               // length-sorted source items of the pull pass
    int32_t n_gpl_chunks = 0, n_gpl_heavy = 0;
    float* msg = nullptr; int32_t msg_hd = 0;       // [E][msg_hd] per-edge message rows (store path)
    uint32_t* stash = nullptr; int32_t stash_words = 0;   // [E][stash_words] per-edge records (stash path: Layer::stash)
    int32_t* csc_dst = nullptr;                     // [E] destination row of every slot (stash path)
    int32_t* csc_src = nullptr;                     // [E + pad] table row of every slot (slot-parallel source-major pass)
    gat::SlotRuns runs;                             // its device-side index (csrc null = not built)
    float* gfull = nullptr;                         // [n_rows][HDmax] dL/dh_pre incl. LReLU' (stash path: gathered by the pull pass)
    uint8_t* hbits = nullptr;                       // [n_rows][HD_last/N] LReLU'(h_pre) decisions of the last layer (EdgeBwdArgs::hbits)
    int32_t dbg = 0;                                // GAT_DBG timing experiments (0 = product behaviour)
    gat::WorkList work;                             // host copy of the item list
    int4* items = nullptr; int4* slot_info = nullptr;
    float* part_acc = nullptr; float* part_mz = nullptr;
    float* ga_partial = nullptr; int32_t ga_blocks = 0;
    float* gw_scratch = nullptr; int64_t gw_scratch_floats = 0; std::vector<int64_t> gw_off;       // [L] first float of each layer's grad_w slab region
    float* hb_partial = nullptr;
    double* loss_partial = nullptr; int32_t* correct_partial = nullptr;
    float* loss_out = nullptr; int32_t* correct_out = nullptr;
    float* clip_scratch = nullptr;
    float* y = nullptr;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    std::vector<gat::Pending> ev_pending;
    int64_t k_launches[GAT_K_COUNT] = {0};
    double k_ms[GAT_K_COUNT] = {0};
    std::vector<void*> owned;   // every hipMalloc of this context
};

namespace gat {

static int dmalloc(gat_ctx* c, void** p, size_t bytes) {
    if (bytes == 0) bytes = 4;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(GAT_E_NOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    }
    c->owned.push_back(*p);
    return 0;
}
template <class T>
static int dalloc(gat_ctx* c, T** p, int64_t count) { return dmalloc(c, (void**)p, (size_t)count * sizeof(T)); }

static void dfree(gat_ctx* c, void* p) {
    if (!p) return;
    auto it = std::find(c->owned.begin(), c->owned.end(), p);
    if (it != c->owned.end()) { c->owned.erase(it); (void)hipFree(p); }
}

// ---- event timing ------------------------------------------------------------------------------
static int flush_events(gat_ctx* c) {
    if (c->ev_pending.empty()) return 0;
    GAT_HIP(hipStreamSynchronize(c->stream));
    for (auto& p : c->ev_pending) {
        float ms = 0.f;
        GAT_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        c->k_ms[p.k] += ms;
        c->k_launches[p.k] += 1;
        c->ev_free.emplace_back(p.e0, p.e1);
    }
    c->ev_pending.clear();
    return 0;
}
struct Scope {      // brackets the launches of one kernel class with a HIP event pair (on the stream they are launched on)
    gat_ctx* c; int k; hipEvent_t e0 = nullptr, e1 = nullptr; bool on; hipStream_t st;
    Scope(gat_ctx* c_, int k_, hipStream_t st_ = nullptr) : c(c_), k(k_), on(c_->cfg.collect_timing != 0), st(st_ ? st_ : c_->stream) {
        if (!on) return;
        if (c->ev_pending.size() >= 2048) (void)flush_events(c);
        if (c->ev_free.empty()) {
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
        } else { e0 = c->ev_free.back().first; e1 = c->ev_free.back().second; c->ev_free.pop_back(); }
        (void)hipEventRecord(e0, st);
    }
    ~Scope() {
        if (!on) return;
        (void)hipEventRecord(e1, st);
        c->ev_pending.push_back({k, e0, e1});
    }
};

static bool bf16(const gat_ctx* c) { return c->cfg.storage_dtype == GAT_DTYPE_BF16; }
static int64_t st_bytes(const gat_ctx* c) { return bf16(c) ? 2 : 4; }       // element size of PL / message rows

static int check_layer(gat_ctx* c, int32_t l) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (l < 0 || l >= c->cfg.num_layers) return fail(GAT_E_INVALID, "layer index out of range");
    if (!c->buffers_ready) return fail(GAT_E_STATE, "set_graph / set_features / set_labels must come first");
    return 0;
}

static float* W_of(gat_ctx* c, int l) { return c->params + c->layers[l].w_off; }
static float* a_of(gat_ctx* c, int l) { return c->params + c->nW + c->layers[l].a_off; }
static float* Wo_of(gat_ctx* c) { return c->params + c->nW + c->nA; }
static float* gW_of(gat_ctx* c, int l) { return c->grads + c->layers[l].w_off; }
static float* ga_of(gat_ctx* c, int l) { return c->grads + c->nW + c->layers[l].a_off; }
static float* gWo_of(gat_ctx* c) { return c->grads + c->nW + c->nA; }
static const float* Xin_of(gat_ctx* c, int l) { return l == 0 ? c->X0 : c->layers[l - 1].hout; }
static int32_t ldX_of(gat_ctx* c, int l) { return l == 0 ? c->ld0 : c->layers[l].F; }
// rows of in_dim floats -> rows of ld0 floats with zeros behind column in_dim
static int upload_rows_padded(gat_ctx* c, float* dst, const float* src, int64_t n_rows, int32_t in_dim, hipMemcpyKind kind) {
    if (c->ld0 == in_dim) {
        GAT_HIP(hipMemcpyAsync(dst, src, (size_t)n_rows * in_dim * sizeof(float), kind, c->stream));
        return 0;
    }
    GAT_HIP(hipMemsetAsync(dst, 0, (size_t)n_rows * c->ld0 * sizeof(float), c->stream));
    GAT_HIP(hipMemcpy2DAsync(dst, (size_t)c->ld0 * sizeof(float), src, (size_t)in_dim * sizeof(float), (size_t)in_dim * sizeof(float), (size_t)n_rows, kind, c->stream));
    return 0;
}
static float* gPL_of(gat_ctx* c, int l) { return (c->ov_active && (l & 1)) ? c->gPL_alt : c->gPL; }
static float* gPR_of(gat_ctx* c, int l) { return (c->ov_active && (l & 1)) ? c->gPR_alt : c->gPR; }

// (re)allocate everything that depends on the graph size
static int ensure_buffers(gat_ctx* c) {
    if (c->buffers_ready) return 0;
    if (!(c->have_graph && c->have_x && c->have_labels)) return 0;
    const int L = c->cfg.num_layers;
    const int64_t N = c->n_rows, E = c->n_edges, T = c->n_table;
    for (int l = 0; l < L; ++l) {
        Layer& y = c->layers[l];
        if (bf16(c) && !edge_fast_path(y.H, y.D, T))
            return fail(GAT_E_UNSUPPORTED, "bf16 storage needs every layer on the wave-per-row path (H*D in {8,16,32,64}, D a power of two)");
        if (!y.PL_bound) GAT_TRY(dmalloc(c, (void**)&y.PL, (size_t)(T * y.HD * st_bytes(c))));
        GAT_TRY(dalloc(c, &y.PR, N * y.HD));
        // attn_coeff [E][H] is only materialised for parity taps and for layers on the generic
        // path; the fast path's backward recomputes it from the per-(row, head) softmax stats.
        if (c->cfg.keep_taps || !edge_fast_path(y.H, y.D, c->n_table)) GAT_TRY(dalloc(c, &y.alpha, E * y.H));
        GAT_TRY(dalloc(c, &y.hpre, N * y.HD));
        GAT_TRY(dalloc(c, &y.hout, N * (l == L - 1 ? y.D : y.HD)));
        // the last layer's g is formed inside its edge backward from gH (except in the flat-index mode, E:598)
        if (l < L - 1 || c->cfg.flat_lrelu_index) GAT_TRY(dalloc(c, &y.g, N * y.HD));
        GAT_TRY(dalloc(c, &y.mstat, N * y.H));
        GAT_TRY(dalloc(c, &y.zstat, N * y.H));
        if (c->cfg.keep_taps) {
            GAT_TRY(dalloc(c, &y.ge, E * y.H));
            GAT_TRY(dalloc(c, &y.score, E * y.H));
            GAT_TRY(dalloc(c, &y.galpha, E * y.H));
        }
    }
    if (!c->gPL_bound) GAT_TRY(dalloc(c, &c->gPL, T * c->HDmax));
    GAT_TRY(dalloc(c, &c->gPR, N * c->HDmax));
    if (!c->cfg.flat_lrelu_index) {
        // the last layer's pull pass rebuilds g from gH and one decision byte per lane: both in ONE 64-byte record per
        // node, so that an edge costs one cache line there instead of two (the edge passes are bound by the number of
        // L1 misses in flight, not by bytes: DESIGN §4)
        const Layer& yl = c->layers[L - 1];
        const int w = edge_stash_words(yl.H, yl.D);
        c->gh_stride = (w > 0 && w <= 32 && yl.D <= 8 && !bf16(c) && !c->cfg.keep_taps) ? 16 : yl.D;
        GAT_TRY(dalloc(c, &c->gH, N * c->gh_stride));
    }
    // work items (rows / hub-row segments) of the wave-per-item kernels
    GAT_TRY(dalloc(c, &c->items, std::max<int64_t>(c->work.n_items, 1)));
    GAT_TRY(dalloc(c, &c->slot_info, std::max<int32_t>(c->work.n_slots + c->work.n_split, 1)));
    GAT_TRY(dalloc(c, &c->part_acc, (int64_t)std::max<int32_t>(c->work.n_slots, 1) * c->HDmax));
    GAT_TRY(dalloc(c, &c->part_mz, (int64_t)std::max<int32_t>(c->work.n_slots, 1) * 2 * c->Hmax));
    if (c->work.n_items > 0)
        GAT_HIP(hipMemcpyAsync(c->items, c->work.items.data(), c->work.items.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice, c->stream));
    if (c->work.n_slots > 0)
        GAT_HIP(hipMemcpyAsync(c->slot_info, c->work.slot_info.data(), c->work.slot_info.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    // Store-then-sum backward for the layers on the wave-per-row fast path: needs the source-major
    // slot index and an [E][H*D] scratch.  GAT_BWD_ATOMICS=1 forces the float-atomic variant (A/B).
    // Which form the per-edge scatter takes, per layer: the stash path (64-B records + one gathered row of g[dst] in
    // the source-major pass) where the shape has one, else H*D-float message rows.  GAT_BWD_STASH=0 forces message
    // rows everywhere (the A/B of DESIGN §4); the taps variant of the backward always uses message rows.
    int32_t msg_hd = 0, stash_words = 0;
    const char* no_stash = choice_env("GAT_BWD_STASH");
    for (int l = 0; l < L; ++l) {
        Layer& y = c->layers[l];
        if (!edge_fast_path(y.H, y.D, c->n_table)) continue;
        const int w = edge_stash_words(y.H, y.D);
        // bf16 storage: only where the message row (H*D*2 bytes) is larger than a record — at H*D = 32 the record path
        // trades a 64-B streamed row for a 32-B record plus a 64-B random gather and loses (10 M / 250 M shape: the
        // source-major pass 8.4 -> 13.4 ms per step, the destination-major pass unchanged)
        y.stash = w > 0 && !c->cfg.keep_taps && !(no_stash && no_stash[0] == '0') && !(bf16(c) && y.HD < 64);
        if (y.stash) stash_words = std::max(stash_words, w); else msg_hd = std::max(msg_hd, y.HD);
    }
    const char* force = choice_env("GAT_BWD_ATOMICS");
    if ((msg_hd > 0 || stash_words > 0) && E > 0 && !(force && force[0] == '1')) {
        float* m = nullptr;
        // E + 1 rows / records: the last one takes the stores of the group-per-row kernels' padded lanes; kPullPad records of
        // padding behind the records and the destination list: the pull pass reads whole 16-slot chunks without clamping
        if (hipMalloc((void**)&m, std::max<size_t>((size_t)(E + kPullPad) * msg_hd * (size_t)st_bytes(c), (size_t)(E + kPullPad) * stash_words * sizeof(uint32_t))) == hipSuccess) {
            c->owned.push_back(m);
            if (msg_hd > 0) { c->msg = m; c->msg_hd = msg_hd; }
            if (stash_words > 0) {                  // records and message rows are never live at the same time: one buffer
                c->stash = reinterpret_cast<uint32_t*>(m); c->stash_words = stash_words;
                if (msg_hd == 0) { c->msg = m; c->msg_hd = 0; }
                GAT_TRY(dalloc(c, &c->gfull, N * c->HDmax));
                if (c->layers[L - 1].stash && c->gH != nullptr && c->gh_stride == 16) c->hbits = reinterpret_cast<uint8_t*>(c->gH) + 32;
                GAT_TRY(dalloc(c, &c->csc_dst, E + kPullPad));
                GAT_HIP(hipMemsetAsync(c->csc_dst + E, 0, (size_t)kPullPad * sizeof(int32_t), c->stream));      // padding: row 0 (any valid row)
                GAT_HIP(hipMemsetAsync(reinterpret_cast<uint32_t*>(m) + (size_t)E * stash_words, 0, (size_t)kPullPad * stash_words * sizeof(uint32_t), c->stream));
            }
            GAT_TRY(dalloc(c, &c->csc_pos, E));
            GAT_TRY(dalloc(c, &c->csc_ptr, T + 1));
            // slot-parallel source-major pass (gat_csc.hip "runs"): the source of every slot + the lists crossing a run boundary.
            // GAT_PULL_RUN=<slots per run> (a multiple of 32; 0 = do not build)
            static const int run_env = [] { const char* e = choice_env("GAT_PULL_RUN"); return e ? atoi(e) : -1; }();
            const int32_t run = run_env >= 0 ? (run_env / 32) * 32 : 64;
            if (run > 0) GAT_TRY(dalloc(c, &c->csc_src, E + kPullPad));
            GAT_TRY(build_csc(c->col_idx, E, T, c->csc_pos, c->csc_ptr, c->csc_src, c->stream));
            if (c->csc_dst) GAT_TRY(build_csc_dst(c->row_ptr, c->csc_pos, c->csc_dst, N, E, c->stream));
            HeavyList hl;
            GAT_TRY(build_heavy_list(c->csc_ptr, T, E, &hl, c->stream, run));
            if (hl.run > 0 && c->csc_src != nullptr) {
                SlotRuns& R = c->runs;
                R.run = hl.run; R.n_runs = hl.n_runs; R.n_open = (int32_t)(hl.open.size() / 4); R.n_empty = (int64_t)hl.empty.size();
                int4* d_open = nullptr; int32_t* d_empty = nullptr;
                GAT_TRY(dalloc(c, &d_open, std::max<int64_t>(R.n_open, 1)));
                GAT_TRY(dalloc(c, &d_empty, std::max<int64_t>(R.n_empty, 1)));
                GAT_TRY(dalloc(c, &R.part, std::max<int64_t>(2 * R.n_runs, 1) * c->HDmax));
                if (R.n_open > 0) GAT_HIP(hipMemcpyAsync(d_open, hl.open.data(), hl.open.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                if (R.n_empty > 0) GAT_HIP(hipMemcpyAsync(d_empty, hl.empty.data(), hl.empty.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                GAT_HIP(hipStreamSynchronize(c->stream));
                R.open = d_open; R.empty = d_empty; R.csrc = c->csc_src;
            }
            c->n_gpl_chunks = (int32_t)(hl.chunks.size() / 4); c->n_gpl_heavy = (int32_t)(hl.heavy.size() / 4);
            if (c->stash != nullptr && !hl.items.empty()) {
                c->n_pull_items = (int64_t)(hl.items.size() / 4);
                GAT_TRY(dalloc(c, &c->pull_items, c->n_pull_items));
                GAT_HIP(hipMemcpyAsync(c->pull_items, hl.items.data(), hl.items.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                GAT_HIP(hipStreamSynchronize(c->stream));
            }
            if (c->n_gpl_heavy > 0) {
                GAT_TRY(dalloc(c, &c->gpl_chunks, c->n_gpl_chunks));
                GAT_TRY(dalloc(c, &c->gpl_heavy, c->n_gpl_heavy));
                GAT_TRY(dalloc(c, &c->gpl_part, (int64_t)c->n_gpl_chunks * c->HDmax));
                GAT_HIP(hipMemcpyAsync(c->gpl_chunks, hl.chunks.data(), hl.chunks.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                GAT_HIP(hipMemcpyAsync(c->gpl_heavy, hl.heavy.data(), hl.heavy.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                GAT_HIP(hipStreamSynchronize(c->stream));
            }
        } else {
            (void)hipGetLastError();      // not enough HBM for the scratch: atomics variant
            for (int l = 0; l < L; ++l) c->layers[l].stash = false;
        }
    } else {
        for (int l = 0; l < L; ++l) c->layers[l].stash = false;
    }
    // one region per layer (grad_a block partials, grad_w slabs): their reductions are queued until the end of the
    // backward and run as one launch (reduce_batch_flush), so a later layer must not overwrite an earlier layer's slabs
    GAT_TRY(dalloc(c, &c->ga_partial, (int64_t)L * 2048 * c->HDmax));  // per layer >= any edge_backward_blocks()
    int64_t gw = 0;
    c->gw_off.assign((size_t)L, 0);
    for (int l = 0; l < L; ++l) { c->gw_off[(size_t)l] = gw; gw += grad_w_scratch_floats(N, c->layers[l].F, c->layers[l].HD); }
    gw = std::max<int64_t>(gw, 1);
    if (c->Xtab) gw = std::max(gw, grad_w_scratch_floats(T, c->layers[0].F, c->layers[0].HD));
    // split-K projection (few rows, long K): sized from the exact (rows, part) pairs gat_layer_project launches
    for (int l = 0; l < L; ++l) {
        const Layer& y = c->layers[l];
        if (l == 0 && c->Xtab) gw = std::max({gw, project_scratch_floats(T, y.F, y.HD, kPartLeft), project_scratch_floats(N, y.F, y.HD, kPartRight)});
        else gw = std::max(gw, project_scratch_floats(N, y.F, y.HD, kPartBoth));
    }
    GAT_TRY(dalloc(c, &c->gw_scratch, gw));
    c->gw_scratch_floats = gw;
    const int C = c->cfg.num_classes, DL = c->layers[L - 1].D;
    GAT_TRY(dalloc(c, &c->hb_partial, (int64_t)head_bwd_blocks(N, C, DL) * C * DL));
    GAT_TRY(dalloc(c, &c->loss_partial, head_blocks(N)));
    GAT_TRY(dalloc(c, &c->correct_partial, head_blocks(N)));
    GAT_TRY(dalloc(c, &c->loss_out, 1));
    GAT_TRY(dalloc(c, &c->correct_out, 1));
    GAT_TRY(dalloc(c, &c->y, N * C));
    c->buffers_ready = true;
    // The cliff of gatv2_abi.h "Limits": a layer whose SHAPE has wave-per-row kernels but whose gathered table is 4 GiB or
    // more drops to the generic float-atomic kernels.  Not an error (same results) — but never silent: the call that
    // completed the context returns 0 with this text in gat_last_error(), and gat_layer_path() reports it per layer.
    std::string slow;
    for (int l = 0; l < L; ++l) {
        const Layer& y = c->layers[l];
        if (!edge_fast_path(y.H, y.D, T) && edge_fast_path(y.H, y.D, 1)) slow += (slow.empty() ? "" : ", ") + std::to_string(l) + " (H*D = " + std::to_string(y.HD) + ")";
    }
    if (!slow.empty())
        set_error("warning: layer(s) " + slow + ": source table of " + std::to_string(T) + " rows is >= 4 GiB — these layers run on the generic "
                  "float-atomic kernels (several times slower); shard the graph by destination range to stay on the wave-per-row path (gat_layer_path reports it per layer)");
    return 0;
}

}  // namespace gat

using namespace gat;

extern "C" {

const char* gat_last_error(void) { return g_err.c_str(); }
int gat_abi_version(void) { return GAT_ABI_VERSION; }
int gat_switches(char* buf, int64_t cap) {
    if (!buf || cap <= 0) return fail(GAT_E_INVALID, "gat_switches: null buffer");
    std::string t = choices_text();
#ifdef GAT_EXPERIMENTS
    t = t.empty() ? "[experiment library]" : "[experiment library] " + t;
#endif
    if ((int64_t)t.size() + 1 > cap) return fail(GAT_E_INVALID, "gat_switches: buffer too small");
    memcpy(buf, t.c_str(), t.size() + 1);
    return 0;
}
int gat_device_count(int* count) {
    if (!count) return fail(GAT_E_INVALID, "null count");
    GAT_HIP(hipGetDeviceCount(count));
    return 0;
}
int gat_mem_info(size_t* free_bytes, size_t* total_bytes) {
    GAT_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}

int gat_create(const gat_config* cfg, gat_ctx** out) {
    if (!cfg || !out) return fail(GAT_E_INVALID, "gat_create: null argument");
    if (cfg->num_layers <= 0) return fail(GAT_E_INVALID, "Number of layers must be > 0");
    if (!cfg->heads || !cfg->outdims) return fail(GAT_E_INVALID, "--heads and --outdims are required (no defaults, E:954-955)");
    if (cfg->in_dim <= 0 || cfg->num_classes <= 0) return fail(GAT_E_INVALID, "in_dim and num_classes must be > 0");
    GAT_HIP(hipSetDevice(cfg->device));
    std::unique_ptr<gat_ctx> c(new gat_ctx());
    c->cfg = *cfg;
    c->heads.assign(cfg->heads, cfg->heads + cfg->num_layers);
    c->outdims.assign(cfg->outdims, cfg->outdims + cfg->num_layers);
    c->cfg.heads = c->heads.data();
    c->cfg.outdims = c->outdims.data();
    if (c->cfg.negative_slope == 0.0f) c->cfg.negative_slope = 0.01f;
    if (!(c->cfg.negative_slope > 0.0f && c->cfg.negative_slope <= 1.0f))
        return fail(GAT_E_INVALID, "negative_slope must be in (0, 1] (the kernels use LReLU(x) = max(x, slope*x))");
    c->layers.resize(cfg->num_layers);
    int64_t woff = 0, aoff = 0;
    for (int l = 0; l < cfg->num_layers; ++l) {
        Layer& y = c->layers[l];
        y.H = c->heads[l]; y.D = c->outdims[l];
        if (y.H <= 0 || y.D <= 0) return fail(GAT_E_INVALID, "heads/outdims must be positive");
        y.HD = y.H * y.D;
        y.F = (l == 0) ? cfg->in_dim : c->layers[l - 1].HD;            // E:1115-1118
        y.w_off = woff; y.a_off = aoff;
        woff += (int64_t)y.HD * 2 * y.F;                                 // E:1248-1254
        aoff += y.HD;
        c->HDmax = std::max(c->HDmax, y.HD);
        c->Hmax = std::max(c->Hmax, y.H);
    }
    c->nW = woff; c->nA = aoff;
    c->nWo = (int64_t)cfg->num_classes * c->layers.back().D;
    if (cfg->storage_dtype != GAT_DTYPE_F32 && cfg->storage_dtype != GAT_DTYPE_BF16)
        return fail(GAT_E_INVALID, "storage_dtype must be GAT_DTYPE_F32 or GAT_DTYPE_BF16");
#ifdef GAT_EXPERIMENTS                               // the release library does not know the name: a stray variable cannot change results
    if (const char* d = getenv("GAT_DBG")) c->dbg = atoi(d);
#endif
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { GAT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    gat_ctx* p = c.get();
    const int64_t np = c->nW + c->nA + c->nWo;
    GAT_TRY(dalloc(p, &p->params, np));
    GAT_TRY(dalloc(p, &p->grads, np + 4));
    GAT_TRY(dalloc(p, &p->clip_scratch, 4));
    GAT_HIP(hipMemsetAsync(p->params, 0, np * sizeof(float), p->stream));
    GAT_HIP(hipMemsetAsync(p->grads, 0, np * sizeof(float), p->stream));
    *out = c.release();
    return 0;
}

int gat_destroy(gat_ctx* c) {
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->ev_pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto& p : c->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    c->comm.reset();
    halo_free(&c->halo);
    for (hipEvent_t e : c->comm_events) (void)hipEventDestroy(e);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    for (hipEvent_t e : c->ev_fork) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_join) (void)hipEventDestroy(e);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    if (c->pinned_tail) (void)hipHostFree(c->pinned_tail);
    for (void* p : c->owned) (void)hipFree(p);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int gat_sync(gat_ctx* c) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    GAT_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- data -------------------------------------------------------------------------------------------------
static int set_graph_common(gat_ctx* c, const int32_t* row_ptr, const int32_t* col_idx, int64_t n_rows,
                            int64_t n_edges, int64_t n_table, int64_t table_row0, hipMemcpyKind kind) {
    if (!c || !row_ptr || (!col_idx && n_edges > 0)) return fail(GAT_E_INVALID, "gat_set_graph: null argument");
    if (c->have_graph) return fail(GAT_E_STATE, "gat_set_graph: graph already set (create a new context)");
    if (n_rows <= 0 || n_edges < 0) return fail(GAT_E_INVALID, "gat_set_graph: bad sizes");
    if (n_edges > 0x7fffffffLL || n_rows >= 0x7fffffffLL || n_table > 0x7fffffffLL)
        return fail(GAT_E_UNSUPPORTED, "gat_set_graph: int32 CSR limits (E:1045-1046) exceeded");
    if (table_row0 < 0 || table_row0 + n_rows > n_table) return fail(GAT_E_INVALID, "gat_set_graph: shard rows outside the table");
    if (kind == hipMemcpyHostToDevice) {
        if (row_ptr[0] != 0 || (int64_t)row_ptr[n_rows] != n_edges)
            return fail(GAT_E_INVALID, "Invalid row_ptr: must start at 0 and end at the edge count");
        for (int64_t i = 0; i < n_rows; ++i)
            if (row_ptr[i + 1] < row_ptr[i]) return fail(GAT_E_INVALID, "Invalid row_ptr: not monotone");
        for (int64_t e = 0; e < n_edges; ++e)
            if (col_idx[e] < 0 || (int64_t)col_idx[e] >= n_table) return fail(GAT_E_INVALID, "col_idx entry outside the node table");
    }
    c->n_rows = n_rows; c->n_edges = n_edges; c->n_table = n_table; c->table_row0 = table_row0;
    GAT_TRY(dalloc(c, &c->row_ptr, n_rows + 1));
    GAT_TRY(dalloc(c, &c->col_idx, n_edges));
    GAT_HIP(hipMemcpyAsync(c->row_ptr, row_ptr, (n_rows + 1) * sizeof(int32_t), kind, c->stream));
    if (n_edges > 0) GAT_HIP(hipMemcpyAsync(c->col_idx, col_idx, n_edges * sizeof(int32_t), kind, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    if (kind == hipMemcpyHostToDevice) {
        build_worklist(row_ptr, n_rows, c->work);
    } else {
        std::vector<int32_t> h(n_rows + 1);
        GAT_HIP(hipMemcpy(h.data(), c->row_ptr, (n_rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
        build_worklist(h.data(), n_rows, c->work);
    }
    c->have_graph = true;
    return ensure_buffers(c);
}
int gat_set_graph(gat_ctx* c, const int32_t* row_ptr, const int32_t* col_idx, int64_t n_rows, int64_t n_edges,
                  int64_t n_table, int64_t table_row0) {
    return set_graph_common(c, row_ptr, col_idx, n_rows, n_edges, n_table, table_row0, hipMemcpyHostToDevice);
}
int gat_set_graph_device(gat_ctx* c, const int32_t* row_ptr, const int32_t* col_idx, int64_t n_rows,
                         int64_t n_edges, int64_t n_table, int64_t table_row0) {
    return set_graph_common(c, row_ptr, col_idx, n_rows, n_edges, n_table, table_row0, hipMemcpyDeviceToDevice);
}

static int set_features_common(gat_ctx* c, const float* x, int64_t n_rows, int32_t in_dim, hipMemcpyKind kind) {
    if (!c || !x) return fail(GAT_E_INVALID, "gat_set_features: null argument");
    if (in_dim != c->cfg.in_dim) return fail(GAT_E_INVALID, "gat_set_features: in_dim differs from the config");
    if (c->have_graph && n_rows != c->n_rows) return fail(GAT_E_INVALID, "gat_set_features: row count differs from the graph");
    c->ld0 = (in_dim + 3) / 4 * 4;
    if (!c->X0) GAT_TRY(dalloc(c, &c->X0, n_rows * c->ld0));
    GAT_TRY(upload_rows_padded(c, c->X0, x, n_rows, in_dim, kind));
    GAT_HIP(hipStreamSynchronize(c->stream));
    c->have_x = true;
    return ensure_buffers(c);
}
int gat_set_features(gat_ctx* c, const float* x, int64_t n_rows, int32_t in_dim) {
    return set_features_common(c, x, n_rows, in_dim, hipMemcpyHostToDevice);
}
int gat_set_features_device(gat_ctx* c, const float* x, int64_t n_rows, int32_t in_dim) {
    return set_features_common(c, x, n_rows, in_dim, hipMemcpyDeviceToDevice);
}
// The layer-0 input is static, so a shard may hold it for EVERY row of the source table instead of
// exchanging layer-0 projections: PL_0 is then computed locally for the whole table and
// gradW_left_0 is accumulated from the shard's partial gPL_0 table (the parameter-gradient
// all-reduce completes the sum) — layer 0 needs neither the all-gather nor the reduce-scatter.
static int set_source_features_common(gat_ctx* c, const float* x, int64_t n_table, int32_t in_dim, hipMemcpyKind kind) {
    if (!c || !x) return fail(GAT_E_INVALID, "gat_set_source_features: null argument");
    if (in_dim != c->cfg.in_dim) return fail(GAT_E_INVALID, "gat_set_source_features: in_dim differs from the config");
    if (!c->have_graph) return fail(GAT_E_STATE, "gat_set_source_features: set the graph first");
    if (n_table != c->n_table) return fail(GAT_E_INVALID, "gat_set_source_features: row count differs from the source table");
    if (c->buffers_ready) return fail(GAT_E_STATE, "gat_set_source_features: call before the features/labels complete the context");
    if (c->X0) return fail(GAT_E_STATE, "gat_set_source_features: features already set (the table replaces gat_set_features)");
    c->ld0 = (in_dim + 3) / 4 * 4;
    GAT_TRY(dalloc(c, &c->Xtab, n_table * c->ld0));
    GAT_TRY(upload_rows_padded(c, c->Xtab, x, n_table, in_dim, kind));
    GAT_HIP(hipStreamSynchronize(c->stream));
    c->X0 = c->Xtab + c->table_row0 * c->ld0;      // the shard's own rows are rows table_row0.. of the table
    c->have_x = true;
    return ensure_buffers(c);
}
int gat_set_source_features(gat_ctx* c, const float* x, int64_t n_table, int32_t in_dim) {
    return set_source_features_common(c, x, n_table, in_dim, hipMemcpyHostToDevice);
}
int gat_set_source_features_device(gat_ctx* c, const float* x, int64_t n_table, int32_t in_dim) {
    return set_source_features_common(c, x, n_table, in_dim, hipMemcpyDeviceToDevice);
}
int gat_layer_path(gat_ctx* c, int32_t l, int32_t* path) {
    if (!c || !path) return fail(GAT_E_INVALID, "null argument");
    if (l < 0 || l >= c->cfg.num_layers) return fail(GAT_E_INVALID, "layer index out of range");
    if (!c->have_graph) return fail(GAT_E_STATE, "gat_layer_path: set the graph first");
    const Layer& y = c->layers[l];
    *path = edge_fast_path(y.H, y.D, c->n_table) ? GAT_PATH_FAST : (edge_fast_path(y.H, y.D, 1) ? GAT_PATH_GENERIC_SIZE : GAT_PATH_GENERIC_SHAPE);
    return 0;
}
int gat_layer_exchange(gat_ctx* c, int32_t l, int32_t* needed) {
    if (!c || !needed) return fail(GAT_E_INVALID, "null argument");
    if (l < 0 || l >= c->cfg.num_layers) return fail(GAT_E_INVALID, "layer index out of range");
    *needed = (c->n_table != c->n_rows) && !(l == 0 && c->Xtab);
    return 0;
}
static void graph_drop_fwd(gat_ctx* c);             // a captured step (gat_step_graph) holds pointers: re-arm when they change
static int set_labels_common(gat_ctx* c, const int32_t* labels, int64_t n_rows, hipMemcpyKind kind) {
    if (!c || !labels) return fail(GAT_E_INVALID, "gat_set_labels: null argument");
    if (c->have_graph && n_rows != c->n_rows) return fail(GAT_E_INVALID, "Invalid labels length");
    if (kind == hipMemcpyHostToDevice)
        for (int64_t i = 0; i < n_rows; ++i)
            if (labels[i] < 0 || labels[i] >= c->cfg.num_classes) return fail(GAT_E_INVALID, "label outside [0, num_classes)");
    if (c->labels && n_rows != c->n_labels) return fail(GAT_E_INVALID, "gat_set_labels: length differs from the labels set before");
    if (!c->labels) GAT_TRY(dalloc(c, &c->labels, n_rows));
    c->n_labels = n_rows;
    GAT_HIP(hipMemcpyAsync(c->labels, labels, n_rows * sizeof(int32_t), kind, c->stream));
    // an active training mask applies to the NEW labels too: labels_eff holds masked copies, not a view
    if (c->labels_eff != nullptr) GAT_TRY(launch_apply_mask(c->labels, c->train_mask, c->labels_eff_buf, n_rows, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    c->have_labels = true;
    return ensure_buffers(c);
}
// ---- train / validation masks (the reference trains and evaluates on ALL nodes, README R:134: "later") ----------
static int upload_mask(gat_ctx* c, const uint8_t* mask, int64_t n_rows, uint8_t** dst) {
    if (!c->have_labels) return fail(GAT_E_STATE, "set the labels first");
    if (n_rows != c->n_labels || (c->have_graph && n_rows != c->n_rows)) return fail(GAT_E_INVALID, "mask length differs from the node count");
    if (!*dst) GAT_TRY(dalloc(c, dst, n_rows));
    GAT_HIP(hipMemcpyAsync(*dst, mask, (size_t)n_rows, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int gat_set_train_mask(gat_ctx* c, const uint8_t* mask, int64_t n_rows) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (!mask) {                                    // back to the reference's behaviour: every node trains
        c->labels_eff = nullptr;                    // (the buffer stays owned by the context)
        graph_drop_fwd(c);
        return 0;
    }
    GAT_TRY(upload_mask(c, mask, n_rows, &c->train_mask));
    if (!c->labels_eff_buf) GAT_TRY(dalloc(c, &c->labels_eff_buf, n_rows));      // one buffer, reused by later calls
    GAT_TRY(launch_apply_mask(c->labels, c->train_mask, c->labels_eff_buf, n_rows, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    c->labels_eff = c->labels_eff_buf;
    graph_drop_fwd(c);                              // a captured step holds the old label pointer
    return 0;
}
int gat_eval_mask(gat_ctx* c, const uint8_t* mask, int64_t n_rows, double* loss_sum, int32_t* n_correct, int32_t* n_nodes) {
    GAT_TRY(check_layer(c, 0));
    if (!mask) return fail(GAT_E_INVALID, "gat_eval_mask: null mask");
    if (!c->y_valid) GAT_TRY(gat_head_forward(c, nullptr, nullptr));     // after a fused head step: y = f(H_L, Wo), both still there
    GAT_TRY(upload_mask(c, mask, n_rows, &c->mask_tmp));
    const int blocks = 256;
    if (!c->eval_loss) { GAT_TRY(dalloc(c, &c->eval_loss, blocks)); GAT_TRY(dalloc(c, &c->eval_cnt, 2 * blocks)); }
    GAT_TRY(launch_eval_mask(c->y, c->labels, c->mask_tmp, n_rows, c->cfg.num_classes, c->eval_loss, c->eval_cnt, blocks, c->stream));
    std::vector<double> hl(blocks); std::vector<int32_t> hc(2 * blocks);
    GAT_HIP(hipMemcpyAsync(hl.data(), c->eval_loss, blocks * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(hipMemcpyAsync(hc.data(), c->eval_cnt, 2 * blocks * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    double l = 0.0; int64_t cr = 0, nn = 0;
    for (int b = 0; b < blocks; ++b) { l += hl[b]; cr += hc[2 * b]; nn += hc[2 * b + 1]; }
    if (loss_sum) *loss_sum = l;
    if (n_correct) *n_correct = (int32_t)cr;
    if (n_nodes) *n_nodes = (int32_t)nn;
    return 0;
}
int gat_set_labels(gat_ctx* c, const int32_t* labels, int64_t n_rows) {
    return set_labels_common(c, labels, n_rows, hipMemcpyHostToDevice);
}
int gat_set_labels_device(gat_ctx* c, const int32_t* labels, int64_t n_rows) {
    return set_labels_common(c, labels, n_rows, hipMemcpyDeviceToDevice);
}

// ---- parameters --------------------------------------------------------------------------------------------
static int group_span(gat_ctx* c, int group, int64_t* off, int64_t* cnt) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    switch (group) {
        case GAT_PARAM_W: *off = 0; *cnt = c->nW; return 0;
        case GAT_PARAM_A: *off = c->nW; *cnt = c->nA; return 0;
        case GAT_PARAM_WO: *off = c->nW + c->nA; *cnt = c->nWo; return 0;
        default: return fail(GAT_E_INVALID, "unknown parameter group");
    }
}
int gat_param_count(gat_ctx* c, int group, int64_t* count) {
    int64_t off;
    return group_span(c, group, &off, count);
}
static int copy_group(gat_ctx* c, float* base, int group, float* host, int64_t count, bool to_device) {
    int64_t off, cnt;
    GAT_TRY(group_span(c, group, &off, &cnt));
    if (count != cnt || !host) return fail(GAT_E_INVALID, "parameter group size mismatch");
    if (to_device) GAT_HIP(hipMemcpyAsync(base + off, host, cnt * sizeof(float), hipMemcpyHostToDevice, c->stream));
    else GAT_HIP(hipMemcpyAsync(host, base + off, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int gat_params_set(gat_ctx* c, int group, const float* host, int64_t count) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    return copy_group(c, c->params, group, const_cast<float*>(host), count, true);
}
int gat_params_get(gat_ctx* c, int group, float* host, int64_t count) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    return copy_group(c, c->params, group, host, count, false);
}
int gat_grads_get(gat_ctx* c, int group, float* host, int64_t count) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    return copy_group(c, c->grads, group, host, count, false);
}
int gat_grads_set(gat_ctx* c, int group, const float* host, int64_t count) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    return copy_group(c, c->grads, group, const_cast<float*>(host), count, true);
}
int gat_grads_device(gat_ctx* c, void** d_ptr, int64_t* count) {
    if (!c || !d_ptr || !count) return fail(GAT_E_INVALID, "null argument");
    *d_ptr = c->grads; *count = c->nW + c->nA + c->nWo;
    return 0;
}

int gat_grads_export(gat_ctx* c, void* d_dst, int64_t count) {
    if (!c || !d_dst) return fail(GAT_E_INVALID, "null argument");
    if (count != c->nW + c->nA + c->nWo) return fail(GAT_E_INVALID, "packed gradient size mismatch");
    GAT_HIP(hipMemcpyAsync(d_dst, c->grads, count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int gat_result_export(gat_ctx* c, void* d_dst3) {
    if (!c || !d_dst3) return fail(GAT_E_INVALID, "null argument");
    if (!c->buffers_ready) return fail(GAT_E_STATE, "gat_result_export: no forward pass yet");
    return launch_pack_result(c->loss_out, c->correct_out, (float*)d_dst3, c->stream);
}
int gat_grads_import(gat_ctx* c, const void* d_src, int64_t count) {
    if (!c || !d_src) return fail(GAT_E_INVALID, "null argument");
    if (count != c->nW + c->nA + c->nWo) return fail(GAT_E_INVALID, "packed gradient size mismatch");
    GAT_HIP(hipMemcpyAsync(c->grads, d_src, count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int gat_params_init(gat_ctx* c, uint64_t seed) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    // Same distribution as xavier_init_kernel_curand (E:205-242): U(-lim, lim] with lim = sqrt(6/(2F+D)) for the W rows
    // and a of a layer, sqrt(6/(C+D_L)) for W_o; own counter-based stream ON THE DEVICE, since the reference's cuRAND
    // XORWOW stream is seeded with time(NULL) (E:1305) and unreproducible.  Draw order: per layer W then a, then W_o.
    const uint64_t s0 = seed * 0x9E3779B97F4A7C15ull + 0x67617432ull;
    uint64_t draw = 0;
    for (int l = 0; l < c->cfg.num_layers; ++l) {
        const Layer& y = c->layers[l];
        const float lim = sqrtf(6.0f / (float)(2 * y.F + y.D));
        const int64_t nw = (int64_t)y.HD * 2 * y.F;
        GAT_TRY(launch_xavier_init(c->params + y.w_off, nw, s0, draw, lim, c->stream));
        draw += (uint64_t)nw;
        GAT_TRY(launch_xavier_init(c->params + c->nW + y.a_off, y.HD, s0, draw, lim, c->stream));
        draw += (uint64_t)y.HD;
    }
    const float limo = sqrtf(6.0f / (float)(c->cfg.num_classes + c->layers.back().D));
    GAT_TRY(launch_xavier_init(c->params + c->nW + c->nA, c->nWo, s0, draw, limo, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- phases --------------------------------------------------------------------------------------------------
int gat_layer_project(gat_ctx* c, int32_t l) {
    GAT_TRY(check_layer(c, l));
    Layer& y = c->layers[l];
    Scope t(c, GAT_K_PROJECT);
    if (l == 0 && c->Xtab) {      // replicated input: whole PL table from the table rows, PR from the shard's rows
        GAT_TRY(launch_project(c->Xtab, W_of(c, l), y.PL, nullptr, c->n_table, y.F, y.HD, kPartLeft, bf16(c), c->gw_scratch, c->gw_scratch_floats, c->stream, c->ld0));
        return launch_project(c->X0, W_of(c, l), nullptr, y.PR, c->n_rows, y.F, y.HD, kPartRight, bf16(c), c->gw_scratch, c->gw_scratch_floats, c->stream, c->ld0);
    }
    float* own_rows = reinterpret_cast<float*>(reinterpret_cast<char*>(y.PL) + c->table_row0 * y.HD * st_bytes(c));
    return launch_project(Xin_of(c, l), W_of(c, l), own_rows, y.PR, c->n_rows, y.F, y.HD, kPartBoth, bf16(c), c->gw_scratch, c->gw_scratch_floats, c->stream, ldX_of(c, l));
}

static EdgeFwdArgs plan_forward_edges(gat_ctx* c, int32_t l);
int gat_layer_forward_edges(gat_ctx* c, int32_t l) {
    GAT_TRY(check_layer(c, l));
    const EdgeFwdArgs a = plan_forward_edges(c, l);
    Scope t(c, GAT_K_EDGE_FWD);
    return launch_edge_forward(a, c->stream);
}
static EdgeFwdArgs plan_forward_edges(gat_ctx* c, int32_t l) {
    Layer& y = c->layers[l];
    EdgeFwdArgs a{};
    a.row_ptr = c->row_ptr; a.col_idx = c->col_idx; a.PL = y.PL; a.PR = y.PR; a.a = a_of(c, l);
    a.alpha = y.alpha; a.score = y.score; a.hpre = y.hpre; a.hout = y.hout; a.mstat = y.mstat; a.zstat = y.zstat;
    a.n_rows = c->n_rows; a.n_table = c->n_table; a.bf16 = bf16(c); a.H = y.H; a.D = y.D; a.is_last = (l == c->cfg.num_layers - 1);
    a.slope = c->cfg.negative_slope;
    a.items = c->items; a.n_items = c->work.n_items; a.slot_info = c->slot_info; a.n_slots = c->work.n_slots; a.n_split = c->work.n_split;
    a.part_acc = c->part_acc; a.part_mz = c->part_mz;
    return a;
}

int gat_head_forward(gat_ctx* c, float* loss_sum, int32_t* n_correct) {
    GAT_TRY(check_layer(c, 0));
    const Layer& y = c->layers.back();
    HeadArgs a{};
    a.Wo = Wo_of(c); a.HL = y.hout; a.labels = c->labels_eff ? c->labels_eff : c->labels; a.y = c->y;
    a.loss_partial = c->loss_partial; a.correct_partial = c->correct_partial;
    a.loss_out = c->loss_out; a.correct_out = c->correct_out;
    a.n_rows = c->n_rows; a.C = c->cfg.num_classes; a.DL = y.D;
    {
        Scope t(c, GAT_K_HEAD_FWD);
        GAT_TRY(launch_head_forward(a, c->stream));
    }
    c->y_valid = true;
    if (loss_sum || n_correct) {
        float hl = 0.f; int32_t hc = 0;
        GAT_HIP(hipMemcpyAsync(&hl, c->loss_out, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        GAT_HIP(hipMemcpyAsync(&hc, c->correct_out, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        GAT_HIP(hipStreamSynchronize(c->stream));
        if (loss_sum) *loss_sum = hl;
        if (n_correct) *n_correct = hc;
    }
    return 0;
}

int gat_head_backward(gat_ctx* c) {
    GAT_TRY(check_layer(c, 0));
    const Layer& y = c->layers.back();
    HeadBwdArgs a{};
    a.Wo = Wo_of(c); a.HL = y.hout; a.y = c->y; a.labels = c->labels_eff ? c->labels_eff : c->labels; a.hpre = y.hpre; a.g = y.g; a.gh_out = c->gH; a.gh_stride = c->gh_stride;
    a.gradWo = gWo_of(c); a.partial = c->hb_partial; a.n_rows = c->n_rows; a.C = c->cfg.num_classes;
    a.DL = y.D; a.H = y.H; a.slope = c->cfg.negative_slope; a.flat_index = c->cfg.flat_lrelu_index;
    Scope t(c, GAT_K_HEAD_BWD);
    return launch_head_backward(a, c->stream);
}

// Arguments of layer l's edge backward (shared by gat_layer_backward_edges and the fused last layer of gat_step).
struct BwdPlan { EdgeBwdArgs a; bool store, stash, last_g; };
static int plan_backward_edges(gat_ctx* c, int32_t l, BwdPlan* P) {
    Layer& y = c->layers[l];
    const bool store = c->msg != nullptr && edge_fast_path(y.H, y.D, c->n_table);
    const bool stash = store && y.stash && c->stash != nullptr;
    EdgeBwdArgs a{};
    a.row_ptr = c->row_ptr; a.col_idx = c->col_idx; a.PL = y.PL; a.PR = y.PR; a.a = a_of(c, l);
    a.alpha = y.alpha; a.mstat = y.mstat; a.zstat = y.zstat;
    a.hpre = y.hpre; a.g = y.g; a.gPL = gPL_of(c, l); a.gPR = gPR_of(c, l); a.ge = y.ge; a.galpha = y.galpha;
    a.g_raw = l < c->cfg.num_layers - 1;           // hidden layers: written by launch_grad_x without the LReLU' factor
    a.gh = (l == c->cfg.num_layers - 1) ? c->gH : nullptr; a.gh_stride = c->gh_stride; a.hb_stride = 64;
    a.pos = store ? c->csc_pos : nullptr; a.msg = store ? c->msg : nullptr;
    a.stash = stash ? c->stash : nullptr; a.gfull = stash ? c->gfull : nullptr; a.stash_spare = (uint32_t)c->n_edges;
    // last layer: the pull pass rebuilds g from gH and the decision bytes (GAT_PULL_LAST=0: gathers gfull like a hidden layer, A/B)
    // Worth it only when the g rows the pull pass would gather do not stay in the caches (Products shape 627 MB: 2.73 -> 2.29 ms;
    // Arxiv shape 43 MB: the extra loads per slot cost more than the smaller rows save, 1.31 -> 1.25 ms per step without)
    static const int pull_last = [] { const char* e = choice_env("GAT_PULL_LAST"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    // With the slot-parallel pull pass (short lists: shards) the records pay from 64 MB of g rows on: one cache line per slot instead
    // of two, and that pass is bound by lines per second (Products P = 8 shard, 78 MB: 0.84 -> 0.78 ms per step; Arxiv 43 MB: equal)
    const bool runs_form = c->runs.csrc != nullptr && c->n_edges < 8 * c->n_table && c->n_edges >= ((int64_t)512 << 10);
    const bool big_rows = (int64_t)c->n_rows * y.H * y.D * 4 > ((int64_t)(runs_form ? 64 : 128) << 20);
    const bool last_g = stash && a.gh != nullptr && c->hbits != nullptr && !bf16(c) && (pull_last >= 0 ? pull_last == 1 : big_rows) &&
                        c->n_rows < ((int64_t)1 << 26);             // the pull pass addresses the 64-byte node records with 32-bit offsets
    a.hbits = last_g ? c->hbits : nullptr;
    a.items = c->items; a.n_items = c->work.n_items; a.slot_info = c->slot_info; a.n_slots = c->work.n_slots; a.n_split = c->work.n_split;
    a.part_acc = c->part_acc;
    a.dbg = c->dbg;
    a.ga_partial = c->ga_partial + (int64_t)l * 2048 * c->HDmax; a.n_rows = c->n_rows; a.n_table = c->n_table; a.bf16 = bf16(c); a.H = y.H; a.D = y.D;
    a.ga_blocks = edge_fast_path(y.H, y.D, c->n_table) ? edge_backward_blocks(c->work.n_items, y.H, y.D, store, y.ge != nullptr, bf16(c), stash)
                                           : edge_backward_blocks(c->n_rows * 4, y.H, y.D, false, false, false);
    a.slope = c->cfg.negative_slope;
    P->a = a; P->store = store; P->stash = stash; P->last_g = last_g;
    return 0;
}
// The source-major pass of layer l (records -> gPL, or message rows -> gPL)
static int sum_backward_edges(gat_ctx* c, int32_t l, const BwdPlan& P) {
    Layer& y = c->layers[l];
    if (P.stash) {
        Scope t(c, GAT_K_GPL_SUM);
        return launch_gpl_pull(c->csc_ptr, c->stash, c->csc_dst, c->gfull, bf16(c), P.last_g ? c->gH : nullptr, P.last_g ? c->hbits : nullptr, c->gh_stride, 64,
                               a_of(c, l), c->cfg.negative_slope, gPL_of(c, l), c->n_table,
                               c->n_edges, y.H, y.D, c->gpl_chunks, c->n_gpl_chunks, c->gpl_heavy, c->n_gpl_heavy, c->gpl_part,
                               c->pull_items, c->n_pull_items, c->runs.csrc ? &c->runs : nullptr, c->n_edges + kPullPad, c->stream);
    }
    if (P.store) {
        Scope t(c, GAT_K_GPL_SUM);
        return launch_gpl_sum(c->csc_ptr, c->msg, gPL_of(c, l), c->n_table, c->n_edges, y.HD, bf16(c), c->gpl_chunks, c->n_gpl_chunks,
                              c->gpl_heavy, c->n_gpl_heavy, c->gpl_part, c->stream, c->runs.csrc ? &c->runs : nullptr);
    }
    return 0;
}
int gat_layer_backward_edges(gat_ctx* c, int32_t l) {
    GAT_TRY(check_layer(c, l));
    Layer& y = c->layers[l];
    BwdPlan P;
    GAT_TRY(plan_backward_edges(c, l, &P));
    if (!P.store) {
        Scope t(c, GAT_K_MISC);
        GAT_HIP(hipMemsetAsync(gPL_of(c, l), 0, (size_t)c->n_table * y.HD * sizeof(float), c->stream));
    }
    {
        Scope t(c, GAT_K_EDGE_BWD);
        GAT_TRY(launch_edge_backward(P.a, c->stream));
    }
    GAT_TRY(sum_backward_edges(c, l, P));
    Scope t(c, GAT_K_MISC);
    return launch_reduce_partials_add(P.a.ga_partial, P.a.ga_blocks, y.HD, ga_of(c, l), c->stream);
}

// grad_w of layer l on stream st (the context's, or the side stream of the overlapped backward)
static int backward_grad_w(gat_ctx* c, int32_t l, hipStream_t st) {
    Layer& y = c->layers[l];
    const float* gPL_rows = gPL_of(c, l) + c->table_row0 * y.HD;
    Scope t(c, GAT_K_GRAD_W, st);
    if (l == 0 && c->Xtab) {  // partial gPL over the whole table x replicated input; the gradient all-reduce sums shards
        GAT_TRY(launch_grad_w(gPL_of(c, l), nullptr, c->Xtab, gW_of(c, l), c->gw_scratch, c->n_table, y.F, y.HD, kPartLeft, st, c->ld0));
        return launch_grad_w(nullptr, gPR_of(c, l), c->X0, gW_of(c, l), c->gw_scratch, c->n_rows, y.F, y.HD, kPartRight, st, c->ld0);
    }
    return launch_grad_w(gPL_rows, gPR_of(c, l), Xin_of(c, l), gW_of(c, l), c->gw_scratch + c->gw_off[(size_t)l], c->n_rows, y.F, y.HD, kPartBoth, st, ldX_of(c, l));
}
static int backward_grad_x(gat_ctx* c, int32_t l) {
    if (l == 0) return 0;                                             // E:1528
    Layer& y = c->layers[l];
    const float* gPL_rows = gPL_of(c, l) + c->table_row0 * y.HD;
    Scope t(c, GAT_K_GRAD_X);
    // plain dL/d(input) of this layer: the LReLU'(h_pre) factor of E:888-892 is applied by the edge backward
    // of layer l-1, which reads h_pre anyway (the epilogue's extra read of h_pre cost 0.45 of 1.03 ms)
    return launch_grad_x(gPL_rows, gPR_of(c, l), W_of(c, l), nullptr, c->layers[l - 1].g, c->n_rows, y.F, y.HD,
                         c->cfg.negative_slope, c->stream);
}
int gat_layer_backward_dense(gat_ctx* c, int32_t l) {
    GAT_TRY(check_layer(c, l));
    GAT_TRY(backward_grad_w(c, l, c->stream));
    return backward_grad_x(c, l);
}

// ---- whole step (single shard, or a shard with a transport attached) ------------------------------------------
static int64_t table_slice(gat_ctx* c, const Layer& y) { return (c->n_table / c->comm->world) * y.HD; }      // fp32 rows
static int64_t pl_slice(gat_ctx* c, const Layer& y) { return table_slice(c, y) * st_bytes(c) / 4; }        // PL rows, as floats
static bool needs_exchange(gat_ctx* c, int l) { return c->n_table != c->n_rows && !(l == 0 && c->Xtab); }
static int check_step(gat_ctx* c, const char* who) {
    GAT_TRY(check_layer(c, 0));
    if (c->n_table != c->n_rows && !c->comm)
        return fail(GAT_E_STATE, std::string(who) + ": sharded context — attach a transport (gat_comm_init_*) or drive "
                                                   "the phase API with the exchange steps");
    return 0;
}
// Chunk-pipelined forward exchange of one layer (gat_comm_option GAT_COMM_PIPELINE = K > 1): the shard's rows are
// projected in K row chunks on the compute stream; as soon as chunk k exists, its part of every rank's table slice is
// exchanged on a second stream, while chunk k+1 is being projected.  The edge pass waits for the last part.  Chunk
// boundaries are the same slice coordinates on every rank (multiples of 128 rows of max_rows), so the parts have fixed
// counts.  Every PL / PR row is produced by the same sequence of matrix instructions as in one launch: results are bitwise those of the
// unchunked exchange (tests/test_shard.py).  What can hide behind the exchange this way is the projection itself; the
// edge pass needs the whole table.
static int forward_exchange_pipelined(gat_ctx* c, int l) {
    Layer& y = c->layers[l];
    const int K = c->comm_chunks;
    const int64_t max_rows = c->n_table / c->comm->world;
    const int64_t rpc = ((max_rows + K - 1) / K + 127) / 128 * 128;       // rows per chunk, a multiple of the GEMM's row tile
    if (!c->comm_stream) GAT_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    while ((int)c->comm_events.size() < K + 1) {
        hipEvent_t e;
        GAT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->comm_events.push_back(e);
    }
    const int64_t rowf = y.HD * st_bytes(c) / 4;                           // floats per PL row (bf16 rows: half)
    char* own_rows = reinterpret_cast<char*>(y.PL) + c->table_row0 * y.HD * st_bytes(c);
    Scope t(c, GAT_K_EXCHANGE);                                            // timed as a whole: projection chunks + their exchanges
    for (int k = 0; k < K; ++k) {
        const int64_t r0 = (int64_t)k * rpc, r1s = std::min<int64_t>(r0 + rpc, max_rows), r1 = std::min<int64_t>(r1s, c->n_rows);
        if (r0 >= max_rows) break;
        if (r1 > r0)
            GAT_TRY(launch_project(Xin_of(c, l) + r0 * ldX_of(c, l), W_of(c, l), reinterpret_cast<float*>(own_rows + r0 * y.HD * st_bytes(c)),
                                   y.PR + r0 * y.HD, r1 - r0, y.F, y.HD, kPartBoth, bf16(c), nullptr, 0, c->stream, ldX_of(c, l)));
        GAT_HIP(hipEventRecord(c->comm_events[k], c->stream));
        GAT_HIP(hipStreamWaitEvent(c->comm_stream, c->comm_events[k], 0));
        GAT_TRY(c->comm->all_gather_part(y.PL, pl_slice(c, y), r0 * rowf, (r1s - r0) * rowf, c->comm_stream));
    }
    GAT_HIP(hipEventRecord(c->comm_events[K], c->comm_stream));
    GAT_HIP(hipStreamWaitEvent(c->stream, c->comm_events[K], 0));
    return 0;
}
static int forward_phases(gat_ctx* c, int l_end = -1, bool last_edges = true) {
    // layers [0, l_end) completely; last_edges = false: the last of them stops after its projection (+ exchange)
    if (l_end < 0) l_end = c->cfg.num_layers;
    for (int l = 0; l < l_end; ++l) {
        const bool edges = last_edges || l < l_end - 1;
        // the chunked projection runs the streaming kernel; a layer whose one-launch projection takes the split-K kernel
        // (few rows, F > 128: another summation order) keeps the plain exchange, so that K chunks stay bitwise K = 1
        if (c->comm && needs_exchange(c, l) && c->comm_chunks > 1 && !c->halo_on &&
            project_scratch_floats(c->n_rows, c->layers[l].F, c->layers[l].HD, kPartBoth) == 0) {
            GAT_TRY(check_layer(c, l));
            GAT_TRY(forward_exchange_pipelined(c, l));
            if (edges) GAT_TRY(gat_layer_forward_edges(c, l));
            continue;
        }
        GAT_TRY(gat_layer_project(c, l));
        if (c->comm && needs_exchange(c, l)) {
            Scope t(c, GAT_K_EXCHANGE);
            if (c->halo_on) GAT_TRY(halo_forward(c->comm.get(), c->halo, c->layers[l].PL, c->layers[l].HD * st_bytes(c) / 4, c->stream));
            else GAT_TRY(c->comm->all_gather(c->layers[l].PL, pl_slice(c, c->layers[l]), c->stream));
        }
        if (edges) GAT_TRY(gat_layer_forward_edges(c, l));
    }
    return 0;
}
static void head_args(gat_ctx* c, HeadArgs* f, HeadBwdArgs* b) {
    const Layer& y = c->layers.back();
    f->Wo = Wo_of(c); f->HL = y.hout; f->labels = c->labels_eff ? c->labels_eff : c->labels; f->y = c->y;
    f->loss_partial = c->loss_partial; f->correct_partial = c->correct_partial;
    f->loss_out = c->loss_out; f->correct_out = c->correct_out;
    f->n_rows = c->n_rows; f->C = c->cfg.num_classes; f->DL = y.D;
    b->Wo = Wo_of(c); b->HL = y.hout; b->y = c->y; b->labels = c->labels_eff ? c->labels_eff : c->labels; b->hpre = y.hpre; b->g = y.g; b->gh_out = c->gH; b->gh_stride = c->gh_stride;
    b->gradWo = gWo_of(c); b->partial = c->hb_partial; b->n_rows = c->n_rows; b->C = c->cfg.num_classes;
    b->DL = y.D; b->H = y.H; b->slope = c->cfg.negative_slope; b->flat_index = c->cfg.flat_lrelu_index;
}
// gat_step: the head's forward and backward as ONE kernel when nothing needs the class probabilities in HBM
static bool fused_head(gat_ctx* c) {
    if (c->cfg.keep_taps) return false;
    HeadArgs f{}; HeadBwdArgs b{};
    head_args(c, &f, &b);
    return head_step_supported(b);
}
static int head_step(gat_ctx* c, bool with_gh = true) {
    HeadArgs f{}; HeadBwdArgs b{};
    head_args(c, &f, &b);
    if (!with_gh) b.gh_out = nullptr;               // loss, #correct and grad_Wo only (the fused last layer formed gH itself)
    Scope t(c, GAT_K_HEAD_BWD);
    c->y_valid = false;
    return launch_head_step(f, b, c->stream);
}
// gat_step: the last layer's forward edge pass, the output head and its backward edge pass fused per destination row
// (edge_last_fused_kernel) — single shard, fp32, H*D = 64 / D = 8 record path.  The idea: the separate backward finds nothing
// of the forward's rows left in the caches when a pass's gathers exceed the Infinity Cache; fused per row, the second walk of
// a row's sources comes a few microseconds after the first.
static bool fused_last(gat_ctx* c) {
    if (!fused_head(c) || bf16(c) || c->comm || c->n_table != c->n_rows || c->cfg.flat_lrelu_index) return false;
    const Layer& y = c->layers.back();
    if (!edge_fast_path(y.H, y.D, c->n_table) || !y.stash || c->stash == nullptr || c->gH == nullptr) return false;
    if (!edge_last_fused_supported(y.H, y.D, c->cfg.num_classes) || c->dbg != 0) return false;
    // MEASURED, NOT THE DEFAULT (DESIGN §4 "Round 3"): on the Products shape the fused launch takes 5.0 ms for the 76 % of the edges
    // that sit in unsplit rows — what the separate forward + backward take for them — and the split rows' segments, launched on
    // their own, lose what they used to hide behind the bulk: 20.95 vs 20.78 ms per step.  GAT_FUSE_LAST=1 enables.
    static const int env = [] { const char* e = choice_env("GAT_FUSE_LAST"); return e ? (e[0] == '0' ? 0 : 1) : 0; }();
    return env == 1;
}
static int last_layer_fused(gat_ctx* c) {
    const int l = c->cfg.num_layers - 1;
    Layer& y = c->layers[l];
    BwdPlan P;
    GAT_TRY(plan_backward_edges(c, l, &P));
    const EdgeFwdArgs f = plan_forward_edges(c, l);
    const int32_t n_seg = c->work.n_slots;          // the segments of split rows come first in the work list
    float* ga_a = P.a.ga_partial;                   // this layer's region (2048 rows): the segments' launch, then the fused one,
    int blocks_a = 0;                               // back to back — ONE reduction job (two jobs into one output would race)
    if (n_seg > 0) {                                // split rows: forward segments + fix-up, their gH, backward segments + fix-up
        EdgeFwdArgs fs = f;
        fs.n_items = n_seg;
        {
            Scope t(c, GAT_K_EDGE_FWD);
            GAT_TRY(launch_edge_forward(fs, c->stream));
        }
        {
            Scope t(c, GAT_K_HEAD_BWD);
            GAT_TRY(launch_head_rows(c->slot_info, c->work.n_slots, c->work.n_split, Wo_of(c), y.hout, c->labels_eff ? c->labels_eff : c->labels,
                                     c->gH, c->gh_stride, c->cfg.num_classes, y.D, c->stream));
        }
        EdgeBwdArgs bs = P.a;
        bs.n_items = n_seg;
        bs.ga_partial = ga_a;
        bs.ga_blocks = blocks_a = std::min(1024, edge_backward_blocks(n_seg, y.H, y.D, P.store, false, false, P.stash));
        Scope t(c, GAT_K_EDGE_BWD);
        GAT_TRY(launch_edge_backward(bs, c->stream));
    }
    EdgeLastArgs a{};
    a.f = f; a.f.items = c->items + n_seg; a.f.n_items = c->work.n_items - n_seg;
    float* ga_b = ga_a + (int64_t)blocks_a * y.HD;
    a.b = P.a; a.b.ga_partial = ga_b; a.b.ga_blocks = a.f.n_items > 0 ? std::min(edge_last_fused_blocks(a.f.n_items), 2048 - blocks_a) : 0;    // the layer's region holds 2048 partial rows
    a.Wo = Wo_of(c); a.labels = c->labels_eff ? c->labels_eff : c->labels; a.gh_out = c->gH; a.C = c->cfg.num_classes;
    {
        Scope t(c, GAT_K_EDGE_FUSED);
        GAT_TRY(launch_edge_last_fused(a, c->stream));
    }
    GAT_TRY(head_step(c, false));                   // loss, #correct, grad_Wo from the stored H (all rows)
    GAT_TRY(sum_backward_edges(c, l, P));
    Scope t(c, GAT_K_MISC);
    return launch_reduce_partials_add(ga_a, blocks_a + a.b.ga_blocks, y.HD, ga_of(c, l), c->stream);
}
// grad_w of a layer feeds nothing but the parameter gradients, yet on one stream it sits between the layer's source-major pass
// and grad_x -> the next layer's edge passes (E:1517 vs 1533).  With GAT_OVERLAP=1 it runs on a side stream: forked after the
// layer's gPL is complete, joined before that layer's buffer pair is written again (two layers later) or at the end of the
// backward.  MEASURED, NOT THE DEFAULT (DESIGN §4 "Round 4", profiles/r04/experiments/overlap_*): Products shape, same box,
// A/B/A/B 20.98 / 21.18 / 21.21 / 21.42 ms per step and 21.15 / 21.29 / 21.17 / 22.36 with a lowest-priority side stream — the
// 0.31 ms of layer 1's grad_w leave the critical path, and the backward edge pass it runs beside (a persistent grid sized to
// the chip, items dealt statically: any CU that shares its wave slots and fabric requests becomes the tail) slows by
// 0.33-0.36 ms, grad_x by 0.06, grad_w itself from 0.79 to 1.29; under hipGraph replay the fork / join costs more than it hides
// (Arxiv shape 1.12 -> 1.16 ms, Pubmed 0.245 -> 0.263, Cora 0.188 -> 0.202).  The edge passes are bound by fabric requests
// per second, which the dense kernels (4-6 TB/s of row streaming) consume too: there is no idle unit to fill.  Needs the second gPL / gPR pair (odd layers), hence not with a caller-bound gPL table
// or a replicated layer-0 input (its two grad_w launches share one slab region).  Same kernels, same operands: bitwise the
// serial order's gradients.  Captured by gat_step_graph as a fork / join in the graph.
static int overlap_prepare(gat_ctx* c) {
    static const int env = [] { const char* e = choice_env("GAT_OVERLAP"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const int L = c->cfg.num_layers;
    const bool want = env == 1;
    if (!want || L < 2 || c->gPL_bound || c->Xtab) return 0;
    if (!c->gPL_alt) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c->stream, &cs);
        if (cs != hipStreamCaptureStatusNone) return 0;               // first use inside a capture: allocate on the next eager step
        GAT_TRY(dalloc(c, &c->gPL_alt, c->n_table * c->HDmax));
        GAT_TRY(dalloc(c, &c->gPR_alt, c->n_rows * c->HDmax));
        {   // lowest priority: grad_w should fill what the edge passes leave free (their tails), not take their wave slots
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            static const bool prio = [] { const char* e = choice_env("GAT_OVERLAP_PRIO"); return !(e && e[0] == '0'); }();
            if (prio) GAT_HIP(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, least));
            else GAT_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        }
        for (int l = 0; l < L; ++l) {
            hipEvent_t a, b;
            GAT_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            GAT_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            c->ev_fork.push_back(a); c->ev_join.push_back(b);
        }
    }
    return 1;
}
struct OverlapScope {       // ov_active only while backward_phases runs (the phase API keeps the single pair)
    gat_ctx* c;
    explicit OverlapScope(gat_ctx* c_, bool on) : c(c_) { c->ov_active = on; }
    ~OverlapScope() { c->ov_active = false; }
};
static int backward_phases(gat_ctx* c, bool head_done = false, bool last_edges_done = false) {
    // last_edges_done: the last layer's edge passes (incl. its source-major pass) have run already (fused last layer)
    const int L = c->cfg.num_layers;
    const int ov = last_edges_done ? 0 : overlap_prepare(c);          // (the fused last layer wrote the first pair before this call)
    if (ov < 0) return ov;
    OverlapScope scope(c, ov == 1);
    std::vector<char> forked((size_t)L, 0);
    if (!head_done) GAT_TRY(gat_head_backward(c));
    for (int l = L - 1; l >= 0; --l) {
        if (ov && l + 2 < L && forked[(size_t)l + 2]) {               // this layer rewrites the pair layer l+2's grad_w reads
            GAT_HIP(hipStreamWaitEvent(c->stream, c->ev_join[(size_t)l + 2], 0));
            forked[(size_t)l + 2] = 0;
        }
        if (!(last_edges_done && l == L - 1)) GAT_TRY(gat_layer_backward_edges(c, l));
        if (c->comm && needs_exchange(c, l)) {
            Scope t(c, GAT_K_EXCHANGE);
            if (c->halo_on) GAT_TRY(halo_backward(c->comm.get(), c->halo, gPL_of(c, l), c->layers[l].HD, c->stream));
            else if (c->comm_gpl_bf16) GAT_TRY(c->comm->reduce_scatter_bf16(gPL_of(c, l), table_slice(c, c->layers[l]), c->stream));
            else GAT_TRY(c->comm->reduce_scatter(gPL_of(c, l), table_slice(c, c->layers[l]), c->stream));
        }
        if (ov && l > 0) {                                            // layer 0's grad_w is the last kernel: nothing to hide behind
            GAT_HIP(hipEventRecord(c->ev_fork[(size_t)l], c->stream));
            GAT_HIP(hipStreamWaitEvent(c->side_stream, c->ev_fork[(size_t)l], 0));
            GAT_TRY(backward_grad_w(c, l, c->side_stream));
            GAT_HIP(hipEventRecord(c->ev_join[(size_t)l], c->side_stream));
            forked[(size_t)l] = 1;
            GAT_TRY(backward_grad_x(c, l));
        } else {
            GAT_TRY(backward_grad_w(c, l, c->stream));
            GAT_TRY(backward_grad_x(c, l));
        }
    }
    for (int l = 0; l < L; ++l)
        if (forked[(size_t)l]) GAT_HIP(hipStreamWaitEvent(c->stream, c->ev_join[(size_t)l], 0));        // join: the slab reductions follow
    return 0;
}
// tail of the packed gradient buffer -> host values (after an all-reduce the sums over shards)
static int read_result_tail(gat_ctx* c, float* loss_sum, int32_t* n_correct) {
    float t[3] = {0.f, 0.f, 0.f};
    GAT_HIP(hipMemcpyAsync(t, c->grads + c->nW + c->nA + c->nWo, sizeof(t), hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    if (loss_sum) *loss_sum = t[0];
    if (n_correct) *n_correct = (int32_t)(t[1] + 4096.0f * t[2] + 0.5f);
    return 0;
}
int gat_forward(gat_ctx* c, float* loss_sum, int32_t* n_correct) {
    GAT_TRY(check_step(c, "gat_forward"));
    GAT_TRY(forward_phases(c));
    if (!c->comm) return gat_head_forward(c, loss_sum, n_correct);
    GAT_TRY(gat_head_forward(c, nullptr, nullptr));
    if (!loss_sum && !n_correct) return 0;
    float* tail = c->grads + c->nW + c->nA + c->nWo;
    GAT_TRY(launch_pack_result(c->loss_out, c->correct_out, tail, c->stream));
    {
        Scope t(c, GAT_K_EXCHANGE);
        GAT_TRY(c->comm->all_reduce(tail, 3, c->stream));
    }
    return read_result_tail(c, loss_sum, n_correct);
}
// The gradient buffer ACCUMULATES until gat_zero_grad (E:1262-1266, 1631-1633).  With a transport the step's
// contribution must be summed over shards exactly once: the accumulated (already reduced) values are set
// aside, the step runs into a zeroed buffer, that is all-reduced, and the old values are added back —
// without this, a second step before gat_zero_grad would reduce the earlier sums again (x world).
static int reduce_begin(gat_ctx* c) {
    if (!c->comm) return 0;
    const int64_t np = c->nW + c->nA + c->nWo;
    if (!c->grads_prev) GAT_TRY(dalloc(c, &c->grads_prev, np));
    Scope t(c, GAT_K_MISC);
    GAT_HIP(hipMemcpyAsync(c->grads_prev, c->grads, (size_t)np * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    GAT_HIP(hipMemsetAsync(c->grads, 0, (size_t)np * sizeof(float), c->stream));
    return 0;
}
static int reduce_end(gat_ctx* c, int64_t count) {          // count: n_params, or n_params + 3 with the result tail
    if (!c->comm) return 0;
    {
        Scope t(c, GAT_K_EXCHANGE);
        GAT_TRY(c->comm->all_reduce(c->grads, count, c->stream));
    }
    Scope t(c, GAT_K_MISC);
    return launch_reduce_partials_add(c->grads_prev, 1, c->nW + c->nA + c->nWo, c->grads, c->stream);
}
// Slab reductions of the backward (grad_Wo, grad_a and grad_W of every layer) and the result pack as one launch at the
// end (ReduceBatch).  Not with replicated layer-0 input: its two grad_w launches share one slab region.  GAT_REDUCE_BATCH=0: A/B.
struct BatchScope {
    bool on;
    explicit BatchScope(gat_ctx* c) {
        static const bool env_on = [] { const char* e = choice_env("GAT_REDUCE_BATCH"); return !(e && e[0] == '0'); }();
        on = env_on && !c->Xtab;
        if (on) reduce_batch_begin();
    }
    // host_tail: pinned copy of the packed result written by the same kernel (*host_done = it was: no copy node needed)
    int finish(gat_ctx* c, bool pack, float* host_tail = nullptr, bool* host_done = nullptr) {
        float* tail = c->grads + c->nW + c->nA + c->nWo;
        if (host_done) *host_done = on && pack && host_tail != nullptr;
        if (!on) return pack ? launch_pack_result(c->loss_out, c->correct_out, tail, c->stream) : 0;
        on = false;
        if (pack) reduce_batch_pack(c->loss_out, c->correct_out, tail, host_tail);
        Scope t(c, GAT_K_MISC);
        return reduce_batch_flush(c->stream);
    }
    ~BatchScope() { if (on) reduce_batch_abort(); }       // error return in between: nothing stays queued
};
int gat_backward(gat_ctx* c) {
    GAT_TRY(check_step(c, "gat_backward"));
    GAT_TRY(reduce_begin(c));
    {
        BatchScope batch(c);
        GAT_TRY(backward_phases(c));
        GAT_TRY(batch.finish(c, false));
    }
    return reduce_end(c, c->nW + c->nA + c->nWo);
}
// The whole step as ONE graph launch: small graphs (Cora / Pubmed / Arxiv shapes) are launch-bound — ~25
// kernels of a few microseconds each — so the sequence is captured once from the context's stream and
// replayed.  First call after arming runs eagerly (fills the occupancy caches, which may not be queried
// during capture), the second captures, later ones replay.  Everything the step touches has a fixed
// address once the context is complete; gat_bind_table re-arms.
static void graph_drop(gat_ctx* c);
static void graph_drop_fwd(gat_ctx* c) { graph_drop(c); }
static void graph_drop(gat_ctx* c) {
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->graph_state == 2) c->graph_state = 1;
    c->graph_warm = 0;
}
static int step_body(gat_ctx* c, float* host_tail = nullptr, bool* host_done = nullptr) {
    if (host_done) *host_done = false;
    if (fused_last(c)) {
        GAT_TRY(forward_phases(c, c->cfg.num_layers, false));      // every layer; the last one stops after its projection
        BatchScope batch(c);
        GAT_TRY(last_layer_fused(c));
        GAT_TRY(backward_phases(c, true, true));
        return batch.finish(c, true, host_tail, host_done);
    }
    GAT_TRY(forward_phases(c));
    BatchScope batch(c);
    if (fused_head(c)) {
        GAT_TRY(head_step(c));
        GAT_TRY(backward_phases(c, true));
    } else {
        GAT_TRY(gat_head_forward(c, nullptr, nullptr));
        GAT_TRY(backward_phases(c));
    }
    return batch.finish(c, true, host_tail, host_done);
}
static int step_graph(gat_ctx* c, float* loss_sum, int32_t* n_correct) {
    const int64_t np = c->nW + c->nA + c->nWo;
    if (c->graph_state == 1 && c->graph_warm == 0) {            // eager warm-up
        GAT_TRY(step_body(c));
        c->graph_warm = 1;
        return (loss_sum || n_correct) ? read_result_tail(c, loss_sum, n_correct) : 0;
    }
    if (c->graph_state == 1) {                                  // capture
        if (!c->pinned_tail) GAT_HIP(hipHostMalloc((void**)&c->pinned_tail, 4 * sizeof(float)));
        GAT_HIP(hipStreamSynchronize(c->stream));
        GAT_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        bool host_done = false;                             // the batch kernel's pack block wrote the pinned copy itself
        int rc = step_body(c, c->pinned_tail, &host_done);
        if (rc == 0 && !host_done && hipMemcpyAsync(c->pinned_tail, c->grads + np, 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = fail(GAT_E_STATE, "gat_step_graph: capture of the result copy failed");
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc != 0 || e != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            c->graph_state = 0;                                     // fall back to eager for good
            return rc != 0 ? rc : fail((int)e, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
        c->graph = g;
        GAT_HIP(hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
        c->graph_state = 2;
    }
    GAT_HIP(hipGraphLaunch(c->graph_exec, c->stream));
    if (!loss_sum && !n_correct) return 0;
    GAT_HIP(hipStreamSynchronize(c->stream));
    if (loss_sum) *loss_sum = c->pinned_tail[0];
    if (n_correct) *n_correct = (int32_t)(c->pinned_tail[1] + 4096.0f * c->pinned_tail[2] + 0.5f);
    return 0;
}
int gat_step_graph(gat_ctx* c, int32_t enable) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (enable && c->comm) return fail(GAT_E_UNSUPPORTED, "gat_step_graph: not with a transport attached (the exchanges are issued eagerly)");
    if (enable && c->cfg.collect_timing) return fail(GAT_E_UNSUPPORTED, "gat_step_graph: not with collect_timing (event pairs cannot be read back from a replay)");
    graph_drop(c);
    c->graph_state = enable ? 1 : 0;
    return 0;
}
int gat_step(gat_ctx* c, float* loss_sum, int32_t* n_correct) {
    GAT_TRY(check_step(c, "gat_step"));
    if (c->graph_state != 0) return step_graph(c, loss_sum, n_correct);
    GAT_TRY(reduce_begin(c));
    GAT_TRY(step_body(c));                              // ends with the packed {loss, correct} behind the gradients
    GAT_TRY(reduce_end(c, c->nW + c->nA + c->nWo + 3));
    if (!loss_sum && !n_correct) return 0;
    return read_result_tail(c, loss_sum, n_correct);
}

// ---- transports ----------------------------------------------------------------------------------------------
static int check_comm_target(gat_ctx* c, int32_t world, int32_t rank) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (c->comm) return fail(GAT_E_STATE, "a transport is already attached");
    if (!c->have_graph) return fail(GAT_E_STATE, "gat_comm_init: set the graph first");
    if (world < 1 || rank < 0 || rank >= world) return fail(GAT_E_INVALID, "gat_comm_init: bad world / rank");
    if (c->n_table % world != 0 || c->table_row0 != (int64_t)rank * (c->n_table / world) || c->n_rows > c->n_table / world)
        return fail(GAT_E_INVALID, "gat_comm_init: the source table must be [world][max_rows] with this shard's rows at rank*max_rows");
    return 0;
}
int gat_comm_unique_id(void* id_out) { return comm_unique_id(id_out); }
int gat_comm_init_rccl(gat_ctx* c, int32_t world, int32_t rank, const void* id) {
    GAT_TRY(check_comm_target(c, world, rank));
    Comm* cm = nullptr;
    GAT_TRY(comm_create_rccl(world, rank, id, &cm));
    c->comm.reset(cm);
    return 0;
}
int gat_comm_init_host(gat_ctx* c, int32_t world, int32_t rank, const char* shm_name, int64_t bytes_per_rank) {
    GAT_TRY(check_comm_target(c, world, rank));
    Comm* cm = nullptr;
    GAT_TRY(comm_create_host(world, rank, shm_name, bytes_per_rank, &cm));
    c->comm.reset(cm);
    return 0;
}
int gat_comm_option(gat_ctx* c, int32_t option, int32_t value) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (option == GAT_COMM_GPL_BF16) {
        if (value != 0 && c->halo_on) return fail(GAT_E_UNSUPPORTED, "gat_comm_option: GAT_COMM_GPL_BF16 and GAT_COMM_HALO exclude each other (the halo backward travels as fp32)");
        c->comm_gpl_bf16 = value != 0;
        return 0;
    }
    if (option == GAT_COMM_HALO) {                       // collective: every rank of the transport makes the same call
        if (value < 0 || value > 2) return fail(GAT_E_INVALID, "gat_comm_option: GAT_COMM_HALO takes 0 (off), 1 (on) or 2 (on where it pays)");
        c->halo_on = false;
        if (value == 0) { halo_free(&c->halo); return 0; }
        if (!c->comm) return fail(GAT_E_STATE, "gat_comm_option: GAT_COMM_HALO needs a transport (gat_comm_init_*) first");
        if (c->comm_gpl_bf16) return fail(GAT_E_UNSUPPORTED, "gat_comm_option: GAT_COMM_GPL_BF16 and GAT_COMM_HALO exclude each other");
        if (c->HDmax % 2 != 0 && bf16(c)) return fail(GAT_E_UNSUPPORTED, "gat_comm_option: GAT_COMM_HALO with bf16 rows needs an even H*D");
        GAT_TRY(halo_build(c->comm.get(), c->col_idx, c->n_edges, c->n_table, c->HDmax, &c->halo, c->stream));
        // value 2: only where fewer than half of the rows would travel — a halo costs a pack and an unpack pass over the rows it moves
        // on each side (4 row transfers through HBM at ~5 TB/s against one over ~1 TB/s of xGMI links: break-even near 0.5)
        c->halo_on = value == 1 || c->halo.referenced_fraction < 0.5;
        return 0;
    }
    if (option == GAT_COMM_PIPELINE) {
        if (value < 1 || value > 64) return fail(GAT_E_INVALID, "gat_comm_option: GAT_COMM_PIPELINE takes 1..64 chunks");
        c->comm_chunks = value;
        return 0;
    }
    return fail(GAT_E_INVALID, "gat_comm_option: unknown option");
}
int gat_comm_halo_info(gat_ctx* c, int32_t* active, int64_t* rows_received, int64_t* rows_sent, double* referenced_fraction) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (active) *active = c->halo_on ? 1 : 0;
    if (rows_received) *rows_received = c->halo.n_need;
    if (rows_sent) *rows_sent = c->halo.n_send;
    if (referenced_fraction) *referenced_fraction = c->halo.world > 0 ? c->halo.referenced_fraction : 1.0;
    return 0;
}
int gat_zero_grad(gat_ctx* c) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    Scope t(c, GAT_K_MISC);
    GAT_HIP(hipMemsetAsync(c->grads, 0, (size_t)(c->nW + c->nA + c->nWo) * sizeof(float), c->stream));
    return 0;
}
int gat_clip(gat_ctx* c, float threshold) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    Scope t(c, GAT_K_MISC);
    GAT_TRY(launch_clip(c->grads, c->nW, threshold, c->clip_scratch, c->stream));
    GAT_TRY(launch_clip(c->grads + c->nW, c->nA, threshold, c->clip_scratch + 1, c->stream));
    return launch_clip(c->grads + c->nW + c->nA, c->nWo, threshold, c->clip_scratch + 2, c->stream);
}
int gat_step_sgd(gat_ctx* c, float lr) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    Scope t(c, GAT_K_MISC);
    return launch_sgd(c->params, c->grads, lr, c->nW + c->nA + c->nWo, c->stream);
}
int gat_step_adam(gat_ctx* c, float lr, float b1, float b2, float eps, int32_t t_) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    if (!(b1 > 0.f && b1 < 1.f && b2 > 0.f && b2 < 1.f))
        return fail(GAT_E_INVALID, "For Adam optimizer, beta1 and beta2 must be in (0,1).");
    const int64_t np = c->nW + c->nA + c->nWo;
    if (!c->adam_m) {
        GAT_TRY(dalloc(c, &c->adam_m, np));
        GAT_TRY(dalloc(c, &c->adam_v, np));
        GAT_HIP(hipMemsetAsync(c->adam_m, 0, np * sizeof(float), c->stream));
        GAT_HIP(hipMemsetAsync(c->adam_v, 0, np * sizeof(float), c->stream));
    }
    Scope t(c, GAT_K_MISC);
    return launch_adam(c->params, c->grads, c->adam_m, c->adam_v, lr, np, b1, b2, eps, t_, c->stream);
}

// ---- exchange tables ----------------------------------------------------------------------------------------------
int gat_table(gat_ctx* c, int which, int32_t l, void** d_ptr, int64_t* n_rows, int64_t* row_floats) {
    GAT_TRY(check_layer(c, l));
    if (!d_ptr) return fail(GAT_E_INVALID, "null argument");
    if (which == GAT_TABLE_PL) *d_ptr = c->layers[l].PL;
    else if (which == GAT_TABLE_GPL) *d_ptr = c->gPL;                 // (the phase API always uses the first pair)
    else return fail(GAT_E_INVALID, "unknown table");
    if (n_rows) *n_rows = c->n_table;
    if (row_floats) *row_floats = which == GAT_TABLE_PL ? c->layers[l].HD * st_bytes(c) / 4 : c->layers[l].HD;
    return 0;
}
int gat_bind_table(gat_ctx* c, int which, int32_t l, void* d_ptr, int64_t bytes) {
    if (!c || !d_ptr) return fail(GAT_E_INVALID, "null argument");
    if (l < 0 || l >= c->cfg.num_layers) return fail(GAT_E_INVALID, "layer index out of range");
    if (!c->have_graph) return fail(GAT_E_STATE, "gat_bind_table: set the graph first");
    graph_drop(c);                                       // captured addresses would be stale
    if (which == GAT_TABLE_PL) {
        Layer& y = c->layers[l];
        if (bytes < (int64_t)c->n_table * y.HD * st_bytes(c)) return fail(GAT_E_INVALID, "bound PL table too small");
        if (!y.PL_bound) dfree(c, y.PL);
        y.PL = (float*)d_ptr; y.PL_bound = true;
        return 0;
    }
    if (which == GAT_TABLE_GPL) {
        if (bytes < (int64_t)c->n_table * c->HDmax * (int64_t)sizeof(float)) return fail(GAT_E_INVALID, "bound gPL table too small (needs n_table*max(H*D) floats)");
        if (!c->gPL_bound) dfree(c, c->gPL);
        c->gPL = (float*)d_ptr; c->gPL_bound = true;
        return 0;
    }
    return fail(GAT_E_INVALID, "unknown table");
}

// ---- taps ----------------------------------------------------------------------------------------------------------
static int d2h(gat_ctx* c, void* host, const void* dev, size_t bytes) {
    GAT_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    GAT_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
int gat_tap(gat_ctx* c, int tensor, int32_t l, void* host, int64_t count) {
    GAT_TRY(check_layer(c, l));
    if (!host) return fail(GAT_E_INVALID, "null destination");
    const Layer& y = c->layers[l];
    const int64_t N = c->n_rows, E = c->n_edges;
    const bool last = (l == c->cfg.num_layers - 1);
    auto need = [&](int64_t n) { return n == count ? 0 : fail(GAT_E_INVALID, "gat_tap: count mismatch"); };
    auto transposed = [&](const float* src, int64_t A_, int32_t B_, bool eh) -> int {
        if (!src) return fail(GAT_E_STATE, "gat_tap: tensor not kept (create the context with keep_taps=1)");
        float* tmp = nullptr;
        GAT_HIP(hipMalloc((void**)&tmp, (size_t)std::max<int64_t>(A_ * B_, 1) * sizeof(float)));
        int rc = eh ? launch_transpose_eh_to_he(src, tmp, A_, B_, c->stream) : launch_transpose_nh_to_hn(src, tmp, A_, B_, c->stream);
        if (rc == 0) rc = d2h(c, host, tmp, (size_t)A_ * B_ * sizeof(float));
        (void)hipFree(tmp);
        return rc;
    };
    switch (tensor) {
        case GAT_TAP_SRC: GAT_TRY(need(E)); return d2h(c, host, c->col_idx, E * sizeof(int32_t));
        case GAT_TAP_DST: {
            GAT_TRY(need(E));
            int32_t *s = nullptr, *d = nullptr;
            GAT_HIP(hipMalloc((void**)&s, std::max<int64_t>(E, 1) * sizeof(int32_t)));
            GAT_HIP(hipMalloc((void**)&d, std::max<int64_t>(E, 1) * sizeof(int32_t)));
            int rc = launch_csr_to_coo(c->row_ptr, c->col_idx, s, d, N, E, 0, c->stream);
            if (rc == 0) rc = d2h(c, host, d, E * sizeof(int32_t));
            (void)hipFree(s); (void)hipFree(d);
            return rc;
        }
        case GAT_TAP_ALPHA: GAT_TRY(need(E * y.H)); return transposed(y.alpha, E, y.H, true);
        case GAT_TAP_MAX: {
            GAT_TRY(need(N * y.H));
            GAT_TRY(transposed(y.mstat, N, y.H, false));
            if (edge_fast_path(y.H, y.D, c->n_table)) {          // kept in the log2 domain on the device
                float* f = static_cast<float*>(host);
                for (int64_t i = 0; i < N * y.H; ++i) f[i] = std::max(f[i] * 0.6931471805599453f, -1e9f);
            }
            return 0;
        }
        case GAT_TAP_GE: GAT_TRY(need(E * y.H)); return transposed(y.ge, E, y.H, true);
        case GAT_TAP_SCORE: GAT_TRY(need(E * y.H)); return transposed(y.score, E, y.H, true);
        case GAT_TAP_GALPHA: GAT_TRY(need(E * y.H)); return transposed(y.galpha, E, y.H, true);
        case GAT_TAP_GX:                              // what launch_grad_x of layer l left in layer l-1's g buffer
            if (l < 1) return fail(GAT_E_INVALID, "gat_tap: GAT_TAP_GX exists for layers >= 1 (E:1528 skips layer 0)");
            GAT_TRY(need(N * y.F));
            return d2h(c, host, c->layers[l - 1].g, N * y.F * sizeof(float));
        case GAT_TAP_SUM: GAT_TRY(need(N * y.H)); return transposed(y.zstat, N, y.H, false);
        case GAT_TAP_HPRE: GAT_TRY(need(N * y.HD)); return d2h(c, host, y.hpre, N * y.HD * sizeof(float));
        case GAT_TAP_HOUT: { const int64_t n = N * (last ? y.D : y.HD); GAT_TRY(need(n)); return d2h(c, host, y.hout, n * sizeof(float)); }
        case GAT_TAP_Y:
            GAT_TRY(need(N * c->cfg.num_classes));
            if (!c->y_valid) GAT_TRY(gat_head_forward(c, nullptr, nullptr));     // after a fused head step: y = f(H_L, Wo), both still there
            return d2h(c, host, c->y, N * c->cfg.num_classes * sizeof(float));
        case GAT_TAP_G: {
            GAT_TRY(need(N * y.HD));
            if (last && c->gH) {                        // formed on the fly by the kernels: same expression, same order
                std::vector<float> hp((size_t)(N * y.HD)), gh((size_t)(N * c->gh_stride));
                GAT_TRY(d2h(c, hp.data(), y.hpre, hp.size() * sizeof(float)));
                GAT_TRY(d2h(c, gh.data(), c->gH, gh.size() * sizeof(float)));
                float* out = static_cast<float*>(host);
                const float inv_heads = 1.0f / (float)y.H;
                for (int64_t i = 0; i < N * y.HD; ++i)
                    out[i] = gh[(size_t)((i / y.HD) * c->gh_stride + i % y.D)] * (hp[(size_t)i] > 0.f ? 1.0f : c->cfg.negative_slope) * inv_heads;
                return 0;
            }
            GAT_TRY(d2h(c, host, y.g, N * y.HD * sizeof(float)));
            if (l < c->cfg.num_layers - 1) {         // stored without the LReLU'(h_pre) factor (EdgeBwdArgs::g_raw)
                std::vector<float> hp((size_t)(N * y.HD));
                GAT_TRY(d2h(c, hp.data(), y.hpre, hp.size() * sizeof(float)));
                float* gh = static_cast<float*>(host);
                for (size_t i = 0; i < hp.size(); ++i) gh[i] *= hp[i] > 0.f ? 1.0f : c->cfg.negative_slope;
            }
            return 0;
        }
        case GAT_TAP_PL: {
            GAT_TRY(need(c->n_table * y.HD));
            if (!bf16(c)) return d2h(c, host, y.PL, c->n_table * y.HD * sizeof(float));
            std::vector<uint16_t> hb((size_t)(c->n_table * y.HD));          // bf16 rows -> fp32 for the caller
            GAT_TRY(d2h(c, hb.data(), y.PL, hb.size() * sizeof(uint16_t)));
            for (size_t i = 0; i < hb.size(); ++i) {
                const uint32_t u = (uint32_t)hb[i] << 16;
                memcpy(static_cast<float*>(host) + i, &u, sizeof(float));
            }
            return 0;
        }
        case GAT_TAP_PR: GAT_TRY(need(N * y.HD)); return d2h(c, host, y.PR, N * y.HD * sizeof(float));
        default: return fail(GAT_E_INVALID, "gat_tap: unknown tensor id");
    }
}

// ---- op-level entry points ------------------------------------------------------------------------------------------
int gat_op_csr_to_coo(const int32_t* d_row_ptr, const int32_t* d_col_idx, int32_t* d_src, int32_t* d_dst,
                      int64_t n_rows, int64_t n_edges, void* stream) {
    if (!d_row_ptr || !d_src || !d_dst || (!d_col_idx && n_edges > 0)) return fail(GAT_E_INVALID, "null argument");
    return launch_csr_to_coo(d_row_ptr, d_col_idx, d_src, d_dst, n_rows, n_edges, 0, (hipStream_t)stream);
}

struct TmpBufs {
    std::vector<void*> p;
    ~TmpBufs() { for (void* q : p) (void)hipFree(q); }
    int get(float** out, int64_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (size_t)std::max<int64_t>(n, 1) * sizeof(float));
        if (e != hipSuccess) return fail(GAT_E_NOMEM, hipGetErrorString(e));
        p.push_back(q); *out = (float*)q;
        return 0;
    }
    // device work list for caller-provided CSR (op-level entry points have no context)
    WorkList w;
    int4* items = nullptr; int4* slot_info = nullptr; float* part_acc = nullptr; float* part_mz = nullptr;
    int worklist(const int32_t* d_row_ptr, int64_t n, int32_t hd, int32_t h, hipStream_t s) {
        std::vector<int32_t> rp(n + 1);
        GAT_HIP(hipMemcpyAsync(rp.data(), d_row_ptr, (n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GAT_HIP(hipStreamSynchronize(s));
        build_worklist(rp.data(), n, w);
        float *fi, *fs;
        GAT_TRY(get(&fi, std::max<int64_t>(w.n_items, 1) * 4));
        GAT_TRY(get(&fs, std::max<int64_t>(w.n_slots + w.n_split, 1) * 4));
        GAT_TRY(get(&part_acc, (int64_t)std::max<int32_t>(w.n_slots, 1) * hd));
        GAT_TRY(get(&part_mz, (int64_t)std::max<int32_t>(w.n_slots, 1) * 2 * h));
        items = (int4*)fi; slot_info = (int4*)fs;
        if (w.n_items) GAT_HIP(hipMemcpyAsync(items, w.items.data(), w.items.size() * 4, hipMemcpyHostToDevice, s));
        if (w.n_slots) GAT_HIP(hipMemcpyAsync(slot_info, w.slot_info.data(), w.slot_info.size() * 4, hipMemcpyHostToDevice, s));
        GAT_HIP(hipStreamSynchronize(s));
        return 0;
    }
};

int gat_op_layer_forward(const int32_t* d_row_ptr, const int32_t* d_col_idx, const float* d_x, const float* d_w,
                         const float* d_a, float* d_attn_coeff, float* d_hpre, float* d_hout, int64_t n, int64_t e,
                         int32_t f, int32_t h, int32_t d, int32_t is_last, float slope, void* stream) {
    if (!d_row_ptr || !d_x || !d_w || !d_a || !d_attn_coeff || !d_hpre || !d_hout) return fail(GAT_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int HD = h * d;
    TmpBufs t;
    float *PL, *PR, *alpha, *ms, *zs;
    GAT_TRY(t.get(&PL, n * HD)); GAT_TRY(t.get(&PR, n * HD)); GAT_TRY(t.get(&alpha, e * h));
    GAT_TRY(t.get(&ms, n * h)); GAT_TRY(t.get(&zs, n * h));
    GAT_TRY(launch_project(d_x, d_w, PL, PR, n, f, HD, kPartBoth, false, nullptr, 0, s));
    EdgeFwdArgs a{};
    a.row_ptr = d_row_ptr; a.col_idx = d_col_idx; a.PL = PL; a.PR = PR; a.a = d_a; a.alpha = alpha;
    a.mstat = ms; a.zstat = zs;
    a.hpre = d_hpre; a.hout = d_hout; a.n_rows = n; a.n_table = n; a.H = h; a.D = d; a.is_last = is_last; a.slope = slope;
    GAT_TRY(t.worklist(d_row_ptr, n, HD, h, s));
    a.items = t.items; a.n_items = t.w.n_items; a.slot_info = t.slot_info; a.n_slots = t.w.n_slots; a.n_split = t.w.n_split;
    a.part_acc = t.part_acc; a.part_mz = t.part_mz;
    GAT_TRY(launch_edge_forward(a, s));
    GAT_TRY(launch_transpose_eh_to_he(alpha, d_attn_coeff, e, h, s));
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

int gat_op_layer_backward(const int32_t* d_row_ptr, const int32_t* d_col_idx, const float* d_x, const float* d_w,
                          const float* d_a, const float* d_attn_coeff, const float* d_hpre, const float* d_g,
                          float* d_grad_w, float* d_grad_a, const float* d_hpre_prev, float* d_g_prev, int64_t n,
                          int64_t e, int32_t f, int32_t h, int32_t d, float slope, void* stream) {
    if (!d_row_ptr || !d_x || !d_w || !d_a || !d_attn_coeff || !d_hpre || !d_g || !d_grad_w || !d_grad_a)
        return fail(GAT_E_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int HD = h * d;
    TmpBufs t;
    if (!d_col_idx) {                  // e == 0: the kernels still clamp their index prefetch to edge 0 — give them one to read
        float* dummy = nullptr;
        GAT_TRY(t.get(&dummy, 1));
        GAT_HIP(hipMemsetAsync(dummy, 0, sizeof(float), s));
        d_col_idx = reinterpret_cast<const int32_t*>(dummy);
    }
    float *PL, *PR, *alpha, *gPL, *gPR, *gap, *scr;
    const int blocks = 2048;      // capacity; the grid actually used is computed below
    GAT_TRY(t.get(&PL, n * HD)); GAT_TRY(t.get(&PR, n * HD)); GAT_TRY(t.get(&alpha, e * h));
    GAT_TRY(t.get(&gPL, n * HD)); GAT_TRY(t.get(&gPR, n * HD)); GAT_TRY(t.get(&gap, (int64_t)blocks * HD));
    GAT_TRY(t.get(&scr, grad_w_scratch_floats(n, f, HD)));
    GAT_TRY(launch_project(d_x, d_w, PL, PR, n, f, HD, kPartBoth, false, nullptr, 0, s));
    GAT_TRY(launch_transpose_he_to_eh(d_attn_coeff, alpha, e, h, s));
    GAT_TRY(t.worklist(d_row_ptr, n, HD, h, s));
    // softmax stats of this layer (the fast-path backward recomputes alpha from them): one
    // forward edge pass into scratch outputs
    float *ms, *zs, *hp_tmp, *ho_tmp;
    GAT_TRY(t.get(&ms, n * h)); GAT_TRY(t.get(&zs, n * h)); GAT_TRY(t.get(&hp_tmp, n * HD)); GAT_TRY(t.get(&ho_tmp, n * HD));
    {
        EdgeFwdArgs fa{};
        fa.row_ptr = d_row_ptr; fa.col_idx = d_col_idx; fa.PL = PL; fa.PR = PR; fa.a = d_a;
        fa.alpha = edge_fast_path(h, d, n) ? nullptr : alpha; fa.hpre = hp_tmp; fa.hout = ho_tmp; fa.mstat = ms; fa.zstat = zs;
        fa.n_rows = n; fa.n_table = n; fa.H = h; fa.D = d; fa.is_last = 0; fa.slope = slope;
        fa.items = t.items; fa.n_items = t.w.n_items; fa.slot_info = t.slot_info; fa.n_slots = t.w.n_slots; fa.n_split = t.w.n_split;
        fa.part_acc = t.part_acc; fa.part_mz = t.part_mz;
        GAT_TRY(launch_edge_forward(fa, s));
    }
    GAT_HIP(hipMemsetAsync(gPL, 0, (size_t)n * HD * sizeof(float), s));
    EdgeBwdArgs a{};
    a.mstat = ms; a.zstat = zs;
    a.row_ptr = d_row_ptr; a.col_idx = d_col_idx; a.PL = PL; a.PR = PR; a.a = d_a; a.alpha = alpha;
    a.hpre = d_hpre; a.g = d_g; a.gPL = gPL; a.gPR = gPR; a.ge = nullptr; a.ga_partial = gap;
    a.ga_blocks = edge_fast_path(h, d, n) ? edge_backward_blocks(t.w.n_items, h, d, false, false, false)
                                       : edge_backward_blocks(n * 4, h, d, false, false, false);
    a.n_rows = n; a.n_table = n; a.H = h; a.D = d; a.slope = slope;
    a.items = t.items; a.n_items = t.w.n_items; a.slot_info = t.slot_info; a.n_slots = t.w.n_slots; a.n_split = t.w.n_split;
    a.part_acc = t.part_acc;
    GAT_TRY(launch_edge_backward(a, s));
    GAT_TRY(launch_reduce_partials_add(gap, a.ga_blocks, HD, d_grad_a, s));
    GAT_TRY(launch_grad_w(gPL, gPR, d_x, d_grad_w, scr, n, f, HD, kPartBoth, s));
    if (d_hpre_prev && d_g_prev) GAT_TRY(launch_grad_x(gPL, gPR, d_w, d_hpre_prev, d_g_prev, n, f, HD, slope, s));
    GAT_HIP(hipStreamSynchronize(s));
    return 0;
}

// ---- measurement -------------------------------------------------------------------------------------------------------
static const char* kNames[GAT_K_COUNT] = {"project_gemm", "edge_forward", "head_forward", "head_backward",
                                          "edge_backward", "gpl_sum", "grad_w_gemm", "grad_x_gemm", "misc",
                                          "exchange", "edge_last_fused"};
const char* gat_kernel_name(int k) { return (k >= 0 && k < GAT_K_COUNT) ? kNames[k] : "?"; }
int gat_kernel_stats(gat_ctx* c, int k, int64_t* launches, double* total_ms) {
    if (!c || k < 0 || k >= GAT_K_COUNT) return fail(GAT_E_INVALID, "bad argument");
    GAT_TRY(flush_events(c));
    if (launches) *launches = c->k_launches[k];
    if (total_ms) *total_ms = c->k_ms[k];
    return 0;
}
int gat_kernel_stats_reset(gat_ctx* c) {
    if (!c) return fail(GAT_E_INVALID, "null context");
    GAT_TRY(flush_events(c));
    for (int k = 0; k < GAT_K_COUNT; ++k) { c->k_launches[k] = 0; c->k_ms[k] = 0.0; }
    return 0;
}

// SURVEY §8(d), literally: every tensor the restructured algorithm must move once per step, with
// b = storage bytes (4 fp32 / 2 bf16) on every float term — the single figure `roofline.achieved` is quoted
// against.  It is a property of the WORKLOAD, not of this implementation: it counts the alpha write/read
// (E*H per direction) although the training path recomputes alpha from the softmax stats instead, it counts
// the gPL scatter once although the no-atomics scatter stores and re-reads a per-edge record, and in bf16 mode
// it prices PR / h_pre / g at 2 bytes although they are kept in fp32 here.  Pure host arithmetic (no device).
int gat_algorithmic_bytes_shape(const gat_config* cfg, int64_t n_rows, int64_t n_edges, int64_t n_table,
                                int32_t replicated_input, double* bytes_step, double* per_kernel) {
    if (!cfg || !cfg->heads || !cfg->outdims || cfg->num_layers <= 0) return fail(GAT_E_INVALID, "gat_algorithmic_bytes_shape: bad config");
    double k[GAT_K_COUNT] = {0};
    const double N = (double)n_rows, E = (double)n_edges;
    const double b = cfg->storage_dtype == GAT_DTYPE_BF16 ? 2.0 : 4.0;
    const int L = cfg->num_layers;
    for (int l = 0; l < L; ++l) {
        const double H = cfg->heads[l], D = cfg->outdims[l], HD = H * D;
        const double F = l == 0 ? (double)cfg->in_dim : (double)cfg->heads[l - 1] * cfg->outdims[l - 1];
        const double Dout = (l == L - 1) ? D : HD;
        const double NL = (l == 0 && replicated_input) ? (double)n_table : N;     // rows of the W_left half (shards)
        k[GAT_K_PROJECT] += b * (std::max(N, NL) * F + 2 * HD * F + (N + NL) * HD);          // X, W read; PL + PR written
        k[GAT_K_EDGE_FWD] += 4 * (N + 1) + 4 * E + b * (E * HD + N * HD + E * H + N * HD + N * Dout);   // PL[src], PR, alpha, h_pre, H
        // g, PL[src], PR, alpha, gPR in the destination-major pass; the gPL scatter (counted once, E*HD) belongs to the
        // source-major pass that performs it here (it gathers g[dst] per edge and sums per source)
        k[GAT_K_EDGE_BWD] += 4 * (N + 1) + 4 * E + b * (N * HD + E * HD + N * HD + E * H + N * HD);
        k[GAT_K_GPL_SUM] += b * E * HD;
        k[GAT_K_GRAD_W] += b * ((N + NL) * HD + std::max(N, NL) * F + 2 * HD * F);            // gPL + gPR re-read, X, grad_W
        if (l > 0) k[GAT_K_GRAD_X] += b * (2 * N * HD + 2 * N * F);                            // gPL + gPR, gX write, h_pre_{l-1}
    }
    const double head = 2.0 * 4.0 * N * ((double)cfg->outdims[L - 1] + 2.0 * cfg->num_classes + 2.0);
    k[GAT_K_HEAD_FWD] = head / 2; k[GAT_K_HEAD_BWD] = head / 2;
    double tot = 0;
    for (int i = 0; i < GAT_K_COUNT; ++i) { tot += k[i]; if (per_kernel) per_kernel[i] = k[i]; }
    if (bytes_step) *bytes_step = tot;
    return 0;
}
// The same model with every RANDOM row access priced at what the memory system can serve: the fabric moves 128-byte requests
// (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ in every PMC pass of this build), so a gathered or scattered row of b*H*D bytes costs
// ceil(b*H*D / 128) * 128.  Identical to SURVEY 8d's figure wherever a row is a multiple of 128 B (fp32 at H*D = 64: 256 B);
// for bf16 rows at H*D = 32 (BASELINE config 5: 64-byte rows) the three per-edge row terms double.  Sequential streams
// (indices, attention coefficients, node-major rows) are unchanged.
int gat_request_bytes_shape(const gat_config* cfg, int64_t n_rows, int64_t n_edges, int64_t n_table,
                            int32_t replicated_input, double* bytes_step, double* per_kernel) {
    double k[GAT_K_COUNT] = {0};
    GAT_TRY(gat_algorithmic_bytes_shape(cfg, n_rows, n_edges, n_table, replicated_input, nullptr, k));
    const double E = (double)n_edges, b = cfg->storage_dtype == GAT_DTYPE_BF16 ? 2.0 : 4.0;
    for (int l = 0; l < cfg->num_layers; ++l) {
        const double row = b * (double)cfg->heads[l] * cfg->outdims[l];
        const double extra = E * (std::ceil(row / 128.0) * 128.0 - row);
        k[GAT_K_EDGE_FWD] += extra; k[GAT_K_EDGE_BWD] += extra; k[GAT_K_GPL_SUM] += extra;      // PL[src] twice, the gPL scatter once
    }
    double tot = 0;
    for (int i = 0; i < GAT_K_COUNT; ++i) { tot += k[i]; if (per_kernel) per_kernel[i] = k[i]; }
    if (bytes_step) *bytes_step = tot;
    return 0;
}
int gat_algorithmic_bytes(gat_ctx* c, double* bytes_step, double* per_kernel) {
    if (!c || !c->have_graph) return fail(GAT_E_STATE, "graph not set");
    GAT_TRY(gat_algorithmic_bytes_shape(&c->cfg, c->n_rows, c->n_edges, c->n_table, c->Xtab != nullptr, bytes_step, per_kernel));
    if (per_kernel && c->buffers_ready && fused_last(c)) {
        // gat_step runs the last layer's forward and backward edge passes (and forms gH) in ONE kernel class: its share of
        // the SAME byte model moves there (the step total is unchanged; the head class keeps its bytes: head_step still runs)
        const int L = c->cfg.num_layers;
        const double N = (double)c->n_rows, E = (double)c->n_edges, H = c->layers[L - 1].H, D = c->layers[L - 1].D, HD = H * D;
        const double fwd = 4 * (N + 1) + 4 * E + 4.0 * (E * HD + N * HD + E * H + N * HD + N * D);
        const double bwd = 4 * (N + 1) + 4 * E + 4.0 * (N * HD + E * HD + N * HD + E * H + N * HD);
        per_kernel[GAT_K_EDGE_FWD] -= fwd; per_kernel[GAT_K_EDGE_BWD] -= bwd; per_kernel[GAT_K_EDGE_FUSED] += fwd + bwd;
    }
    return 0;
}

}  // extern "C"
