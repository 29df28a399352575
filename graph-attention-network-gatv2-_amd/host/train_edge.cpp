// train_edge — drop-in for the reference's edge-based GATv2 trainer (GATv2_edge_based.cu `main`,
// cited E:<line>): same command-line flags and defaults, same four-text-file CSR dataset, same
// stdout/stderr lines, but every device operation goes through the C ABI of include/gatv2_abi.h
// (libgatv2_hip.so: hand-written HIP for MI355X).  This file is plain host C++: no HIP headers.
//
//   ./train_edge --dataset cora --data-root /path/to/datasets --num-layers 2 --heads 8,8 ...
//                --outdims 8,8 --epochs 20 --optimizer adam --lr 0.01 --clip
//
// Additive flags (not in the reference): --seed N (parameter init; default time(NULL) like
// E:1305), --load-params FILE / --dump-params FILE (raw fp32: W | a | Wo in the reference
// layouts), --device N, --cache (binary cache of the parsed text files, written next to them),
// --ranks P [--transport rccl|host]: destination-range sharding over P GPUs of the node, one forked
// process per GPU (devices --device .. --device+P-1), exchanges inside the library (RCCL over xGMI;
// "host" stages them through shared memory and lets ranks share a GPU — for testing).  Every rank
// reads the dataset, keeps its destination range (host/shard_plan.h) and the replicated input
// features; rank 0 prints.  Same numbers as one GPU up to fp32 summation order.
// --halo 0|1|2 (with --ranks): the table exchanges move only the rows the receiving shard's edges reference (gat_comm_option
// GAT_COMM_HALO; 2 = only where fewer than half of the rows would travel).  Same numbers as the full exchange.
// --dtype f32|bf16: storage type of the gathered / exchanged source table and the per-edge message
// rows (arithmetic stays fp32).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "gatv2_abi.h"
#include "shard_plan.h"

namespace {

struct Options {
    int epochs = 200;                 // E:935
    int layers = 2;                   // E:936
    bool clip = false;                // E:937
    std::string optimizer = "sgd";    // E:938
    float lr = 0.0001f, beta1 = 0.9f, beta2 = 0.999f;   // E:939
    bool beta1_given = false, beta2_given = false;
    std::vector<int32_t> heads, outdims;
    bool heads_given = false, outdims_given = false;
    std::string dataset = "pubmed";   // E:1050
    std::string data_root = "./data"; // E:1051
    // additive
    uint64_t seed = 0; bool seed_given = false;
    std::string load_params, dump_params;
    int device = 0;
    bool cache = false;
    int ranks = 1;
    std::string transport = "rccl";
    int halo = 0;
    std::string dtype = "f32";
    std::string train_mask, val_mask;     // text files of N 0/1 values (beyond the reference: README R:134 "later")
};

struct RankEnv {                      // one forked process per GPU
    int world = 1, rank = 0;
    struct Shared { volatile int id_ready; char id[GAT_COMM_ID_BYTES]; }* shared = nullptr;
    std::string shm_name;
};

[[noreturn]] void die(const std::string& msg) {
    std::cerr << msg;
    std::exit(1);
}

bool split_ints(const std::string& csv, int count, std::vector<int32_t>& out) {
    out.clear();
    std::stringstream ss(csv);
    std::string item;
    for (int i = 0; i < count; ++i) {
        if (!std::getline(ss, item, ',')) return false;
        out.push_back(std::stoi(item));
    }
    return true;
}

// The reference scans argv three times (E:943-953, 958-1010, 1053-1061): --num-layers is looked
// up first so that --heads/--outdims can be sized; unknown flags are ignored.
Options parse_args(int argc, char** argv) {
    Options o;
    for (int i = 1; i + 1 < argc; ++i)
        if (std::string(argv[i]) == "--num-layers") {
            o.layers = std::stoi(argv[i + 1]);
            if (o.layers <= 0) die("Error: Number of layers must be > 0\n");
            break;
        }
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const bool has_val = i + 1 < argc;
        if (a == "--epochs" && has_val) o.epochs = std::stoi(argv[++i]);
        else if (a == "--heads" && has_val) {
            if (!split_ints(argv[++i], o.layers, o.heads))
                die("Error: --heads must have " + std::to_string(o.layers) + " values.\n");
            o.heads_given = true;
        } else if (a == "--outdims" && has_val) {
            if (!split_ints(argv[++i], o.layers, o.outdims))
                die("Error: --ooutdims must have " + std::to_string(o.layers) + " values.\n");   // sic, E:982
            o.outdims_given = true;
        } else if (a == "--clip") o.clip = true;
        else if (a == "--optimizer" && has_val) {
            o.optimizer = argv[++i];
            if (o.optimizer != "sgd" && o.optimizer != "adam") die("Invalid optimizer choice. Use 'sgd' or 'adam'\n");
        } else if (a == "--beta1" && has_val) { o.beta1 = std::stof(argv[++i]); o.beta1_given = true; }
        else if (a == "--beta2" && has_val) { o.beta2 = std::stof(argv[++i]); o.beta2_given = true; }
        else if (a == "--lr" && has_val) o.lr = std::stof(argv[++i]);
        else if (a == "--dataset" && has_val) o.dataset = argv[++i];
        else if (a == "--data-root" && has_val) o.data_root = argv[++i];
        else if (a == "--seed" && has_val) { o.seed = std::strtoull(argv[++i], nullptr, 0); o.seed_given = true; }
        else if (a == "--train-mask" && has_val) o.train_mask = argv[++i];
        else if (a == "--val-mask" && has_val) o.val_mask = argv[++i];
        else if (a == "--load-params" && has_val) o.load_params = argv[++i];
        else if (a == "--dump-params" && has_val) o.dump_params = argv[++i];
        else if (a == "--device" && has_val) o.device = std::stoi(argv[++i]);
        else if (a == "--cache") o.cache = true;
        else if (a == "--ranks" && has_val) {
            o.ranks = std::stoi(argv[++i]);
            if (o.ranks < 1) die("Error: --ranks must be >= 1\n");
        } else if (a == "--dtype" && has_val) {
            o.dtype = argv[++i];
            if (o.dtype != "f32" && o.dtype != "bf16") die("Invalid dtype choice. Use 'f32' or 'bf16'\n");
        } else if (a == "--halo" && has_val) {
            o.halo = std::stoi(argv[++i]);
            if (o.halo < 0 || o.halo > 2) die("Invalid halo choice. Use 0, 1 or 2\n");
        } else if (a == "--transport" && has_val) {
            o.transport = argv[++i];
            if (o.transport != "rccl" && o.transport != "host") die("Invalid transport choice. Use 'rccl' or 'host'\n");
        }
    }
    if (o.optimizer == "adam") {
        if (o.beta1 <= 0.0f || o.beta1 >= 1.0f || o.beta2 <= 0.0f || o.beta2 >= 1.0f)
            die("Error: For Adam optimizer, beta1 and beta2 must be in (0,1).\n");
    } else if (o.beta1_given || o.beta2_given) {
        std::cerr << "Warning: beta1/beta2 specified but ignored for SGD optimizer.\n";
    }
    // The reference leaves head[]/out_dim[] uninitialised without these flags.
    if (!o.heads_given) die("Error: --heads must have " + std::to_string(o.layers) + " values.\n");
    if (!o.outdims_given) die("Error: --ooutdims must have " + std::to_string(o.layers) + " values.\n");
    const char* env_root = std::getenv("DATA_ROOT");             // E:1064-1067
    if (env_root && o.data_root == "./data") o.data_root = env_root;
    if (!o.data_root.empty() && o.data_root.back() != '/' && o.data_root.back() != '\\') o.data_root += '/';
    return o;
}

// ---- dataset text files (R:20-27; E:24-64) ----------------------------------------------------
// Whole-file read + strtof/strtol scanning: the reference's iostream loaders are the start-up
// bottleneck on large graphs; the accepted syntax (whitespace separated numbers) is the same.
bool slurp(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { out.clear(); return false; }
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

void load_features(const std::string& path, std::vector<float>& x, int64_t& n, int& dim) {
    std::string buf;
    slurp(path, buf);
    n = 0; dim = 0;
    const char* p = buf.c_str();
    const char* end = p + buf.size();
    while (p < end) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', end - p));
        if (!eol) eol = end;
        int count = 0;
        const char* q = p;
        while (q < eol) {
            char* next = nullptr;
            const float v = std::strtof(q, &next);
            if (next == q || next > eol) break;
            x.push_back(v);
            ++count;
            q = next;
        }
        if (dim == 0) dim = count;
        else if (count != dim) {
            std::cerr << "Inconsistent input_dim on line " << n << std::endl;   // E:43
            std::exit(1);
        }
        ++n;
        p = eol + 1;
    }
}

void load_ints(const std::string& path, std::vector<int32_t>& v) {
    std::string buf;
    slurp(path, buf);
    const char* p = buf.c_str();
    for (;;) {
        char* next = nullptr;
        const long val = std::strtol(p, &next, 10);
        if (next == p) break;
        v.push_back((int32_t)val);
        p = next;
    }
}

// ---- binary cache: magic, N, F, E, then the four arrays as raw little-endian data ----------------
constexpr uint64_t kCacheMagic = 0x3143564154414755ull;   // "UGATAVC1"
bool newer(const std::string& a, const std::string& b) {
    struct stat sa{}, sb{};
    if (stat(a.c_str(), &sa) != 0 || stat(b.c_str(), &sb) != 0) return false;
    return sa.st_mtime >= sb.st_mtime;
}
bool load_cache(const std::string& file, const std::string& dir, std::vector<float>& x, int64_t& n, int& f,
                std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<int32_t>& lab) {
    for (const char* t : {"features.txt", "row_ptr.txt", "col_idx.txt", "labels.txt"})
        if (!newer(file, dir + t)) return false;
    std::ifstream in(file, std::ios::binary);
    uint64_t h[4] = {0, 0, 0, 0};
    if (!in.read(reinterpret_cast<char*>(h), sizeof(h)) || h[0] != kCacheMagic) return false;
    n = (int64_t)h[1]; f = (int)h[2];
    x.resize((size_t)n * f); rp.resize(n + 1); ci.resize(h[3]); lab.resize(n);
    in.read(reinterpret_cast<char*>(x.data()), x.size() * sizeof(float));
    in.read(reinterpret_cast<char*>(rp.data()), rp.size() * sizeof(int32_t));
    in.read(reinterpret_cast<char*>(ci.data()), ci.size() * sizeof(int32_t));
    in.read(reinterpret_cast<char*>(lab.data()), lab.size() * sizeof(int32_t));
    return (bool)in;
}
void save_cache(const std::string& file, const std::vector<float>& x, int64_t n, int f, const std::vector<int32_t>& rp,
                const std::vector<int32_t>& ci, const std::vector<int32_t>& lab) {
    std::ofstream out(file, std::ios::binary);
    const uint64_t h[4] = {kCacheMagic, (uint64_t)n, (uint64_t)f, (uint64_t)ci.size()};
    out.write(reinterpret_cast<const char*>(h), sizeof(h));
    out.write(reinterpret_cast<const char*>(x.data()), x.size() * sizeof(float));
    out.write(reinterpret_cast<const char*>(rp.data()), rp.size() * sizeof(int32_t));
    out.write(reinterpret_cast<const char*>(ci.data()), ci.size() * sizeof(int32_t));
    out.write(reinterpret_cast<const char*>(lab.data()), lab.size() * sizeof(int32_t));
}

void check(int rc, const char* what) {
    if (rc != 0) {
        std::fprintf(stderr, "Error launching %s: %s\n", what, gat_last_error());   // style of E:1191 …, but fatal
        std::exit(1);
    }
}

// E:929-933.  The reference prints this block BEFORE it looks at argv (E:943), so its argument-error exits
// still show it; main() therefore calls it ahead of parse_args.  With --ranks P > 1 the parent must not touch
// the HIP runtime before forking, so there rank 0 prints it at the start of run() instead.
size_t g_free_before = 0;
bool g_tracker_printed = false;
void print_memory_tracker_before() {
    size_t total_mem = 0;
    if (gat_mem_info(&g_free_before, &total_mem) != 0)     // like the reference: report and go on (E:1189-1192)
        std::fprintf(stderr, "Error launching gat_mem_info: %s\n", gat_last_error());
    std::printf("\n[Memory Tracker] Before allocation:\n");
    std::printf("  Total GPU memory: %.2f MB\n", total_mem / (1024.0 * 1024.0));
    std::printf("  Free GPU memory : %.2f MB\n", g_free_before / (1024.0 * 1024.0));
    g_tracker_printed = true;
}

int run(const Options& o, const RankEnv& env) {
    if (!g_tracker_printed) print_memory_tracker_before();
    const size_t free_before = g_free_before;
    size_t total_mem = 0;

    const int L = o.layers;
    std::cout << "Configuration:\n"
              << "  Number of layers: " << L << "\n"
              << "  Epochs: " << o.epochs << "\n"
              << "  Attention heads: [";
    for (int l = 0; l < L; ++l) std::cout << o.heads[l] << (l < L - 1 ? ", " : "");
    std::cout << "]\n  Output dimensions: [";
    for (int l = 0; l < L; ++l) std::cout << o.outdims[l] << (l < L - 1 ? ", " : "");
    std::cout << "]\n"
              << "  Gradient clipping: " << (o.clip ? "true" : "false") << "\n"
              << "  Optimizer: " << o.optimizer << "\n"
              << "  Learning rate: " << o.lr << "\n\n";

    const std::string path = o.data_root + o.dataset + "/";
    std::cout << "Using dataset: " << o.dataset << std::endl;
    std::cout << "Dataset path: " << path << std::endl;

    std::vector<float> x;
    int64_t N = 0; int F0 = 0;
    std::vector<int32_t> row_ptr, col_idx, labels;
    // Optional binary cache of the parsed text files (additive: --cache).  Parsing features.txt is the
    // start-up bottleneck at Products scale (~2 GB of text); the cache is rewritten whenever a text
    // file is newer than it.
    const std::string cache = path + "gatv2_cache.bin";
    if (!(o.cache && load_cache(cache, path, x, N, F0, row_ptr, col_idx, labels))) {
        load_features(path + "features.txt", x, N, F0);
        load_ints(path + "row_ptr.txt", row_ptr);
        if ((int64_t)row_ptr.size() != N + 1) { std::cerr << "Invalid row_ptr length\n"; return 1; }
        load_ints(path + "col_idx.txt", col_idx);
        load_ints(path + "labels.txt", labels);
        if ((int64_t)labels.size() != N) { std::cerr << "Invalid labels length\n"; return 1; }
        if (o.cache) save_cache(cache, x, N, F0, row_ptr, col_idx, labels);
    }
    const int64_t E = (int64_t)col_idx.size();

    int max_degree = 0;
    for (int64_t i = 0; i < N; ++i) max_degree = std::max(max_degree, row_ptr[i + 1] - row_ptr[i]);
    std::cout << "Max degree = " << max_degree << std::endl;
    const int C = *std::max_element(labels.begin(), labels.end()) + 1;          // E:1106-1107
    std::cout << "Number of classes = " << C << std::endl;
    std::cout << "Graph loaded: " << N << " nodes, " << E << " edges, "
              << "input_feature_vector_dim = " << F0 << std::endl;

    // ---- the reference's buffer-size report (E:1153-1350), same formulas ----
    std::vector<int64_t> in_dim(L);
    in_dim[0] = F0;
    for (int l = 1; l < L; ++l) in_dim[l] = (int64_t)o.heads[l - 1] * o.outdims[l - 1];
    const size_t fsz = sizeof(float);
    std::printf("\nsize of input features : %zu MB\n", (size_t)(N * F0 * fsz) / (1024 * 1024));
    std::printf("\nsize of graph data (CSR,COO, labels) : %zu KB\n",
                (size_t)(((N + 1) + E + N + 2 * E) * sizeof(int)) / 1024);
    size_t total_out = 0, total_heads = 0, total_w = 0, total_a = 0, total_ig = 0;
    for (int l = 0; l < L; ++l) {
        total_out += (l == L - 1) ? (size_t)N * o.outdims[l] : (size_t)N * o.heads[l] * o.outdims[l];
        total_heads += o.heads[l];
        total_w += (size_t)o.heads[l] * o.outdims[l] * 2 * in_dim[l];
        total_a += (size_t)o.heads[l] * o.outdims[l];
        total_ig += (size_t)N * o.heads[l] * o.outdims[l] * fsz;
    }
    const int max_heads = *std::max_element(o.heads.begin(), o.heads.end());
    std::printf("\nTotal size of layer outputs(pre & post activation): %zu MB\n", (2 * total_out * fsz) / (1024 * 1024));
    std::printf("Total size of attention scores & coeffs: %zu MB\n", (2 * total_heads * E * fsz) / (1024 * 1024));
    std::printf("\nTotal size of parameters and their gradients: %zu MB\n",
                ((total_w + total_a + (size_t)C * o.outdims[L - 1]) * 2 * fsz) / (1024 * 1024));
    std::printf("\nloss and accuracy storage: %zu MB\n", ((size_t)N * (sizeof(float) + sizeof(int))) / (1024 * 1024));
    std::printf("Total size of intermediate gradients: %zu MB\n",
                ((total_ig + (size_t)max_heads * E * 2) * fsz) / (1024 * 1024));     // E:1350 (its 4x quirk kept)

    // ---- device context ----
    gat_config cfg{};
    cfg.num_layers = L; cfg.heads = o.heads.data(); cfg.outdims = o.outdims.data();
    cfg.in_dim = F0; cfg.num_classes = C; cfg.negative_slope = 0.01f; cfg.device = o.device;
    cfg.storage_dtype = o.dtype == "bf16" ? GAT_DTYPE_BF16 : GAT_DTYPE_F32;
    int n_devices = 0;
    check(gat_device_count(&n_devices), "gat_device_count");
    if (env.world > 1) {
        if (o.device + env.world <= n_devices) cfg.device = o.device + env.rank;      // one GPU per rank
        else if (o.transport == "rccl") die("Error: --ranks " + std::to_string(env.world) + " needs that many GPUs from --device on "
                                            "(RCCL cannot share a GPU between ranks); found " + std::to_string(n_devices) + "\n");
    }
    gat_ctx* ctx = nullptr;
    check(gat_create(&cfg, &ctx), "gat_create");
    int64_t nW = 0, nA = 0, nWo = 0;
    gat_param_count(ctx, GAT_PARAM_W, &nW); gat_param_count(ctx, GAT_PARAM_A, &nA); gat_param_count(ctx, GAT_PARAM_WO, &nWo);
    if (env.world == 1) {
        check(gat_set_graph(ctx, row_ptr.data(), col_idx.data(), N, E, N, 0), "csr_to_coo_kernel");
        check(gat_set_features(ctx, x.data(), N, F0), "gat_set_features");
        check(gat_set_labels(ctx, labels.data(), N), "gat_set_labels");
    } else {
        // this rank's destination range; sources as rows of the padded [world][max_rows] table; the static
        // input features replicated for every table row (layer 0 then needs no exchange)
        if (env.world > N) die("Error: more ranks than graph rows\n");
        const gatshard::Plan plan = gatshard::make_plan(row_ptr.data(), N, env.world, env.rank);
        std::vector<int32_t> rp_l, ci_l;
        gatshard::local_csr(plan, row_ptr.data(), col_idx.data(), rp_l, ci_l);
        check(gat_set_graph(ctx, rp_l.data(), ci_l.data(), plan.n_rows(), (int64_t)ci_l.size(), plan.n_table(), plan.table_row0()),
              "csr_to_coo_kernel");
        {
            const std::vector<float> xt = gatshard::table_features(plan, x.data(), F0);
            check(gat_set_source_features(ctx, xt.data(), plan.n_table(), F0), "gat_set_source_features");
        }
        check(gat_set_labels(ctx, labels.data() + plan.row0(), plan.n_rows()), "gat_set_labels");
        if (o.transport == "rccl") {
            if (env.rank == 0) {
                check(gat_comm_unique_id(env.shared->id), "gat_comm_unique_id");
                __sync_synchronize();
                env.shared->id_ready = 1;
            } else {
                while (!env.shared->id_ready) usleep(1000);
                __sync_synchronize();
            }
            check(gat_comm_init_rccl(ctx, env.world, env.rank, env.shared->id), "gat_comm_init_rccl");
        } else {
            int64_t hd_max = 0;
            for (int l = 0; l < L; ++l) hd_max = std::max<int64_t>(hd_max, (int64_t)o.heads[l] * o.outdims[l]);
            const int64_t bytes = std::max<int64_t>(plan.n_table() * hd_max + 64, nW + nA + nWo + 3) * (int64_t)sizeof(float);   // (+ 64: block offsets of a halo exchange)
            check(gat_comm_init_host(ctx, env.world, env.rank, env.shm_name.c_str(), bytes), "gat_comm_init_host");
        }
        if (o.halo != 0) check(gat_comm_option(ctx, GAT_COMM_HALO, o.halo), "gat_comm_option(GAT_COMM_HALO)");      // collective: every rank
    }
    // the 4 GiB cliff of gatv2_abi.h "Limits" is never silent: ask every layer which kernels it runs on
    if (env.rank == 0) {
        std::string slow;
        for (int l = 0; l < L; ++l) {
            int32_t path = GAT_PATH_FAST;
            if (gat_layer_path(ctx, l, &path) == 0 && path == GAT_PATH_GENERIC_SIZE) slow += (slow.empty() ? "" : ", ") + std::to_string(l);
        }
        if (!slow.empty())
            fprintf(stderr, "Warning: layer(s) %s: the gathered source table is >= 4 GiB; these layers run on the generic float-atomic kernels "
                            "(same results, several times slower) — use --ranks to shard by destination range\n", slow.c_str());
    }
    // optional splits: loss / accuracy / gradient over the training nodes, an extra validation line per epoch
    std::vector<uint8_t> train_m, val_m;
    int64_t n_train = N;
    auto load_mask = [&](const std::string& file, std::vector<uint8_t>& m, const char* what) {
        std::vector<int32_t> v;
        load_ints(file, v);
        if ((int64_t)v.size() != N) die(std::string("Invalid ") + what + " length\n");
        m.resize((size_t)N);
        for (int64_t i = 0; i < N; ++i) m[(size_t)i] = v[(size_t)i] != 0;
    };
    if (!o.train_mask.empty()) {
        load_mask(o.train_mask, train_m, "train mask");
        n_train = 0;
        for (uint8_t b : train_m) n_train += b;
        if (n_train == 0) die("Error: --train-mask selects no node\n");
        const int64_t r0 = env.world == 1 ? 0 : gatshard::make_plan(row_ptr.data(), N, env.world, env.rank).row0();
        const int64_t nr = env.world == 1 ? N : gatshard::make_plan(row_ptr.data(), N, env.world, env.rank).n_rows();
        check(gat_set_train_mask(ctx, train_m.data() + r0, nr), "gat_set_train_mask");
        std::printf("Training nodes: %lld of %lld\n", (long long)n_train, (long long)N);
    }
    if (!o.val_mask.empty()) {
        if (env.world != 1) die("Error: --val-mask needs --ranks 1\n");
        load_mask(o.val_mask, val_m, "val mask");
    }
    check(gat_params_init(ctx, o.seed), "xavier_init_kernel");
    if (!o.load_params.empty()) {
        std::vector<float> p(nW + nA + nWo);
        std::ifstream f(o.load_params, std::ios::binary);
        if (!f.read(reinterpret_cast<char*>(p.data()), p.size() * sizeof(float))) die("Error: cannot read --load-params file\n");
        check(gat_params_set(ctx, GAT_PARAM_W, p.data(), nW), "gat_params_set");
        check(gat_params_set(ctx, GAT_PARAM_A, p.data() + nW, nA), "gat_params_set");
        check(gat_params_set(ctx, GAT_PARAM_WO, p.data() + nW + nA, nWo), "gat_params_set");
    }

    size_t free_after = 0;
    check(gat_mem_info(&free_after, &total_mem), "gat_mem_info");
    std::printf("\n[Memory Tracker] After all allocations:\n");
    std::printf("  Free GPU memory : %.2f MB\n", free_after / (1024.0 * 1024.0));
    std::printf("  Approx. GPU memory allocated by this program: %.2f MB\n",
                (double)(free_before - free_after) / (1024.0 * 1024.0));

    check(gat_zero_grad(ctx), "gat_zero_grad");
    for (int epoch = 1; epoch <= o.epochs; ++epoch) {
        const auto start = std::chrono::high_resolution_clock::now();
        std::printf("\nEpoch %d\n", epoch);
        float loss_sum = 0.f; int32_t n_correct = 0;
        check(gat_forward(ctx, &loss_sum, &n_correct), "gatv2 forward");
        std::printf("\nAvg Loss: %f, Accuracy: %.2f%%\n", loss_sum / n_train, 100.0f * (static_cast<float>(n_correct) / n_train));
        if (!val_m.empty()) {
            double vl = 0.0; int32_t vc = 0, vn = 0;
            check(gat_eval_mask(ctx, val_m.data(), N, &vl, &vc, &vn), "gat_eval_mask");
            std::printf("Val Loss: %f, Val Accuracy: %.2f%%\n", vn ? vl / vn : 0.0, vn ? 100.0f * (static_cast<float>(vc) / vn) : 0.0f);
        }
        check(gat_backward(ctx), "gatv2 backward");
        if (o.clip) check(gat_clip(ctx, 5.0f), "clip_grad_norm");                   // E:1563
        if (o.optimizer == "adam") check(gat_step_adam(ctx, o.lr, o.beta1, o.beta2, 1e-8f, epoch), "adam_update_kernel");
        else check(gat_step_sgd(ctx, o.lr), "sgd_update_kernel");
        check(gat_zero_grad(ctx), "gat_zero_grad");
        check(gat_sync(ctx), "gat_sync");
        const std::chrono::duration<double, std::milli> elapsed = std::chrono::high_resolution_clock::now() - start;
        std::cout << " total time: " << elapsed.count() << " ms" << std::endl;
    }

    if (!o.dump_params.empty() && env.rank == 0) {
        std::vector<float> p(nW + nA + nWo);
        check(gat_params_get(ctx, GAT_PARAM_W, p.data(), nW), "gat_params_get");
        check(gat_params_get(ctx, GAT_PARAM_A, p.data() + nW, nA), "gat_params_get");
        check(gat_params_get(ctx, GAT_PARAM_WO, p.data() + nW + nA, nWo), "gat_params_get");
        std::ofstream f(o.dump_params, std::ios::binary);
        f.write(reinterpret_cast<const char*>(p.data()), p.size() * sizeof(float));
    }
    gat_destroy(ctx);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    bool multi = false;                                    // --ranks P > 1: see print_memory_tracker_before
    for (int i = 1; i + 1 < argc; ++i)
        if (std::string(argv[i]) == "--ranks" && std::atoi(argv[i + 1]) > 1) multi = true;
    if (!multi) print_memory_tracker_before();
    Options o = parse_args(argc, argv);
    if (!o.seed_given) { o.seed = (uint64_t)time(nullptr); o.seed_given = true; }     // E:1305; one seed for all ranks
    if (o.ranks == 1) return run(o, RankEnv{});
    // One process per GPU, forked before anything touches the HIP runtime.
    void* mem = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (mem == MAP_FAILED) die("Error: mmap failed\n");
    std::memset(mem, 0, 4096);
    RankEnv env;
    env.world = o.ranks;
    env.shared = static_cast<RankEnv::Shared*>(mem);
    env.shm_name = "/gatv2_" + std::to_string((long)getpid());
    std::fflush(nullptr);
    std::vector<pid_t> kids;
    for (int r = 0; r < o.ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) die("Error: fork failed\n");
        if (pid == 0) {
            env.rank = r;
            if (r != 0 && !std::freopen("/dev/null", "w", stdout)) std::_Exit(1);      // rank 0 prints
            const int rc = run(o, env);
            std::fflush(nullptr);
            std::_Exit(rc);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (size_t left = kids.size(); left > 0; --left) {
        int status = 0;
        const pid_t done = wait(&status);
        if (done < 0) break;
        const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
        if (!ok && rc == 0) {           // a rank failed: the others would wait for it in an exchange
            rc = 1;
            for (pid_t k : kids) if (k != done) kill(k, SIGTERM);
        }
    }
    return rc;
}
