"""ctypes binding of ``include/gatv2_abi.h`` (libgatv2_hip.so) — the Python host side.

This is the same seam the C++ ``train_edge`` host uses; Python is here for tests, the benchmark
and ``torch.distributed`` plumbing.  There is NO CPU fallback: if the HIP library is missing or
fails to load, importing callers get a loud ``GatLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgatv2_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "gatv2_abi.h"))


class GatLibraryError(RuntimeError):
    pass


class GatError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[gat status {code}] {msg}")
        self.code = code


class _Config(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32),
        ("heads", C.POINTER(C.c_int32)),
        ("outdims", C.POINTER(C.c_int32)),
        ("in_dim", C.c_int32),
        ("num_classes", C.c_int32),
        ("negative_slope", C.c_float),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("flat_lrelu_index", C.c_int32),
        ("collect_timing", C.c_int32),
        ("keep_taps", C.c_int32),
        ("storage_dtype", C.c_int32),
    ]


# enums of the header
PARAM_W, PARAM_A, PARAM_WO = 0, 1, 2
TABLE_PL, TABLE_GPL = 0, 1
COMM_GPL_BF16 = 1
COMM_PIPELINE = 2
COMM_HALO = 3
(TAP_SRC, TAP_DST, TAP_ALPHA, TAP_HPRE, TAP_HOUT, TAP_Y, TAP_G, TAP_GE, TAP_MAX, TAP_SUM, TAP_PL,
 TAP_PR, TAP_SCORE, TAP_GALPHA, TAP_GX) = range(15)
(K_PROJECT, K_EDGE_FWD, K_HEAD_FWD, K_HEAD_BWD, K_EDGE_BWD, K_GPL_SUM, K_GRAD_W, K_GRAD_X, K_MISC,
 K_EXCHANGE, K_EDGE_FUSED, K_COUNT) = range(12)
COMM_ID_BYTES = 128
PATH_GENERIC_SHAPE, PATH_FAST, PATH_GENERIC_SIZE = 0, 1, 2

_lib: Optional[C.CDLL] = None


def build_library(force: bool = False) -> str:
    """Compile libgatv2_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library(preload_torch: bool = True) -> C.CDLL:
    """dlopen the C-ABI library.  When torch is importable it is imported first so that this
    library binds to the HIP runtime torch already loaded (same soname), which is what lets
    torch.distributed/RCCL operate on the context's buffers and stream."""
    global _lib
    if _lib is not None:
        return _lib
    # GATV2_LIB: another build of the same ABI — the experiment library (csrc `make experiments`, libgatv2_hip_exp.so: the
    # timing-only GAT_DBG variants live only there) for tools/bench_ab.sh and the experiment-kernel tests
    lib_path = os.environ.get("GATV2_LIB") or LIB_PATH
    if not os.path.exists(lib_path):
        raise GatLibraryError(
            f"{lib_path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)")
    if preload_torch:
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: fall back to the system HIP runtime
            pass
    try:
        lib = C.CDLL(lib_path, mode=C.RTLD_LOCAL)
    except OSError as e:
        raise GatLibraryError(f"cannot load {lib_path}: {e}") from e
    _declare(lib)
    _lib = lib
    return lib


def switches() -> str:
    """The choice switches (environment) this process has read and found set: "NAME=VALUE ..." (gat_switches)."""
    buf = C.create_string_buffer(4096)
    _chk(load_library().gat_switches(buf, 4096))
    return buf.value.decode()


def _declare(lib: C.CDLL) -> None:
    i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
    P = C.POINTER
    lib.gat_last_error.restype = C.c_char_p
    lib.gat_last_error.argtypes = []
    lib.gat_kernel_name.restype = C.c_char_p
    lib.gat_kernel_name.argtypes = [C.c_int]
    sigs = {
        "gat_abi_version": [],
        "gat_device_count": [P(C.c_int)],
        "gat_switches": [C.c_char_p, i64],
        "gat_create": [P(_Config), P(vp)],
        "gat_destroy": [vp],
        "gat_sync": [vp],
        "gat_mem_info": [P(C.c_size_t), P(C.c_size_t)],
        "gat_set_graph": [vp, vp, vp, i64, i64, i64, i64],
        "gat_set_features": [vp, vp, i64, i32],
        "gat_set_labels": [vp, vp, i64],
        "gat_set_train_mask": [vp, vp, i64],
        "gat_eval_mask": [vp, vp, i64, P(C.c_double), P(i32), P(i32)],
        "gat_set_graph_device": [vp, vp, vp, i64, i64, i64, i64],
        "gat_set_features_device": [vp, vp, i64, i32],
        "gat_set_labels_device": [vp, vp, i64],
        "gat_set_source_features": [vp, vp, i64, i32],
        "gat_set_source_features_device": [vp, vp, i64, i32],
        "gat_param_count": [vp, C.c_int, P(i64)],
        "gat_params_init": [vp, C.c_uint64],
        "gat_params_set": [vp, C.c_int, vp, i64],
        "gat_params_get": [vp, C.c_int, vp, i64],
        "gat_grads_get": [vp, C.c_int, vp, i64],
        "gat_grads_set": [vp, C.c_int, vp, i64],
        "gat_grads_device": [vp, P(vp), P(i64)],
        "gat_grads_export": [vp, vp, i64],
        "gat_grads_import": [vp, vp, i64],
        "gat_result_export": [vp, vp],
        "gat_comm_unique_id": [vp],
        "gat_comm_init_rccl": [vp, i32, i32, vp],
        "gat_comm_init_host": [vp, i32, i32, C.c_char_p, i64],
        "gat_comm_option": [vp, i32, i32],
        "gat_comm_halo_info": [vp, P(i32), P(i64), P(i64), P(C.c_double)],
        "gat_step": [vp, P(f32), P(i32)],
        "gat_step_graph": [vp, i32],
        "gat_forward": [vp, P(f32), P(i32)],
        "gat_backward": [vp],
        "gat_zero_grad": [vp],
        "gat_clip": [vp, f32],
        "gat_step_sgd": [vp, f32],
        "gat_step_adam": [vp, f32, f32, f32, f32, i32],
        "gat_layer_project": [vp, i32],
        "gat_layer_forward_edges": [vp, i32],
        "gat_head_forward": [vp, P(f32), P(i32)],
        "gat_head_backward": [vp],
        "gat_layer_backward_edges": [vp, i32],
        "gat_layer_backward_dense": [vp, i32],
        "gat_layer_exchange": [vp, i32, P(i32)],
        "gat_layer_path": [vp, i32, P(i32)],
        "gat_table": [vp, C.c_int, i32, P(vp), P(i64), P(i64)],
        "gat_bind_table": [vp, C.c_int, i32, vp, i64],
        "gat_tap": [vp, C.c_int, i32, vp, i64],
        "gat_op_csr_to_coo": [vp, vp, vp, vp, i64, i64, vp],
        "gat_op_layer_forward": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, f32, vp],
        "gat_op_layer_backward": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, f32, vp],
        "gat_op_edge_score": [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i64, f32, vp],
        "gat_op_max_sum": [vp, vp, i64, i32, i64, vp, vp, vp],
        "gat_op_attn_coeff": [vp, vp, vp, vp, vp, vp, i64, i32, i64, vp],
        "gat_op_aggregate": [vp, vp, vp, vp, vp, vp, i64, i32, i64, i32, i32, vp],
        "gat_op_post_activation": [vp, vp, i64, i32, i32, i32, f32, vp],
        "gat_op_output_head": [vp, vp, vp, vp, i64, i32, i32, vp],
        "gat_op_loss_accuracy": [vp, vp, vp, vp, i64, i32, vp],
        "gat_op_output_gradients": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp],
        "gat_op_grad_attn_coeff": [i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp],
        "gat_op_grad_attn_score": [vp, vp, vp, vp, vp, i64, i32, i64, vp],
        "gat_op_grad_parameters": [i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i64, vp],
        "gat_op_features_input_gradients": [i64, i32, i64, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "gat_op_preact_gradient": [i64, f32, i32, vp, vp, vp],
        "gat_kernel_stats": [vp, C.c_int, P(i64), P(C.c_double)],
        "gat_kernel_stats_reset": [vp],
        "gat_algorithmic_bytes": [vp, P(C.c_double), P(C.c_double)],
        "gat_algorithmic_bytes_shape": [P(_Config), i64, i64, i64, i32, P(C.c_double), P(C.c_double)],
        "gat_request_bytes_shape": [P(_Config), i64, i64, i64, i32, P(C.c_double), P(C.c_double)],
        "gat_synth_sources_device": [vp, vp, vp, i64, i64, C.c_uint64, vp, vp],
        "gat_synth_features_device": [C.c_uint64, i64, i64, i32, i32, vp, vp],
        "gat_synth_labels_device": [C.c_uint64, i64, i64, i32, vp, vp],
        "gat_synth_argsort_u64": [vp, i64, vp, vp],
    }
    for name, argt in sigs.items():
        fn = getattr(lib, name)          # AttributeError here == symbol missing from the .so
        fn.argtypes = argt
        fn.restype = C.c_int


EXPORTED_SYMBOLS = None  # filled lazily by declared_symbols()


def declared_symbols() -> List[str]:
    """Function names declared in include/gatv2_abi.h (parsed from the header text)."""
    import re
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gat_[a-z0-9_]+)\s*\(", txt)))


def _chk(rc: int) -> None:
    if rc != 0:
        raise GatError(rc, load_library().gat_last_error().decode("utf-8", "replace"))


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class GatContext:
    """One GPU's (one rank's) context: mirrors the reference's per-process state — graph,
    features, labels, parameters and every intermediate of the epoch loop (E:1151-1357)."""

    def __init__(self, heads: Sequence[int], outdims: Sequence[int], in_dim: int, num_classes: int, *,
                 device: int = 0, stream: int = 0, negative_slope: float = 0.01,
                 flat_lrelu_index: bool = False, collect_timing: bool = False, keep_taps: bool = False,
                 dtype: str = "f32"):
        if len(heads) != len(outdims) or len(heads) == 0:
            raise ValueError("--heads and --outdims must both have num_layers values")
        self.lib = load_library()
        self.heads, self.outdims = list(map(int, heads)), list(map(int, outdims))
        self.L, self.in_dim, self.C = len(self.heads), int(in_dim), int(num_classes)
        self._h = (C.c_int32 * self.L)(*self.heads)
        self._d = (C.c_int32 * self.L)(*self.outdims)
        cfg = _Config(self.L, self._h, self._d, self.in_dim, self.C, negative_slope, device,
                      C.c_void_p(stream or None), int(flat_lrelu_index), int(collect_timing), int(keep_taps),
                      {"f32": 0, "bf16": 1}[dtype])
        self.storage_bytes = 2 if dtype == "bf16" else 4        # element size of the PL exchange table
        self._ctx = C.c_void_p()
        _chk(self.lib.gat_create(C.byref(cfg), C.byref(self._ctx)))
        self.n_rows = self.n_edges = self.n_table = 0
        self.table_row0 = 0

    # -- lifecycle
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.gat_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        _chk(self.lib.gat_sync(self._ctx))

    # -- shapes
    @property
    def in_dims(self) -> List[int]:
        d = [self.in_dim]
        for l in range(1, self.L):
            d.append(self.heads[l - 1] * self.outdims[l - 1])
        return d

    def param_count(self, group: int) -> int:
        n = C.c_int64()
        _chk(self.lib.gat_param_count(self._ctx, group, C.byref(n)))
        return n.value

    # -- data
    def set_graph(self, row_ptr, col_idx, n_table: Optional[int] = None, table_row0: int = 0):
        rp = np.ascontiguousarray(row_ptr, np.int32)
        ci = np.ascontiguousarray(col_idx, np.int32)
        n, e = len(rp) - 1, len(ci)
        nt = n if n_table is None else int(n_table)
        _chk(self.lib.gat_set_graph(self._ctx, _np_ptr(rp), _np_ptr(ci), n, e, nt, table_row0))
        self.n_rows, self.n_edges, self.n_table, self.table_row0 = n, e, nt, table_row0

    def set_graph_device(self, d_row_ptr: int, d_col_idx: int, n_rows: int, n_edges: int,
                         n_table: Optional[int] = None, table_row0: int = 0):
        nt = n_rows if n_table is None else int(n_table)
        _chk(self.lib.gat_set_graph_device(self._ctx, C.c_void_p(d_row_ptr), C.c_void_p(d_col_idx), n_rows,
                                           n_edges, nt, table_row0))
        self.n_rows, self.n_edges, self.n_table, self.table_row0 = n_rows, n_edges, nt, table_row0

    def set_features(self, x):
        x = np.ascontiguousarray(x, np.float32)
        _chk(self.lib.gat_set_features(self._ctx, _np_ptr(x), x.shape[0], x.shape[1]))

    def set_features_device(self, d_x: int, n_rows: int, in_dim: int):
        _chk(self.lib.gat_set_features_device(self._ctx, C.c_void_p(d_x), n_rows, in_dim))

    def set_source_features(self, x_table):
        """Replicated layer-0 input: features of every source-table row, [n_table][F0] (instead of
        set_features; layer 0 then needs no exchange)."""
        x = np.ascontiguousarray(x_table, np.float32)
        _chk(self.lib.gat_set_source_features(self._ctx, _np_ptr(x), x.shape[0], x.shape[1]))

    def set_source_features_device(self, d_x: int, n_table: int, in_dim: int):
        _chk(self.lib.gat_set_source_features_device(self._ctx, C.c_void_p(d_x), n_table, in_dim))

    def set_labels(self, labels):
        lab = np.ascontiguousarray(labels, np.int32)
        _chk(self.lib.gat_set_labels(self._ctx, _np_ptr(lab), len(lab)))

    def set_train_mask(self, mask):
        """mask: bool/uint8 [n_rows] (None: every node trains, the reference's behaviour)."""
        if mask is None:
            _chk(self.lib.gat_set_train_mask(self._ctx, None, 0))
            return
        m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
        _chk(self.lib.gat_set_train_mask(self._ctx, _np_ptr(m), len(m)))

    def eval_mask(self, mask):
        """-> (loss sum, #correct, #nodes) of the last forward over the nodes of `mask`."""
        m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
        l, c, n = C.c_double(), C.c_int32(), C.c_int32()
        _chk(self.lib.gat_eval_mask(self._ctx, _np_ptr(m), len(m), C.byref(l), C.byref(c), C.byref(n)))
        return l.value, c.value, n.value

    def set_labels_device(self, d_labels: int, n_rows: int):
        _chk(self.lib.gat_set_labels_device(self._ctx, C.c_void_p(d_labels), n_rows))

    # -- parameters
    def params_init(self, seed: int):
        _chk(self.lib.gat_params_init(self._ctx, seed))

    def params_set(self, group: int, arr):
        a = np.ascontiguousarray(arr, np.float32).reshape(-1)
        _chk(self.lib.gat_params_set(self._ctx, group, _np_ptr(a), a.size))

    def params_get(self, group: int) -> np.ndarray:
        a = np.empty(self.param_count(group), np.float32)
        _chk(self.lib.gat_params_get(self._ctx, group, _np_ptr(a), a.size))
        return a

    def grads_get(self, group: int) -> np.ndarray:
        a = np.empty(self.param_count(group), np.float32)
        _chk(self.lib.gat_grads_get(self._ctx, group, _np_ptr(a), a.size))
        return a

    def grads_device(self):
        p, n = C.c_void_p(), C.c_int64()
        _chk(self.lib.gat_grads_device(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def grads_export(self, d_dst: int, count: int):
        _chk(self.lib.gat_grads_export(self._ctx, C.c_void_p(d_dst), count))

    def result_export(self, d_dst3: int):
        _chk(self.lib.gat_result_export(self._ctx, C.c_void_p(d_dst3)))

    def grads_import(self, d_src: int, count: int):
        _chk(self.lib.gat_grads_import(self._ctx, C.c_void_p(d_src), count))

    @property
    def n_params(self) -> int:
        return self.param_count(PARAM_W) + self.param_count(PARAM_A) + self.param_count(PARAM_WO)

    # -- step
    def forward(self, want_loss: bool = True):
        if not want_loss:
            _chk(self.lib.gat_forward(self._ctx, None, None))
            return None
        loss, corr = C.c_float(), C.c_int32()
        _chk(self.lib.gat_forward(self._ctx, C.byref(loss), C.byref(corr)))
        return loss.value, corr.value

    def backward(self):
        _chk(self.lib.gat_backward(self._ctx))

    def step(self, want_loss: bool = True):
        """forward + backward (with a transport attached: exchanges and the single all-reduce inside)."""
        if not want_loss:
            _chk(self.lib.gat_step(self._ctx, None, None))
            return None
        loss, corr = C.c_float(), C.c_int32()
        _chk(self.lib.gat_step(self._ctx, C.byref(loss), C.byref(corr)))
        return loss.value, corr.value

    def step_graph(self, enable: bool = True):
        """step() as one replayed hipGraph launch (small, launch-bound graphs)."""
        _chk(self.lib.gat_step_graph(self._ctx, int(enable)))

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _chk(load_library().gat_comm_unique_id(buf))
        return buf.raw

    def comm_init_rccl(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be COMM_ID_BYTES long")
        _chk(self.lib.gat_comm_init_rccl(self._ctx, world, rank, C.c_char_p(unique_id)))

    def comm_init_host(self, world: int, rank: int, shm_name: str, bytes_per_rank: int):
        _chk(self.lib.gat_comm_init_host(self._ctx, world, rank, shm_name.encode(), bytes_per_rank))

    def comm_option(self, option: int, value: int):
        _chk(self.lib.gat_comm_option(self._ctx, option, value))

    def comm_halo_info(self):
        """(active, rows received, rows sent, referenced fraction) of GAT_COMM_HALO."""
        act, rr, rs, fr = C.c_int32(), C.c_int64(), C.c_int64(), C.c_double()
        _chk(self.lib.gat_comm_halo_info(self._ctx, C.byref(act), C.byref(rr), C.byref(rs), C.byref(fr)))
        return bool(act.value), rr.value, rs.value, fr.value

    def zero_grad(self):
        _chk(self.lib.gat_zero_grad(self._ctx))

    def clip(self, threshold: float = 5.0):
        _chk(self.lib.gat_clip(self._ctx, threshold))

    def step_sgd(self, lr: float):
        _chk(self.lib.gat_step_sgd(self._ctx, lr))

    def step_adam(self, lr: float, beta1: float, beta2: float, eps: float, t: int):
        _chk(self.lib.gat_step_adam(self._ctx, lr, beta1, beta2, eps, t))

    # -- phases
    def layer_project(self, l: int):
        _chk(self.lib.gat_layer_project(self._ctx, l))

    def layer_forward_edges(self, l: int):
        _chk(self.lib.gat_layer_forward_edges(self._ctx, l))

    def head_forward(self, want_loss: bool = True):
        if not want_loss:
            _chk(self.lib.gat_head_forward(self._ctx, None, None))
            return None
        loss, corr = C.c_float(), C.c_int32()
        _chk(self.lib.gat_head_forward(self._ctx, C.byref(loss), C.byref(corr)))
        return loss.value, corr.value

    def head_backward(self):
        _chk(self.lib.gat_head_backward(self._ctx))

    def layer_backward_edges(self, l: int):
        _chk(self.lib.gat_layer_backward_edges(self._ctx, l))

    def layer_backward_dense(self, l: int):
        _chk(self.lib.gat_layer_backward_dense(self._ctx, l))

    def layer_exchange(self, l: int) -> bool:
        need = C.c_int32(0)
        _chk(self.lib.gat_layer_exchange(self._ctx, l, C.byref(need)))
        return bool(need.value)

    def layer_path(self, l: int) -> int:
        """PATH_FAST / PATH_GENERIC_SHAPE / PATH_GENERIC_SIZE: the edge kernels layer l runs on."""
        p = C.c_int32(0)
        _chk(self.lib.gat_layer_path(self._ctx, l, C.byref(p)))
        return p.value

    def last_message(self) -> str:
        """gat_last_error(): after a successful call either stale text or a "warning: ..." left by that call."""
        return self.lib.gat_last_error().decode("utf-8", "replace")

    def table(self, which: int, l: int):
        p, n, w = C.c_void_p(), C.c_int64(), C.c_int64()
        _chk(self.lib.gat_table(self._ctx, which, l, C.byref(p), C.byref(n), C.byref(w)))
        return p.value, n.value, w.value

    def bind_table(self, which: int, l: int, d_ptr: int, nbytes: int):
        _chk(self.lib.gat_bind_table(self._ctx, which, l, C.c_void_p(d_ptr), nbytes))

    # -- taps (reference layouts)
    def tap(self, tensor: int, l: int = 0) -> np.ndarray:
        N, E = self.n_rows, self.n_edges
        H, D = self.heads[l], self.outdims[l]
        last = l == self.L - 1
        shapes = {
            TAP_SRC: ((E,), np.int32), TAP_DST: ((E,), np.int32),
            TAP_ALPHA: ((H, E), np.float32), TAP_GE: ((H, E), np.float32),
            TAP_MAX: ((H, N), np.float32), TAP_SUM: ((H, N), np.float32),
            TAP_HPRE: ((N, H, D), np.float32), TAP_G: ((N, H, D), np.float32),
            TAP_HOUT: ((N, D if last else H * D), np.float32),
            TAP_Y: ((N, self.C), np.float32),
            TAP_PL: ((self.n_table, H * D), np.float32), TAP_PR: ((N, H * D), np.float32),
            TAP_SCORE: ((H, E), np.float32), TAP_GALPHA: ((H, E), np.float32),
            TAP_GX: ((N, self.heads[l - 1] * self.outdims[l - 1] if l > 0 else 0), np.float32),
        }
        shape, dt = shapes[tensor]
        out = np.empty(shape, dt)
        _chk(self.lib.gat_tap(self._ctx, tensor, l, _np_ptr(out), out.size))
        return out

    # -- measurement
    def kernel_stats(self):
        out = {}
        for k in range(K_COUNT):
            n, ms = C.c_int64(), C.c_double()
            _chk(self.lib.gat_kernel_stats(self._ctx, k, C.byref(n), C.byref(ms)))
            out[self.lib.gat_kernel_name(k).decode()] = (n.value, ms.value)
        return out

    def kernel_stats_reset(self):
        _chk(self.lib.gat_kernel_stats_reset(self._ctx))

    def algorithmic_bytes(self):
        tot = C.c_double()
        per = (C.c_double * K_COUNT)()
        _chk(self.lib.gat_algorithmic_bytes(self._ctx, C.byref(tot), per))
        return tot.value, {self.lib.gat_kernel_name(k).decode(): per[k] for k in range(K_COUNT)}


def algorithmic_bytes_shape(heads: Sequence[int], outdims: Sequence[int], in_dim: int, num_classes: int, n_rows: int,
                            n_edges: int, *, dtype: str = "f32", n_table: Optional[int] = None, replicated_input: bool = False):
    """SURVEY 8d's algorithmic HBM bytes of one forward+backward step from the shape alone (host arithmetic: works
    without a GPU).  -> (bytes_step, {kernel class: bytes})"""
    lib = load_library()
    L = len(heads)
    h = (C.c_int32 * L)(*heads); d = (C.c_int32 * L)(*outdims)
    cfg = _Config()
    cfg.num_layers, cfg.heads, cfg.outdims = L, h, d
    cfg.in_dim, cfg.num_classes = int(in_dim), int(num_classes)
    cfg.storage_dtype = 1 if dtype == "bf16" else 0
    tot = C.c_double()
    per = (C.c_double * K_COUNT)()
    _chk(lib.gat_algorithmic_bytes_shape(C.byref(cfg), n_rows, n_edges, n_rows if n_table is None else n_table,
                                         int(replicated_input), C.byref(tot), per))
    return tot.value, {lib.gat_kernel_name(k).decode(): per[k] for k in range(K_COUNT)}


def request_bytes_shape(heads: Sequence[int], outdims: Sequence[int], in_dim: int, num_classes: int, n_rows: int,
                        n_edges: int, *, dtype: str = "f32", n_table: Optional[int] = None, replicated_input: bool = False):
    """The same byte model with every per-edge gathered / scattered row rounded up to whole 128-byte fabric requests
    (gat_request_bytes_shape): the roofline the memory system can serve.  -> (bytes_step, {kernel class: bytes})"""
    lib = load_library()
    L = len(heads)
    h = (C.c_int32 * L)(*heads); d = (C.c_int32 * L)(*outdims)
    cfg = _Config()
    cfg.num_layers, cfg.heads, cfg.outdims = L, h, d
    cfg.in_dim, cfg.num_classes = int(in_dim), int(num_classes)
    cfg.storage_dtype = 1 if dtype == "bf16" else 0
    tot = C.c_double()
    per = (C.c_double * K_COUNT)()
    _chk(lib.gat_request_bytes_shape(C.byref(cfg), n_rows, n_edges, n_rows if n_table is None else n_table,
                                     int(replicated_input), C.byref(tot), per))
    return tot.value, {lib.gat_kernel_name(k).decode(): per[k] for k in range(K_COUNT)}


def mem_info():
    lib = load_library()
    f, t = C.c_size_t(), C.c_size_t()
    _chk(lib.gat_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def device_count() -> int:
    lib = load_library()
    n = C.c_int()
    _chk(lib.gat_device_count(C.byref(n)))
    return n.value
