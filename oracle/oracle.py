"""ctypes front end of the CPU checker — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Loads ``oracle/libgatv2_oracle.so`` (built by ``oracle/Makefile``) and wires its per-kernel
restatements into one forward+backward "step" in exactly the launch order of the reference's
epoch loop (GATv2_edge_based.cu, cited E:<line>): forward E:1374-1460, backward E:1463-1557.

Allowed importers: tests/, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline leg.
With respect to the reference's own fixtures the parity is *unpinned* (the reference has no
tests, goldens or datasets); the restatement is pinned by autograd/finite differences instead
(tests/test_oracle.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """GAT_ORACLE_NATIVE=1 (set by bench.py's cpu_baseline leg before the first load): the -O3 -march=native
    build of BASELINE.md §3, compiled on the machine that times it; falls back to the portable build if the
    compiler is missing there."""
    native = os.environ.get("GAT_ORACLE_NATIVE") == "1"
    src = os.path.join(_HERE, "gatv2_oracle.cpp")
    if native:
        so = os.path.join(_HERE, "libgatv2_oracle_native.so")
        try:
            # always rebuilt: -march=native must mean THIS machine, not the one a stale file came from
            subprocess.check_call(["make", "-B", "-C", _HERE, "libgatv2_oracle_native.so"], stdout=subprocess.DEVNULL)
            return so
        except (OSError, subprocess.CalledProcessError):
            pass
    so = os.path.join(_HERE, "libgatv2_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgatv2_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_flags() -> str:
    return ("-O3 -march=native -ffp-contract=fast -fopenmp" if _LIB is not None and "native" in str(_LIB._name)
            else "-O2 -mavx2 -mfma -ffp-contract=fast -fopenmp")


def effective_cpus() -> int:
    """Host CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  (The GPU boxes of the build pool show
    256 CPUs and grant a job 16 of them through cpu.max: an OpenMP team of 256 on a 16-CPU quota is throttled to FEWER operations
    per second than a team of 16 — measured — so the oracle sizes its team to the quota, and the CPU baselines report that number.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p_ > 0:
                n = min(n, max(1, int(q / p_ + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    user_threads = os.environ.get("OMP_NUM_THREADS")
    if user_threads is None:                           # before libgomp is loaded: the team size of every parallel region
        os.environ["OMP_NUM_THREADS"] = str(effective_cpus())
    L = C.CDLL(build())
    try:                                               # ... and explicitly: an OpenMP runtime loaded earlier (torch) has read the variable already
        L.orc_set_num_threads.argtypes = [C.c_int]; L.orc_set_num_threads.restype = None
        L.orc_set_num_threads(int(user_threads) if user_threads and user_threads.isdigit() else effective_cpus())
    except AttributeError:
        pass
    i, f = C.c_int, C.c_float
    sig = {
        "orc_num_threads": ([], i),
        "orc_csr_to_coo": ([i32p, i32p, i32p, i32p, i], None),
        "orc_edge_score": ([f32p, i32p, i32p, f32p, f32p, f32p, i, i, i, i, i, f], None),
        "orc_presum": ([f32p, i32p, i32p, f32p, f32p, i, i, i, i, i], None),
        "orc_max_sum": ([i32p, f32p, i, i, i, f32p, f32p], None),
        "orc_attn_coeff": ([i32p, f32p, f32p, f32p, f32p, i, i, i], None),
        "orc_aggregate": ([i32p, i32p, f32p, f32p, f32p, f32p, i, i, i, i, i], None),
        "orc_post_activation": ([f32p, f32p, i, i, i, i, f], None),
        "orc_output_head": ([f32p, f32p, f32p, f32p, i, i, i], None),
        "orc_loss_accuracy": ([f32p, i32p, f32p, i32p, i, i], None),
        "orc_reduce_loss": ([f32p, i32p, i, C.POINTER(f), C.POINTER(C.c_double), C.POINTER(i)], None),
        "orc_output_gradients": ([f32p, i32p, f32p, f32p, f32p, f32p, f32p, i, i, i, i, f, i], None),
        "orc_grad_attn_coeff": ([i, i, i, i, i32p, i32p, f32p, f32p, f32p, f32p], None),
        "orc_grad_attn_score": ([i32p, i32p, f32p, f32p, f32p, i, i, i], None),
        "orc_grad_parameters": ([i, i, i32p, i32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, i, i, f], None),
        "orc_features_input_gradients": ([i, i, i, i, i, f, i32p, i32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p], None),
        "orc_features_input_gradients_mt": ([i, i, i, i, i, f, i32p, i32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p], None),
        "orc_preact_gradient": ([i, f, i, f32p, f32p], None),
        "orc_sgd": ([f32p, f32p, f, C.c_int64], None),
        "orc_adam": ([f32p, f32p, f32p, f32p, f, C.c_int64, f, f, f, i], None),
        "orc_clip_grad_norm": ([f32p, C.c_int64, f], f),
        "orc_step_restructured": ([i, i32p, i32p, i, i, i, i, i32p, i32p, i32p, f32p, f32p, f32p, f32p, f,
                                   C.POINTER(C.c_double), C.POINTER(i), f32p, f32p, f32p, i,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], i),
    }
    for name, (argt, rest) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argt
        fn.restype = rest
    _LIB = L
    return L


NEG_SLOPE = 0.01  # E:1143


@dataclass
class Config:
    """Layer/head configuration as the reference's CLI gives it (E:954-987, 1115-1118)."""
    heads: List[int]
    outdims: List[int]
    in_dim0: int
    num_classes: int

    @property
    def L(self) -> int:
        return len(self.heads)

    @property
    def in_dims(self) -> List[int]:
        d = [self.in_dim0]
        for l in range(1, self.L):
            d.append(self.heads[l - 1] * self.outdims[l - 1])
        return d

    @property
    def w_offsets(self) -> List[int]:          # E:1242-1254
        off, o = [], 0
        for l in range(self.L):
            off.append(o)
            o += self.heads[l] * self.outdims[l] * 2 * self.in_dims[l]
        return off + [o]

    @property
    def a_offsets(self) -> List[int]:
        off, o = [], 0
        for l in range(self.L):
            off.append(o)
            o += self.heads[l] * self.outdims[l]
        return off + [o]

    @property
    def wo_size(self) -> int:
        return self.num_classes * self.outdims[-1]


def xavier_params(cfg: Config, seed: int):
    """Seeded parameters with the reference's init distribution (E:205-242): U(-lim, lim]
    with lim = sqrt(6/(2F+D)) for W rows and a, sqrt(6/(C+D_L)) for W_o.  The reference's
    own stream (cuRAND XORWOW seeded with time(NULL), E:1305) cannot be reproduced."""
    rng = np.random.default_rng(seed)
    W = np.empty(cfg.w_offsets[-1], np.float32)
    a = np.empty(cfg.a_offsets[-1], np.float32)
    for l in range(cfg.L):
        F, D, H = cfg.in_dims[l], cfg.outdims[l], cfg.heads[l]
        lim = np.float32(np.sqrt(np.float32(6.0) / np.float32(2 * F + D)))
        n = H * D * 2 * F
        W[cfg.w_offsets[l]:cfg.w_offsets[l] + n] = (1.0 - rng.random(n, np.float32)) * 2 * lim - lim
        a[cfg.a_offsets[l]:cfg.a_offsets[l] + H * D] = (1.0 - rng.random(H * D, np.float32)) * 2 * lim - lim
    limo = np.float32(np.sqrt(np.float32(6.0) / np.float32(cfg.num_classes + cfg.outdims[-1])))
    Wo = ((1.0 - rng.random(cfg.wo_size, np.float32)) * 2 * limo - limo).astype(np.float32)
    return W, a, Wo


@dataclass
class StepResult:
    loss_sum_f32: float = 0.0
    loss_sum_f64: float = 0.0
    n_correct: int = 0
    src: np.ndarray = None
    dst: np.ndarray = None
    taps: Dict[str, list] = field(default_factory=dict)
    y: np.ndarray = None
    gradW: np.ndarray = None
    grada: np.ndarray = None
    gradWo: np.ndarray = None


def step(cfg: Config, row_ptr, col_idx, labels, X0, W, a, Wo, *, flat_lrelu_index=False,
         hpre_init: Optional[List[np.ndarray]] = None, backward=True, mt_baseline=False) -> StepResult:
    """One forward(+backward) over all layers incl. output head and loss, literal algorithm.

    ``hpre_init`` (Q1, faithful accumulate): per-layer arrays the aggregation adds INTO; the
    default is the intended semantics, zeros.  All edge tensors come back in the reference
    layout: ``[H][E]`` head-major; node tensors ``[N][H][D]``.
    """
    Lb = lib()
    N, E = len(row_ptr) - 1, len(col_idx)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    labels = np.ascontiguousarray(labels, np.int32)
    X0 = np.ascontiguousarray(X0, np.float32)
    W = np.ascontiguousarray(W, np.float32); a = np.ascontiguousarray(a, np.float32)
    Wo = np.ascontiguousarray(Wo, np.float32)
    r = StepResult()
    src = np.empty(E, np.int32); dst = np.empty(E, np.int32)
    Lb.orc_csr_to_coo(row_ptr, col_idx, src, dst, N)                      # E:1186
    r.src, r.dst = src, dst
    Lc, Cn = cfg.L, cfg.num_classes
    Hmax = max(cfg.heads)
    mx = np.empty(Hmax * N, np.float32); sm = np.empty(Hmax * N, np.float32)
    taps = {k: [None] * Lc for k in ("score", "alpha", "hpre", "H", "max", "sum", "g", "galpha", "ge", "gx")}
    Xin = X0
    for l in range(Lc):                                                    # E:1375
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Wl = W[cfg.w_offsets[l]:cfg.w_offsets[l + 1]]
        al = a[cfg.a_offsets[l]:cfg.a_offsets[l + 1]]
        score = np.empty(H * E, np.float32); alpha = np.empty(H * E, np.float32)
        Lb.orc_edge_score(Xin, col_idx, dst, Wl, al, score, N, F, D, H, E, NEG_SLOPE)   # E:1386
        Lb.orc_max_sum(row_ptr, score, N, H, E, mx, sm)                    # E:1398
        Lb.orc_attn_coeff(dst, score, mx, sm, alpha, E, H, N)              # E:1407
        hpre = (np.zeros(N * H * D, np.float32) if hpre_init is None
                else np.ascontiguousarray(hpre_init[l], np.float32).reshape(-1).copy())
        Lb.orc_aggregate(row_ptr, src, alpha, Xin, Wl, hpre, N, H, E, F, D)  # E:1416
        last = l == Lc - 1
        Hout = np.empty(N * (D if last else H * D), np.float32)
        Lb.orc_post_activation(hpre, Hout, N, H, D, int(last), NEG_SLOPE)  # E:1428
        taps["score"][l] = score.reshape(H, E); taps["alpha"][l] = alpha.reshape(H, E)
        taps["hpre"][l] = hpre.reshape(N, H, D)
        taps["H"][l] = Hout.reshape(N, D if last else H * D)
        taps["max"][l] = mx[:H * N].reshape(H, N).copy(); taps["sum"][l] = sm[:H * N].reshape(H, N).copy()
        Xin = Hout                                                         # E:1435
    DL = cfg.outdims[-1]
    z = np.empty(N * Cn, np.float32); y = np.empty(N * Cn, np.float32)
    Lb.orc_output_head(Wo, Xin, z, y, N, Cn, DL)                           # E:1446
    losses = np.empty(N, np.float32); corrects = np.empty(N, np.int32)
    Lb.orc_loss_accuracy(y, labels, losses, corrects, N, Cn)               # E:1457
    sf, sd, nc = C.c_float(), C.c_double(), C.c_int()
    Lb.orc_reduce_loss(losses, corrects, N, C.byref(sf), C.byref(sd), C.byref(nc))  # E:542-543
    r.loss_sum_f32, r.loss_sum_f64, r.n_correct = sf.value, sd.value, nc.value
    r.y = y.reshape(N, Cn)
    r.taps = taps
    if not backward:
        return r
    gradW = np.zeros_like(W); grada = np.zeros_like(a); gradWo = np.zeros_like(Wo)   # E:1262-1266
    g = [np.zeros(N * cfg.heads[l] * cfg.outdims[l], np.float32) for l in range(Lc)]  # E:1339-1345
    HLm1, DLm1 = cfg.heads[-1], cfg.outdims[-1]
    Lb.orc_output_gradients(y, labels, taps["hpre"][-1].reshape(-1), taps["H"][-1].reshape(-1), Wo,
                            gradWo, g[-1], N, Cn, DLm1, HLm1, NEG_SLOPE, int(flat_lrelu_index))  # E:1468
    for l in range(Lc - 1, -1, -1):                                        # E:1483
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Wl = W[cfg.w_offsets[l]:cfg.w_offsets[l + 1]]
        al = a[cfg.a_offsets[l]:cfg.a_offsets[l + 1]]
        Xl = (taps["H"][l - 1].reshape(-1) if l > 0 else X0.reshape(-1))   # E:1490
        alpha = taps["alpha"][l].reshape(-1)
        galpha = np.empty(H * E, np.float32); ge = np.empty(H * E, np.float32)
        Lb.orc_grad_attn_coeff(E, H, F, D, src, dst, Xl, Wl, g[l], galpha)                 # E:1489
        Lb.orc_grad_attn_score(row_ptr, dst, alpha, galpha, ge, N, H, E)                   # E:1503
        gWl = gradW[cfg.w_offsets[l]:cfg.w_offsets[l + 1]]
        gal = grada[cfg.a_offsets[l]:cfg.a_offsets[l + 1]]
        Lb.orc_grad_parameters(E, H, src, dst, Xl, g[l], ge, alpha, Wl, al, gWl, gal, F, D, NEG_SLOPE)  # E:1517
        taps["g"][l] = g[l].reshape(N, H, D); taps["galpha"][l] = galpha.reshape(H, E)
        taps["ge"][l] = ge.reshape(H, E)
        if l == 0:
            break                                                          # E:1528
        fn = Lb.orc_features_input_gradients_mt if mt_baseline else Lb.orc_features_input_gradients
        fn(N, H, E, F, D, NEG_SLOPE, src, dst, alpha, Xl, Wl, g[l], ge, al, g[l - 1])      # E:1533
        taps["gx"][l] = g[l - 1].reshape(N, F).copy()
        Lb.orc_preact_gradient(N, NEG_SLOPE, F, taps["hpre"][l - 1].reshape(-1), g[l - 1])  # E:1546
    r.gradW, r.grada, r.gradWo = gradW, grada, gradWo
    return r


def presum_signs(cfg: Config, row_ptr, col_idx, X0, W, ref: "StepResult", values: bool = False):
    """Per layer the sign decisions (s > 0) of the literal restatement, in both float orders the reference
    uses: -> (pos_lr[l], pos_il[l]) bool arrays [E][H][D]; see orc_presum.  values=True: the float32 pre-activations
    s themselves instead of their signs (the tests bound |s| where two paths decide differently).  Test
    infrastructure for the LeakyReLU-kink bookkeeping of tests/parity.py."""
    Lb = lib()
    E = len(col_idx)
    src = np.ascontiguousarray(col_idx, np.int32)
    dst = np.ascontiguousarray(ref.dst, np.int32)
    W = np.ascontiguousarray(W, np.float32)
    out_lr, out_il = [], []
    for l in range(cfg.L):
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Xl = np.ascontiguousarray(ref.taps["H"][l - 1] if l > 0 else X0, np.float32).reshape(-1)
        Wl = W[cfg.w_offsets[l]:cfg.w_offsets[l + 1]]
        s = np.empty(E * H * D, np.float32)
        Lb.orc_presum(Xl, src, dst, Wl, s, F, D, H, E, 0)
        out_lr.append(s.reshape(E, H, D).copy() if values else (s > 0).reshape(E, H, D))
        Lb.orc_presum(Xl, src, dst, Wl, s, F, D, H, E, 1)
        out_il.append(s.reshape(E, H, D).copy() if values else (s > 0).reshape(E, H, D))
    return out_lr, out_il


def step_restructured(cfg: Config, row_ptr, col_idx, labels, X0, W, a, Wo, acc64: bool = False, decisions=None):
    """One forward + backward with the RESTRUCTURED algorithm (the one the HIP path uses) on the host
    cores: bench.py's second CPU line; not a restatement of the reference's kernels.
    decisions = (sbits, hbits): packed LeakyReLU' decisions to use (see orc_step_restructured); the result then
    carries ``flips`` = (counts int64 [L][2], max |value| at a differing decision float32 [L][2]).
    -> (loss_sum, n_correct, gradW, grada, gradWo[, flips])"""
    Lb = lib()
    row_ptr = np.ascontiguousarray(row_ptr, np.int32); col_idx = np.ascontiguousarray(col_idx, np.int32)
    labels = np.ascontiguousarray(labels, np.int32); X0 = np.ascontiguousarray(X0, np.float32)
    W = np.ascontiguousarray(W, np.float32); a = np.ascontiguousarray(a, np.float32)
    Wo = np.ascontiguousarray(Wo, np.float32)
    heads = np.asarray(cfg.heads, np.int32); outdims = np.asarray(cfg.outdims, np.int32)
    gW, ga, gWo = np.zeros_like(W), np.zeros_like(a), np.zeros_like(Wo)
    loss, corr = C.c_double(), C.c_int()
    sb = hb = fc = fm = None
    cnt = np.zeros((cfg.L, 2), np.int64); mx = np.zeros((cfg.L, 2), np.float32)
    if decisions is not None:
        sbits = np.ascontiguousarray(decisions[0], np.uint8); hbits = np.ascontiguousarray(decisions[1], np.uint8)
        N, E = len(row_ptr) - 1, len(col_idx)
        assert sbits.size == sum((E * h * d + 7) // 8 for h, d in zip(cfg.heads, cfg.outdims))
        assert hbits.size == sum((N * h * d + 7) // 8 for h, d in zip(cfg.heads, cfg.outdims))
        sb, hb = sbits.ctypes.data, hbits.ctypes.data
        fc, fm = cnt.ctypes.data, mx.ctypes.data
    rc = Lb.orc_step_restructured(cfg.L, heads, outdims, cfg.in_dim0, cfg.num_classes, len(row_ptr) - 1, len(col_idx),
                                  row_ptr, col_idx, labels, X0, W, a, Wo, NEG_SLOPE, C.byref(loss), C.byref(corr),
                                  gW, ga, gWo, int(acc64), sb, hb, fc, fm)
    assert rc == 0
    if decisions is not None:
        return loss.value, corr.value, gW, ga, gWo, (cnt, mx)
    return loss.value, corr.value, gW, ga, gWo
