"""Deterministic synthetic datasets of the shapes BASELINE.json names (the real files are a
Google-Drive link in the reference's README, R:21, and there is no network).

Counter-based (splitmix64 of seed/stream/index), so any row range can be generated on any rank
and is identical everywhere.  Graph law: in-degree by largest-remainder
apportionment of exactly E edges over w_i = (pi(i)+r0)^-beta; every edge's source drawn from the
same law under an independent permutation; sources ascending inside a row; self-loops and
duplicates kept.  CSR rows are destinations, columns sources — the reference's convention
(GATv2_edge_based.cu E:74-82).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

GRAPH_SEED = 0x67617432
_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    x ^= x >> _U64(30); x *= _U64(0xBF58476D1CE4E5B9)
    x ^= x >> _U64(27); x *= _U64(0x94D049BB133111EB)
    x ^= x >> _U64(31)
    return x


def _hash(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _mix(np.array([(seed ^ (stream * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF], np.uint64))[0]
        return _mix(idx.astype(np.uint64) * _GOLD + base)


def _uniform01(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    return (_hash(seed, stream, idx) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _argsort_u64(keys: np.ndarray) -> np.ndarray:
    return np.argsort(keys, kind="stable")


def _perm(seed: int, stream: int, n: int, argsort=_argsort_u64) -> np.ndarray:
    """pi[i] = rank of node i under a seeded hash order."""
    order = argsort(_hash(seed, stream, np.arange(n, dtype=np.uint64)))
    pi = np.empty(n, np.int64)
    pi[order] = np.arange(n, dtype=np.int64)
    return pi


def graph_tables(n: int, e: int, seed: int = GRAPH_SEED, beta: float = 0.75, r0: float = 100.0, argsort=_argsort_u64):
    """The N-sized part of the graph law: -> (row_ptr int32[n+1], cdf float64[n] over rank weights,
    node_of_rank int32[n]).  Cheap (a few argsorts of n values); the E-sized part follows in powerlaw_graph (host)
    or gat_synth_sources_device (GPU, same arrays bit for bit).  `argsort`: stable ascending argsort of uint64 keys
    (the device generator passes one that sorts on the GPU)."""
    rank_w = (np.arange(n, dtype=np.float64) + r0) ** (-beta)        # weight by rank
    pi_in = _perm(seed, 1, n, argsort)
    w = rank_w[pi_in]
    quota = w * (e / w.sum())
    deg = np.floor(quota).astype(np.int64)
    rem = int(e - deg.sum())
    if rem > 0:
        frac = quota - deg
        # largest remainders first, ties in node order: == argsort(-frac, stable).  frac is in [0, 1): the bit
        # patterns of non-negative doubles order like the values, their complements the other way round
        top = argsort(~np.ascontiguousarray(frac).view(np.uint64))[:rem]
        deg[top] += 1
    row_ptr = np.zeros(n + 1, np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    assert row_ptr[-1] == e
    # sources: inverse CDF over rank weights, rank -> node through an independent permutation
    pi_src = _perm(seed, 2, n, argsort)
    node_of_rank = np.empty(n, np.int64)
    node_of_rank[pi_src] = np.arange(n, dtype=np.int64)
    cdf = np.cumsum(rank_w)
    cdf /= cdf[-1]
    return row_ptr.astype(np.int32), cdf, node_of_rank.astype(np.int32)


def powerlaw_graph(n: int, e: int, seed: int = GRAPH_SEED, beta: float = 0.75, r0: float = 100.0
                   ) -> Tuple[np.ndarray, np.ndarray]:
    """-> (row_ptr int32[n+1], col_idx int32[e])"""
    row_ptr, cdf, node_of_rank = graph_tables(n, e, seed, beta, r0)
    deg = np.diff(row_ptr.astype(np.int64))
    col = np.empty(e, np.int64)
    step = 1 << 24
    for lo in range(0, e, step):
        hi = min(e, lo + step)
        u = _uniform01(seed, 3, np.arange(lo, hi, dtype=np.uint64))
        r = np.searchsorted(cdf, u, side="right")
        np.minimum(r, n - 1, out=r)
        col[lo:hi] = node_of_rank[r]
    # ascending sources inside each row: sort by (row, src)
    dst = np.repeat(np.arange(n, dtype=np.int64), deg)
    key = dst * np.int64(n) + col
    key.sort(kind="stable")
    col = (key % np.int64(n)).astype(np.int32)
    return row_ptr, col


def features(n: int, f: int, seed: int = GRAPH_SEED + 1, rows: Optional[Tuple[int, int]] = None,
             kind: str = "uniform") -> np.ndarray:
    """uniform: U[-1,1) fp32.  bow: sparse binary rows (~18 nnz) row-normalised (Cora-like)."""
    lo, hi = rows if rows is not None else (0, n)
    out = np.empty((hi - lo, f), np.float32)
    chunk = max(1, (1 << 24) // max(f, 1))
    for a in range(lo, hi, chunk):
        b = min(hi, a + chunk)
        idx = (np.arange(a, b, dtype=np.uint64)[:, None] * _U64(f) + np.arange(f, dtype=np.uint64)[None, :])
        u = _uniform01(seed, 7, idx.reshape(-1)).reshape(b - a, f)
        if kind == "bow":
            m = (u < (18.0 / f)).astype(np.float32)
            s = m.sum(axis=1, keepdims=True)
            out[a - lo:b - lo] = m / np.maximum(s, 1.0)
        else:
            out[a - lo:b - lo] = (u * 2.0 - 1.0).astype(np.float32)
    return out


def labels(n: int, c: int, seed: int = GRAPH_SEED + 2, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    lo, hi = rows if rows is not None else (0, n)
    lab = (_hash(seed, 11, np.arange(lo, hi, dtype=np.uint64)) % _U64(c)).astype(np.int32)
    if lo == 0 and hi > 0:
        lab[0] = c - 1          # C = max(label)+1 (E:1107) must come out as requested
    return lab


# name -> (nodes, edges, features, classes, feature kind); BASELINE.json "configs"
SHAPES: Dict[str, Tuple[int, int, int, int, str]] = {
    "cora": (2708, 5429, 1433, 7, "bow"),
    "pubmed": (19717, 44338, 500, 3, "uniform"),
    "arxiv": (169343, 1166243, 128, 40, "uniform"),
    "products": (2450000, 61900000, 100, 47, "uniform"),
    "pl10m": (10000000, 250000000, 128, 47, "uniform"),
}


def make_dataset(name: str, scale: float = 1.0, seed: int = GRAPH_SEED):
    """-> dict(row_ptr, col_idx, x, labels, n, e, f, c).  scale<1 shrinks nodes and edges alike."""
    n, e, f, c, kind = SHAPES[name]
    if scale != 1.0:
        n, e = max(16, int(n * scale)), max(16, int(e * scale))
    rp, ci = powerlaw_graph(n, e, seed)
    return dict(row_ptr=rp, col_idx=ci, x=features(n, f, seed + 1, kind=kind), labels=labels(n, c, seed + 2),
                n=n, e=e, f=f, c=c, name=name)


def make_dataset_device(name: str, device, scale: float = 1.0, seed: int = GRAPH_SEED, stream: int = 0, beta: float = 0.75,
                        sorted_sources: bool = False):
    """The same dataset generated ON the GPU (csrc/gat_synth.hip): the host builds the N-sized tables, the device
    draws and sorts the E sources and fills features and labels.  -> dict(row_ptr (host int32), d_col_idx, d_x,
    d_labels (torch tensors on `device`: feed their data_ptr() to GatContext.set_*_device), n, e, f, c).  Bit-for-bit
    make_dataset's arrays (tests/test_synth_device.py)."""
    import ctypes as C
    import torch
    from . import abi
    lib = abi.load_library()
    n, e, f, c, kind = SHAPES[name]
    if scale != 1.0:
        n, e = max(16, int(n * scale)), max(16, int(e * scale))
    def device_argsort(keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        order = np.empty(len(keys), np.int32)
        abi._chk(lib.gat_synth_argsort_u64(keys.ctypes.data_as(C.c_void_p), len(keys), order.ctypes.data_as(C.c_void_p), None))
        return order

    rp, cdf, nor = graph_tables(n, e, seed, beta, argsort=device_argsort)
    if sorted_sources:          # experiment (bench.py --sorted-sources): source popularity decreasing with the node id, i.e. what a
        nor = np.arange(n, dtype=np.int32)      # library-side popularity ordering of the table rows would make of ANY graph
    d_col = torch.empty(max(e, 1), dtype=torch.int32, device=device)
    d_x = torch.empty((n, f), dtype=torch.float32, device=device)
    d_lab = torch.empty(n, dtype=torch.int32, device=device)
    st = C.c_void_p(stream or None)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    abi._chk(lib.gat_synth_sources_device(vp(cdf), vp(nor), vp(rp), n, e, seed, C.c_void_p(d_col.data_ptr()), st))
    abi._chk(lib.gat_synth_features_device(seed + 1, 0, n, f, 1 if kind == "bow" else 0, C.c_void_p(d_x.data_ptr()), st))
    abi._chk(lib.gat_synth_labels_device(seed + 2, 0, n, c, C.c_void_p(d_lab.data_ptr()), st))
    torch.cuda.synchronize(device)
    return dict(row_ptr=rp, d_col_idx=d_col[:e], d_x=d_x, d_labels=d_lab, n=n, e=e, f=f, c=c, name=name)


def write_text_dataset(ds: dict, root: str, name: Optional[str] = None) -> str:
    """Write the four whitespace text files the reference loads (R:20-27, E:1079-1099)."""
    d = os.path.join(root, name or ds["name"])
    os.makedirs(d, exist_ok=True)
    np.savetxt(os.path.join(d, "features.txt"), ds["x"], fmt="%.9g")
    np.savetxt(os.path.join(d, "row_ptr.txt"), ds["row_ptr"], fmt="%d")
    np.savetxt(os.path.join(d, "col_idx.txt"), ds["col_idx"], fmt="%d")
    np.savetxt(os.path.join(d, "labels.txt"), ds["labels"], fmt="%d")
    return d
