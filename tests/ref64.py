"""fp64 numpy restatement of the GATv2 step in the RESTRUCTURED form (s = PL[src] + PR[dst]),
with the LeakyReLU' decisions passed in as explicit boolean masks instead of being taken from the sign of
the fp64 values.  Test infrastructure only.

Why: LeakyReLU' is discontinuous at 0 (E:599, 774, 855, 890).  When some |s| or |h_pre| is within fp32
round-off of 0, two correct fp32 evaluation orders may take different sides, and the gradients then differ by a
finite, exactly computable amount.  tests/parity.py finds those entries by comparing the sign decisions of the
HIP path with those of the literal oracle, and uses this module to evaluate what the difference does to every
gradient tensor: backward(masks of the HIP path) - backward(masks of the oracle).  Everything is linear in
the masks' effect once the forward is fixed, so the correction is exact to fp64 round-off.

The backward follows the reference's formulas (it ignores the +1e-8 epsilons, like E:682-693 does).
"""
import numpy as np

SLOPE = 0.01


def _segsum(vals, row_ptr):
    """Sum of vals[e] over the CSR row of e: [E, ...] -> [N, ...] (rows without edges: 0)."""
    n = len(row_ptr) - 1
    out = np.zeros((n,) + vals.shape[1:], vals.dtype)
    deg = np.diff(row_ptr)
    nz = np.flatnonzero(deg > 0)
    if len(nz):
        out[nz] = np.add.reduceat(vals, row_ptr[:-1][nz].astype(np.int64), axis=0)
    return out


def _scatter(vals, idx, n):
    """out[idx[e]] += vals[e]  ([E, ...] -> [n, ...]) via a stable sort + segmented sum."""
    order = np.argsort(idx, kind="stable")
    sidx = idx[order]
    ptr = np.searchsorted(sidx, np.arange(n + 1))
    return _segsum(vals[order], ptr)


def forward(cfg, row_ptr, col_idx, labels, X, W, a, Wo):
    row_ptr = np.asarray(row_ptr, np.int64)
    src = np.asarray(col_idx, np.int64)
    N = len(row_ptr) - 1
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(row_ptr))
    x = np.asarray(X, np.float64)
    W = np.asarray(W, np.float64); a = np.asarray(a, np.float64)
    layers = []
    for l in range(cfg.L):
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Wl = W[cfg.w_offsets[l]:cfg.w_offsets[l + 1]].reshape(H, D, 2 * F)
        al = a[cfg.a_offsets[l]:cfg.a_offsets[l + 1]].reshape(H, D)
        PL = np.einsum("nf,hkf->nhk", x, Wl[:, :, :F])
        PR = np.einsum("nf,hkf->nhk", x, Wl[:, :, F:])
        s = PL[src] + PR[dst]                                           # [E,H,D]
        sc = (al * np.where(s > 0, s, SLOPE * s)).sum(-1)                # [E,H]
        m = np.full((N, H), -1e9)
        np.maximum.at(m, dst, sc)
        p = np.exp(sc - m[dst])
        Z = _segsum(p, row_ptr)
        alpha = p / (Z[dst] + 1e-8)
        hpre = _segsum(alpha[..., None] * PL[src], row_ptr)             # [N,H,D]
        act = np.where(hpre > 0, hpre, SLOPE * hpre)
        last = l == cfg.L - 1
        xn = act.mean(1) if last else act.reshape(N, H * D)
        layers.append(dict(x=x, Wl=Wl, al=al, PL=PL, PR=PR, s=s, alpha=alpha, hpre=hpre))
        x = xn
    C = cfg.num_classes
    Wom = np.asarray(Wo, np.float64).reshape(C, cfg.outdims[-1])
    z = x @ Wom.T
    ez = np.exp(z - z.max(1, keepdims=True))
    y = ez / (ez.sum(1, keepdims=True) + 1e-8)
    return dict(layers=layers, HL=x, Wo=Wom, y=y, src=src, dst=dst, row_ptr=row_ptr, labels=np.asarray(labels), N=N)


def backward(cfg, fw, mask_s_params, mask_s_gx, mask_h, node_mask=None):
    """mask_s_params[l], mask_s_gx[l]: bool [E,H,D], True = LeakyReLU' took the positive branch at s — as used for
    the parameter gradients (E:774) and for the input-feature gradients (E:855); mask_h[l]: bool [N,H,D] for
    LeakyReLU'(h_pre) (E:599 last layer, E:890 hidden layers); node_mask: optional [N] 0/1 training split.
    -> dict of every gradient tensor (fp64)."""
    N, src, dst, rp = fw["N"], fw["src"], fw["dst"], fw["row_ptr"]
    L = cfg.L
    y = fw["y"]
    dz = y.copy()
    dz[np.arange(N), fw["labels"]] -= 1.0                                # E:571-573 (sum loss, no 1/N)
    if node_mask is not None:                                            # training split (gat_set_train_mask): dz = 0 outside
        dz *= np.asarray(node_mask, np.float64)[:, None]
    gradWo = dz.T @ fw["HL"]
    gH = dz @ fw["Wo"]                                                   # [N,DL]
    HL_ = cfg.heads[-1]
    g = gH[:, None, :] * np.where(mask_h[L - 1], 1.0, SLOPE) / HL_      # E:597-603
    out = dict(g=[None] * L, galpha=[None] * L, ge=[None] * L, gx=[None] * L, gradWo=gradWo.reshape(-1))
    gradW = np.zeros(cfg.w_offsets[-1]); grada = np.zeros(cfg.a_offsets[-1])
    for l in range(L - 1, -1, -1):
        y_ = fw["layers"][l]
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        PLs, alpha, s, al, Wl, x = y_["PL"][src], y_["alpha"], y_["s"], y_["al"], y_["Wl"], y_["x"]
        out["g"][l] = g
        galpha = (g[dst] * PLs).sum(-1)                                  # [E,H]  E:632-646
        dot = _segsum(galpha * alpha, rp)                                # [N,H]
        ge = alpha * (galpha - dot[dst])                                 # E:682-693
        out["galpha"][l] = galpha.T.copy(); out["ge"][l] = ge.T.copy()   # reference layout [H][E]
        grada[cfg.a_offsets[l]:cfg.a_offsets[l + 1]] = (ge[..., None] * np.where(s > 0, s, SLOPE * s)).sum(0).reshape(-1)
        gs9 = ge[..., None] * al * np.where(mask_s_params[l], 1.0, SLOPE)
        gPL = _scatter(g[dst] * alpha[..., None] + gs9, src, N)
        gPR = _segsum(gs9, rp)
        gW = np.concatenate([np.einsum("nhk,nf->hkf", gPL, x), np.einsum("nhk,nf->hkf", gPR, x)], axis=2)
        gradW[cfg.w_offsets[l]:cfg.w_offsets[l + 1]] = gW.reshape(-1)
        if l == 0:
            break                                                        # E:1528
        gs10 = ge[..., None] * al * np.where(mask_s_gx[l], 1.0, SLOPE)
        gPL10 = _scatter(g[dst] * alpha[..., None] + gs10, src, N)
        gPR10 = _segsum(gs10, rp)
        gx = np.einsum("nhk,hkf->nf", gPL10, Wl[:, :, :F]) + np.einsum("nhk,hkf->nf", gPR10, Wl[:, :, F:])
        out["gx"][l] = gx
        Hp, Dp = cfg.heads[l - 1], cfg.outdims[l - 1]
        g = (gx * np.where(mask_h[l - 1].reshape(N, F), 1.0, SLOPE)).reshape(N, Hp, Dp)   # E:888-892
    out["gradW"], out["grada"] = gradW, grada
    return out
