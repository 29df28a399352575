"""Independent fp64 torch restatement of the GATv2 forward the reference implements,
used to pin the oracle through autograd.  Intended semantics (zeroed h_pre, per-head LReLU')."""
import numpy as np
import torch

SLOPE = 0.01


def forward(cfg, row_ptr, col_idx, labels, X, W, a, Wo, dtype=torch.float64):
    """-> dict(loss, alpha[l] [H,E], hpre[l] [N,H,D], y) with W, a, Wo as leaf tensors (requires_grad)."""
    N = len(row_ptr) - 1
    deg = np.diff(row_ptr)
    dst = torch.from_numpy(np.repeat(np.arange(N), deg)).long()
    src = torch.from_numpy(np.asarray(col_idx)).long()
    Wt = torch.tensor(np.asarray(W), dtype=dtype, requires_grad=True)
    at = torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)
    Wot = torch.tensor(np.asarray(Wo), dtype=dtype, requires_grad=True)
    x = torch.tensor(np.asarray(X), dtype=dtype)
    out = {"alpha": [], "hpre": [], "W": Wt, "a": at, "Wo": Wot}
    for l in range(cfg.L):
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Wl = Wt[cfg.w_offsets[l]:cfg.w_offsets[l + 1]].view(H, D, 2 * F)
        al = at[cfg.a_offsets[l]:cfg.a_offsets[l + 1]].view(H, D)
        PL = torch.einsum("nf,hkf->nhk", x, Wl[:, :, :F])
        PR = torch.einsum("nf,hkf->nhk", x, Wl[:, :, F:])
        s = PL[src] + PR[dst]
        e = (al * torch.nn.functional.leaky_relu(s, SLOPE)).sum(-1)              # [E,H]
        m = torch.full((N, H), -1e9, dtype=dtype).scatter_reduce(0, dst[:, None].expand(-1, H), e.detach(),
                                                                  "amax", include_self=True)
        p = torch.exp(e - m[dst])
        Z = torch.zeros((N, H), dtype=dtype).index_add(0, dst, p)
        alpha = p / (Z[dst] + 1e-8)
        hpre = torch.zeros((N, H, D), dtype=dtype).index_add(0, dst, alpha[..., None] * PL[src])
        act = torch.nn.functional.leaky_relu(hpre, SLOPE)
        x = act.mean(1) if l == cfg.L - 1 else act.reshape(N, H * D)
        out["alpha"].append(alpha.t())
        out["hpre"].append(hpre)
    z = x @ Wot.view(cfg.num_classes, cfg.outdims[-1]).t()
    ez = torch.exp(z - z.max(dim=1, keepdim=True).values.detach())
    y = ez / (ez.sum(1, keepdim=True) + 1e-8)
    lab = torch.from_numpy(np.asarray(labels)).long()
    loss = -torch.log(torch.clamp(y[torch.arange(N), lab], min=1e-12)).sum()
    out.update(loss=loss, y=y)
    return out
