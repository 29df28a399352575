"""numpy stand-in for GatContext's phase API (TEST INFRASTRUCTURE).

Implements the same restructured algorithm the HIP path uses (PL/PR projections, destination-
segmented softmax, O(E) softmax backward, table-based exchange) in float64 numpy on host memory,
so that the exchange logic of ``shard.ShardedGat`` (partition, table-id remap, all-gather,
reduce-scatter, gradient all-reduce) can be driven with ``gloo`` on a machine without a GPU and
checked against the literal oracle.  Never imported by the product.
"""
import ctypes

import numpy as np

SLOPE = 0.01


def _view(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))


def lrelu(x):
    return np.where(x > 0, x, SLOPE * x)


def dlrelu(x):
    return np.where(x > 0, 1.0, SLOPE)


class FakeContext:
    def __init__(self, heads, outdims, in_dim, num_classes):
        self.heads, self.outdims, self.in_dim, self.C = list(heads), list(outdims), in_dim, num_classes
        self.L = len(heads)
        self.hd = [h * d for h, d in zip(heads, outdims)]
        self.F = [in_dim] + self.hd[:-1]
        self.w_off = np.cumsum([0] + [self.hd[l] * 2 * self.F[l] for l in range(self.L)])
        self.a_off = np.cumsum([0] + self.hd)
        self.nW, self.nA, self.nWo = int(self.w_off[-1]), int(self.a_off[-1]), num_classes * outdims[-1]
        self.grads = np.zeros(self.nW + self.nA + self.nWo, np.float64)
        self.tables = {}

    @property
    def n_params(self):
        return self.nW + self.nA + self.nWo

    # -- data
    def set_graph(self, rp, ci, n_table=None, table_row0=0):
        self.rp, self.ci = np.asarray(rp, np.int64), np.asarray(ci, np.int64)
        self.n = len(rp) - 1
        self.n_table = self.n if n_table is None else n_table
        self.row0 = table_row0
        self.dst = np.repeat(np.arange(self.n), np.diff(self.rp))

    def set_features(self, x):
        self.X0 = np.asarray(x, np.float64)

    def set_source_features(self, x_table):
        self.Xtab = np.asarray(x_table, np.float64)
        assert self.Xtab.shape[0] == self.n_table
        self.X0 = self.Xtab[self.row0:self.row0 + self.n]

    def layer_exchange(self, l):
        return self.n_table != self.n and not (l == 0 and getattr(self, "Xtab", None) is not None)

    def set_labels(self, lab):
        self.labels = np.asarray(lab, np.int64)

    def params_set(self, group, arr):
        setattr(self, ("W", "a", "Wo")[group], np.asarray(arr, np.float64).copy())

    def bind_table(self, which, l, ptr, nbytes):
        if which == 1:
            for k in range(self.L):
                self.tables[(1, k)] = _view(ptr, self.n_table * self.hd[k])
        else:
            self.tables[(0, l)] = _view(ptr, self.n_table * self.hd[l])

    def grads_export(self, ptr, n):
        _view(ptr, n)[:] = self.grads.astype(np.float32)

    def result_export(self, ptr):
        _view(ptr, 3)[:] = [self._loss, self._correct & 4095, self._correct >> 12]

    def grads_import(self, ptr, n):
        self.grads[:] = _view(ptr, n)

    def zero_grad(self):
        self.grads[:] = 0

    # -- helpers
    def _Wl(self, l):
        H, D, F = self.heads[l], self.outdims[l], self.F[l]
        return self.W[self.w_off[l]:self.w_off[l + 1]].reshape(H * D, 2 * F)

    def _al(self, l):
        return self.a[self.a_off[l]:self.a_off[l + 1]]

    def _X(self, l):
        return self.X0 if l == 0 else self.hout[l - 1]

    def _segsum(self, vals):
        out = np.zeros((self.n,) + vals.shape[1:])
        np.add.at(out, self.dst, vals)
        return out

    # -- phases
    def layer_project(self, l):
        if l == 0:
            self.PR, self.hpre, self.hout, self.alpha, self.g = {}, {}, {}, {}, {}
        W, F, HD = self._Wl(l), self.F[l], self.hd[l]
        X = self._X(l)
        tab = self.tables[(0, l)].reshape(self.n_table, HD)
        if l == 0 and getattr(self, "Xtab", None) is not None:
            tab[:] = (self.Xtab @ W[:, :F].T).astype(np.float32)
        else:
            tab[self.row0:self.row0 + self.n] = (X @ W[:, :F].T).astype(np.float32)
        self.PR[l] = X @ W[:, F:].T

    def layer_forward_edges(self, l):
        H, D, HD = self.heads[l], self.outdims[l], self.hd[l]
        PL = self.tables[(0, l)].reshape(self.n_table, HD).astype(np.float64)
        v = PL[self.ci]
        s = v + self.PR[l][self.dst]
        e = (self._al(l) * lrelu(s)).reshape(-1, H, D).sum(-1)                      # [E, H]
        m = np.full((self.n, H), -1e9)
        np.maximum.at(m, self.dst, e)
        p = np.exp(e - m[self.dst])
        Z = self._segsum(p)
        alpha = p / (Z[self.dst] + 1e-8)
        hpre = self._segsum(np.repeat(alpha, D, axis=1) * v)
        self.alpha[l], self.hpre[l] = alpha, hpre
        act = lrelu(hpre)
        self.hout[l] = act.reshape(self.n, H, D).mean(1) if l == self.L - 1 else act

    def head_forward(self, want_loss=True):
        Wo = self.Wo.reshape(self.C, self.outdims[-1])
        z = self.hout[self.L - 1] @ Wo.T
        ez = np.exp(z - z.max(1, keepdims=True))
        self.y = ez / (ez.sum(1, keepdims=True) + 1e-8)
        pl = self.y[np.arange(self.n), self.labels]
        self._loss = float(-np.log(np.maximum(pl, 1e-12)).sum())
        self._correct = int((self.y.argmax(1) == self.labels).sum())
        return self._loss, self._correct

    def head_backward(self):
        L, H, D = self.L - 1, self.heads[-1], self.outdims[-1]
        Wo = self.Wo.reshape(self.C, D)
        dz = self.y.copy()
        dz[np.arange(self.n), self.labels] -= 1.0
        self.grads[self.nW + self.nA:] += (dz.T @ self.hout[L]).reshape(-1)
        gH = dz @ Wo
        self.g[L] = (np.tile(gH, (1, H)) * dlrelu(self.hpre[L]) / H)

    def layer_backward_edges(self, l):
        H, D, HD = self.heads[l], self.outdims[l], self.hd[l]
        PL = self.tables[(0, l)].reshape(self.n_table, HD).astype(np.float64)
        g, a = self.g[l], self._al(l)
        v = PL[self.ci]
        s = v + self.PR[l][self.dst]
        gd = g[self.dst]
        alpha = self.alpha[l]
        galpha = (gd * v).reshape(-1, H, D).sum(-1)
        dot = (g * self.hpre[l]).reshape(self.n, H, D).sum(-1)
        ge = alpha * (galpha - dot[self.dst])
        ge_c = np.repeat(ge, D, axis=1)
        gs = ge_c * a * dlrelu(s)
        self.grads[self.nW + self.a_off[l]:self.nW + self.a_off[l + 1]] += (ge_c * lrelu(s)).sum(0)
        self.gPR = self._segsum(gs)
        msg = gd * np.repeat(alpha, D, axis=1) + gs
        gpl = np.zeros((self.n_table, HD))
        np.add.at(gpl, self.ci, msg)
        self.tables[(1, l)].reshape(self.n_table, HD)[:] = gpl.astype(np.float32)

    def layer_backward_dense(self, l):
        HD, F = self.hd[l], self.F[l]
        gPL = self.tables[(1, l)].reshape(self.n_table, HD)[self.row0:self.row0 + self.n].astype(np.float64)
        X, W = self._X(l), self._Wl(l)
        if l == 0 and getattr(self, "Xtab", None) is not None:
            gl = self.tables[(1, l)].reshape(self.n_table, HD).astype(np.float64).T @ self.Xtab
        else:
            gl = gPL.T @ X
        gW = np.concatenate([gl, self.gPR.T @ X], axis=1)                           # [HD, 2F]
        self.grads[self.w_off[l]:self.w_off[l + 1]] += gW.reshape(-1)
        if l > 0:
            gX = gPL @ W[:, :F] + self.gPR @ W[:, F:]
            self.g[l - 1] = gX * dlrelu(self.hpre[l - 1])
