"""The dense kernels (projection, grad_w) run on the bf16 matrix pipe with every fp32 operand cut into three bf16
pieces (csrc/gat_gemm_kernels.hip: rowgemm_x3_kernel, project_splitk_x3_kernel, gradw_x3_kernel).  The claim is that
this is an fp32-accurate product — error per product < 2^-24, less than the rounding of an fp32 multiply:
tests/test_split_pieces.py — and NOT a reduced-precision one.
These tests hold the kernels to that, against an fp64 product of the same fp32 inputs, with the yardstick the reference
itself sets: its per-edge float loop (E:303-316) is a chain of K fused multiply-adds, emulated here in numpy on the same
data (`_chain_ratio`).  With  ratio = max |got - exact| / sum_k |a_k| |b_k|  over all outputs:

    ratio(kernel) <= 1.5 * ratio(fp32 fma chain) + 1e-7     and     ratio(kernel) <= 1e-6 * max(1, sqrt(K / 100))

(measured on the K = 100 case: kernel 6.9e-7, chain 8.5e-7).  One bf16 pass (2^-9 per product) misses this by three
orders of magnitude, a three-term bf16x3 (2^-17, biased) by one.

The projection (E:303-316 hoisted out of the per-edge loop) is read through the PL / PR taps; grad_w (E:770-782 summed
over edges) through gat_layer_backward_dense on a caller-bound gPL table filled with known values.
"""
import numpy as np
import pytest

import parity
from conftest import small_graph

pytestmark = pytest.mark.gpu

BOUND = 1e-6


def _chain_ratio(a, b, exact, scale):
    """a [M,K], b [N,K] fp32: the error ratio of acc = fma(a_k, b_k, acc), k ascending, in fp32 (product exact, one rounding per step)."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    acc = np.zeros((a.shape[0], b.shape[0]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a64[:, k:k + 1] * b64[:, k][None, :]).astype(np.float32)
    return _ratio(acc, exact, scale)


def _ratio(got, exact, scale):
    return float(np.max(np.abs(got.astype(np.float64) - exact) / np.maximum(scale, 1e-300)))


@pytest.mark.parametrize("n,f,heads,outdims,what", [
    (5000, 100, [8, 8], [8, 8], "row-streaming kernel, K = 100 (Products shape)"),
    (3000, 128, [8, 8], [8, 8], "K = 128"),
    (1500, 67, [8, 8], [8, 8], "K not a multiple of 4 (scalar loads)"),
    (900, 1433, [8, 8], [8, 8], "split-K kernel, K = 1433 (Cora shape)"),
    (2500, 500, [4, 4], [16, 8], "split-K kernel, 64 output columns per half"),
])
def test_projection_is_an_fp32_accurate_product(pkg, orc, n, f, heads, outdims, what):
    A = pkg.abi
    rng = np.random.default_rng(n + f)
    rp, ci = small_graph(rng, n, 4 * n, hub=(3, min(200, n - 1)), empty=(0, 5))
    # wide dynamic range: magnitudes over 2^+-12, so low-order pieces of large values sit beside high-order pieces of small ones
    x = (rng.standard_normal((n, f)) * np.exp2(rng.integers(-12, 13, (n, f)))).astype(np.float32)
    lab = rng.integers(0, 5, n).astype(np.int32)
    cfg = orc.Config(heads, outdims, f, 5)
    W, a, Wo = orc.xavier_params(cfg, 7)
    W = (W * np.exp2(rng.integers(-6, 7, W.shape))).astype(np.float32)
    with pkg.GatContext(heads, outdims, f, 5, keep_taps=True) as ctx:
        ctx.set_graph(rp, ci); ctx.set_features(x); ctx.set_labels(lab)
        for g, arr in enumerate((W, a, Wo)):
            ctx.params_set(g, arr)
        ctx.layer_project(0)
        ctx.sync()
        PL, PR = ctx.tap(A.TAP_PL, 0), ctx.tap(A.TAP_PR, 0)
    HD = heads[0] * outdims[0]
    Wl = W[:HD * 2 * f].reshape(HD, 2 * f).astype(np.float64)
    x64 = x.astype(np.float64)
    for name, got, w in (("PL", PL, Wl[:, :f]), ("PR", PR, Wl[:, f:])):
        exact = x64 @ w.T
        scale = np.abs(x64) @ np.abs(w).T
        r = _ratio(got, exact, scale)
        rc = _chain_ratio(x, w.astype(np.float32), exact, scale)
        parity.record(f"{name} error / sum|x||w| (K={f})", r, BOUND * max(1.0, (f / 100) ** 0.5), fp32_fma_chain=rc)
        assert r <= BOUND * max(1.0, (f / 100) ** 0.5) and r <= 1.5 * rc + 1e-7, \
            f"{what}: {name} error {r:.3g} x sum|x||w| (fp32 fma chain: {rc:.3g})"
        # how much of the TIGHTER of the two bounds is used: the worst case of round 2 sat at 83 % (K = 128) — recorded, and held
        # below 90 % so that a kernel change that eats the remaining slack is noticed before it fails
        margin = r / min(BOUND * max(1.0, (f / 100) ** 0.5), 1.5 * rc + 1e-7)
        parity.record(f"{name} share of the tighter bound (K={f})", margin, 0.9)
        assert margin <= 0.9, f"{what}: {name} uses {100 * margin:.0f} % of its error bound"
    # and the bound means something: rounding the operands to bf16 once misses it by orders of magnitude
    xb = (x.view(np.uint32) & 0xFFFF0000).view(np.float32).astype(np.float64)
    r_bf16 = _ratio(xb @ Wl[:, :f].T, x64 @ Wl[:, :f].T, np.abs(x64) @ np.abs(Wl[:, :f]).T)
    assert r_bf16 > 100 * BOUND


@pytest.mark.parametrize("n,f,what", [
    (6000, 100, "128 x 128 block, F = 100"),
    (4000, 64, "128 x 64 block, F = 64"),
    (2000, 333, "three column blocks, scalar loads"),
])
def test_grad_w_is_an_fp32_accurate_product(pkg, orc, n, f, what):
    torch = pytest.importorskip("torch")
    A = pkg.abi
    rng = np.random.default_rng(n * 3 + f)
    heads, outdims = [8, 8], [8, 8]
    HD = 64
    rp, ci = small_graph(rng, n, 4 * n, hub=(3, 200), empty=(0, 5))
    x = (rng.standard_normal((n, f)) * np.exp2(rng.integers(-10, 11, (n, f)))).astype(np.float32)
    lab = rng.integers(0, 5, n).astype(np.int32)
    cfg = orc.Config(heads, outdims, f, 5)
    W, a, Wo = orc.xavier_params(cfg, 9)
    gpl = (rng.standard_normal((n, HD)) * np.exp2(rng.integers(-10, 11, (n, HD)))).astype(np.float32)
    with pkg.GatContext(heads, outdims, f, 5) as ctx:
        ctx.set_graph(rp, ci); ctx.set_features(x); ctx.set_labels(lab)
        for g, arr in enumerate((W, a, Wo)):
            ctx.params_set(g, arr)
        ctx.zero_grad()
        ctx.step()                                        # allocates every buffer; leaves some gPR behind
        ctx.sync()
        t = torch.from_numpy(gpl).to("cuda:0")
        ctx.bind_table(A.TABLE_GPL, 0, t.data_ptr(), t.numel() * 4)
        ctx.zero_grad()
        ctx.layer_backward_dense(0)
        ctx.sync()
        gW = ctx.grads_get(A.PARAM_W)[:HD * 2 * f].reshape(HD, 2 * f)
    exact = gpl.astype(np.float64).T @ x.astype(np.float64)                    # gradW_left[c][f] = sum_n gPL[n][c] X[n][f]
    scale = np.abs(gpl.astype(np.float64)).T @ np.abs(x.astype(np.float64))
    r = _ratio(gW[:, :f], exact, scale)
    # K = n nodes, summed in slabs of a few hundred nodes and then across slabs: more accurate than one chain over all nodes
    rc = _chain_ratio(np.ascontiguousarray(gpl.T), np.ascontiguousarray(x.T), exact, scale)
    parity.record(f"gradW_left error / sum|g||x| (K={n})", r, BOUND * max(1.0, (n / 100) ** 0.5), fp32_fma_chain=rc)
    assert r <= BOUND * max(1.0, (n / 100) ** 0.5) and r <= 1.5 * rc + 1e-7, \
        f"{what}: gradW_left error {r:.3g} x sum|g||x| (fp32 fma chain: {rc:.3g})"
    margin = r / min(BOUND * max(1.0, (n / 100) ** 0.5), 1.5 * rc + 1e-7)
    parity.record(f"gradW_left share of the tighter bound (K={n})", margin, 0.9)
    assert margin <= 0.9, f"{what}: gradW_left uses {100 * margin:.0f} % of its error bound"
