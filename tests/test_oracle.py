"""CPU tests that pin the oracle (the reference ships no fixtures): autograd in
fp64, finite differences, and the invariants of the algorithm."""
import numpy as np
import pytest
import torch

from conftest import small_graph
import torch_ref


def _case(orc, seed, n, e, heads, outdims, f, c, hub=None, empty=()):
    rng = np.random.default_rng(seed)
    rp, ci = small_graph(rng, n, e, hub=hub, empty=empty)
    x = rng.standard_normal((n, f)).astype(np.float32)
    lab = rng.integers(0, c, n).astype(np.int32)
    lab[0] = c - 1
    cfg = orc.Config(list(heads), list(outdims), f, c)
    W, a, Wo = orc.xavier_params(cfg, seed + 1)
    return cfg, rp, ci, lab, x, W, a, Wo


def test_csr_to_coo_bit_exact(orc):
    rng = np.random.default_rng(0)
    rp, ci = small_graph(rng, 200, 3000, hub=(7, 700), empty=(0, 5, 199))
    src = np.empty_like(ci); dst = np.empty_like(ci)
    orc.lib().orc_csr_to_coo(rp, ci, src, dst, 200)
    assert np.array_equal(src, ci)
    assert np.array_equal(dst, np.repeat(np.arange(200, dtype=np.int32), np.diff(rp)))


@pytest.mark.parametrize("heads,outdims,f", [((8, 1), (8, 8), 64), ((3, 1), (4, 8), 5), ((1, 1), (64, 4), 100),
                                             ((8, 8), (8, 8), 33)])
def test_oracle_matches_autograd(orc, heads, outdims, f):
    """Forward and every parameter gradient of the restatement == autograd of an independent fp64
    forward.  Deviation is the reference's own: it ignores its +1e-8 epsilons in the backward."""
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 3, 40, 160, heads, outdims, f, 5, hub=(3, 70), empty=(0, 11))
    r = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
    t = torch_ref.forward(cfg, rp, ci, lab, x, W, a, Wo)
    t["loss"].backward()
    assert abs(r.loss_sum_f64 - t["loss"].item()) < 1e-4 * max(1.0, abs(t["loss"].item()))
    for l in range(cfg.L):
        assert np.abs(r.taps["alpha"][l] - t["alpha"][l].detach().numpy()).max() < 1e-5
        assert np.abs(r.taps["hpre"][l] - t["hpre"][l].detach().numpy()).max() < 1e-4
    assert np.abs(r.y - t["y"].detach().numpy()).max() < 1e-5
    for got, ref in ((r.gradW, t["W"].grad), (r.grada, t["a"].grad), (r.gradWo, t["Wo"].grad)):
        ref = ref.numpy()
        assert np.abs(got - ref).max() <= 1e-4 * max(1e-3, np.abs(ref).max())


def test_finite_difference(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 5, 24, 90, (3, 1), (4, 4), 6, 4)
    r = orc.step(cfg, rp, ci, lab, x, W, a, Wo)

    def loss64(Wv, av, Wov):
        with torch.no_grad():
            return torch_ref.forward(cfg, rp, ci, lab, x, Wv, av, Wov)["loss"].item()
    rng = np.random.default_rng(9)
    h = 1e-5
    for name, base, grad in (("W", W, r.gradW), ("a", a, r.grada), ("Wo", Wo, r.gradWo)):
        for i in rng.choice(base.size, 4, replace=False):
            up = base.astype(np.float64).copy(); dn = up.copy()
            up[i] += h; dn[i] -= h
            args = {"W": (up, a, Wo), "a": (W, up, Wo), "Wo": (W, a, up)}[name]
            args_dn = {"W": (dn, a, Wo), "a": (W, dn, Wo), "Wo": (W, a, dn)}[name]
            fd = (loss64(*args) - loss64(*args_dn)) / (2 * h)
            assert abs(fd - grad[i]) < 2e-3 * max(1.0, abs(fd)), (name, i, fd, grad[i])


def test_invariants(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 7, 60, 400, (8, 8), (8, 8), 16, 7, hub=(9, 300), empty=(0, 30, 59))
    r = orc.step(cfg, rp, ci, lab, x, W, a, Wo, backward=False)
    deg = np.diff(rp)
    dst = np.repeat(np.arange(60), deg)
    for l in range(2):
        al = r.taps["alpha"][l]
        sums = np.zeros((8, 60)); np.add.at(sums.T, dst, al.T)
        assert np.allclose(sums[:, deg > 0], 1.0, atol=1e-5)
        assert np.all(r.taps["hpre"][l][deg == 0] == 0.0)          # zero in-degree: no NaN, output 0
        assert np.all(r.taps["max"][l][:, deg == 0] == np.float32(-1e9))
        assert np.all(r.taps["sum"][l][:, deg == 0] == 0.0)
    assert np.isfinite(r.loss_sum_f32)
    assert np.allclose(r.y.sum(1), 1.0, atol=1e-5)


def test_q2_flat_index_coincides_for_single_head_last_layer(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 11, 30, 120, (4, 1), (8, 8), 10, 3)
    r0 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, flat_lrelu_index=False)
    r1 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, flat_lrelu_index=True)
    assert np.array_equal(r0.gradW, r1.gradW) and np.array_equal(r0.grada, r1.grada)
    # and diverges (O(1)) with several heads in the last layer — SURVEY Q2
    cfg2, rp, ci, lab, x, W, a, Wo = _case(orc, 11, 30, 120, (4, 3), (8, 8), 10, 3)
    q0 = orc.step(cfg2, rp, ci, lab, x, W, a, Wo, flat_lrelu_index=False)
    q1 = orc.step(cfg2, rp, ci, lab, x, W, a, Wo, flat_lrelu_index=True)
    assert np.abs(q0.gradW - q1.gradW).max() > 1e-6


def test_q1_faithful_accumulate(orc):
    """The reference never zeroes d_h (E:1208 vs E:422): a second pass adds onto the first."""
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 13, 20, 70, (2, 1), (4, 4), 6, 3)
    r0 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, backward=False)
    stale = [t.copy() for t in r0.taps["hpre"]]
    r1 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, backward=False, hpre_init=stale)
    assert np.allclose(r1.taps["hpre"][0], 2 * r0.taps["hpre"][0], rtol=1e-6)


def test_mt_baseline_variant_matches(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, 17, 50, 300, (8, 8), (8, 8), 12, 4)
    r0 = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
    r1 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, mt_baseline=True)
    assert np.allclose(r0.gradW, r1.gradW, rtol=1e-4, atol=1e-5 * np.abs(r0.gradW).max())


def test_optimizer_and_clip(orc):
    L = orc.lib()
    rng = np.random.default_rng(1)
    p = rng.standard_normal(1000).astype(np.float32); g = rng.standard_normal(1000).astype(np.float32) * 3
    g2 = g.copy()
    norm = L.orc_clip_grad_norm(g2, g2.size, 5.0)
    assert abs(norm - np.linalg.norm(g)) < 1e-2 and abs(np.linalg.norm(g2) - 5.0) < 1e-3
    p1 = p.copy(); L.orc_sgd(p1, g, 0.1, p1.size)
    assert np.allclose(p1, p - 0.1 * g, atol=1e-6)
    m = np.zeros_like(p); v = np.zeros_like(p); p2 = p.copy()
    L.orc_adam(p2, g, m, v, 0.01, p2.size, 0.9, 0.999, 1e-8, 1)
    assert np.allclose(p2, p - 0.01 * np.sign(g), atol=1e-4)        # first Adam step == lr*sign(g)


@pytest.mark.parametrize("acc64", [False, True])
@pytest.mark.parametrize("heads,outdims", [((8, 8), (8, 8)), ((3, 1), (4, 5)), ((2, 2, 1), (4, 4, 8))])
def test_restructured_cpu_step_matches_literal(orc, heads, outdims, acc64):
    """bench.py's second CPU line (PL/PR projections, O(E) softmax backward, message rows summed
    source-major — the HIP path's algorithm on host cores) against the literal restatement."""
    import parity
    from conftest import small_graph
    rng = np.random.default_rng(8)
    n, f, c = 70, 9, 4
    rp, ci = small_graph(rng, n, 500, hub=(3, 90), empty=(0, 11))
    x = rng.standard_normal((n, f)).astype(np.float32)
    lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
    cfg = orc.Config(list(heads), list(outdims), f, c)
    W, a, Wo = orc.xavier_params(cfg, 2)
    ref = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
    loss, correct, gW, ga, gWo = orc.step_restructured(cfg, rp, ci, lab, x, W, a, Wo, acc64=acc64)
    assert abs(loss - ref.loss_sum_f64) < 1e-4 * n and correct == ref.n_correct
    for name, got, want in (("gradW", gW, ref.gradW), ("grada", ga, ref.grada), ("gradWo", gWo, ref.gradWo)):
        parity.check_rel(name, got, want, 1e-4)
