"""Behaviour at the boundary for inputs the parity fixtures never hold:
  * a non-finite feature poisons exactly what it can reach, as in the reference, and nothing hangs or faults (the kind of
    non-finite value differs: NaN here where the reference's fp32 loops, E:303-316, may hold +-inf — gatv2_abi.h at
    gat_set_features);
  * a source table of 4 GiB or more drops a layer to the generic kernels, and says so (gat_layer_path + a "warning:" text);
  * labels replaced after a training mask are masked again;
  * a replicated-input shard with F > 128 and n_rows < 32641 <= n_table projects both halves correctly (the split-K
    scratch is sized per launch shape)."""
import numpy as np
import pytest

from conftest import small_graph

pytestmark = pytest.mark.gpu


def _ctx(pkg, heads=(8, 8), outdims=(8, 8), n=200, e=1500, f=12, c=4, seed=2, **kw):
    rng = np.random.default_rng(seed)
    rp, ci = small_graph(rng, n, e, hub=(5, 300), empty=(0,))
    x = rng.standard_normal((n, f)).astype(np.float32)
    lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
    ctx = pkg.GatContext(list(heads), list(outdims), f, c, **kw)
    ctx.set_graph(rp, ci)
    return ctx, rp, ci, x, lab


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan, 3.4e38])
def test_nonfinite_feature_poisons_its_two_hop_neighbourhood_and_nothing_else(pkg, orc, bad):
    """What the reference does with a non-finite feature (its plain fp32 loops, restated by the oracle): the node's
    projections become +-inf / NaN, every row that aggregates it within two layers ends with NaN probabilities, and the
    loss stays FINITE because E:527 clamps with fmaxf(prob, 1e-12f), which drops a NaN.  Here: the same, except that
    the three-piece dense product turns +-inf into NaN at once — so a row the reference rescues (an edge whose score
    is -inf gets alpha = 0 there) can be NaN here.  Asserted: the step returns; the poisoned rows are a superset of
    the oracle's and a subset of the two-hop neighbourhood; every other row agrees at 1e-4; the context stays usable."""
    A = pkg.abi
    ctx, rp, ci, x, lab = _ctx(pkg)
    n = len(lab)
    with ctx:
        x[17, 3] = bad
        if bad == 3.4e38:
            x[17, :] = bad                       # within half a bf16 ulp of FLT_MAX: the three-piece cut overflows (documented)
        ctx.set_features(x); ctx.set_labels(lab)
        cfg = orc.Config([8, 8], [8, 8], x.shape[1], 4)
        W, a, Wo = orc.xavier_params(cfg, 1)
        for g, arr in enumerate((W, a, Wo)):
            ctx.params_set(g, arr)
        ctx.zero_grad()
        loss, correct = ctx.step()                # returns: nothing hangs or faults
        assert 0 <= correct <= n
        y = ctx.tap(A.TAP_Y)
        with np.errstate(all="ignore"):
            ref = orc.step(cfg, rp, ci, lab, x, W, a, Wo, backward=False)
        mine = ~np.isfinite(y).all(1)
        theirs = ~np.isfinite(ref.y).all(1)
        dst = np.repeat(np.arange(n), np.diff(rp))
        hop1 = np.zeros(n, bool); hop1[17] = True; hop1[dst[ci == 17]] = True
        hop2 = hop1.copy(); hop2[dst[hop1[ci]]] = True
        assert mine[17] and theirs[17]
        assert not (theirs & ~mine).any(), "a row the reference poisons is clean here"
        assert not (mine & ~hop2).any(), "a row outside the two-hop neighbourhood is poisoned"
        ok = ~mine & ~theirs
        assert ok.sum() > 0 and np.abs(y[ok] - ref.y[ok]).max() < 1e-4
        if (mine == theirs).all():                # same poisoned set => same clamped loss (E:527)
            assert np.isfinite(loss) and abs(loss - ref.loss_sum_f64) < 1e-4 * n
        assert ctx.grads_get(A.PARAM_W).shape[0] > 0
        # the context stays usable: finite features again -> finite loss equal to a clean run
        x[17, :] = 0.5
        ctx.set_features(x); ctx.zero_grad()
        loss2, _ = ctx.step()
        ref2 = orc.step(cfg, rp, ci, lab, x, W, a, Wo, backward=False)
        assert np.isfinite(loss2) and abs(loss2 - ref2.loss_sum_f64) < 1e-4 * n


def test_labels_set_after_a_train_mask_are_masked_again(pkg):
    A = pkg.abi
    ctx, rp, ci, x, lab = _ctx(pkg)
    with ctx:
        n = len(lab)
        ctx.set_features(x); ctx.set_labels(lab)
        ctx.params_init(3)
        mask = np.arange(n) % 3 == 0
        ctx.set_train_mask(mask)
        ctx.zero_grad(); l1, c1 = ctx.step(); g1 = ctx.grads_get(A.PARAM_WO).copy()
        lab2 = (lab + 1) % 4
        ctx.set_labels(lab2)                      # new labels, mask still active
        ctx.zero_grad(); l2, c2 = ctx.step(); g2 = ctx.grads_get(A.PARAM_WO).copy()
        # reference: a fresh context given lab2 and the same mask
        ctx2, *_ = _ctx(pkg)
        with ctx2:
            ctx2.set_features(x); ctx2.set_labels(lab2)
            for g in range(3):
                ctx2.params_set(g, ctx.params_get(g))
            ctx2.set_train_mask(mask)
            ctx2.zero_grad(); l3, c3 = ctx2.step(); g3 = ctx2.grads_get(A.PARAM_WO).copy()
        assert l2 == l3 and c2 == c3 and np.array_equal(g2, g3)
        assert l2 != l1
        with pytest.raises(A.GatError):           # a mask of another length is refused, not read out of range
            ctx.set_train_mask(np.ones(n + 5, bool))


def test_table_of_4gib_says_it_left_the_fast_path(pkg):
    A = pkg.abi
    n_table = (1 << 24) + 64                      # x 64 floats x 4 B >= 4 GiB
    rng = np.random.default_rng(4)
    n, f, c = 64, 8, 3
    rp, ci = small_graph(rng, n, 400, n_src=n)
    ci = (ci.astype(np.int64) * (n_table // n)).astype(np.int32)     # sources spread over the whole table
    with pkg.GatContext([8, 8], [8, 8], f, c) as ctx:
        ctx.set_graph(rp, ci, n_table=n_table, table_row0=0)
        assert ctx.layer_path(0) == A.PATH_GENERIC_SIZE and ctx.layer_path(1) == A.PATH_GENERIC_SIZE
        ctx.set_features(rng.standard_normal((n, f)).astype(np.float32))
        ctx.set_labels(rng.integers(0, c, n).astype(np.int32))
        msg = ctx.last_message()
        assert msg.startswith("warning:") and "4 GiB" in msg and "generic" in msg, msg
    with pkg.GatContext([8, 8], [8, 8], f, c) as ctx:
        ctx.set_graph(rp, (ci // (n_table // n)).astype(np.int32))
        assert ctx.layer_path(0) == A.PATH_FAST
    with pkg.GatContext([3, 2], [8, 8], f, c) as ctx:
        ctx.set_graph(rp, (ci // (n_table // n)).astype(np.int32))
        assert ctx.layer_path(0) == A.PATH_GENERIC_SHAPE


def test_replicated_input_shard_with_long_k_projects_both_halves(pkg):
    """n_rows = 500 takes the split-K kernel (N = H*D columns, K = 200), the 33,000-row table half does not: the scratch
    must hold the former although it is the smaller launch (gat_abi.hip:348)."""
    A = pkg.abi
    rng = np.random.default_rng(6)
    n_table, n, f, c = 33000, 500, 200, 4
    rp, ci = small_graph(rng, n, 3000, n_src=n_table)
    xt = rng.standard_normal((n_table, f)).astype(np.float32)
    with pkg.GatContext([8], [8], f, c) as ctx:
        ctx.set_graph(rp, ci, n_table=n_table, table_row0=1024)
        ctx.set_source_features(xt)
        ctx.set_labels(rng.integers(0, c, n).astype(np.int32))
        ctx.params_init(9)
        W = ctx.params_get(A.PARAM_W).reshape(64, 2 * f).astype(np.float64)
        ctx.layer_project(0)
        PL = ctx.tap(A.TAP_PL, 0); PR = ctx.tap(A.TAP_PR, 0)
        wantL = xt.astype(np.float64) @ W[:, :f].T
        wantR = xt[1024:1024 + n].astype(np.float64) @ W[:, f:].T
        assert np.abs(PL - wantL).max() < 1e-4 * np.abs(wantL).max()
        assert np.abs(PR - wantR).max() < 1e-4 * np.abs(wantR).max()
