"""seeded Xavier init on the device with the reference's distribution (E:205-242), and train /
validation masks (the reference trains and evaluates on all nodes, README R:134 "later")."""
import numpy as np
import pytest

import parity
import ref64
from conftest import small_graph

pytestmark = pytest.mark.gpu


def test_params_init_distribution_per_group(pkg):
    """W rows and a of layer l: U(-lim, lim] with lim = sqrt(6/(2F+D)) (E:208-229; fan_in 2F, fan_out D);
    W_o: sqrt(6/(C+D_L)) (E:236); curand_uniform is (0, 1] => values in (-lim, lim] (E:217-218)."""
    A = pkg.abi
    heads, outdims, f, c = [8, 4, 2], [8, 16, 4], 300, 11
    with pkg.GatContext(heads, outdims, f, c) as ctx:
        ctx.params_init(1234)
        W, a, Wo = ctx.params_get(A.PARAM_W), ctx.params_get(A.PARAM_A), ctx.params_get(A.PARAM_WO)
        ctx.params_init(1234)
        assert np.array_equal(W, ctx.params_get(A.PARAM_W))                     # seeded: reproducible
        ctx.params_init(1235)
        assert not np.array_equal(W, ctx.params_get(A.PARAM_W))
    in_dims = [f, heads[0] * outdims[0], heads[1] * outdims[1]]
    wo, ao = 0, 0
    for l in range(3):
        H, D, F = heads[l], outdims[l], in_dims[l]
        lim = np.float32(np.sqrt(np.float32(6.0) / np.float32(2 * F + D)))
        Wl = W[wo:wo + H * D * 2 * F]; al = a[ao:ao + H * D]
        wo += H * D * 2 * F; ao += H * D
        for name, v in ((f"W{l}", Wl), (f"a{l}", al)):
            assert v.max() <= lim and v.min() > -lim, (name, v.min(), v.max(), lim)
            if v.size > 2000:                       # uniform: mean 0, variance lim^2/3, the range is actually used
                assert abs(v.mean()) < 4 * lim / np.sqrt(3 * v.size)
                assert abs(v.var() / (lim * lim / 3) - 1) < 0.1
                assert v.max() > 0.99 * lim and v.min() < -0.99 * lim
    limo = np.float32(np.sqrt(np.float32(6.0) / np.float32(c + outdims[-1])))
    assert Wo.max() <= limo and Wo.min() > -limo
    # all three groups come from ONE stream without overlap: no value sequence of W shows up again in a / W_o
    assert not np.isin(a, W[:100000]).all()


def test_train_mask_restricts_loss_and_gradient_and_eval_mask_reports_a_split(pkg, orc):
    A = pkg.abi
    rng = np.random.default_rng(21)
    n, f, c = 220, 14, 5
    rp, ci = small_graph(rng, n, 2600, hub=(9, 300), empty=(0, 7))
    x = rng.standard_normal((n, f)).astype(np.float32)
    lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
    train = rng.random(n) < 0.6
    val = ~train & (rng.random(n) < 0.5)
    cfg = orc.Config([8, 8], [8, 8], f, c)
    W, a, Wo = orc.xavier_params(cfg, 3)
    fw = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)
    y = fw["y"]
    nll = -np.log(np.maximum(y[np.arange(n), lab], 1e-12))
    pred = y.argmax(1)
    for keep_taps in (False, True):
        with pkg.GatContext(cfg.heads, cfg.outdims, f, c, keep_taps=keep_taps) as ctx:
            ctx.set_graph(rp, ci); ctx.set_features(x); ctx.set_labels(lab)
            for g, arr in enumerate((W, a, Wo)):
                ctx.params_set(g, arr)
            ctx.set_train_mask(train)
            ctx.zero_grad()
            loss, correct = ctx.step() if not keep_taps else ctx.forward()
            if keep_taps:
                ctx.backward()
            parity.check_abs(f"taps={int(keep_taps)} masked loss/N", loss / train.sum(), nll[train].sum() / train.sum())
            assert correct == int((pred[train] == lab[train]).sum())
            vl, vc, vn = ctx.eval_mask(val)
            assert vn == int(val.sum()) and vc == int((pred[val] == lab[val]).sum())
            parity.check_abs("val loss/N", vl / max(vn, 1), nll[val].sum() / max(vn, 1))
            # gradients: fp64 reference with dz = 0 outside the mask, evaluated with the HIP path's own LeakyReLU' decisions
            src = ci.astype(np.int64); dst = fw["dst"]
            gs, gh = [], []
            for l in range(2):
                s_ = (ctx.tap(A.TAP_PL, l)[src] + ctx.tap(A.TAP_PR, l)[dst]).astype(np.float32)
                gs.append((s_ > 0).reshape(len(src), 8, 8)); gh.append(ctx.tap(A.TAP_HPRE, l) > 0)
            b = ref64.backward(cfg, fw, gs, gs, gh, node_mask=train)
            for grp, k in ((A.PARAM_WO, "gradWo"), (A.PARAM_A, "grada"), (A.PARAM_W, "gradW")):
                parity.check_rel(f"taps={int(keep_taps)} masked {k}", ctx.grads_get(grp), b[k])
            # unmasking restores the reference's behaviour
            ctx.set_train_mask(None)
            loss_all, correct_all = ctx.forward()
            parity.check_abs("unmasked loss/N", loss_all / n, nll.sum() / n)
            assert correct_all == int((pred == lab).sum())
