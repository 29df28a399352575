"""One op-level entry point per reference kernel (gatv2_abi.h "per-kernel entry points", csrc/gat_ops.hip), each called
with caller-owned DEVICE buffers in the reference layouts and compared with the oracle function that restates that
kernel (oracle/gatv2_oracle.cpp orc_*), fed the oracle's own inputs for that stage — so a difference is the op's, not an
upstream stage's.  Shapes: the bench preset (H*D = 64), a generic (H, D) = (3, 4)/(2, 5), graphs with a hub row, empty
rows at both ends and E % 256 != 0.  Tolerances: the contract's (1e-4; gradients relative to max-abs, with
the LeakyReLU-kink bookkeeping of tests/parity.py where a stage takes sign decisions)."""
import ctypes as C

import numpy as np
import pytest

import parity
from conftest import small_graph
from parity import TOL, check_abs, check_rel

pytestmark = pytest.mark.gpu

SLOPE = 0.01
SHAPES = [((8, 8), (8, 8), 20, 130, 1111, (4, 300), (0, 129)),      # heads, outdims, F, N, E, hub, empty rows
          ((3, 2), (4, 5), 7, 70, 403, (2, 70), (0, 69)),
          ((8, 1), (8, 8), 40, 257, 2999, (100, 300), (3,))]


class Case:
    def __init__(self, orc, heads, outdims, f, n, e, hub, empty, seed=31):
        rng = np.random.default_rng(seed)
        self.rp, self.ci = small_graph(rng, n, e, hub=hub, empty=empty)
        self.n, self.e, self.f, self.c = n, len(self.ci), f, 4
        self.x = rng.standard_normal((n, f)).astype(np.float32)
        self.lab = rng.integers(0, self.c, n).astype(np.int32); self.lab[0] = self.c - 1
        self.cfg = orc.Config(list(heads), list(outdims), f, self.c)
        self.W, self.a, self.Wo = orc.xavier_params(self.cfg, seed + 1)
        self.ref = orc.step(self.cfg, self.rp, self.ci, self.lab, self.x, self.W, self.a, self.Wo)

    def layer(self, l):
        cfg = self.cfg
        H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
        Wl = self.W[cfg.w_offsets[l]:cfg.w_offsets[l + 1]]
        al = self.a[cfg.a_offsets[l]:cfg.a_offsets[l + 1]]
        X = self.x if l == 0 else self.ref.taps["H"][l - 1]
        return H, D, F, Wl, al, np.ascontiguousarray(X, np.float32)


@pytest.fixture(scope="module", params=range(len(SHAPES)), ids=lambda i: "heads%s_outdims%s" % SHAPES[i][:2])
def case(request, orc):
    return Case(orc, *SHAPES[request.param])


@pytest.fixture(scope="module")
def dev():
    import torch
    return torch.device("cuda:0")


def _t(arr, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to(dev)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


def _ok(lib, rc):
    assert rc == 0, lib.gat_last_error()


def test_edge_score_max_sum_coeff_aggregate_activation(pkg, orc, case, dev):
    """a2 - a6 of every layer, each fed the oracle's tensors of the stage before."""
    import torch
    lib = pkg.abi.load_library()
    n, e = case.n, case.e
    d_ci, d_dst, d_rp = _t(case.ci, dev), _t(case.ref.dst, dev), _t(case.rp, dev)
    for l in range(case.cfg.L):
        H, D, F, Wl, al, X = case.layer(l)
        last = l == case.cfg.L - 1
        d_x, d_w, d_a = _t(X, dev), _t(Wl, dev), _t(al, dev)
        score = torch.full((H * e,), 7.0, device=dev)
        _ok(lib, lib.gat_op_edge_score(_p(d_x), _p(d_ci), _p(d_dst), _p(d_w), _p(d_a), _p(score), n, F, D, H, e, C.c_float(SLOPE), None))
        check_abs(f"a2 score[{l}]", score.cpu().numpy().reshape(H, e), case.ref.taps["score"][l])
        r_score = _t(case.ref.taps["score"][l].reshape(-1), dev)
        mx = torch.empty(H * n, device=dev); sm = torch.empty(H * n, device=dev)
        _ok(lib, lib.gat_op_max_sum(_p(d_rp), _p(r_score), n, H, e, _p(mx), _p(sm), None))
        assert np.array_equal(mx.cpu().numpy().reshape(H, n), case.ref.taps["max"][l])          # a maximum: exact
        check_rel(f"a3 sum[{l}]", sm.cpu().numpy().reshape(H, n), case.ref.taps["sum"][l], 1e-5)
        r_mx, r_sm = _t(case.ref.taps["max"][l].reshape(-1), dev), _t(case.ref.taps["sum"][l].reshape(-1), dev)
        alpha = torch.empty(H * e, device=dev)
        _ok(lib, lib.gat_op_attn_coeff(_p(d_ci), _p(d_dst), _p(r_score), _p(r_mx), _p(r_sm), _p(alpha), e, H, n, None))
        check_abs(f"a4 alpha[{l}]", alpha.cpu().numpy().reshape(H, e), case.ref.taps["alpha"][l], 1e-6)
        r_alpha = _t(case.ref.taps["alpha"][l].reshape(-1), dev)
        hpre = torch.zeros(n * H * D, device=dev)                                                # the caller zeroes (Q1)
        _ok(lib, lib.gat_op_aggregate(_p(d_ci), _p(d_dst), _p(r_alpha), _p(d_x), _p(d_w), _p(hpre), n, H, e, F, D, None))
        want = case.ref.taps["hpre"][l]
        check_rel(f"a5 hpre[{l}]", hpre.cpu().numpy().reshape(n, H, D), want, TOL)
        deg = np.diff(case.rp)
        assert (hpre.cpu().numpy().reshape(n, -1)[deg == 0] == 0).all()                          # zero in-degree rows stay 0
        _ok(lib, lib.gat_op_aggregate(_p(d_ci), _p(d_dst), _p(r_alpha), _p(d_x), _p(d_w), _p(hpre), n, H, e, F, D, None))
        check_rel(f"a5 accumulates[{l}]", hpre.cpu().numpy().reshape(n, H, D), 2.0 * want, TOL)   # atomicAdd INTO the buffer (E:422)
        r_hpre = _t(want.reshape(-1), dev)
        out = torch.empty(n * (D if last else H * D), device=dev)
        _ok(lib, lib.gat_op_post_activation(_p(r_hpre), _p(out), n, H, D, int(last), C.c_float(SLOPE), None))
        check_abs(f"a6 H[{l}]", out.cpu().numpy().reshape(case.ref.taps["H"][l].shape), case.ref.taps["H"][l], 1e-6)


def test_output_head_loss_and_output_gradients(pkg, orc, case, dev):
    import torch
    lib = pkg.abi.load_library()
    Lb = orc.lib()
    n, Cn = case.n, case.c
    H, D = case.cfg.heads[-1], case.cfg.outdims[-1]
    HL = np.ascontiguousarray(case.ref.taps["H"][-1], np.float32)
    d_wo, d_HL = _t(case.Wo, dev), _t(HL, dev)
    z = torch.empty(n * Cn, device=dev); y = torch.empty(n * Cn, device=dev)
    _ok(lib, lib.gat_op_output_head(_p(d_wo), _p(d_HL), _p(z), _p(y), n, Cn, D, None))
    check_abs("C12 y", y.cpu().numpy().reshape(n, Cn), case.ref.y, 1e-6)
    assert torch.equal(z, y)                                                  # both hold the probabilities (E:502-509)
    r_y = _t(case.ref.y.reshape(-1), dev); d_lab = _t(case.lab, dev)
    loss = torch.empty(n, device=dev); corr = torch.empty(n, dtype=torch.int32, device=dev)
    _ok(lib, lib.gat_op_loss_accuracy(_p(r_y), _p(d_lab), _p(loss), _p(corr), n, Cn, None))
    want_l = np.empty(n, np.float32); want_c = np.empty(n, np.int32)
    Lb.orc_loss_accuracy(np.ascontiguousarray(case.ref.y.reshape(-1)), case.lab, want_l, want_c, n, Cn)
    check_abs("C13 loss", loss.cpu().numpy(), want_l, 1e-5)
    assert np.array_equal(corr.cpu().numpy(), want_c)
    hL = np.ascontiguousarray(case.ref.taps["hpre"][-1].reshape(-1), np.float32)
    d_hL = _t(hL, dev)
    for flat in (0, 1):
        gWo = torch.zeros(Cn * D, device=dev); ghL = torch.empty(n * H * D, device=dev)
        _ok(lib, lib.gat_op_output_gradients(_p(r_y), _p(d_lab), _p(d_hL), _p(d_HL), _p(d_wo), _p(gWo), _p(ghL), n, Cn, D, H, C.c_float(SLOPE), flat, None))
        want_wo = np.zeros(Cn * D, np.float32); want_g = np.zeros(n * H * D, np.float32)
        Lb.orc_output_gradients(np.ascontiguousarray(case.ref.y.reshape(-1)), case.lab, hL, HL.reshape(-1), case.Wo, want_wo, want_g, n, Cn, D, H,
                                SLOPE, flat)
        check_rel(f"C14 gradWo flat={flat}", gWo.cpu().numpy(), want_wo)
        check_rel(f"C14 g_L flat={flat}", ghL.cpu().numpy(), want_g)
        _ok(lib, lib.gat_op_output_gradients(_p(r_y), _p(d_lab), _p(d_hL), _p(d_HL), _p(d_wo), _p(gWo), _p(ghL), n, Cn, D, H, C.c_float(SLOPE), flat, None))
        check_rel(f"C14 gradWo accumulates flat={flat}", gWo.cpu().numpy(), 2.0 * want_wo)      # atomicAdd INTO grad_d_wo (E:581)


def test_backward_ops(pkg, orc, case, dev):
    """a7 - a11 of every layer on the oracle's upstream tensors.  a9 / a10 take LeakyReLU' decisions on fl32(PL + PR); a
    context on the same inputs runs the same projection kernel, so its taps tell whether any decision differs from the
    oracle's (none does for these seeds: asserted)."""
    import torch
    A = pkg.abi
    lib = A.load_library()
    n, e, cfg, ref = case.n, case.e, case.cfg, case.ref
    with pkg.GatContext(cfg.heads, cfg.outdims, case.f, case.c) as ctx:
        ctx.set_graph(case.rp, case.ci); ctx.set_features(case.x); ctx.set_labels(case.lab)
        for g, arr in enumerate((case.W, case.a, case.Wo)):
            ctx.params_set(g, arr)
        ctx.zero_grad(); ctx.step()
        fl = parity.find_flips(orc, cfg, case.rp, case.ci, case.x, case.W, ref, ctx, A)
    # every op below is fed the ORACLE's upstream tensors, so a flipped decision acts on one layer only — not what the
    # step-level correction of tests/parity.py models; the seeds are chosen so that no decision sits on a kink here
    assert fl.total == 0, fl.summary()
    exp = dict(gradW=ref.gradW, grada=ref.grada, gx=ref.taps["gx"])
    d_ci, d_dst, d_rp = _t(case.ci, dev), _t(ref.dst, dev), _t(case.rp, dev)
    for l in range(cfg.L - 1, -1, -1):
        H, D, F, Wl, al, X = case.layer(l)
        d_x, d_w, d_a = _t(X, dev), _t(Wl, dev), _t(al, dev)
        d_g = _t(ref.taps["g"][l].reshape(-1), dev)
        galpha = torch.empty(H * e, device=dev)
        _ok(lib, lib.gat_op_grad_attn_coeff(e, H, F, D, _p(d_ci), _p(d_dst), _p(d_x), _p(d_w), _p(d_g), _p(galpha), n, None))
        check_rel(f"a7 galpha[{l}]", galpha.cpu().numpy().reshape(H, e), ref.taps["galpha"][l])
        r_alpha = _t(ref.taps["alpha"][l].reshape(-1), dev); r_galpha = _t(ref.taps["galpha"][l].reshape(-1), dev)
        ge = torch.empty(H * e, device=dev)
        _ok(lib, lib.gat_op_grad_attn_score(_p(d_rp), _p(d_dst), _p(r_alpha), _p(r_galpha), _p(ge), n, H, e, None))
        check_rel(f"a8 ge[{l}]", ge.cpu().numpy().reshape(H, e), ref.taps["ge"][l])
        r_ge = _t(ref.taps["ge"][l].reshape(-1), dev)
        gw = torch.zeros(H * D * 2 * F, device=dev); ga = torch.zeros(H * D, device=dev)
        _ok(lib, lib.gat_op_grad_parameters(e, H, _p(d_ci), _p(d_dst), _p(d_x), _p(d_g), _p(r_ge), _p(r_alpha), _p(d_w), _p(d_a), _p(gw), _p(ga),
                                            F, D, C.c_float(SLOPE), n, None))
        scale = float(np.abs(exp["gradW"]).max())
        check_rel(f"a9 gradW[{l}]", gw.cpu().numpy(), exp["gradW"][cfg.w_offsets[l]:cfg.w_offsets[l + 1]], floor=1e-3 * scale)
        check_rel(f"a9 grada[{l}]", ga.cpu().numpy(), exp["grada"][cfg.a_offsets[l]:cfg.a_offsets[l + 1]], floor=1e-3 * scale)
        if l == 0:
            break                                                             # E:1528
        gx = torch.zeros(n * F, device=dev)
        _ok(lib, lib.gat_op_features_input_gradients(n, H, e, F, D, C.c_float(SLOPE), _p(d_ci), _p(d_dst), _p(r_alpha), _p(d_x), _p(d_w), _p(d_g),
                                                     _p(r_ge), _p(d_a), _p(gx), None))
        check_rel(f"a10 gx[{l}]", gx.cpu().numpy().reshape(n, F), exp["gx"][l])
        r_gx = _t(np.ascontiguousarray(ref.taps["gx"][l], np.float32).reshape(-1), dev)
        r_hp = _t(ref.taps["hpre"][l - 1].reshape(-1), dev)
        _ok(lib, lib.gat_op_preact_gradient(n, C.c_float(SLOPE), F, _p(r_hp), _p(r_gx), None))
        check_rel(f"a11 g[{l - 1}]", r_gx.cpu().numpy().reshape(ref.taps["g"][l - 1].shape), ref.taps["g"][l - 1], 1e-6)


def test_ops_refuse_bad_arguments_and_take_empty_graphs(pkg, dev):
    import torch
    A = pkg.abi
    lib = A.load_library()
    assert lib.gat_op_max_sum(None, None, 10, 8, 0, None, None, None) != 0
    assert "null" in lib.gat_last_error().decode()
    assert lib.gat_op_post_activation(None, None, 10, 8, 8, 0, C.c_float(SLOPE), None) != 0
    # a graph without edges: every row has max = -1e9f, sum = 0 (E:336), nothing else is touched
    n, H = 9, 4
    rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    mx = torch.empty(H * n, device=dev); sm = torch.empty(H * n, device=dev)
    _ok(lib, lib.gat_op_max_sum(_p(rp), None, n, H, 0, _p(mx), _p(sm), None))
    assert (mx.cpu().numpy() == np.float32(-1e9)).all() and (sm.cpu().numpy() == 0).all()
