"""CPU tests of the parity bookkeeping itself (tests/parity.py, tests/ref64.py, orc_presum): the fp64
masked reference must be a correct backward (== torch autograd of an independent forward), must follow the
oracle when given the oracle's sign decisions, and a flipped LeakyReLU' decision must move the gradients by
exactly the analytic jump."""
import numpy as np
import pytest
import torch

import parity
import ref64
import torch_ref
from conftest import small_graph


def _case(orc, seed=3, n=40, e=200, heads=(4, 2), outdims=(4, 8), f=7, c=5):
    rng = np.random.default_rng(seed)
    rp, ci = small_graph(rng, n, e, hub=(3, 70), empty=(0, 11))
    x = rng.standard_normal((n, f)).astype(np.float32)
    lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
    cfg = orc.Config(list(heads), list(outdims), f, c)
    W, a, Wo = orc.xavier_params(cfg, seed + 1)
    return cfg, rp, ci, lab, x, W, a, Wo


def _natural_masks(fw):
    ms = [y["s"] > 0 for y in fw["layers"]]
    mh = [y["hpre"] > 0 for y in fw["layers"]]
    return ms, mh


@pytest.mark.parametrize("heads,outdims", [((4, 2), (4, 8)), ((8, 8, 1), (8, 8, 4)), ((3, 1), (4, 5))])
def test_ref64_is_the_autograd_gradient(orc, heads, outdims):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, heads=heads, outdims=outdims)
    fw = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)
    ms, mh = _natural_masks(fw)
    b = ref64.backward(cfg, fw, ms, ms, mh)
    t = torch_ref.forward(cfg, rp, ci, lab, x, W, a, Wo)
    t["loss"].backward()
    # the reference's backward drops its own +1e-8 epsilons: agreement to ~1e-7, not 1e-15
    for name, got, want in (("W", b["gradW"], t["W"].grad), ("a", b["grada"], t["a"].grad), ("Wo", b["gradWo"], t["Wo"].grad)):
        want = want.numpy()
        assert np.abs(got - want).max() <= 1e-6 * max(1e-3, np.abs(want).max()), name
    for l in range(cfg.L):
        assert np.abs(fw["layers"][l]["alpha"].T - t["alpha"][l].detach().numpy()).max() < 1e-12


def test_ref64_follows_the_oracle_with_its_masks(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, n=60, e=400, heads=(8, 8), outdims=(8, 8), f=12)
    ref = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
    lr, il = orc.presum_signs(cfg, rp, ci, x, W, ref)
    fw = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)
    # the oracle's fp32 sign decisions agree with the fp64 signs except within round-off of 0
    for l in range(cfg.L):
        s = fw["layers"][l]["s"]
        for m in (lr[l], il[l]):
            bad = m != (s > 0)
            assert bad.sum() <= 4 and (np.abs(s[bad]) < 1e-5).all()
    b = ref64.backward(cfg, fw, lr, il, [ref.taps["hpre"][l] > 0 for l in range(cfg.L)])
    for name, got, want in (("gradW", b["gradW"], ref.gradW), ("grada", b["grada"], ref.grada), ("gradWo", b["gradWo"], ref.gradWo)):
        assert parity.rel_err(want, got) < 1e-5, name
    for l in range(cfg.L):
        for k in ("g", "galpha", "ge"):
            assert parity.rel_err(ref.taps[k][l], b[k][l].reshape(ref.taps[k][l].shape)) < 1e-5, (k, l)
    assert parity.rel_err(ref.taps["gx"][1], b["gx"][1]) < 1e-5


def test_one_flipped_decision_moves_gradw_by_the_analytic_jump(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, heads=(2, 1), outdims=(4, 4), f=5)
    fw = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)
    ms, mh = _natural_masks(fw)
    b0 = ref64.backward(cfg, fw, ms, ms, mh)
    e, h, k, l = 17, 1, 2, 0
    m2 = [m.copy() for m in ms]
    m2[l][e, h, k] = ~m2[l][e, h, k]
    b1 = ref64.backward(cfg, fw, m2, ms, mh)                 # parameter-gradient path only
    H, D, F = cfg.heads[l], cfg.outdims[l], cfg.in_dims[l]
    ge = b0["ge"][l][h, e]
    al = fw["layers"][l]["al"][h, k]
    sign = 1.0 if m2[l][e, h, k] else -1.0                   # slope -> 1 (+) or 1 -> slope (-)
    jump = sign * ge * al * (1.0 - ref64.SLOPE)
    d = (b1["gradW"] - b0["gradW"])[cfg.w_offsets[l]:cfg.w_offsets[l + 1]].reshape(H, D, 2 * F)
    want = np.zeros_like(d)
    want[h, k, :F] = jump * fw["layers"][l]["x"][fw["src"][e]]
    want[h, k, F:] = jump * fw["layers"][l]["x"][fw["dst"][e]]
    assert np.abs(d - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    assert np.array_equal(b1["grada"], b0["grada"]) and np.array_equal(b1["gradWo"], b0["gradWo"])


def test_expected_gradients_without_flips_is_the_oracle(orc):
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc)
    ref = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
    lr, il = orc.presum_signs(cfg, rp, ci, x, W, ref)
    oh = [ref.taps["hpre"][l] > 0 for l in range(cfg.L)]
    fl = parity.Flips(cfg, lr, oh, lr, lr, oh)
    assert fl.total == 0
    exp = parity.expected_gradients(cfg, rp, ci, lab, x, W, a, Wo, ref, fl)
    assert exp["gradW"] is ref.gradW
    # one hidden-layer h_pre decision flipped on the "HIP" side: the correction is finite and confined to what layer 0 feeds
    gh = [m.copy() for m in oh]
    gh[0][5, 1, 2] = ~gh[0][5, 1, 2]
    fl = parity.Flips(cfg, lr, gh, lr, lr, oh)
    assert fl.total == 1
    exp = parity.expected_gradients(cfg, rp, ci, lab, x, W, a, Wo, ref, fl)
    assert np.array_equal(exp["gradWo"], ref.gradWo)                        # the head does not see layer 0's kink
    assert np.array_equal(exp["g"][1], ref.taps["g"][1])
    d = exp["g"][0] - ref.taps["g"][0]
    assert np.count_nonzero(d) == 1 and d[5, 1, 2] != 0


def test_forced_decisions_in_the_restructured_checker(orc):
    """orc_step_restructured(decisions=...): its own decisions packed and fed back change nothing and report no
    flips; one flipped s-decision changes the parameter gradients by exactly what tests/ref64.py computes for that
    flip (the two kink mechanisms agree), and is reported with the |s| it happened at."""
    cfg, rp, ci, lab, x, W, a, Wo = _case(orc, n=60, e=400, heads=(8, 8), outdims=(8, 8), f=12)
    base = orc.step_restructured(cfg, rp, ci, lab, x, W, a, Wo, acc64=True)
    fw = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)           # fp64 twin: same signs except within round-off of 0
    PLs = [y["PL"].reshape(len(rp) - 1, -1).astype(np.float32) for y in fw["layers"]]
    PRs = [y["PR"].reshape(len(rp) - 1, -1).astype(np.float32) for y in fw["layers"]]
    hps = [y["hpre"].astype(np.float32) for y in fw["layers"]]
    sb, hb = parity.pack_decisions(cfg, rp, ci, PLs, PRs, hps, chunk=64)
    same = orc.step_restructured(cfg, rp, ci, lab, x, W, a, Wo, acc64=True, decisions=(sb, hb))
    cnt, mx = same[5]
    assert cnt.sum() <= 2 and (mx < 1e-5).all()                 # fp64-vs-fp32 sign differences, if any, sit at the kink
    if cnt.sum() == 0:
        for u, v in zip(base[2:5], same[2:5]):
            assert np.array_equal(u, v)
    # flip one layer-1 decision: edge 17, channel 9
    e, c, l = 17, 9, 1
    HD0 = cfg.heads[0] * cfg.outdims[0]
    off = (len(ci) * HD0 + 7) // 8 * 8 + e * 64 + c             # bit index inside the concatenated block
    sb2 = sb.copy(); sb2[off >> 3] ^= np.uint8(1 << (off & 7))
    flipped = orc.step_restructured(cfg, rp, ci, lab, x, W, a, Wo, acc64=True, decisions=(sb2, hb))
    assert flipped[5][0][1, 0] == cnt[1, 0] + 1
    ms = [y["s"] > 0 for y in fw["layers"]]; mh = [y["hpre"] > 0 for y in fw["layers"]]
    m2 = [m.copy() for m in ms]; m2[l][e, c // 8, c % 8] ^= True
    d64 = ref64.backward(cfg, fw, m2, m2, mh)
    b64 = ref64.backward(cfg, fw, ms, ms, mh)
    for k, i in (("gradW", 2), ("grada", 3), ("gradWo", 4)):
        want = d64[k] - b64[k]
        got = flipped[i].astype(np.float64) - same[i].astype(np.float64)
        assert np.abs(got - want).max() <= 2e-5 * max(1e-9, np.abs(b64[k]).max()), k
    assert np.abs(d64["gradW"] - b64["gradW"]).max() > 0
