"""Randomised small cases for the training-path kernels (group-per-row forward / backward, per-edge records, pull pass
in both forms): shapes with and without a record path, graphs with empty rows at both ends, hub rows beyond one
segment, hub SOURCES beyond the chunking threshold, item counts that are not a multiple of the groups per wave, very
few nodes.  Every case is held to the oracle at the contract's tolerances with the kink bookkeeping of tests/parity.py."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = """
    import sys, numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    import __graft_entry__ as entry
    import parity, ref64
    pkg = entry.load_package(); orc = entry.load_oracle(); A = pkg.abi
    SHAPES = [((8, 8), (8, 8)), ((4, 4), (8, 8)), ((8, 1), (8, 8)), ((2, 4), (4, 8)), ((16, 8), (4, 4)), ((4, 2), (16, 8)),
              ((1, 8), (8, 4)), ((8, 8, 8), (8, 4, 8)), ((3, 2), (8, 8))]
    rng = np.random.default_rng({seed})
    skipped = ran = draws = 0
    while ran < {cases} and draws < 4 * {cases}:       # an ill-conditioned draw (below) is REDRAWN: {cases} cases are checked per setting
        case = draws
        draws += 1
        heads, outdims = SHAPES[rng.integers(len(SHAPES))]
        n = int(rng.choice([1, 2, 3, 5, 17, 64, 130, 257]))
        f = int(rng.choice([1, 3, 5, 8, 20, 33, 130, 257]))     # odd widths: the padded feature pitch; > 128: the split-K projection
        c = int(rng.integers(2, 6))
        deg = rng.integers(0, 9, n)
        if n > 4 and rng.random() < 0.7:
            deg[rng.integers(n)] = int(rng.choice([70, 300, 700]))          # hub row (beyond one 64-edge segment on small graphs)
        if rng.random() < 0.5:
            deg[0] = 0
        if rng.random() < 0.5:
            deg[-1] = 0
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        hot = int(rng.integers(n))                                           # a hub SOURCE: long slot list in the pull pass
        cols = []
        for d in deg:
            s = rng.integers(0, n, int(d))
            s[rng.random(int(d)) < 0.4] = hot
            cols.append(np.sort(s))
        ci = (np.concatenate(cols) if rp[-1] else np.zeros(0)).astype(np.int32)
        x = rng.standard_normal((n, f)).astype(np.float32)
        lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
        cfg = orc.Config(list(heads), list(outdims), f, c)
        W, a, Wo = orc.xavier_params(cfg, int(rng.integers(1 << 30)))
        ref = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
        # conditioning: on a few of these tiny graphs a gradient tensor is a near-total cancellation (e.g. grad_a when every
        # row's attention is almost one-hot) and the fp32 ORACLE itself is off by a good part of the tolerance against the
        # same formulas in fp64 (same LeakyReLU decisions) — such a case says nothing about the kernels and is skipped
        lr_, il_ = orc.presum_signs(cfg, rp, ci, x, W, ref)
        fw64 = ref64.forward(cfg, rp, ci, lab, x, W, a, Wo)
        b64 = ref64.backward(cfg, fw64, lr_, il_, [ref.taps["hpre"][l] > 0 for l in range(cfg.L)])
        cond = max(parity.rel_err(getattr(ref, k), b64[k]) for k in ("gradW", "grada", "gradWo"))
        if cond > 2e-5:
            skipped += 1
            continue
        ran += 1
        parity.set_test("fuzz{tag} case %d: heads %s outdims %s n %d e %d" % (case, heads, outdims, n, len(ci)))
        with pkg.GatContext(cfg.heads, cfg.outdims, f, c) as ctx:
            ctx.set_graph(rp, ci); ctx.set_features(x); ctx.set_labels(lab)
            for g, arr in enumerate((W, a, Wo)):
                ctx.params_set(g, arr)
            ctx.zero_grad()
            loss, correct = ctx.step()
            parity.check_abs("loss/N", loss / n, ref.loss_sum_f64 / n)
            assert correct == ref.n_correct, (case, correct, ref.n_correct)
            for l in range(cfg.L):
                parity.check_rel("hpre[%d]" % l, ctx.tap(A.TAP_HPRE, l), ref.taps["hpre"][l], 1e-4, floor=1e-6)
            scale = float(max(np.abs(ref.gradW).max(), 1e-20))
            parity.check_context_gradients(orc, A, cfg, rp, ci, lab, x, W, a, Wo, ref, ctx, floor=1e-3 * scale)
    parity.flush()
    assert ran == {cases}, (ran, draws, skipped)           # every setting checks the full number of well-conditioned cases
    assert skipped <= max(draws // 3, 2), (skipped, draws)  # a statement about the DRAW: ill-conditioned graphs must stay the exception
    print("OK ran", ran, "draws", draws, "skipped", skipped)
"""


SETTINGS = [({}, ""), ({"GAT_PULL_GROUPS": "1", "GAT_GPL_HEAVY": "16", "GAT_SEG_EDGES": "16"}, " groups, tiny segments"),
            ({"GAT_PULL_GROUPS": "0", "GAT_PULL_LAST": "0"}, " waves"), ({"GAT_ROWGROUP": "0"}, " chunked"),
            ({"GAT_PULL_LAST": "1", "GAT_PULL_GROUPS": "1", "GAT_GPL_HEAVY": "16", "GAT_SEG_EDGES": "16"}, " last-layer records, groups"),
            ({"GAT_PULL_LAST": "1", "GAT_PULL_GROUPS": "0", "GAT_ROWGROUP": "0"}, " last-layer records, waves, chunked"),
            ({"GAT_FUSE_LAST": "1", "GAT_PULL_LAST": "1"}, " last layer fused per row"),
            # slot-parallel source-major pass (gat_csc.hip "runs") forced, gfull and node-record variants, short runs
            ({"GAT_PULL_RUNS": "1", "GAT_PULL_RUN": "32", "GAT_PULL_LAST": "0"}, " slot runs of 32"),
            ({"GAT_PULL_RUNS": "1", "GAT_PULL_LAST": "1", "GAT_SEG_EDGES": "16"}, " slot runs, last-layer records"),
            ({"GAT_PULL_RUNS": "1", "GAT_BWD_STASH": "0"}, " slot runs, message rows")]
CASES = 150


@pytest.fixture(scope="module")
def fuzz_runs():
    """All settings at once, five subprocesses at a time (tests/conftest.py run_snippets_parallel): each is ~35 s of CPU oracle +
    fp64 reference around a few milliseconds of GPU work."""
    from conftest import run_snippets_parallel
    jobs = {}
    for env, tag in SETTINGS:
        code = textwrap.dedent(SNIPPET.format(root=ROOT, tests=os.path.join(ROOT, "tests"), seed=20260 + len(tag), cases=CASES, tag=tag))
        jobs[tag] = (code, env)
    return run_snippets_parallel(jobs)


# 10 switch settings x 150 checked cases (ill-conditioned draws are redrawn, not forgiven: VERDICT r3 / ADVICE r3); a setting is ~10 s of
# oracle + fp64 reference with a team of four threads — it was ~38 s for 9 cases while the oracle ran one thread per visible CPU
@pytest.mark.parametrize("env,tag", SETTINGS)
def test_random_small_cases(fuzz_runs, env, tag):
    out = fuzz_runs[tag]
    assert out.returncode == 0 and "OK ran %d" % CASES in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
