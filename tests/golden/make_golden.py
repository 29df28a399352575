#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs + expected outputs of one forward+backward step on tiny
graphs, computed by the CPU oracle (oracle/gatv2_oracle.cpp, the restatement of
GATv2_edge_based.cu pinned by autograd — the reference itself ships no vectors and cannot run
here).  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as entry  # noqa: E402
from conftest import small_graph  # noqa: E402

CASES = {
    # name: (seed, N, E, heads, outdims, F, C, hub, empty)
    "tiny_2x8h": (101, 12, 23, (8, 8), (8, 8), 5, 3, None, (0, 7)),                  # SURVEY's 12-node/23-edge case
    "hub_8h1h": (102, 40, 417, (8, 1), (8, 8), 16, 4, (3, 300), (0, 39)),             # hub > 256, E % 64 != 0
    "generic_3h": (103, 30, 131, (3, 1), (4, 8), 7, 5, (2, 70), ()),                  # H=3: generic kernels
}


def main():
    orc = entry.load_oracle()
    for name, (seed, n, e, heads, outdims, f, c, hub, empty) in CASES.items():
        rng = np.random.default_rng(seed)
        rp, ci = small_graph(rng, n, e, hub=hub, empty=empty)
        x = rng.standard_normal((n, f)).astype(np.float32)
        lab = rng.integers(0, c, n).astype(np.int32); lab[0] = c - 1
        cfg = orc.Config(list(heads), list(outdims), f, c)
        W, a, Wo = orc.xavier_params(cfg, seed)
        r = orc.step(cfg, rp, ci, lab, x, W, a, Wo)
        out = dict(row_ptr=rp, col_idx=ci, x=x, labels=lab, W=W, a=a, Wo=Wo, heads=np.array(heads), outdims=np.array(outdims),
                   num_classes=c, src=r.src, dst=r.dst, y=r.y, loss_sum=np.float64(r.loss_sum_f64), n_correct=r.n_correct,
                   gradW=r.gradW, grada=r.grada, gradWo=r.gradWo)
        for l in range(cfg.L):
            out[f"alpha{l}"] = r.taps["alpha"][l]; out[f"hpre{l}"] = r.taps["hpre"][l]; out[f"H{l}"] = r.taps["H"][l]
            out[f"g{l}"] = r.taps["g"][l]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "loss/N", r.loss_sum_f64 / n)


if __name__ == "__main__":
    main()
