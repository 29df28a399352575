"""The C++ host `train_edge` (drop-in for the reference binary): CLI contract on CPU, and with
-m gpu whole epochs (forward, loss line, backward, clip, optimizer) against the oracle driven
through the same sequence (E:1370-1642)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "graph-attention-network-gatv2-_amd", "train_edge")


def run(args, env=None):
    e = dict(os.environ)
    e.pop("DATA_ROOT", None)
    if env:
        e.update(env)
    return subprocess.run([BIN] + args, capture_output=True, text=True, env=e, timeout=600)


def test_binary_exists():
    assert os.path.exists(BIN), "run __graft_entry__.build()"


@pytest.mark.parametrize("args,msg", [
    (["--num-layers", "0"], "Error: Number of layers must be > 0\n"),
    (["--num-layers", "2", "--heads", "8"], "Error: --heads must have 2 values.\n"),
    (["--num-layers", "2", "--heads", "8,8", "--outdims", "8"], "Error: --ooutdims must have 2 values.\n"),
    (["--heads", "8,8", "--outdims", "8,8", "--optimizer", "rmsprop"], "Invalid optimizer choice. Use 'sgd' or 'adam'\n"),
    (["--heads", "8,8", "--outdims", "8,8", "--optimizer", "adam", "--beta1", "1.5"],
     "Error: For Adam optimizer, beta1 and beta2 must be in (0,1).\n"),
    (["--heads", "8,8"], "Error: --ooutdims must have 2 values.\n"),        # no defaults exist
])
def test_cli_errors(args, msg):
    r = run(args)
    assert r.returncode == 1
    assert r.stderr.endswith(msg), r.stderr
    # the reference prints its "[Memory Tracker] Before allocation" block before it looks at argv (E:929-933 vs
    # E:943): the argument-error exits still show it
    assert r.stdout.startswith("\n[Memory Tracker] Before allocation:\n  Total GPU memory: "), r.stdout
    assert "Configuration:" not in r.stdout


def test_cli_config_echo_and_missing_dataset(tmp_path):
    r = run(["--num-layers", "3", "--heads", "4,1,1", "--outdims", "64,32,16", "--epochs", "20", "--optimizer", "adam",
             "--lr", "0.01", "--clip", "--dataset", "citeseer", "--data-root", str(tmp_path)])
    assert ("Configuration:\n  Number of layers: 3\n  Epochs: 20\n  Attention heads: [4, 1, 1]\n"
            "  Output dimensions: [64, 32, 16]\n  Gradient clipping: true\n  Optimizer: adam\n"
            "  Learning rate: 0.01\n\n") in r.stdout
    assert f"Using dataset: citeseer\nDataset path: {tmp_path}/citeseer/\n" in r.stdout
    assert r.returncode == 1 and "Invalid row_ptr length\n" in r.stderr       # unopened files -> length 0 (E:1084)
    r = run(["--heads", "8,8", "--outdims", "8,8", "--beta1", "0.5"], env={"DATA_ROOT": str(tmp_path)})
    assert "Warning: beta1/beta2 specified but ignored for SGD optimizer.\n" in r.stderr
    assert f"Dataset path: {tmp_path}/pubmed/\n" in r.stdout                   # env DATA_ROOT + default dataset


def test_inconsistent_feature_rows(tmp_path):
    d = tmp_path / "bad"
    d.mkdir()
    (d / "features.txt").write_text("1 2 3\n4 5\n")
    for f in ("row_ptr.txt", "col_idx.txt", "labels.txt"):
        (d / f).write_text("0\n")
    r = run(["--heads", "8,8", "--outdims", "8,8", "--dataset", "bad", "--data-root", str(tmp_path)])
    assert r.returncode == 1 and "Inconsistent input_dim on line 1" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer,clip", [("adam", True), ("sgd", False)])
def test_epochs_match_oracle(pkg, orc, tmp_path, optimizer, clip):
    ds = pkg.synth.make_dataset("cora", scale=0.15)
    pkg.synth.write_text_dataset(ds, str(tmp_path), "tiny")
    # the loaders parse text: use exactly what the binary will read
    x = np.loadtxt(tmp_path / "tiny" / "features.txt", dtype=np.float32, ndmin=2)
    heads, outdims = [8, 8], [8, 8]
    cfg = orc.Config(heads, outdims, ds["f"], ds["c"])
    W, a, Wo = orc.xavier_params(cfg, 7)
    pfile = tmp_path / "p.bin"
    np.concatenate([W, a, Wo]).astype(np.float32).tofile(pfile)
    epochs, lr = 3, (0.01 if optimizer == "adam" else 1e-3)
    args = ["--dataset", "tiny", "--data-root", str(tmp_path), "--num-layers", "2", "--heads", "8,8", "--outdims", "8,8",
            "--epochs", str(epochs), "--optimizer", optimizer, "--lr", str(lr), "--load-params", str(pfile),
            "--dump-params", str(tmp_path / "out.bin")] + (["--clip"] if clip else [])
    r = run(args)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert f"Graph loaded: {ds['n']} nodes, {ds['e']} edges, input_feature_vector_dim = {ds['f']}\n" in out
    assert f"Number of classes = {ds['c']}\n" in out and f"Max degree = {int(np.diff(ds['row_ptr']).max())}\n" in out
    assert "[Memory Tracker] After all allocations:" in out and out.count(" total time: ") == epochs
    got = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(r"Avg Loss: ([0-9.]+), Accuracy: ([0-9.]+)%", out)]
    assert len(got) == epochs
    # the oracle through the same epoch loop (intended semantics: zeroed h_pre each pass)
    L = orc.lib()
    ms = [np.zeros_like(p) for p in (W, a, Wo)]; vs = [np.zeros_like(p) for p in (W, a, Wo)]
    params = [W.copy(), a.copy(), Wo.copy()]
    for ep in range(1, epochs + 1):
        ref = orc.step(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], x, *params)
        assert abs(got[ep - 1][0] - ref.loss_sum_f32 / ds["n"]) < 1e-4, (ep, got[ep - 1], ref.loss_sum_f32 / ds["n"])
        assert abs(got[ep - 1][1] - 100.0 * ref.n_correct / ds["n"]) < 0.011
        grads = [ref.gradW, ref.grada, ref.gradWo]
        for i in range(3):
            if clip:
                L.orc_clip_grad_norm(grads[i], grads[i].size, 5.0)
            if optimizer == "adam":
                L.orc_adam(params[i], grads[i], ms[i], vs[i], lr, params[i].size, 0.9, 0.999, 1e-8, ep)
            else:
                L.orc_sgd(params[i], grads[i], lr, params[i].size)
    final = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    want = np.concatenate(params)
    if optimizer == "sgd":
        assert np.abs(final - want).max() < 1e-4
    else:   # Adam's first steps are ~lr*sign(g): an entry whose gradient is ~0 may step the other way
        assert (np.abs(final - want) > 5e-3).mean() < 0.005


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_match_single_process(pkg, tmp_path, ranks):
    """train_edge --ranks P: P forked processes, destination-range shards, exchanges inside the library.
    The box has one GPU, so the ranks share it and the exchanges use the host-staged transport (RCCL
    refuses two ranks on one device); the sharded epochs must print the same loss/accuracy lines and
    end at the same parameters as the single-process run (sgd: no sign-flip sensitivity)."""
    ds = pkg.synth.make_dataset("cora", scale=0.15)
    pkg.synth.write_text_dataset(ds, str(tmp_path), "tiny")
    base = ["--dataset", "tiny", "--data-root", str(tmp_path), "--num-layers", "2", "--heads", "8,8", "--outdims", "8,8",
            "--epochs", "3", "--optimizer", "sgd", "--lr", "0.001", "--seed", "5", "--clip"]
    one = run(base + ["--dump-params", str(tmp_path / "p1.bin")])
    assert one.returncode == 0, one.stderr
    many = run(base + ["--ranks", str(ranks), "--transport", "host", "--dump-params", str(tmp_path / "pN.bin")])
    assert many.returncode == 0, many.stderr
    pat = r"Avg Loss: ([0-9.]+), Accuracy: ([0-9.]+)%"
    a = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(pat, one.stdout)]
    b = [(float(m.group(1)), float(m.group(2))) for m in re.finditer(pat, many.stdout)]
    assert len(a) == 3 and len(b) == 3                       # rank 0 prints, the others are silent
    for (la, aa), (lb, ab) in zip(a, b):
        assert abs(la - lb) < 1e-4 and abs(aa - ab) < 0.011
    assert many.stdout.count("Graph loaded:") == 1 and many.stdout.count(" total time: ") == 3
    p1 = np.fromfile(tmp_path / "p1.bin", dtype=np.float32)
    pN = np.fromfile(tmp_path / "pN.bin", dtype=np.float32)
    assert p1.shape == pN.shape and np.abs(p1 - pN).max() < 1e-4 * max(1.0, np.abs(p1).max())
    # --halo 1: only the rows each shard's edges reference travel; the epochs end at the SAME parameters, bit for bit
    halo = run(base + ["--ranks", str(ranks), "--transport", "host", "--halo", "1", "--dump-params", str(tmp_path / "pH.bin")])
    assert halo.returncode == 0, halo.stderr
    assert re.findall(pat, halo.stdout) == re.findall(pat, many.stdout)
    assert np.array_equal(np.fromfile(tmp_path / "pH.bin", dtype=np.float32), pN)


@pytest.mark.gpu
def test_rccl_transport_needs_one_gpu_per_rank(pkg, tmp_path):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs")
    ds = pkg.synth.make_dataset("cora", scale=0.05)
    pkg.synth.write_text_dataset(ds, str(tmp_path), "tiny")
    r = run(["--dataset", "tiny", "--data-root", str(tmp_path), "--num-layers", "2", "--heads", "8,8", "--outdims", "8,8",
             "--epochs", "1", "--ranks", "2"])
    assert r.returncode == 1 and "needs that many GPUs" in r.stderr


@pytest.mark.gpu
def test_binary_cache_gives_the_same_run(pkg, tmp_path):
    """--cache: the parsed text files are stored next to them; a second run loads the cache and must print the
    same loss lines; touching a text file invalidates it."""
    ds = pkg.synth.make_dataset("cora", scale=0.1)
    pkg.synth.write_text_dataset(ds, str(tmp_path), "tiny")
    args = ["--dataset", "tiny", "--data-root", str(tmp_path), "--num-layers", "2", "--heads", "8,8", "--outdims", "8,8",
            "--epochs", "2", "--seed", "3", "--cache"]
    pat = r"Avg Loss: ([0-9.]+), Accuracy: ([0-9.]+)%"
    first = run(args)
    assert first.returncode == 0, first.stderr
    cache = tmp_path / "tiny" / "gatv2_cache.bin"
    assert cache.exists() and cache.stat().st_size > ds["n"] * ds["f"] * 4
    second = run(args)
    assert second.returncode == 0 and re.findall(pat, first.stdout) == re.findall(pat, second.stdout)
    assert f"Graph loaded: {ds['n']} nodes, {ds['e']} edges" in second.stdout
    before = cache.stat().st_mtime_ns
    os.utime(tmp_path / "tiny" / "labels.txt", (cache.stat().st_mtime + 3600, cache.stat().st_mtime + 3600))
    third = run(args)                          # a text file newer than the cache: re-parsed and rewritten
    assert third.returncode == 0 and re.findall(pat, first.stdout) == re.findall(pat, third.stdout)
    assert cache.stat().st_mtime_ns > before


@pytest.mark.gpu
def test_train_and_val_masks_through_the_binary(pkg, orc, tmp_path):
    """--train-mask / --val-mask (beyond the reference, README R:134 "later"): the first epoch's printed numbers
    equal the fp64 forward restricted to the splits; without the flags the output is the reference's."""
    import ref64
    ds = pkg.synth.make_dataset("cora", scale=0.2)
    pkg.synth.write_text_dataset(ds, str(tmp_path), "tiny")
    x = np.loadtxt(tmp_path / "tiny" / "features.txt", dtype=np.float32, ndmin=2)
    n = ds["n"]
    rng = np.random.default_rng(4)
    train = rng.random(n) < 0.5
    val = ~train
    np.savetxt(tmp_path / "train.txt", train.astype(int), fmt="%d")
    np.savetxt(tmp_path / "val.txt", val.astype(int), fmt="%d")
    cfg = orc.Config([8, 8], [8, 8], ds["f"], ds["c"])
    W, a, Wo = orc.xavier_params(cfg, 7)
    pfile = tmp_path / "p.bin"
    np.concatenate([W, a, Wo]).astype(np.float32).tofile(pfile)
    r = run(["--dataset", "tiny", "--data-root", str(tmp_path), "--heads", "8,8", "--outdims", "8,8", "--epochs", "1",
             "--load-params", str(pfile), "--train-mask", str(tmp_path / "train.txt"), "--val-mask", str(tmp_path / "val.txt")])
    assert r.returncode == 0, r.stderr
    fw = ref64.forward(cfg, ds["row_ptr"], ds["col_idx"], ds["labels"], x, W, a, Wo)
    nll = -np.log(np.maximum(fw["y"][np.arange(n), ds["labels"]], 1e-12)); ok = fw["y"].argmax(1) == ds["labels"]
    m = re.search(r"Avg Loss: ([0-9.]+), Accuracy: ([0-9.]+)%\nVal Loss: ([0-9.]+), Val Accuracy: ([0-9.]+)%", r.stdout)
    assert m, r.stdout
    assert f"Training nodes: {int(train.sum())} of {n}\n" in r.stdout
    assert abs(float(m.group(1)) - nll[train].mean()) < 1e-4 and abs(float(m.group(2)) - 100.0 * ok[train].mean()) < 0.011
    assert abs(float(m.group(3)) - nll[val].mean()) < 1e-4 and abs(float(m.group(4)) - 100.0 * ok[val].mean()) < 0.011
    bad = run(["--dataset", "tiny", "--data-root", str(tmp_path), "--heads", "8,8", "--outdims", "8,8", "--epochs", "1",
               "--train-mask", str(tmp_path / "tiny" / "row_ptr.txt")])
    assert bad.returncode == 1 and "Invalid train mask length" in bad.stderr
