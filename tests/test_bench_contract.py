"""The bench.py contract the driver depends on: ONE JSON line on stdout with the headline keys, the `roofline` object (live HIP-event
timing of the dominant kernel class) and, at N = 1, the `cpu_baseline` object — on a scaled-down graph so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout (native banners go to stderr)
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = _run("--gpus", "1", "--steps", "4", "--warmup", "2", "--scale", "0.02", "--cpu-sample-scale", "0.0005")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("edges/sec") and d["unit"] == "edges/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 0.02 * 61_900_000 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None                        # PMC traffic is attached only to the full-size workload it was collected on
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and 1 <= c["cores"] <= (os.cpu_count() or 1)
    sys.path.insert(0, ROOT)
    import oracle.oracle as orc
    assert c["cores"] == orc.effective_cpus()          # the CPUs the job may really use, not the ones the box shows


def test_bench_graph_mode_and_bf16_lines():
    d = _run("--workload", "pubmed", "--graph", "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    assert d["config"]["launch"] == "hipGraph replay" and d["value"] > 0 and "cpu_baseline" not in d
    d = _run("--workload", "products", "--dtype", "bf16", "--scale", "0.02", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert d["dtype"].startswith("f32 arithmetic") and d["value"] > 0
