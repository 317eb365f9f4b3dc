"""GPU: bench.py honours its output contract (one JSON line with the roofline and cpu_baseline
objects) on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_contract_short_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "32", "--warmup", "4",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 32 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 32 / (d["ms_per_step"] * 32e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"] * 0.99
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus 2 through its own launcher: two rank processes (sharing the one device of the
    test box, gloo), n_gpus == 2 in the JSON, the reward statistics counted on both ranks."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpus",
                          "--backend", "gloo", "--instances", "512", "--steps", "130", "--warmup", "4",
                          "--stats-every", "64"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == "batch x2"
    assert d["rewards_counted_all_ranks"] == 2 * 512 * (130 * d["timing"]["repeats"] + 4)
    assert "cpu_baseline" not in d


def test_bench_refuses_more_gpus_than_present():
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)],
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 2 and "refusing" in out.stderr
