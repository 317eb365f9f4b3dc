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
    # round 5: the other BASELINE workloads behind the headline, in the same line; who took part in the run
    labels = [e["label"] for e in d["configs"]]
    assert len(labels) == 4 and labels[0].startswith("configs[2]") and labels[1].startswith("configs[3]") \
        and labels[2].startswith("configs[4]") and "AE2D" in labels[3], labels
    for e in d["configs"]:
        assert "error" not in e, e
        assert e["ms_per_step"] > 0 and e["env_steps_per_s"] > 0 and 0.3 < e["roofline"]["frac"] < 1.0, e
        assert abs(e["roofline"]["frac"] - e["roofline"]["achieved"] / 8000.0) < 1e-9
    assert "b1_kernel" in d["configs"][2]["roofline"]["kernel"] and d["configs"][2]["dtype"] == "u64 bit-planes"
    assert "ae_block64_kernel" in d["configs"][3]["roofline"]["other_kernels_us"]
    col = d["collective"]
    assert col["collective_backend_initialised"] is False and col["world_size"] == 1
    assert col["rank_device_map"] == [{"rank": 0, "local_rank": 0, "device": 0}]


def test_bench_single_rank_rccl_group():
    """bench.py --single-rank-group: the N=1 workload with every exchange of the path going through a process group of ONE
    rank on RCCL (all a one-GPU box can execute of the multi-GPU path); the line says so."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-rank-group", "--rnd", "--steps", "70",
                          "--warmup", "4", "--stats-every", "16", "--no-cpu-baseline", "--no-extra-configs"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    col = d["collective"]
    assert col["collective_backend_initialised"] is True and col["backend"] == "nccl" and col["world_size"] == 1
    assert d["n_gpus"] == 1 and "configs" not in d and "cpu_baseline" not in d
    assert d["rewards_counted_all_ranks"] == 4096 * (70 * d["timing"]["repeats"] + 4)


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
