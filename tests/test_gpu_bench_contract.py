"""bench.py prints exactly one JSON line with the fields the driver reads (small batch, 1 step)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line(gpu_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "4096", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"].startswith("f32 (2xf16 split operands") and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 4096 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.01
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["peak"] == 2500.0 and 0 < r["alg_frac"] <= r["issued_frac"] < 1
    assert d["roofline_valu"]["bound"] == "valu" and "range_guard" in d and d["range_guard"]["conv_max"] > 0


def test_bench_train_mode_line(gpu_device):
    """--mode train: the data-parallel training step (forward + global-batch RMSE losses + HIP backward + Adam)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", "2", "--warmup", "1",
                          "--batch", "512"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "windows/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["grad_path"] == "hip"
    assert d["value"] > 0 and abs(d["value"] - 512 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.01
    assert set(d["exchange_ms_per_step"]) >= {"stats_allreduce", "grad_allreduce"}
