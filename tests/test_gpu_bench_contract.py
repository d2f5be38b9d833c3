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
    # `roofline` is the launch family that takes most of the step: the recurrences (matrix pipe) or the attention layers
    # (vector ALU -- a bound the contract has no name for); both families are always in the line under their own names
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["largest_family_by_time"] == max(d["kernels"], key=lambda k: d["kernels"][k]["ms_per_step"])
    assert r["kernel"].startswith(r["largest_family_by_time"].split("(")[0])
    rm, rv = d["roofline_mfma"], d["roofline_valu"]
    assert rm["bound"] == "mfma" and rm["peak"] == 2500.0 and 0 < rm["alg_frac"] <= rm["issued_frac"] < 1
    assert rv["bound"] == "valu" and abs(rv["peak"] - 78.6) < 0.1 and abs(rv["frac_vs_r03_peak"] - 2 * rv["frac"]) < 1e-3
    assert "range_guard" in d and d["range_guard"]["conv_max"] > 0
    assert d["rccl_ranks"] == 0 and len(d["per_rank_windows_per_s"]) == 1


def test_bench_starts_its_own_ranks(gpu_device):
    """`python bench.py --gpus N` with no launcher around it re-executes itself under torch.distributed.run with one rank per
    GPU and an RCCL process group; at N = 1 through --spawn.  More ranks than GPUs must fail loudly, not run fewer."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1",
                          "--batch", "4096", "--no-cpu-baseline", "--no-sub"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and len(d["per_rank_windows_per_s"]) == 1 and d["value"] > 0
    import torch
    too_many = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(too_many), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


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
