"""bench.py's own launcher (`python bench.py --gpus N` with no torch.distributed.run around it): on a host with fewer than N
GPUs it must refuse loudly instead of benchmarking fewer ranks and printing a line (runs on the CPU-only build host: 0 GPUs)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_ranks_than_gpus_is_refused():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "refusing to run fewer ranks" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_gpus_flag_must_match_the_launcher():
    """Under a launcher (WORLD_SIZE set) a --gpus value that disagrees with it is an error, not a warning."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in out.stderr
