"""The N>1 path on CPU: window sharding + max-over-ranks timing + gather, world_size 2 over gloo.
Every rank runs the drop-in module on its shard (CPU tensors -> the package's torch-op path; the same
test with the HIP forward per rank is tests/test_gpu_plumbing.py::test_two_processes_share_the_gpu...)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import Case


def test_shard_range_partitions_everything():
    from sharding import shard_range
    for total in (0, 1, 7, 8, 100, 250001):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and 0 <= (b - a) - (d - c) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mtad-gat-pytorch_amd"), root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharding import gather_windows, max_over_ranks, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    case = Case("syn_v2_embed")            # 37 windows: ragged split 19 + 18
    lo, hi = shard_range(case.x.shape[0], rank, world)
    model = case.build_model()
    with torch.no_grad():
        p, r = model(case.x[lo:hi])
    p_all = gather_windows(p, case.x.shape[0])
    r_all = gather_windows(r, case.x.shape[0])
    t = max_over_ranks(1.0 + rank)          # slowest rank wins
    dist.barrier()
    if rank == 0:
        torch.save(dict(p=p_all, r=r_all, t=t, shard=(lo, hi)), out_path)
    dist.destroy_process_group()


def test_two_rank_sharded_forward_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    case = Case("syn_v2_embed")
    assert res["t"] == 2.0 and res["shard"] == (0, 19)
    # windows are independent: sharded == unsharded == reference
    assert (res["p"] - case.preds).abs().max().item() <= 2e-6
    assert (res["r"] - case.recons).abs().max().item() <= 2e-6
