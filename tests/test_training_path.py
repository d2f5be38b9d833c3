"""The training step on CPU tensors (the package's torch-op path, _torchpath.py): gradients equal to autograd through the
oracle, every parameter gets a gradient, and the data-parallel step reproduces the single-process
global-batch gradient (gloo, world_size 2).  CPU only; the GPU variant is in test_gpu_training.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import mtad_gat_oracle as oracle

KW = [
    dict(n_features=6, window_size=10, out_dim=6, kernel_size=3, use_gatv2=True, gru_n_layers=2, gru_hid_dim=8,
         forecast_n_layers=2, forecast_hid_dim=7, recon_n_layers=2, recon_hid_dim=9, dropout=0.0, alpha=0.2),
    dict(n_features=5, window_size=8, out_dim=2, kernel_size=5, use_gatv2=False, feat_gat_embed_dim=3,
         time_gat_embed_dim=4, gru_hid_dim=11, forecast_n_layers=1, forecast_hid_dim=6, recon_hid_dim=5,
         dropout=0.0, alpha=0.3),
]


def _model(kw, seed=0):
    from mtad_gat import MTAD_GAT
    torch.manual_seed(seed)
    m = MTAD_GAT(**kw)
    with torch.no_grad():
        m.feature_gat.bias.normal_()
        m.temporal_gat.bias.normal_()
    return m.train()


def _loss(preds, recons, x, y):
    return torch.sqrt(F.mse_loss(y, preds)) + torch.sqrt(F.mse_loss(x[:, :, : recons.shape[2]], recons))


@pytest.mark.parametrize("kw", KW)
def test_gradients_match_autograd_through_the_oracle(kw):
    m = _model(kw)
    x = torch.rand(5, kw["window_size"], kw["n_features"])
    y = torch.rand(5, kw["out_dim"])
    p, r = m(x)
    _loss(p, r, x, y).backward()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    po, ro = oracle.forward(x, sd, alpha=kw["alpha"])
    _loss(po, ro, x, y).backward()
    assert (p - po).abs().max().item() <= 1e-6 and (r - ro).abs().max().item() <= 1e-6
    for name, prm in m.named_parameters():
        assert prm.grad is not None, name
        assert (prm.grad - sd[name].grad).abs().max().item() <= 1e-5, name


def test_dropout_is_active_only_in_training():
    kw = dict(KW[0], dropout=0.5)
    m = _model(kw)
    x = torch.rand(4, kw["window_size"], kw["n_features"])
    torch.manual_seed(1)
    a = m(x)[0]
    torch.manual_seed(2)
    b = m(x)[0]
    assert not torch.equal(a, b)                 # different masks
    torch.manual_seed(1)
    assert torch.equal(a, m(x)[0])               # reproducible under the torch generator


def _dp_worker(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mtad-gat-pytorch_amd"), root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sharding import dp_training_step, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    kw = KW[0]
    m = _model(kw)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(7, kw["window_size"], kw["n_features"], generator=g)      # ragged: 4 + 3
    y = torch.rand(7, 1, kw["n_features"], generator=g)
    lo, hi = shard_range(7, rank, world)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)          # lr 0: keep weights, inspect .grad
    rm = dp_training_step(m, x[lo:hi], y[lo:hi], opt)
    if rank == 0:
        torch.save(dict(grads={n: p.grad.clone() for n, p in m.named_parameters()}, rmse=rm), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_equals_global_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    kw = KW[0]
    m = _model(kw)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(7, kw["window_size"], kw["n_features"], generator=g)
    y = torch.rand(7, 1, kw["n_features"], generator=g)
    p, r = m(x)
    fl = torch.sqrt(F.mse_loss(y.squeeze(1), p))
    rl = torch.sqrt(F.mse_loss(x, r))
    (fl + rl).backward()                                   # reference semantics, training.py:122-126
    assert abs(res["rmse"][0] - float(fl)) <= 1e-6 and abs(res["rmse"][1] - float(rl)) <= 1e-6
    for n, prm in m.named_parameters():
        assert (prm.grad - res["grads"][n]).abs().max().item() <= 2e-6, n


def _dp_count_worker(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mtad-gat-pytorch_amd"), root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sharding
    from sharding import dp_training_step, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    kw = KW[0]
    m = _model(kw)
    frozen = m.forecasting_model.layers[0].bias
    frozen.requires_grad_(False)                           # a parameter without a gradient: stays without one (copy path)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(7, kw["window_size"], kw["n_features"], generator=g)
    y = torch.rand(7, 1, kw["n_features"], generator=g)
    lo, hi = shard_range(7, rank, world)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    log = []                                               # ("coll", name) / ("item",) / ("backward",) in call order
    real = {n: getattr(dist, n) for n in ("all_reduce", "broadcast", "all_gather", "reduce", "all_gather_into_tensor", "barrier")}
    for n, fn in real.items():
        setattr(dist, n, (lambda n, fn: lambda *a, **k: (log.append(("coll", n)), fn(*a, **k))[1])(n, fn))
    real_item, real_float, real_bwd = torch.Tensor.item, torch.Tensor.__float__, torch.Tensor.backward
    torch.Tensor.item = lambda self: (log.append(("item",)), real_item(self))[1]
    torch.Tensor.__float__ = lambda self: (log.append(("item",)), real_float(self))[1]
    torch.Tensor.backward = lambda self, *a, **k: (log.append(("backward",)), real_bwd(self, *a, **k))[1]
    per_step = []
    try:
        for step in range(3):
            del log[:]
            dp_training_step(m, x[lo:hi], y[lo:hi], opt)
            per_step.append(list(log))
    finally:
        for n, fn in real.items():
            setattr(dist, n, fn)
        torch.Tensor.item, torch.Tensor.__float__, torch.Tensor.backward = real_item, real_float, real_bwd
    if rank == 0:
        torch.save(dict(per_step=per_step, frozen_grad_is_none=frozen.grad is None, state=dict(m._dp_state, pinned=None)), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_talks_twice(tmp_path):
    """SURVEY section 8e / reference training.py:122-127: the steady-state data-parallel step is TWO collectives (statistics,
    gradient bucket); what the ranks must agree on rides in the first, the run's seed / shard sizes are settled once in step 0;
    nothing reads a device value on the host before backward() has been enqueued."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dpc.pt")
    mp.spawn(_dp_count_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    first, steady = res["per_step"][0], res["per_step"][1:]
    assert [e[1] for e in first if e[0] == "coll"] == ["all_gather", "all_reduce", "all_reduce"]
    for log in steady:
        assert [e[1] for e in log if e[0] == "coll"] == ["all_reduce", "all_reduce"], log
        b = log.index(("backward",))
        assert ("item",) not in log[:b], "a host read in front of backward()"
        assert [e for e in log[:b] if e[0] == "coll"] == [("coll", "all_reduce")]       # the statistics go first, the bucket after
    assert res["frozen_grad_is_none"]
    assert res["state"]["counts"] == [4, 3] and res["state"]["step"] == 3 and res["state"]["layout_refreshed"] == 0


@pytest.mark.parametrize("name", ["grads_msl_eval", "grads_msl_masks", "grads_smd_eval", "grads_smd_masks"])
def test_torch_op_algebra_matches_the_reference_held_gradients(name):
    """_torchpath.forward (the CPU route of the training step and the other side of tests/test_gpu_backward.py) against
    the gradients the reference itself produced (tests/golden/make_golden.py --grads; loss as training.py:113-126)."""
    import _torchpath
    from helpers import GradCase, dropout_masks_like_the_library, training_loss
    from mtad_gat import MTAD_GAT
    c = GradCase(name)
    m = MTAD_GAT(**c.kwargs)
    m.load_state_dict(c.base.state_dict())
    masks = None
    if c.meta["variant"] == "masks":
        m.train()
        masks = dropout_masks_like_the_library(c.kwargs, c.meta["batch"], c.meta["drop_seed"])
    else:
        m.eval()
    p, r = _torchpath.forward(m, c.x, masks)
    fl, rl = training_loss(p, r, c.x, c.y, c.meta["target_dims"])
    assert abs(fl.item() - c.loss[0]) <= 1e-5 and abs(rl.item() - c.loss[1]) <= 1e-5
    (fl + rl).backward()
    for n, prm in m.named_parameters():
        ref = c.grads[n]
        d = (prm.grad - ref).abs().max().item()
        assert d <= 1e-5 + 1e-4 * ref.abs().max().item() + c.noise[n], (n, d)


def test_layer_by_layer_gru_equals_nn_gru_and_applies_explicit_masks():
    """_torchpath._layered_gru (the torch-op side of the stacked-layer gradient tests): with all-ones masks it is nn.GRU in
    eval mode; with a mask it is the per-layer recurrence on the masked, rescaled sequence."""
    import _torchpath
    torch.manual_seed(3)
    rnn = torch.nn.GRU(7, 11, num_layers=3, batch_first=True, dropout=0.25).eval()
    x = torch.rand(4, 9, 7)
    ref, _ = rnn(x)
    ones = [torch.ones(4, 9, 11), torch.ones(4, 9, 11)]
    p = rnn.dropout
    got = _torchpath._layered_gru(rnn, x, [m * (1.0 - p) for m in ones])      # keep-scale 1 / (1 - p) cancelled
    assert (got - ref).abs().max().item() <= 1e-6
    g = torch.Generator().manual_seed(1)
    masks = [(torch.rand(4, 9, 11, generator=g) > p).float() for _ in range(2)]
    got = _torchpath._layered_gru(rnn, x, masks)
    single = [torch.nn.GRU(7 if l == 0 else 11, 11, batch_first=True) for l in range(3)]
    for l, s1 in enumerate(single):
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            getattr(s1, k + "_l0").data.copy_(getattr(rnn, f"{k}_l{l}").data)
    h = x
    for l, s1 in enumerate(single):
        h, _ = s1(h)
        if l < 2:
            h = h * masks[l] / (1.0 - p)
    assert (got - h).abs().max().item() <= 1e-6
