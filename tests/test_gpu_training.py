"""Training step on the GPU (interim autograd path) next to the HIP inference path."""
import pytest
import torch
import torch.nn.functional as F

from helpers import Case

pytestmark = pytest.mark.gpu


def test_train_steps_then_hip_inference(gpu_device):
    case = Case("syn_v2_embed")
    model = case.build_model().to(gpu_device)
    x = case.x.to(gpu_device)
    y = torch.rand(x.shape[0], case.kwargs["out_dim"], device=gpu_device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)       # as train.py:92
    with torch.no_grad():
        p0, r0 = model.eval()(x)                               # HIP
    model.train()
    losses = []
    for _ in range(5):
        opt.zero_grad()
        p, r = model(x)                                        # autograd path (dropout 0.2 active)
        loss = torch.sqrt(F.mse_loss(y, p)) + torch.sqrt(F.mse_loss(x[:, :, : r.shape[2]], r))
        loss.backward()
        assert all(prm.grad is not None and torch.isfinite(prm.grad).all() for prm in model.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    # the optimizer updated the parameters in place: the HIP path must pick the new weights up,
    # and agree with the autograd path evaluated in eval mode on the same weights
    model.eval()
    with torch.no_grad():
        p1, r1 = model(x)
    assert not torch.equal(p1, p0)
    from _torchpath import forward as torch_ops_forward
    with torch.no_grad():
        p2, r2 = torch_ops_forward(model, x)
    assert (p1 - p2).abs().max().item() <= 1e-5 and (r1 - r2).abs().max().item() <= 1e-5


def test_data_edits_between_training_steps_reach_the_kernels(gpu_device):
    """`p.data` edits (weight clipping, EMA copies, nn.init on p.data) bump no autograd version counter: the default weight
    key carries the content fingerprint in train() mode as well, so the next step runs on the edited weights; with
    check_weight_contents = "eval_only" the training step trusts the counters and refresh_weights() is the caller's job."""
    from mtad_gat import MTAD_GAT
    case = Case("syn_v2_embed")
    model = MTAD_GAT(**{**case.kwargs, "dropout": 0.0})                  # no dropout: two steps on the same weights agree exactly
    model.load_state_dict(case.state_dict())
    model = model.to(gpu_device).train()
    x = case.x.to(gpu_device)
    p0, _ = model(x)
    with torch.no_grad():
        model.forecasting_model.layers[-1].bias.data.add_(0.25)          # .data: its own version counter
    p1, _ = model(x)
    assert (p1 - p0 - 0.25).abs().max().item() <= 1e-5
    model.check_weight_contents = "eval_only"
    p1b, _ = model(x)                                                    # unchanged weights, no fingerprint: same result
    assert torch.equal(p1b, p1)
    with torch.no_grad():
        model.forecasting_model.layers[-1].bias.data.add_(0.25)
    p2, _ = model(x)                                                     # the documented blind spot of "eval_only"
    assert torch.equal(p2, p1)
    model.refresh_weights()
    p3, _ = model(x)
    assert (p3 - p1 - 0.25).abs().max().item() <= 1e-5
