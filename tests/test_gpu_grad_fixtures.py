"""The HIP training step (mtadgat_forward_train / mtadgat_backward) against gradients HELD BY THE REFERENCE:
tests/golden/grads_*.npz were written by tests/golden/make_golden.py --grads from the unmodified reference model on CPU,
loss exactly as training.py:113-126, at the MSL and SMD-1-1 shapes with the shipped checkpoints -- in eval mode (dropout
off) and in train mode with the library's counter-based dropout masks injected into the reference's own dropout calls.

Batches: 32 / 33 windows (window-per-workgroup recurrences k_gru1 / k_gru1_bwd, <= 1792 windows), 2000 windows
(16-window groups, k_gru16 / k_gru16_bwd, <= 4096) and 4100 windows (hidden-tile-split kernels k_gru_split / k_gru_bwd).

Gate per parameter: |ours - reference| <= 1e-5 + 1e-4 * max|reference| + the reference's own float32 rounding noise
(max |g32 - g64| of the same reference model run in float64, recorded in the fixture; 0.002-0.1 of the gate at b ~ 32).
"""
import pytest
import torch

from helpers import GRAD_CASES, GradCase, dropout_masks_like_the_library, training_loss

pytestmark = pytest.mark.gpu


def _model(case, device):
    from mtad_gat import MTAD_GAT
    m = MTAD_GAT(**case.kwargs)
    m.load_state_dict(case.base.state_dict())
    return m.to(device)


@pytest.mark.parametrize("name", GRAD_CASES)
def test_training_step_matches_the_reference_held_gradients(name, gpu_device):
    c = GradCase(name)
    b, masks_variant = c.meta["batch"], c.meta["variant"] == "masks"
    model = _model(c, gpu_device)
    model.train() if masks_variant else model.eval()
    x, y = c.x.to(gpu_device), c.y.to(gpu_device)
    if masks_variant:
        object.__setattr__(model, "dropout_stream", (c.meta["drop_seed"], 0))
    try:
        preds, recons = model(x)
    finally:
        object.__setattr__(model, "dropout_stream", None)
    assert model.grad_path == "hip", model.grad_path
    if masks_variant and b <= 64:
        # the masks the kernels applied are the ones the fixture generator injected into the reference
        lib = model._engine.dropout_masks(b, c.kwargs["dropout"], c.meta["drop_seed"], gpu_device)
        ours = dropout_masks_like_the_library(c.kwargs, b, c.meta["drop_seed"])
        assert torch.equal(lib["feat"].cpu(), ours["feat"]) and torch.equal(lib["temp"].cpu(), ours["temp"])
        for a, r in zip(lib["fc"], ours["fc"]):
            assert torch.equal(a.cpu(), r)
    assert (preds[:64].detach().cpu().reshape(c.preds_head.shape) - c.preds_head).abs().max().item() <= 1e-5
    assert (recons[:8].detach().cpu() - c.recons_head).abs().max().item() <= 1e-5
    fl, rl = training_loss(preds, recons, x, y, c.meta["target_dims"])
    assert abs(fl.item() - c.loss[0]) <= 1e-5 and abs(rl.item() - c.loss[1]) <= 1e-5
    (fl + rl).backward()                                   # training.py:124-126
    rows, bad = [], []
    for n, p in model.named_parameters():
        ref = c.grads[n]
        d = (p.grad.detach().cpu() - ref).abs().max().item()
        gate = 1e-5 + 1e-4 * ref.abs().max().item() + c.noise[n]
        rows.append(f"{n:45s} |diff|={d:.3e} gate={gate:.3e} (reference's own noise {c.noise[n]:.1e})")
        if not d <= gate:
            bad.append(rows[-1])
    print("\n".join(rows))
    assert not bad, "\n".join(bad)
