"""Differentiable GPU forward: `torch.autograd.Function` over the HIP training forward (keeps the
activations the backward needs in a tape, dropout inside the kernels) and the HIP backward
(`mtadgat_forward_train` / `mtadgat_backward`, include/mtadgat.h).

Configurations without a HIP backward (`Engine.backward_supported()` false; `Engine.why_not()` says why) are evaluated by
the package's torch-op algebra (`_torchpath.py`) with autograd; `MTAD_GAT.grad_path` says which path the last differentiable
call took ("hip" / "torch-ops: <reason>").  Inputs that themselves require a gradient stay on the HIP path: the backward
returns d x as well (`mtadgat_backward_input`).
"""
import warnings

import torch

import _torchpath

# windows per chunk of the training step: tape + workspace are ~1.6 MB per window (W=100, F=55); a batch larger
# than this is processed chunk by chunk, re-running the chunk's forward inside backward (the counter-based
# dropout reproduces its masks) so that memory stays bounded whatever the batch
TRAIN_CHUNK = 8192


def _chunk_windows(eng, device):
    """Windows per chunk of the training step: TRAIN_CHUNK, fewer when tape + backward workspace of that many windows would not
    fit a third of the device memory (wide models: ~20 MB per window at F = 512, W = 256)."""
    cap = getattr(eng, "_train_chunk_cap", None)
    if cap is None:
        probe = 64
        per = (eng.lib.mtadgat_tape_bytes(eng.handle, probe) + eng.lib.mtadgat_backward_workspace_bytes(eng.handle, probe)) / probe
        total = torch.cuda.get_device_properties(device).total_memory
        cap = max(32, int((total / 3) // max(per, 1.0)) // 32 * 32)
        eng._train_chunk_cap = cap
    return max(1, min(TRAIN_CHUNK, cap))


def param_order(model):
    """The model's parameters in the field order of mtadgat_params / mtadgat_grad_offsets."""
    names = ["conv.conv.weight", "conv.conv.bias"]
    for g in ("feature_gat", "temporal_gat"):
        names += [f"{g}.lin.weight", f"{g}.lin.bias", f"{g}.a", f"{g}.bias"]
    for l in range(model.gru.gru.num_layers):
        names += [f"gru.gru.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    for i in range(len(model.forecasting_model.layers)):
        names += [f"forecasting_model.layers.{i}.weight", f"forecasting_model.layers.{i}.bias"]
    for l in range(model.recon_model.decoder.rnn.num_layers):
        names += [f"recon_model.decoder.rnn.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["recon_model.fc.weight", "recon_model.fc.bias"]
    named = dict(model.named_parameters())
    return [named[n] for n in names]


class _HipStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, p, seed, w0, *params):
        b = x.shape[0]
        cw = _chunk_windows(eng, x.device)
        chunks = [(lo, min(lo + cw, b)) for lo in range(0, b, cw)] or [(0, 0)]
        tape = None
        if len(chunks) == 1:
            preds, recons, tape = eng.forward_train(x, p, seed, w0)
        else:
            outs = []
            scratch = getattr(eng, "_train_tape", None)
            for lo, hi in chunks:
                pr, rc, scratch = eng.forward_train(x[lo:hi], p, seed, w0 + lo, tape=scratch)
                outs.append((pr, rc))
            preds = torch.cat([o[0] for o in outs])
            recons = torch.cat([o[1] for o in outs])
            eng._train_tape = scratch            # reused by backward's recomputation
        ctx.eng, ctx.p, ctx.seed, ctx.w0, ctx.chunks, ctx.tape = eng, p, seed, w0, chunks, tape
        ctx.save_for_backward(x)
        ctx.shapes = [(q.shape, q.numel()) for q in params]
        return preds, recons

    @staticmethod
    def backward(ctx, d_preds, d_recons):
        eng = ctx.eng
        (x,) = ctx.saved_tensors
        offs, total = eng.grad_layout()
        grads = torch.zeros(total, dtype=torch.float32, device=x.device)
        d_preds = d_preds.contiguous().float()
        d_recons = d_recons.contiguous().float()
        want_dx = ctx.needs_input_grad[1]
        dx = None
        if x.shape[0]:
            if ctx.tape is not None:
                eng.backward(x, ctx.p, ctx.seed, d_preds, d_recons, ctx.tape, grads, ctx.w0)
                if want_dx:
                    dx = eng.backward_input(x)       # (the convolution's pre-activation gradients are still in the backward's workspace)
            else:
                scratch = getattr(eng, "_train_tape", None)
                parts = []
                for lo, hi in ctx.chunks:
                    xc = x[lo:hi]
                    _, _, scratch = eng.forward_train(xc, ctx.p, ctx.seed, ctx.w0 + lo, tape=scratch)
                    eng.backward(xc, ctx.p, ctx.seed, d_preds[lo:hi].contiguous(), d_recons[lo:hi].contiguous(), scratch, grads, ctx.w0 + lo)
                    if want_dx:
                        parts.append(eng.backward_input(xc))
                if want_dx:
                    dx = torch.cat(parts)
        elif want_dx:
            dx = torch.zeros_like(x)
        ctx.tape = None
        # the parameters' gradients are views of ONE flat buffer (field order of mtadgat_params): autograd adopts them as
        # `.grad` without copying, and a data-parallel step can exchange the whole buffer with a single collective
        # (sharding.dp_training_step looks for it here)
        eng._flat_grads = (grads, list(offs))
        out = [grads[o:o + n].view(shape) for o, (shape, n) in zip(offs, ctx.shapes)]
        return (None, dx, None, None, None, *out)


_warned = set()      # reasons already reported (one warning per reason and process)


def draw_dropout_stream(model):
    """(seed, first global window) of this call's counter-based dropout masks.  One seed per call from torch's (CPU) generator:
    runs are reproducible under torch.manual_seed.  A caller that shards one logical batch over several ranks sets
    `model.dropout_stream = (seed, first_global_window)` for the call (sharding.dp_training_step does): the masks are keyed by
    the global window index, so the shards draw exactly the masks the single-process step over the whole batch would."""
    override = getattr(model, "dropout_stream", None)
    if override is not None:
        return int(override[0]), int(override[1])
    p = float(model.dropout_p) if model.training else 0.0
    return (int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0), 0


def forward(model, eng, x, stream=None):
    """(preds, recons) with autograd history when grad is enabled; dropout active iff model.training.
    stream: draw_dropout_stream(model) when the caller drew it already (MTAD_GAT.forward does, once per call)."""
    why = None
    if not eng.backward_supported():
        why = eng.why_not()
    if why is not None:
        # Not silent: the torch-op route materialises the (b, K, K, 2E) attention tensors and runs MIOpen's GRU -- a caller
        # who expects the HIP training step must learn that this configuration does not have one.
        object.__setattr__(model, "grad_path", "torch-ops: " + why)
        if getattr(model, "strict_hip_training", True):
            raise RuntimeError("this training step cannot run on the HIP kernels: " + why + ".  The inference forward of this "
                               "configuration is unaffected.  MTAD_GAT.strict_hip_training = False evaluates such a step "
                               "by torch ops on the GPU instead (rocBLAS / MIOpen through autograd, much slower).")
        if why not in _warned:
            _warned.add(why)
            warnings.warn("MTAD_GAT training step on the GPU runs through torch ops (autograd), not the HIP backward: " + why +
                          ".  The inference forward of this configuration is unaffected (model.strict_hip_training = False "
                          "opted into this route).", RuntimeWarning, stacklevel=3)
        return _torchpath.forward(model, x.float())
    object.__setattr__(model, "grad_path", "hip")
    p = float(model.dropout_p) if model.training else 0.0
    seed, w0 = stream if stream is not None else draw_dropout_stream(model)
    params = param_order(model)
    # (x keeps its autograd history: when it requires a gradient the backward also returns d x -- mtadgat_backward_input)
    return _HipStep.apply(eng, x.contiguous().float(), p, seed, w0, *params)
