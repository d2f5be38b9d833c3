"""The package's own PyTorch-op evaluation of MTAD_GAT (no HIP, no oracle).

Two jobs:

* **CPU tensors.**  The reference's callers fall back to `device = "cpu"` when no GPU is present or
  `--use_cuda False` is given (reference predict.py:122, training.py:60, prediction.py:45; BASELINE
  config 1 is a CPU plumbing run).  A model / input on the CPU is evaluated here -- explicitly chosen
  by the caller through the tensors' device, never a silent fallback: GPU tensors always take the HIP
  kernels and raise if `libmtadgat.so` is missing.
* **Checker / coverage gaps on the GPU.**  Configurations the HIP backward does not cover yet are
  evaluated here with autograd, and the GPU tests differentiate through this file to check the HIP
  gradients (besides the oracle).

It evaluates the *re-associated algebra the kernels use* (DESIGN.md section 3), not the reference's
formulation: nothing materialises the (b, K, K, 2D) pair tensor; the largest intermediate is
|L_i + R_j| of shape (b, K, K, E).

Dropout follows the reference: on the two attention matrices (modules.py:90, :189), between the
forecasting layers (modules.py:310) and between stacked GRU layers (modules.py:233, :253, inside
nn.GRU); masks come from torch's generator in the reference's order of consumption, so a seeded CPU run
reproduces the reference's masks exactly.  `masks` (optional) injects explicit keep-masks instead:
{"feat": (b,F,F), "temp": (b,W,W), "fc": [(b,hid), ...]} of 0/1 floats, scaled by 1/(1-p) here; with stacked recurrences also
"gru" / "rec": [(b,W,H), ...] for nn.GRU's dropout between the layers.
"""
import contextlib

import torch
import torch.nn.functional as F


def _drop(t, p, training, mask=None):
    if mask is not None:
        return t * mask * (1.0 / (1.0 - p))
    return F.dropout(t, p, training)


def graph_attention(v, layer, training, mask=None):
    """v (b, K, D) node rows -> sigmoid(softmax(e) @ v), e as in modules.py:74-93 / :174-191."""
    alpha = layer.alpha
    if layer.use_gatv2:
        d = v.shape[2]
        w_l, w_r = layer.lin.weight[:, :d], layer.lin.weight[:, d:]
        left = F.linear(v, w_l, layer.lin.bias)                 # (b, K, E)  W_l v_i + b
        right = F.linear(v, w_r)                                # (b, K, E)  W_r v_j
        a = layer.a.squeeze(1)
        # a . LeakyReLU(u) = (1+alpha)/2 a.u + (1-alpha)/2 a.|u|,  u_ij = left_i + right_j
        lin = 0.5 * (1.0 + alpha) * ((left @ a).unsqueeze(2) + (right @ a).unsqueeze(1))
        pair = (left.unsqueeze(2) + right.unsqueeze(1)).abs() @ (0.5 * (1.0 - alpha) * a)
        e = lin + pair
    else:
        e_dim = layer.lin.weight.shape[0]
        p = layer.lin(v)                                        # (b, K, E)
        a = layer.a.squeeze(1)
        e = F.leaky_relu((p @ a[:e_dim]).unsqueeze(2) + (p @ a[e_dim:]).unsqueeze(1), alpha)
    if layer.use_bias:
        e = e + layer.bias
    att = torch.softmax(e, dim=2)
    att = _drop(att, layer.dropout, training, mask)
    return torch.sigmoid(att @ v)


def conv_stage(model, x):
    conv = model.conv.conv
    pad = (conv.kernel_size[0] - 1) // 2
    return F.relu(F.conv1d(F.pad(x.permute(0, 2, 1), (pad, pad)), conv.weight, conv.bias)).permute(0, 2, 1)


def feature_gat_stage(model, xc, training=False, mask=None):
    return graph_attention(xc.permute(0, 2, 1), model.feature_gat, training, mask).permute(0, 2, 1)


def temporal_gat_stage(model, xc, training=False, mask=None):
    return graph_attention(xc, model.temporal_gat, training, mask)


def _no_miopen_rnn(t):
    """On the GPU nn.GRU dispatches to MIOpen's fused RNN, which returned wrong, call-order dependent results for some shape
    sequences in round 6's random-shape runs (tests/test_gpu_fuzz.py; 0.1 .. 0.5 off, gone with the fused RNN disabled): the
    torch-op route uses aten's own GRU there.  (The CPU route is unaffected.)"""
    return torch.backends.cudnn.flags(enabled=False) if t.device.type == "cuda" else contextlib.nullcontext()


def _layered_gru(rnn, x, masks):
    """nn.GRU evaluated layer by layer (single-layer aten::gru calls on the module's own parameters), with the given keep-masks
    (b, W, H) applied -- scaled by 1 / (1 - p) -- to the outputs of every layer but the last: nn.GRU's inter-layer dropout
    (reference modules.py:233 / :253) with the masks made explicit.  Returns the last layer's state sequence."""
    p = rnn.dropout
    out = x
    for l in range(rnn.num_layers):
        w = [getattr(rnn, f"{k}_l{l}") for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        h0 = out.new_zeros(1, out.shape[0], rnn.hidden_size)
        with _no_miopen_rnn(out):
            out, _ = torch._VF.gru(out, h0, w, True, 1, 0.0, False, False, True)
        if l + 1 < rnn.num_layers:
            out = out * masks[l] * (1.0 / (1.0 - p))
    return out


def gru_stage(model, h_cat, masks=None):
    if masks:
        return _layered_gru(model.gru.gru, h_cat, masks)[:, -1, :]
    with _no_miopen_rnn(h_cat):
        _, h = model.gru.gru(h_cat)                             # nn.GRU: h0 = 0, inter-layer dropout in train()
    return h[-1]


def forecast_stage(model, h_end, training=False, masks=None):
    y = h_end
    layers = model.forecasting_model.layers
    for i, lin in enumerate(layers[:-1]):
        y = _drop(F.relu(lin(y)), model.forecasting_model.dropout.p, training, None if masks is None else masks[i])
    return layers[-1](y)


def recon_stage(model, h_end, masks=None):
    w = model.recon_model.window_size
    rep = h_end.repeat_interleave(w, dim=1).view(h_end.shape[0], w, h_end.shape[1])     # the reference's decoder input (modules.py:279)
    if masks:
        dec = _layered_gru(model.recon_model.decoder.rnn, rep, masks)
    else:
        with _no_miopen_rnn(rep):
            dec, _ = model.recon_model.decoder.rnn(rep)
    return model.recon_model.fc(dec)


def forward(model, x, masks=None):
    """Same contract as MTAD_GAT.forward (reference mtad_gat.py:64-79), built from torch ops."""
    training = model.training
    masks = masks or {}
    xc = conv_stage(model, x)
    h_feat = feature_gat_stage(model, xc, training, masks.get("feat"))
    h_temp = temporal_gat_stage(model, xc, training, masks.get("temp"))
    h_end = gru_stage(model, torch.cat([xc, h_feat, h_temp], dim=2), masks.get("gru"))
    preds = forecast_stage(model, h_end, training, masks.get("fc"))
    recons = recon_stage(model, h_end, masks.get("rec"))
    return preds, recons

