"""Differentiable forward for the training step (interim: PyTorch-ROCm ops + autograd).

The HIP kernels in this package implement the forward pass only.  Whenever gradients are
needed (`model.train()` with grad enabled -- `Trainer.fit`, reference training.py:100-130) the
module evaluates the *same re-associated algebra the kernels use*  with
differentiable torch ops, so `loss.backward()` populates `.grad` of every parameter exactly as
with the reference module (SURVEY.md section 8b "Autograd").  Nothing here materialises the
reference's (b, K, K, 2D) pair tensor; the largest intermediate is |L_i + R_j| of shape
(b, K, K, E).  Inference (`eval()` / `no_grad`) never comes through this file.

Dropout follows the reference: on the two attention matrices (modules.py:90, :189), between
the forecasting layers (modules.py:310) and between stacked GRU layers (modules.py:233, :253,
inside nn.GRU); masks come from torch's generator, so runs are reproducible under
`torch.manual_seed` but not bit-identical to the reference's random stream.
"""
import torch
import torch.nn.functional as F


def _graph_attention(v, layer, training):
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
    att = F.dropout(att, layer.dropout, training)
    return torch.sigmoid(att @ v)


def differentiable_forward(model, x):
    """Same contract as MTAD_GAT.forward (reference mtad_gat.py:64-79), built from torch ops."""
    training = model.training
    conv = model.conv.conv
    pad = (conv.kernel_size[0] - 1) // 2
    xc = F.relu(F.conv1d(F.pad(x.permute(0, 2, 1), (pad, pad)), conv.weight, conv.bias)).permute(0, 2, 1)
    h_feat = _graph_attention(xc.permute(0, 2, 1), model.feature_gat, training).permute(0, 2, 1)
    h_temp = _graph_attention(xc, model.temporal_gat, training)
    h_cat = torch.cat([xc, h_feat, h_temp], dim=2)
    _, h = model.gru.gru(h_cat)                                 # nn.GRU: h0 = 0, inter-layer dropout in train()
    h_end = h[-1]
    y = h_end
    layers = model.forecasting_model.layers
    for lin in layers[:-1]:
        y = F.dropout(F.relu(lin(y)), model.forecasting_model.dropout.p, training)
    preds = layers[-1](y)
    w = model.recon_model.window_size
    rep = h_end.repeat_interleave(w, dim=1).view(x.shape[0], w, -1)     # the reference's decoder input (modules.py:279)
    dec, _ = model.recon_model.decoder.rnn(rep)
    recons = model.recon_model.fc(dec)
    return preds, recons
