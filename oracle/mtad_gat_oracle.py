"""CPU oracle for the MTAD-GAT per-window forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`mtad-gat-pytorch_amd/`)
may import this file; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the timed CPU
baseline.

What it is: a functional restatement, on CPU tensors, of the arithmetic that
ML4ITS/mtad-gat-pytorch performs in `MTAD_GAT.forward` (reference
`mtad_gat.py:64-79`) in the reference's own formulation -- the pairwise
(K, K, 2D) attention input is materialised exactly like the reference does --
so that timing it gives the reference CPU path, and comparing against it gives
reference parity.  It consumes a reference-format ``state_dict`` (the key names
of `mtad_gat.py:56-62` / `modules.py`) and derives every dimension from the
tensor shapes, so the three shipped checkpoints load as-is.

The arithmetic lives in PyTorch's ATen (conv / addmm / softmax / GRU); the
reference does not pin a torch version (`requirements.txt:1-9` does not list
torch).  The GRU recurrence is restated from PyTorch's documented gate
equations (torch.nn.GRU: r, z, n gate order; n = tanh(W_in x + b_in +
r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h).

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 8c).  The oracle is pinned instead against outputs of the
reference itself, produced in the build container by importing
`/root/reference/{mtad_gat,modules}.py` (`tests/golden/make_golden.py`,
fixtures committed under `tests/golden/*.npz`), and -- when `/root/reference`
is present -- live in `tests/test_oracle_vs_reference.py`.

dtype: every function computes in the dtype of its inputs, so the same code
gives the float32 reference path and a float64 "truth" used to judge rounding.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration derived from a reference-format state_dict
# --------------------------------------------------------------------------
@dataclass
class OracleConfig:
    n_features: int
    window_size: int
    out_dim: int
    kernel_size: int
    use_gatv2: bool
    feat_embed: int      # rows of feature_gat.lin.weight (already doubled for v2)
    time_embed: int
    gru_n_layers: int
    gru_hid_dim: int
    forecast_n_linear: int   # number of nn.Linear in the forecasting head (= n_layers + 1)
    recon_n_layers: int
    recon_hid_dim: int
    alpha: float = 0.2


def _count_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    n = 0
    while f"{prefix}weight_ih_l{n}" in sd:
        n += 1
    return n


def config_from_state_dict(sd: Dict[str, Tensor], alpha: float = 0.2) -> OracleConfig:
    """Shapes follow `modules.py:12-16, 36-63, 137-164, 228-233, 260-262, 295-305`."""
    conv_w = sd["conv.conv.weight"]                    # (F, F, k)
    n_features, _, kernel = conv_w.shape
    window = sd["temporal_gat.bias"].shape[0]          # (W, W)
    f_lin = sd["feature_gat.lin.weight"]               # v2: (E, 2W)   v1: (E, W)
    use_v2 = f_lin.shape[1] == 2 * window
    n_fc = 0
    while f"forecasting_model.layers.{n_fc}.weight" in sd:
        n_fc += 1
    return OracleConfig(
        n_features=n_features,
        window_size=window,
        out_dim=sd["recon_model.fc.weight"].shape[0],
        kernel_size=kernel,
        use_gatv2=use_v2,
        feat_embed=f_lin.shape[0],
        time_embed=sd["temporal_gat.lin.weight"].shape[0],
        gru_n_layers=_count_layers(sd, "gru.gru."),
        gru_hid_dim=sd["gru.gru.weight_hh_l0"].shape[1],
        forecast_n_linear=n_fc,
        recon_n_layers=_count_layers(sd, "recon_model.decoder.rnn."),
        recon_hid_dim=sd["recon_model.decoder.rnn.weight_hh_l0"].shape[1],
        alpha=alpha,
    )


# --------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------
def conv_layer(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """`ConvLayer.forward`, reference `modules.py:18-22`.

    x (b, W, F) -> zero-pad (k-1)//2 time steps on both sides of *each window*
    -> Conv1d(F->F, k) over time -> ReLU -> (b, W, F).
    """
    k = weight.shape[2]
    pad = (k - 1) // 2
    xt = x.permute(0, 2, 1)                     # (b, F, W)
    xt = F.pad(xt, (pad, pad), value=0.0)
    y = F.relu(F.conv1d(xt, weight, bias))
    return y.permute(0, 2, 1)


def _pairwise_concat(v: Tensor) -> Tensor:
    """`_make_attention_input`, reference `modules.py:97-122` and `:195-217`.

    v (b, K, D) -> (b, K, K, 2D) with out[b, i, j] = v_i || v_j, built like the
    reference builds it (repeat_interleave / repeat / cat) so the CPU baseline
    pays the same materialisation cost.
    """
    K = v.shape[1]
    left = v.repeat_interleave(K, dim=1)        # i-major
    right = v.repeat(1, K, 1)                   # j-minor
    both = torch.cat((left, right), dim=2)
    return both.view(v.shape[0], K, K, 2 * v.shape[2])


def graph_attention(v: Tensor, lin_w: Tensor, lin_b: Tensor, a: Tensor,
                    bias: Optional[Tensor], alpha: float, use_gatv2: bool,
                    drop_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Complete-graph GAT / GATv2 over the K nodes of v (b, K, D).

    Shared body of `FeatureAttentionLayer.forward` (`modules.py:65-95`, nodes =
    features, v = x^T) and `TemporalAttentionLayer.forward` (`modules.py:166-193`,
    nodes = time steps, v = x).  Returns (sigmoid(att @ v), att).

    drop_mask, if given, is the already-scaled dropout multiplier (0 or 1/(1-p))
    applied to the attention matrix (`modules.py:90, 189`); None = eval mode.
    """
    if use_gatv2:
        a_in = _pairwise_concat(v)                                  # (b,K,K,2D)
        a_in = F.leaky_relu(F.linear(a_in, lin_w, lin_b), alpha)    # (b,K,K,E)
        e = torch.matmul(a_in, a).squeeze(3)                        # (b,K,K)
    else:
        wx = F.linear(v, lin_w, lin_b)                              # (b,K,E)
        a_in = _pairwise_concat(wx)                                 # (b,K,K,2E)
        e = F.leaky_relu(torch.matmul(a_in, a), alpha).squeeze(3)
    if bias is not None:
        e = e + bias
    att = torch.softmax(e, dim=2)
    if drop_mask is not None:
        att = att * drop_mask
    h = torch.sigmoid(torch.matmul(att, v))
    return h, att


def feature_attention(xc: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig,
                      drop_mask: Optional[Tensor] = None) -> Tensor:
    """`FeatureAttentionLayer.forward`, reference `modules.py:65-95`: (b,W,F)->(b,W,F)."""
    v = xc.permute(0, 2, 1)                                         # nodes = features
    h, _ = graph_attention(v, sd["feature_gat.lin.weight"], sd["feature_gat.lin.bias"],
                           sd["feature_gat.a"], sd.get("feature_gat.bias"),
                           cfg.alpha, cfg.use_gatv2, drop_mask)
    return h.permute(0, 2, 1)


def temporal_attention(xc: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig,
                       drop_mask: Optional[Tensor] = None) -> Tensor:
    """`TemporalAttentionLayer.forward`, reference `modules.py:166-193`: (b,W,F)->(b,W,F)."""
    h, _ = graph_attention(xc, sd["temporal_gat.lin.weight"], sd["temporal_gat.lin.bias"],
                           sd["temporal_gat.a"], sd.get("temporal_gat.bias"),
                           cfg.alpha, cfg.use_gatv2, drop_mask)
    return h


def gru_stack(x: Tensor, sd: Dict[str, Tensor], prefix: str, n_layers: int) -> Tuple[Tensor, Tensor]:
    """`nn.GRU(batch_first=True)` with h0 = 0 (reference `modules.py:233-236`, `:253-256`).

    x (b, T, in) -> (sequence output of the last layer (b, T, H),
                     final hidden of the last layer (b, H)).
    Gate order r | z | n; inter-layer dropout omitted (eval mode).
    """
    seq = x
    h = None
    for layer in range(n_layers):
        w_ih = sd[f"{prefix}weight_ih_l{layer}"]
        w_hh = sd[f"{prefix}weight_hh_l{layer}"]
        b_ih = sd[f"{prefix}bias_ih_l{layer}"]
        b_hh = sd[f"{prefix}bias_hh_l{layer}"]
        H = w_hh.shape[1]
        b, T, _ = seq.shape
        gx = F.linear(seq, w_ih, b_ih)                         # (b, T, 3H)
        h = torch.zeros(b, H, dtype=seq.dtype)
        outs: List[Tensor] = []
        for t in range(T):
            gh = F.linear(h, w_hh, b_hh)                       # (b, 3H)
            xr, xz, xn = gx[:, t].split(H, dim=1)
            hr, hz, hn = gh.split(H, dim=1)
            r = torch.sigmoid(xr + hr)
            z = torch.sigmoid(xz + hz)
            n = torch.tanh(xn + r * hn)
            h = (1.0 - z) * n + z * h
            outs.append(h)
        seq = torch.stack(outs, dim=1)
    return seq, h


def gru_stack_aten(x: Tensor, sd: Dict[str, Tensor], prefix: str, n_layers: int) -> Tuple[Tensor, Tensor]:
    """Same contract as gru_stack, through ATen's fused `aten::gru` (what the reference's nn.GRU
    dispatches to on CPU).  Used for the timed CPU baseline so the GRU costs what it costs the
    reference; tests check it agrees with the explicit gate equations above."""
    w_ih = sd[f"{prefix}weight_ih_l0"]
    rnn = torch.nn.GRU(w_ih.shape[1], sd[f"{prefix}weight_hh_l0"].shape[1], n_layers, batch_first=True).to(x.dtype)
    rnn.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    out, h = rnn(x)
    return out, h[-1]


def gru_layer(h_cat: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig) -> Tensor:
    """`GRULayer.forward`, reference `modules.py:235-238`; returns h[-1] (b, H).

    (The reference also returns out[-1] -- the last *batch element's* sequence,
    a `batch_first` quirk -- which `mtad_gat.py:73` discards.)
    """
    _, h_end = _GRU_IMPL[0](h_cat, sd, "gru.gru.", cfg.gru_n_layers)
    return h_end


def forecasting_head(h_end: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig) -> Tensor:
    """`Forecasting_Model.forward`, reference `modules.py:307-311` (eval: dropout = id)."""
    x = h_end
    for i in range(cfg.forecast_n_linear - 1):
        x = F.relu(F.linear(x, sd[f"forecasting_model.layers.{i}.weight"],
                            sd[f"forecasting_model.layers.{i}.bias"]))
    i = cfg.forecast_n_linear - 1
    return F.linear(x, sd[f"forecasting_model.layers.{i}.weight"],
                    sd[f"forecasting_model.layers.{i}.bias"])


def reconstruction_head(h_end: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig) -> Tensor:
    """`ReconstructionModel.forward`, reference `modules.py:276-283`.

    Note the reference's `repeat_interleave(W, dim=1).view(b, W, -1)`
    (`modules.py:279`): decoder input element (t, j) is h_end[(t*H + j) // W],
    *not* h_end repeated W times.
    """
    b = h_end.shape[0]
    W = cfg.window_size
    rep = h_end.repeat_interleave(W, dim=1).view(b, W, -1)
    dec, _ = _GRU_IMPL[0](rep, sd, "recon_model.decoder.rnn.", cfg.recon_n_layers)
    return F.linear(dec, sd["recon_model.fc.weight"], sd["recon_model.fc.bias"])


_GRU_IMPL = [gru_stack]


# --------------------------------------------------------------------------
# whole forward
# --------------------------------------------------------------------------
def forward(x: Tensor, sd: Dict[str, Tensor], alpha: float = 0.2,
            return_stages: bool = False, aten_gru: bool = False):
    """`MTAD_GAT.forward`, reference `mtad_gat.py:64-79` (eval mode).

    x (b, W, F) -> (predictions (b, out), recons (b, W, out)); with
    return_stages also a dict of the intermediate tensors.
    """
    cfg = config_from_state_dict(sd, alpha)
    sd = {k: v.to(x.dtype) for k, v in sd.items()}
    _GRU_IMPL[0] = gru_stack_aten if aten_gru else gru_stack
    xc = conv_layer(x, sd["conv.conv.weight"], sd["conv.conv.bias"])
    h_feat = feature_attention(xc, sd, cfg)
    h_temp = temporal_attention(xc, sd, cfg)
    h_cat = torch.cat([xc, h_feat, h_temp], dim=2)
    h_end = gru_layer(h_cat, sd, cfg)
    preds = forecasting_head(h_end, sd, cfg)
    recons = reconstruction_head(h_end, sd, cfg)
    if return_stages:
        return preds, recons, dict(xc=xc, h_feat=h_feat, h_temp=h_temp,
                                   h_cat=h_cat, h_end=h_end)
    return preds, recons


def forward_chunked(x: Tensor, sd: Dict[str, Tensor], alpha: float = 0.2,
                    chunk: int = 64):
    """forward() over batch chunks -- bounds the (b, K, K, 2D) materialisation."""
    ps, rs = [], []
    for i in range(0, x.shape[0], chunk):
        p, r = forward(x[i:i + chunk], sd, alpha)
        ps.append(p)
        rs.append(r)
    return torch.cat(ps), torch.cat(rs)
