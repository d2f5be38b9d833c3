"""The oracle (oracle/mtad_gat_oracle.py) pinned against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import pytest
import torch

from helpers import ALL_CASES, Case
from oracle import mtad_gat_oracle as oracle

SMALL = [c for c in ALL_CASES if c != "syn_c4"]


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_forward_matches_reference(name):
    case = Case(name)
    sd = case.state_dict()
    with torch.no_grad():
        p, r, st = oracle.forward(case.x, sd, alpha=case.kwargs["alpha"], return_stages=True)
    # same ATen kernels, same op order as the reference: agreement at float32 rounding level
    assert (p - case.preds).abs().max().item() <= 2e-6
    assert (r - case.recons).abs().max().item() <= 2e-6
    for k in ("xc", "h_feat", "h_temp", "h_end"):
        assert (st[k] - case.stages[k]).abs().max().item() <= 2e-6, k


@pytest.mark.parametrize("name", SMALL)
def test_oracle_float64_matches_reference_float64(name):
    case = Case(name)
    sd = {k: v.double() for k, v in case.state_dict().items()}
    with torch.no_grad():
        p, r = oracle.forward(case.x.double(), sd, alpha=case.kwargs["alpha"])
    assert (p - case.preds64).abs().max().item() <= 1e-10
    assert (r - case.recons64).abs().max().item() <= 1e-10


@pytest.mark.parametrize("name", ["smap", "syn_v1_small"])
def test_aten_gru_agrees_with_gate_equations(name):
    case = Case(name)
    sd = case.state_dict()
    with torch.no_grad():
        p0, r0 = oracle.forward(case.x, sd, alpha=case.kwargs["alpha"])
        p1, r1 = oracle.forward(case.x, sd, alpha=case.kwargs["alpha"], aten_gru=True)
    assert (p0 - p1).abs().max().item() <= 2e-6 and (r0 - r1).abs().max().item() <= 2e-6


def test_config_is_derived_from_state_dict():
    case = Case("syn_v1_small")
    cfg = oracle.config_from_state_dict(case.state_dict(), alpha=0.2)
    kw = case.kwargs
    assert (cfg.n_features, cfg.window_size, cfg.out_dim, cfg.kernel_size) == (
        kw["n_features"], kw["window_size"], kw["out_dim"], kw["kernel_size"])
    assert cfg.use_gatv2 is False and cfg.gru_n_layers == 2 and cfg.recon_n_layers == 2
    assert cfg.forecast_n_linear == kw["forecast_n_layers"] + 1


def test_reconstruction_input_quirk():
    """Decoder input (t, j) = h_end[(t*H + j) // W], not h_end repeated W times (modules.py:279)."""
    H, W = 6, 4
    h = torch.arange(H, dtype=torch.float32)[None]
    rep = h.repeat_interleave(W, dim=1).view(1, W, -1)
    for t in range(W):
        for j in range(H):
            assert rep[0, t, j].item() == (t * H + j) // W


@pytest.mark.parametrize("name", ["smap_wide", "msl_c1"])
def test_oracle_matches_reference_on_wide_fixtures(name):
    """>= 256 windows incl. a ragged tail (SURVEY.md section 8d's gate), and the C1 input statistics
    (sine + Bernoulli columns, values outside [0,1]) through the MSL checkpoint with its 1.8e23 biases."""
    from helpers import WideCase
    case = WideCase(name)
    with torch.no_grad():
        p, r = oracle.forward(case.x, case.state_dict(), alpha=case.kwargs["alpha"], aten_gru=True)
    assert (p - case.preds).abs().max().item() <= 2e-6
    assert (r - case.recons).abs().max().item() <= 5e-6
