"""Live check of the oracle against the unmodified reference, when /root/reference is mounted
(build container only; the GPU box relies on the committed golden vectors)."""
import os
import sys

import pytest
import torch

from oracle import mtad_gat_oracle as oracle

REF = os.environ.get("MTAD_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "mtad_gat.py")), reason="reference not mounted")


def _ref_class():
    """Import the reference's MTAD_GAT without disturbing our own `mtad_gat` module."""
    import importlib.util
    saved = {k: sys.modules.pop(k) for k in ("mtad_gat", "modules") if k in sys.modules}
    sys.path.insert(0, REF)
    try:
        spec = importlib.util.spec_from_file_location("ref_mtad_gat", os.path.join(REF, "mtad_gat.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(REF)
        for k in ("mtad_gat", "modules"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    return mod.MTAD_GAT


CONFIGS = [
    dict(n_features=5, window_size=9, out_dim=2, kernel_size=3, use_gatv2=True, gru_hid_dim=12,
         forecast_n_layers=1, forecast_hid_dim=10, recon_hid_dim=11),
    dict(n_features=11, window_size=7, out_dim=11, kernel_size=5, use_gatv2=False, feat_gat_embed_dim=4,
         time_gat_embed_dim=9, gru_n_layers=3, gru_hid_dim=17, forecast_n_layers=3, forecast_hid_dim=8,
         recon_n_layers=2, recon_hid_dim=13, alpha=0.3),
    dict(n_features=33, window_size=70, out_dim=1, kernel_size=7, use_gatv2=True, gru_hid_dim=150,
         forecast_n_layers=3, forecast_hid_dim=150, recon_hid_dim=150),
]


@pytest.mark.parametrize("kw", CONFIGS)
def test_oracle_equals_live_reference(kw):
    Ref = _ref_class()
    torch.manual_seed(3)
    ref = Ref(**kw).eval()
    with torch.no_grad():
        ref.feature_gat.bias.normal_()
        ref.temporal_gat.bias.normal_()
        x = torch.rand(4, kw["window_size"], kw["n_features"])
        p_ref, r_ref = ref(x)
        p, r = oracle.forward(x, ref.state_dict(), alpha=kw.get("alpha", 0.2))
    assert (p - p_ref).abs().max().item() <= 2e-6
    assert (r - r_ref).abs().max().item() <= 2e-6


def test_our_constructor_matches_reference_signature_and_init():
    import inspect
    from mtad_gat import MTAD_GAT
    Ref = _ref_class()
    assert str(inspect.signature(MTAD_GAT.__init__)) == str(inspect.signature(Ref.__init__))
    kw = CONFIGS[1]
    torch.manual_seed(11)
    a = Ref(**kw).state_dict()
    torch.manual_seed(11)
    b = MTAD_GAT(**kw).state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k
