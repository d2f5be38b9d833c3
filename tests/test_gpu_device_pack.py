"""mtadgat_update_weights_device: the weight image rebuilt on the GPU from the parameters (after optimizer.step(),
reference training.py:127) must be the image the host packer produces from the same parameters."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CONFIGS = {
    "msl_shape": dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3,
                      forecast_hid_dim=150, recon_hid_dim=150, dropout=0.3, alpha=0.2),
    "odd_shapes": dict(n_features=12, window_size=30, out_dim=12, kernel_size=5, gru_hid_dim=40, forecast_n_layers=2,
                       forecast_hid_dim=36, recon_hid_dim=44, dropout=0.3, alpha=0.2),
    "gat_v1_embed": dict(n_features=9, window_size=16, out_dim=3, kernel_size=5, use_gatv2=False, feat_gat_embed_dim=5,
                         time_gat_embed_dim=3, gru_hid_dim=33, forecast_n_layers=1, forecast_hid_dim=40, recon_hid_dim=35),
    "stacked": dict(n_features=7, window_size=20, out_dim=7, kernel_size=3, gru_n_layers=2, gru_hid_dim=24, recon_n_layers=2,
                    recon_hid_dim=20, forecast_n_layers=2, forecast_hid_dim=16),
    "wide_hidden": dict(n_features=6, window_size=12, out_dim=2, kernel_size=3, gru_hid_dim=200, recon_hid_dim=180,
                        forecast_n_layers=1, forecast_hid_dim=8),          # decoder input folds more than 8 entries per step
    "many_nodes": dict(n_features=5, window_size=140, out_dim=5, kernel_size=3, gru_hid_dim=16, recon_hid_dim=16,
                       forecast_n_layers=1, forecast_hid_dim=8),           # temporal layer beyond the fused kernel
    "long_embedding": dict(n_features=4, window_size=300, out_dim=4, kernel_size=3, gru_hid_dim=16, recon_hid_dim=16,
                           forecast_n_layers=1, forecast_hid_dim=8),       # feature layer: 600 embedding columns = three chunks of
                                                                           # k_gat_colorder's ballot ranks, the last one partial
}


def _perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_((0.05 * torch.randn(p.shape, generator=g)).to(p.device))
        for a in (model.feature_gat.a, model.temporal_gat.a):      # move embedding columns across the sign boundary
            flip = (torch.rand(a.shape, generator=g) < 0.3).to(a.device)
            a.mul_(torch.where(flip, -1.0, 1.0))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_device_repack_equals_the_host_packer(name, gpu_device):
    from mtad_gat import MTAD_GAT
    torch.manual_seed(0)
    model = MTAD_GAT(**CONFIGS[name]).to(gpu_device).eval()
    eng = model._sync_engine(gpu_device)                     # first load: host packer
    for rnd in range(3):
        _perturb(model, 10 + rnd)
        sd = model.state_dict()
        assert eng.update_weights_device(sd, gpu_device), "the library declined the device-side re-pack"
        img_dev = eng.read_packed(gpu_device)
        eng.load_weights(sd, gpu_device, allow_device_pack=False)
        img_host = eng.read_packed(gpu_device)
        # compared as bit patterns (the image also holds integer index maps).  Plain copies and the scaled / summed
        # attention columns are bit-identical; the folded decoder input sums the same terms in another order (host:
        # differences of prefix sums) -- last-bit differences only
        mism = img_dev.view(torch.int32) != img_host.view(torch.int32)
        for off, n in eng.derived_regions():      # split-bf16 packs: produced from the fp32 packs by the same kernel on both paths
            mism[off:off + n] = False
        assert mism.float().mean().item() < 0.02, (name, rnd, int(mism.sum()))
        if mism.any():
            a, b = img_dev[mism], img_host[mism]
            assert torch.isfinite(a).all() and torch.isfinite(b).all()
            assert (a - b).abs().max().item() <= 1e-6 * max(b.abs().max().item(), 1.0), (name, rnd, (a - b).abs().max().item())
        # and the forward on the device-packed image is the forward on the host-packed one
        x = torch.rand(5, CONFIGS[name]["window_size"], CONFIGS[name]["n_features"], device=gpu_device)
        with torch.no_grad():
            ph, rh = eng.forward(x)
            assert eng.update_weights_device(sd, gpu_device)
            pd, rd = eng.forward(x)
        assert (ph - pd).abs().max().item() <= 1e-6 and (rh - rd).abs().max().item() <= 1e-6


@pytest.mark.parametrize("name", ["odd_shapes", "gat_v1_embed"])
def test_training_loop_on_device_packed_weights_tracks_the_host_packed_loop(name, gpu_device, monkeypatch):
    """Adam steps with the image re-packed on the device after every step vs. the same loop through the host packer
    (GATv2 and GAT v1: the HIP training step covers both)."""
    from mtad_gat import MTAD_GAT
    kw = dict(CONFIGS[name], dropout=0.0)

    def run(host_pack):
        torch.manual_seed(0)
        m = MTAD_GAT(**kw).to(gpu_device).train()
        m.device_repack = not host_pack
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(5)
        x = torch.rand(48, kw["window_size"], kw["n_features"], generator=g).to(gpu_device)
        y = torch.rand(48, kw["out_dim"], generator=g).to(gpu_device)
        losses = []
        for _ in range(6):
            opt.zero_grad()
            p, r = m(x)
            assert m.grad_path == "hip"
            loss = torch.sqrt(F.mse_loss(y, p)) + torch.sqrt(F.mse_loss(x[:, :, : r.shape[2]], r))
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, [q.detach().clone() for q in m.parameters()]

    l_dev, p_dev = run(False)
    l_host, p_host = run(True)
    assert l_dev[-1] < l_dev[0]
    for a, b in zip(l_dev, l_host):
        assert abs(a - b) <= 1e-5, (l_dev, l_host)
    for a, b in zip(p_dev, p_host):
        assert (a - b).abs().max().item() <= 1e-5
