"""The two-fp16-piece build of the fused attention layer (reference FeatureAttentionLayer.forward modules.py:65-95 and
TemporalAttentionLayer.forward :166-193): k_gath (csrc/mtadgat_gat.hip: the row-split kernel with the node vectors split once
per window; engine option gat_kernel = 3; in normal use it serves batches of 4096 windows and more -- the 20 000- and
65 573-window tests of test_gpu_parity.py go through it).  The testing hook forces it at fixture size, so every window is
compared with the reference's golden outputs, and odd node counts / embedding widths with the oracle.  (Round 4's
column-sliced k_gat2 lost to it on every shipped shape and was removed in round 5; profiles/ubench_gat2.hip keeps its pair
phase as a microbenchmark.)  Also here: the window convolution on fp16 pieces (k_conv_win)."""
import pytest
import torch

from helpers import Case, WideCase, gate
from oracle import mtad_gat_oracle as oracle

pytestmark = pytest.mark.gpu


def _engine(model, dev):
    return model._sync_engine(dev)


KERNELS = [3]


@pytest.mark.parametrize("gk", KERNELS)
@pytest.mark.parametrize("name", ["msl", "smap", "smd_1_1", "syn_v2_embed", "syn_v1_small"])
def test_fixture_windows_through_the_fp16_piece_kernel(name, gk, gpu_device):
    case = Case(name)
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    x = case.x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("gat_kernel", gk)
        p2, r2 = model(x)
        eng.set_option("gat_kernel", 1)
        p1, r1 = model(x)
        eng.set_option("gat_kernel", 0)
    dp = gate(p2, case.preds, case.preds64, what=f"{name} predictions (k_gath)")
    dr = gate(r2, case.recons, case.recons64, what=f"{name} recons (k_gath)")
    print(f"{name}: k_gath |preds-ref|={dp:.2e} |recons-ref|={dr:.2e}  vs k_gat: {(p2 - p1).abs().max().item():.2e} {(r2 - r1).abs().max().item():.2e}")
    assert (p2 - p1).abs().max().item() <= 2e-6 and (r2 - r1).abs().max().item() <= 2e-6


@pytest.mark.parametrize("gk", KERNELS)
@pytest.mark.parametrize("name", ["msl_wide", "smd_1_1_wide", "msl_c1"])
def test_wide_fixtures_through_the_fp16_piece_kernel(name, gk, gpu_device):
    """300 / 320 windows per shipped checkpoint incl. the C1 input statistics (values outside [0, 1])."""
    case = WideCase(name)
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    x = case.x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("gat_kernel", gk)
        preds, recons = model(x)
        p_split = torch.cat([model(x[lo:lo + 77].contiguous())[0] for lo in range(0, x.shape[0], 77)])
        eng.set_option("gat_kernel", 0)
    assert torch.equal(p_split, preds)                      # windows are independent of their batch
    gate(preds, case.preds, case.preds64, what=f"{name} predictions (k_gath)")
    gate(recons, case.recons, case.recons64, what=f"{name} recons (k_gath)")


SHAPES = [
    # (nodes of the two layers = F and W) accumulator blocks 4..13, embedding widths that leave waves without columns,
    # node dimensions of one to five 32-feature chunks, feature counts that are / are not multiples of 4
    dict(n_features=25, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=40, recon_hid_dim=40),
    dict(n_features=38, window_size=100, out_dim=38, kernel_size=7, gru_hid_dim=40, recon_hid_dim=40),
    dict(n_features=32, window_size=104, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24),
    dict(n_features=41, window_size=73, out_dim=3, kernel_size=5, gru_hid_dim=33, recon_hid_dim=20, feat_gat_embed_dim=3, time_gat_embed_dim=50),
    dict(n_features=64, window_size=65, out_dim=4, kernel_size=3, gru_hid_dim=17, recon_hid_dim=30),
    dict(n_features=100, window_size=28, out_dim=1, kernel_size=3, gru_hid_dim=20, recon_hid_dim=20, time_gat_embed_dim=9),
    dict(n_features=57, window_size=88, out_dim=1, kernel_size=7, gru_hid_dim=20, recon_hid_dim=20, alpha=0.6),
    dict(n_features=80, window_size=96, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24, feat_gat_embed_dim=60, time_gat_embed_dim=50),
]


SHAPES_H = [   # up to 128 nodes, fewer than 25, GAT (v1)
    dict(n_features=128, window_size=96, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24),
    dict(n_features=10, window_size=120, out_dim=1, kernel_size=7, use_gatv2=False, gru_hid_dim=30, recon_hid_dim=30),
    dict(n_features=3, window_size=64, out_dim=1, kernel_size=1, gru_hid_dim=64, forecast_n_layers=2, recon_hid_dim=96),
]


@pytest.mark.parametrize("gk", KERNELS)
@pytest.mark.parametrize("kw", SHAPES + SHAPES_H, ids=lambda k: f"F{k['n_features']}W{k['window_size']}")
def test_shapes_against_the_oracle(kw, gk, gpu_device):
    from mtad_gat import MTAD_GAT
    torch.manual_seed(29)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
        # both signs of `a` must occur in every wave's column slice for the test to mean something
        assert not kw.get("use_gatv2", True) or ((model.temporal_gat.a > 0).any() and (model.temporal_gat.a < 0).any())
    x = torch.rand(19, kw["window_size"], kw["n_features"])
    with torch.no_grad():
        p_ref, r_ref = oracle.forward(x, model.state_dict(), alpha=kw.get("alpha", 0.2))
        m = model.to(gpu_device)
        eng = _engine(m, gpu_device)
        eng.set_option("gat_kernel", gk)
        p, r = m(x.to(gpu_device))
        eng.set_option("gat_kernel", 1)
        p1, r1 = m(x.to(gpu_device))
    gate(p, p_ref, what="preds (k_gath)")
    gate(r, r_ref, what="recons (k_gath)")
    gate(p1, p_ref, what="preds (k_gat)")
    assert (p - p1).abs().max().item() <= 2e-6 and (r - r1).abs().max().item() <= 2e-6


@pytest.mark.parametrize("gk", KERNELS)
def test_large_inputs_fall_back_to_the_row_split_kernel(gk, gpu_device):
    """Convolution outputs of 2^15 and more do not fit the fp16 pieces: the device-side range guard hands the launch to
    k_gat's bf16-piece build (both kernels are enqueued, one of them returns at once)."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(64, 100, 55, generator=g) * 3e4).to(gpu_device)
    with torch.no_grad():
        eng.set_option("gat_kernel", gk)
        p2, r2 = model(x)
        eng.set_option("gat_kernel", 1)
        p1, r1 = model(x)
        eng.set_option("gat_kernel", 0)
    assert torch.isfinite(p2).all() and torch.isfinite(r2).all()
    assert torch.equal(p2, p1) and torch.equal(r2, r1)


@pytest.mark.parametrize("gk", KERNELS)
def test_sign_flips_of_a_reach_the_fp16_piece_pack(gk, gpu_device):
    """In-place weight changes that flip signs of the attention vector `a` change the column order of the pack (positive
    columns first) on the device-side re-pack path; outputs must track the fp32 kernel's."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(40, 100, 55, generator=g).to(gpu_device)
    with torch.no_grad():
        for _ in range(2):
            flip = (torch.rand(model.temporal_gat.a.shape, generator=g) < 0.3).to(gpu_device)
            model.temporal_gat.a.mul_(torch.where(flip, -1.0, 1.0))
            model.feature_gat.a.mul_(-1.0)
            eng = _engine(model, gpu_device)
            eng.set_option("gat_kernel", gk)
            p2, r2 = model(x)
            eng.set_option("gat_kernel", 1)
            p1, r1 = model(x)
            eng.set_option("gat_kernel", 0)
            p_ref, r_ref = oracle.forward(x.cpu(), {k: v.cpu() for k, v in model.state_dict().items()}, alpha=0.2)
            gate(p2, p_ref, what="preds after sign flips")
            gate(r2, r_ref, what="recons after sign flips")
            assert (p2 - p1).abs().max().item() <= 2e-6 and (r2 - r1).abs().max().item() <= 2e-6


@pytest.mark.parametrize("name", ["msl", "smap", "smd_1_1", "syn_v2_embed", "syn_v1_small"])
def test_window_convolution_on_fp16_pieces(name, gpu_device):
    """k_conv_win (csrc/mtadgat_convw.hip; reference ConvLayer.forward modules.py:18-22): one workgroup per window, the window
    scaled by a power of two and split into two fp16 pieces, weights likewise -- forced at fixture size through the testing
    hook; the forward must still match the reference's golden outputs, and the fp32-MFMA convolution's to rounding."""
    case = Case(name)
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    x = case.x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("conv_kernel", 2)
        p2, r2 = model(x)
        eng.set_option("conv_kernel", 1)
        p1, r1 = model(x)
        eng.set_option("conv_kernel", 0)
    gate(p2, case.preds, case.preds64, what=f"{name} predictions (k_conv_win)")
    gate(r2, case.recons, case.recons64, what=f"{name} recons (k_conv_win)")
    assert (p2 - p1).abs().max().item() <= 2e-6 and (r2 - r1).abs().max().item() <= 2e-6


@pytest.mark.parametrize("scale", [1e-6, 1.0, 3e4, 1e9])
def test_window_convolution_at_any_input_scale(scale, gpu_device):
    """The per-window power-of-two scaling makes the fp16 pieces independent of the input's magnitude: the forward must
    track the one on the fp32-MFMA convolution for normalised, tiny and un-normalised series alike, with windows of very
    different magnitude in one batch and an outlier that sets one window's scale."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    g = torch.Generator().manual_seed(23)
    x = torch.rand(96, 100, 55, generator=g) * scale
    x[::3] *= 1e-3                                         # every third window three orders of magnitude smaller
    x[5, 17, 3] = 40.0 * scale
    x = x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("conv_kernel", 2)
        eng.set_option("gat_kernel", 1)
        p2, r2 = model(x)
        eng.set_option("conv_kernel", 1)
        p1, r1 = model(x)
        eng.set_option("conv_kernel", 0)
        eng.set_option("gat_kernel", 0)
    assert torch.isfinite(p2).all() and torch.isfinite(r2).all()
    tol = 2e-6 if scale <= 1.0 else 5e-3 * max(1.0, r1.abs().max().item())     # (pre-activations of ~1e4 and more: see test_gpu_parity)
    assert (p2 - p1).abs().max().item() <= tol and (r2 - r1).abs().max().item() <= tol


# ---- round 5: the window convolution inside the temporal layer's k_gath workgroup (csrc/mtadgat_gath.hip, CONV build) -------------
def _fused(eng, on):
    eng.set_option("conv_fused", 0 if on else 1)


@pytest.mark.parametrize("name", ["msl", "smap", "smd_1_1", "syn_v2_embed", "syn_v1_small"])
def test_convolution_inside_the_temporal_workgroup(name, gpu_device):
    """ConvLayer.forward (modules.py:18-22) computed by the workgroup that runs TemporalAttentionLayer.forward (modules.py:166-193)
    on the same window -- forced at fixture size (conv_kernel = 2, gat_kernel = 3).  Gated against the reference's outputs; and
    it is k_conv_win's arithmetic instruction for instruction, so the forward must equal the two-launch path bit for bit."""
    case = Case(name)
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    x = case.x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("conv_kernel", 2)
        eng.set_option("gat_kernel", 3)
        _fused(eng, True)
        p2, r2 = model(x)
        h2 = eng.forward(x, want_hend=True)[2]
        _fused(eng, False)
        p1, r1 = model(x)
        h1 = eng.forward(x, want_hend=True)[2]
        _fused(eng, True)
        eng.set_option("conv_kernel", 0)
        eng.set_option("gat_kernel", 0)
    gate(p2, case.preds, case.preds64, what=f"{name} predictions (fused convolution)")
    gate(r2, case.recons, case.recons64, what=f"{name} recons (fused convolution)")
    assert torch.equal(p2, p1) and torch.equal(r2, r1) and torch.equal(h2, h1)


@pytest.mark.parametrize("kw", SHAPES + SHAPES_H, ids=lambda k: f"F{k['n_features']}W{k['window_size']}")
def test_fused_convolution_shapes_against_the_oracle(kw, gpu_device):
    from mtad_gat import MTAD_GAT
    torch.manual_seed(31)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    x = torch.rand(21, kw["window_size"], kw["n_features"])
    with torch.no_grad():
        p_ref, r_ref = oracle.forward(x, model.state_dict(), alpha=kw.get("alpha", 0.2))
        m = model.to(gpu_device)
        eng = _engine(m, gpu_device)
        eng.set_option("conv_kernel", 2)
        eng.set_option("gat_kernel", 3)
        p, r = m(x.to(gpu_device))
        _fused(eng, False)
        p1, r1 = m(x.to(gpu_device))
        _fused(eng, True)
    gate(p, p_ref, what="preds (fused convolution)")
    gate(r, r_ref, what="recons (fused convolution)")
    assert torch.equal(p, p1) and torch.equal(r, r1)       # (shapes the fused form does not take run the two launches both times)


def test_fused_convolution_range_guard_is_per_window(gpu_device):
    """Windows whose convolution outputs reach 2^15 do not fit the fp16 pieces: the workgroup flags its window and k_gat's
    bf16-piece build, enqueued behind, serves exactly the flagged ones -- mixed in one batch with ordinary windows."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    eng = _engine(model, gpu_device)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(96, 100, 55, generator=g)
    x[::5] *= 3e4                                          # every fifth window un-normalised
    x[7] *= 1e-4
    x = x.to(gpu_device)
    with torch.no_grad():
        eng.set_option("conv_kernel", 2)
        eng.set_option("gat_kernel", 3)
        p2, r2 = model(x)
        small = model(x[1:2].contiguous())                 # an ordinary window on its own: the fp16-piece path
        eng.set_option("gat_kernel", 1)
        _fused(eng, False)
        p1, r1 = model(x)                                  # fp32 k_gat for every window
        _fused(eng, True)
        eng.set_option("conv_kernel", 0)
        eng.set_option("gat_kernel", 0)
    assert torch.isfinite(p2).all() and torch.isfinite(r2).all()
    # an ordinary window's temporal layer does not see its neighbours (the feature layer and the recurrences still follow the
    # batch-wide recorded maximum: three bf16 pieces in the mixed batch, two fp16 pieces alone -- both within 2e-6)
    assert (p2[1:2] - small[0]).abs().max().item() <= 2e-6 and (r2[1:2] - small[1]).abs().max().item() <= 2e-6
    tol = 5e-3 * max(1.0, r1.abs().max().item())          # (pre-activations of ~1e4 and more in the scaled windows: see test_gpu_parity)
    ordinary = torch.ones(96, dtype=torch.bool)
    ordinary[::5] = False
    assert (p2[ordinary] - p1[ordinary]).abs().max().item() <= 2e-6 and (r2[ordinary] - r1[ordinary]).abs().max().item() <= 2e-6
    assert (p2 - p1).abs().max().item() <= tol and (r2 - r1).abs().max().item() <= tol


def test_fused_convolution_reads_bfloat16_windows_and_series_views(gpu_device):
    """The convolution inside the temporal workgroup takes its window the ways k_conv_win does: float32 or bfloat16 elements,
    materialised windows or views of a device-resident series (arithmetic progression or explicit starts).  Each must equal the
    two-launch path bit for bit."""
    case = Case("smd_1_1")
    model = case.build_model().to(gpu_device)
    model.precision = "fp32"
    model.check_weight_contents = False
    eng = _engine(model, gpu_device)
    W, F = case.kwargs["window_size"], case.kwargs["n_features"]
    g = torch.Generator().manual_seed(41)
    series = torch.rand(400 + W - 1, F, generator=g).to(gpu_device)
    starts = torch.randperm(400, generator=g)[:77].to(gpu_device)
    xb = torch.rand(65, W, F, generator=g).to(gpu_device).to(torch.bfloat16)
    with torch.no_grad():
        eng.set_option("conv_kernel", 2)
        eng.set_option("gat_kernel", 3)
        outs = {}
        for fused in (True, False):
            _fused(eng, fused)
            outs[fused] = (model(xb), model.forward_series(series, start=3, stride=2, count=150), model.forward_series(series, starts=starts))
        _fused(eng, True)
        eng.set_option("conv_kernel", 0)
        eng.set_option("gat_kernel", 0)
        ref = model(torch.stack([series[s:s + W] for s in starts.tolist()]))
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert (outs[True][2][0] - ref[0]).abs().max().item() <= 2e-6 and (outs[True][2][1] - ref[1]).abs().max().item() <= 2e-6
