"""Odd and tiny shapes straight against the oracle (which is pinned by the golden vectors): single
feature / tiny windows / kernel wider than the window / non-multiple-of-anything dims / ragged node
blocks, v1 and v2 attention.  Tolerance 1e-5 (north_star)."""
import pytest
import torch

from helpers import gate
from oracle import mtad_gat_oracle as oracle

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(n_features=1, window_size=3, out_dim=1, kernel_size=3, gru_hid_dim=5, forecast_hid_dim=4, recon_hid_dim=3),
    dict(n_features=2, window_size=5, out_dim=2, kernel_size=9, gru_hid_dim=32, forecast_hid_dim=33, recon_hid_dim=31),
    dict(n_features=3, window_size=64, out_dim=1, kernel_size=1, gru_hid_dim=64, forecast_n_layers=2, recon_hid_dim=96),
    dict(n_features=64, window_size=65, out_dim=64, kernel_size=3, use_gatv2=False, gru_hid_dim=17, recon_hid_dim=150),
    dict(n_features=33, window_size=31, out_dim=5, kernel_size=5, feat_gat_embed_dim=1, time_gat_embed_dim=1,
         gru_hid_dim=97, recon_hid_dim=129, recon_n_layers=3, gru_n_layers=3),
    dict(n_features=20, window_size=128, out_dim=20, kernel_size=7, gru_hid_dim=256, recon_hid_dim=200, alpha=0.7),
    dict(n_features=129, window_size=40, out_dim=3, kernel_size=3, gru_hid_dim=40, recon_hid_dim=40),   # un-fused path (K > 128)
    # temporal layer fused with 6 keys per lane and a staging batch beyond the register budget, feature layer un-fused
    # both layers un-fused with node rows that are NOT whole 16-byte words (the wide kernels' guarded V staging instead of the
    # LDS-DMA tiles), an odd number of 16-column output tiles and a last key tile with a single real row (K = 129 + 16 k)
    dict(n_features=145, window_size=150, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24),
    dict(n_features=128, window_size=96, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24),
    dict(n_features=10, window_size=100, out_dim=1, kernel_size=7, use_gatv2=False, gru_hid_dim=30, recon_hid_dim=30),
    dict(n_features=90, window_size=17, out_dim=4, kernel_size=5, gru_hid_dim=20, recon_hid_dim=20, feat_gat_embed_dim=7),
]


@pytest.mark.parametrize("kw", SHAPES, ids=lambda k: f"F{k['n_features']}W{k['window_size']}")
def test_shape_against_oracle(kw, gpu_device):
    from mtad_gat import MTAD_GAT
    torch.manual_seed(17)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    for b in (1, 35):
        x = torch.rand(b, kw["window_size"], kw["n_features"])
        with torch.no_grad():
            p_ref, r_ref = oracle.forward(x, model.state_dict(), alpha=kw.get("alpha", 0.2))
            m = model.to(gpu_device)
            p, r = m(x.to(gpu_device))
            model = m.cpu()
        gate(p, p_ref, what=f"preds b={b}")
        gate(r, r_ref, what=f"recons b={b}")


@pytest.mark.parametrize("kw", [
    # decoder input spread over more than 8 h_end entries per step (3-chunk folded input), stacked layers
    dict(n_features=4, window_size=5, out_dim=2, kernel_size=3, gru_hid_dim=96, recon_hid_dim=40, gru_n_layers=2, recon_n_layers=2),
    dict(n_features=7, window_size=12, out_dim=7, kernel_size=5, gru_hid_dim=33, recon_hid_dim=65, use_gatv2=False),
], ids=["F4W5", "F7W12"])
def test_large_and_small_batch_kernels_on_odd_shapes(kw, gpu_device):
    """20 000 windows go through the register-resident GRU kernels (70 000: two window groups per wave), 40
    through the hidden-tile-split ones; all must match the oracle."""
    from mtad_gat import MTAD_GAT
    torch.manual_seed(23)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    x = torch.rand(70000, kw["window_size"], kw["n_features"])
    with torch.no_grad():
        p_ref, r_ref = oracle.forward(x[:40], model.state_dict(), alpha=kw.get("alpha", 0.2))
        m = model.to(gpu_device)
        p_big, r_big = m(x[:20000].to(gpu_device))
        m._engine.set_chunk_windows(1 << 17)                       # keep the 70 000 windows in one launch
        p_huge, r_huge = m(x.to(gpu_device))
        p_small, r_small = m(x[:40].to(gpu_device))
    gate(p_big[:40], p_ref, what="preds, large batch")
    gate(r_big[:40], r_ref, what="recons, large batch")
    gate(p_huge[:40], p_ref, what="preds, full-machine batch")
    gate(r_huge[:40], r_ref, what="recons, full-machine batch")
    gate(p_small, p_ref, what="preds, small batch")
    gate(r_small, r_ref, what="recons, small batch")


def test_unsupported_shapes_fail_loudly(gpu_device):
    from mtad_gat import MTAD_GAT
    model = MTAD_GAT(n_features=4, window_size=600, out_dim=1).eval().to(gpu_device)
    with pytest.raises(RuntimeError, match="512"):
        model(torch.rand(1, 600, 4, device=gpu_device))
    model = MTAD_GAT(n_features=4, window_size=10, out_dim=1, gru_hid_dim=300).eval().to(gpu_device)
    with pytest.raises(RuntimeError, match="hidden"):
        model(torch.rand(1, 10, 4, device=gpu_device))


def _random_shapes(count, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(count):
        F = rnd.choice([3, 5, 9, 13, 21, 38, 55, 64])
        W = rnd.choice([8, 17, 30, 64, 100, 120])
        out.append(dict(n_features=F, window_size=W, out_dim=rnd.choice([1, min(F, 4), F]), kernel_size=rnd.choice([3, 5, 7]),
                        use_gatv2=rnd.random() < 0.75, gru_n_layers=rnd.choice([1, 1, 2]), gru_hid_dim=rnd.choice([33, 40, 64, 96, 100, 128, 150, 160]),
                        forecast_n_layers=rnd.choice([1, 3]), forecast_hid_dim=rnd.choice([24, 150]), recon_n_layers=rnd.choice([1, 1, 2]),
                        recon_hid_dim=rnd.choice([35, 44, 70, 100, 150]), alpha=0.2))
    return out


@pytest.mark.parametrize("idx", list(range(10)))
def test_recurrence_kernel_bands_on_random_shapes(idx, gpu_device):
    """Random model shapes through every band of the batch-size dispatch of the default arithmetic -- 3 000 windows
    (hidden-tile split on split operands), 8 192 (its largest batch), 9 000 (chunk-major), also as stride-1 windows of a series
    (shared convolution rows) -- against precision = "fp32_strict" (fp32-MFMA kernels, themselves pinned to the oracle at
    small batches by the tests above): <= 2e-6 of the output scale, and the first windows against the oracle."""
    from mtad_gat import MTAD_GAT
    kw = _random_shapes(10, 77)[idx]
    torch.manual_seed(100 + idx)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    W, F = kw["window_size"], kw["n_features"]
    series = torch.rand(W + 9000 - 1, F)
    with torch.no_grad():
        x_head = torch.stack([series[i:i + W] for i in range(6)])
        p_ref, r_ref = oracle.forward(x_head, model.state_dict(), alpha=kw["alpha"])
        m = model.to(gpu_device)
        sd = series.to(gpu_device)
        for n in (3000, 8192, 9000):
            x = torch.stack([sd[i:i + W] for i in range(n)])
            m.precision = "fp32_strict"
            ps, rs = m(x)
            m.precision = "fp32"
            pd, rd = m(x)
            pser, rser = m.forward_series(sd, start=0, stride=1, count=n)
            tol = 2e-6 * max(1.0, rs.abs().max().item(), ps.abs().max().item())
            assert (pd - ps).abs().max().item() <= tol and (rd - rs).abs().max().item() <= tol, (kw, n)
            assert torch.equal(pser, pd) and torch.equal(rser, rd), (kw, n)
            gate(pd[:6], p_ref, what=f"preds, {n} windows")
            gate(rd[:6], r_ref, what=f"recons, {n} windows")


def test_config4_chunked_batch_against_the_oracle(gpu_device):
    """BASELINE config 4 (F = 512, W = 256, out_dim = 512; the un-fused wide attention path: projections through HBM,
    k_gat_wide) on a batch that the library walks in several chunks (64 = 24 + 24 + 16 windows), every window against
    oracle.forward_chunked (the reference's algorithm on the CPU, two windows at a time: it materialises 0.8 GB of pairwise
    inputs per window)."""
    from mtad_gat import MTAD_GAT
    kw = dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
              recon_hid_dim=150)
    torch.manual_seed(0)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    g = torch.Generator().manual_seed(31)
    x = torch.rand(64, 256, 512, generator=g)
    with torch.no_grad():
        p_ref, r_ref = oracle.forward_chunked(x, model.state_dict(), alpha=0.2, chunk=2)
        m = model.to(gpu_device)
        eng = m._sync_engine(gpu_device)
        eng.set_chunk_windows(24)
        p, r = m(x.to(gpu_device))
        eng.set_chunk_windows(64)
        p1, r1 = m(x.to(gpu_device))
        # the split-bf16 builds of the wide path's convolution and projections (k_conv_x3, k_rowgemm_x3: chunks of >= 65 536 rows
        # by default -- BASELINE config 4's chunk of 896 windows), forced here
        eng.set_option("conv_kernel", 2)
        eng.set_option("rowgemm_kernel", 2)
        try:
            p2, r2 = m(x.to(gpu_device))
        finally:
            eng.set_option("conv_kernel", 0)
            eng.set_option("rowgemm_kernel", 0)
    assert torch.equal(p, p1) and torch.equal(r, r1)            # chunking does not change a window's result
    gate(p, p_ref, what="config 4 preds, 64 windows in 3 chunks")
    gate(r, r_ref, what="config 4 recons, 64 windows in 3 chunks")
    assert not (torch.equal(p2, p1) and torch.equal(r2, r1)), "the split-bf16 kernels did not run"
    gate(p2, p_ref, what="config 4 preds, split-bf16 convolution / projections")
    gate(r2, r_ref, what="config 4 recons, split-bf16 convolution / projections")
    assert (p2 - p1).abs().max().item() <= 2e-6 * max(1.0, p1.abs().max().item()) and (r2 - r1).abs().max().item() <= 2e-6 * max(1.0, r1.abs().max().item())


@pytest.mark.parametrize("precision", ["fp32", "fp32_strict"])
def test_config4_bench_chunk_against_the_oracle(precision, gpu_device):
    """BASELINE config 4 at the dispatch bench.py's `config4_f512_w256` number runs on: ONE call of 896 windows (the library's own
    chunk at this shape: F = 512, W = 256, out_dim = 512) with every engine option at its default, so the chunk goes through the
    LDS-shared split-bf16 GEMMs (k_conv_x3s, k_rowgemm_x3s: 229 376 rows per launch), k_gat_wide / k_gat_wide_os and the
    896-window recurrences.  The first two, two mid-chunk and the last two windows against oracle.forward_chunked (the reference's
    algorithm, modules.py:65-95, :166-193, mtad_gat.py:64-79), in both fp32 arithmetics."""
    from mtad_gat import MTAD_GAT
    kw = dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
              recon_hid_dim=150)
    torch.manual_seed(0)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    n = 896
    g = torch.Generator().manual_seed(47)
    x = torch.rand(n, 256, 512, generator=g)
    pick = torch.tensor([0, 1, 447, 448, n - 2, n - 1])
    with torch.no_grad():
        p_ref, r_ref = oracle.forward_chunked(x[pick], model.state_dict(), alpha=0.2, chunk=2)
        m = model.to(gpu_device)
        m.precision = precision
        eng = m._sync_engine(gpu_device)
        assert eng.chunk_windows() >= n, "the library no longer takes 896 windows of this shape in one chunk: update the test and bench.py"
        p, r = m(x.to(gpu_device))
    gate(p[pick.to(gpu_device)], p_ref, what=f"config 4 preds, one 896-window chunk, {precision}")
    gate(r[pick.to(gpu_device)], r_ref, what=f"config 4 recons, one 896-window chunk, {precision}")


def test_many_row_gemms_share_weight_words_through_lds(gpu_device):
    """Launches of >= 131 072 rows of the split-bf16 row GEMM and of the wide models' convolution run as workgroups of four
    waves that share each chunk's weight words through LDS (k_rowgemm_x3s / k_conv_x3s, csrc/mtadgat_kernels.hip; the Linear
    / Conv1d layers modules.py:18-22, :176-181).  A wide model (W = 256 > 128 nodes: projections through memory; F = 136: the
    straight-from-memory convolution) on 1 024 windows = 262 144 (window, step) rows: bit-equal to the one-wave kernels
    (engine option "gemm_lds" = 1), and the first windows against the oracle."""
    from mtad_gat import MTAD_GAT
    kw = dict(n_features=136, window_size=256, out_dim=136, kernel_size=7, gru_hid_dim=64, forecast_n_layers=1, forecast_hid_dim=64,
              recon_hid_dim=64)
    torch.manual_seed(3)
    model = MTAD_GAT(**kw).eval()
    with torch.no_grad():
        model.feature_gat.bias.normal_()
        model.temporal_gat.bias.normal_()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1024, 256, 136, generator=g)
    with torch.no_grad():
        p_ref, r_ref = oracle.forward_chunked(x[:4], model.state_dict(), alpha=0.2, chunk=2)
        m = model.to(gpu_device)
        eng = m._sync_engine(gpu_device)
        eng.set_chunk_windows(1024)
        xd = x.to(gpu_device)
        p, r = m(xd)
        eng.set_option("gemm_lds", 1)
        try:
            p1, r1 = m(xd)
            eng.set_option("conv_kernel", 1)
            eng.set_option("rowgemm_kernel", 1)
            p2, r2 = m(xd)                               # fp32-MFMA convolution / projections: a different arithmetic
        finally:
            eng.set_option("gemm_lds", 0)
            eng.set_option("conv_kernel", 0)
            eng.set_option("rowgemm_kernel", 0)
    assert torch.equal(p, p1) and torch.equal(r, r1)
    assert not (torch.equal(p2, p1) and torch.equal(r2, r1)), "the split-bf16 kernels did not run"
    assert (p2 - p1).abs().max().item() <= 2e-6 * max(1.0, p1.abs().max().item()) and (r2 - r1).abs().max().item() <= 2e-6 * max(1.0, r1.abs().max().item())
    gate(p[:4], p_ref, what="wide model preds, 1 024 windows per chunk")
    gate(r[:4], r_ref, what="wide model recons, 1 024 windows per chunk")
