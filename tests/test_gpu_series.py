"""GPU-side sliding-window gather and the fused Predictor scoring pass,
against the materialised-window forward (bit-exact: same kernels, same values) and the oracle."""
import pytest
import torch

from helpers import Case, gate
from oracle import mtad_gat_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu_device):
    case = Case("smap")
    model = case.build_model().to(gpu_device)
    g = torch.Generator().manual_seed(3)
    series = torch.rand(100 + 77, case.kwargs["n_features"], generator=g)
    return case, model, series


def _windows(series, starts, w):
    # what SlidingWindowDataset.__getitem__ + default collate produce (reference utils.py:114-117)
    return torch.stack([series[s:s + w] for s in starts])


def test_forward_series_equals_materialised_windows(setup, gpu_device):
    case, model, series = setup
    w = case.kwargs["window_size"]
    sd = series.to(gpu_device)
    with torch.no_grad():
        for starts in (list(range(0, 78)), list(range(2, 78, 3)), [5, 5, 70, 0, 33, 77]):
            x = _windows(series, starts, w).to(gpu_device)
            p_ref, r_ref = model(x)
            if starts == list(range(0, 78)):
                p, r = model.forward_series(sd)
            elif starts == list(range(2, 78, 3)):
                p, r = model.forward_series(sd, start=2, stride=3)
            else:
                p, r = model.forward_series(sd, starts=torch.tensor(starts, dtype=torch.int64, device=gpu_device))
            assert torch.equal(p, p_ref) and torch.equal(r, r_ref), starts
    with pytest.raises(RuntimeError):
        model.forward_series(sd, start=0, stride=1, count=79)          # window 78 would end past the series
    with pytest.raises(RuntimeError):
        model.forward_series(sd, starts=torch.tensor([78], dtype=torch.int64, device=gpu_device))


def test_score_series_equals_predictor_double_forward(setup, gpu_device):
    case, model, series = setup
    w = case.kwargs["window_size"]
    n = series.shape[0] - w
    sd = series.to(gpu_device)
    with torch.no_grad():
        preds, last = model.score_series(sd)
        assert preds.shape == (n, case.kwargs["out_dim"]) and last.shape == (n, case.kwargs["out_dim"])
        # the reference's loop (prediction.py:51-63), batch by batch, through the same module
        ref_p, ref_l = [], []
        for lo in range(0, n, 32):
            idx = list(range(lo, min(lo + 32, n)))
            x = _windows(series, idx, w).to(gpu_device)
            y = torch.stack([series[i + w:i + w + 1] for i in idx]).to(gpu_device)
            y_hat, _ = model(x)
            recon_x = torch.cat((x[:, 1:, :], y), dim=1)
            _, window_recon = model(recon_x)
            ref_p.append(y_hat)
            ref_l.append(window_recon[:, -1, :])
        assert torch.equal(preds, torch.cat(ref_p)) and torch.equal(last, torch.cat(ref_l))
        # and against the oracle (reference formulation on CPU) on a slice
        x = _windows(series, list(range(0, 12)), w)
        y = torch.stack([series[i + w:i + w + 1] for i in range(0, 12)])
        p_o, _ = oracle.forward(x, case.state_dict(), alpha=case.kwargs["alpha"])
        _, r_o = oracle.forward(torch.cat((x[:, 1:, :], y), dim=1), case.state_dict(), alpha=case.kwargs["alpha"])
    gate(preds[:12], p_o, what="score_series forecasts")
    gate(last[:12], r_o[:, -1, :], what="score_series reconstructions")


def test_anomaly_scores_follow_get_score(setup, gpu_device):
    """anomaly_scores() against the arithmetic of Predictor.get_score (reference prediction.py:65-91), restated
    here in numpy on the outputs of score_series()."""
    import numpy as np
    case, model, series = setup
    w = case.kwargs["window_size"]
    sd = series.to(gpu_device)
    with torch.no_grad():
        preds, recons = model.score_series(sd)
        for scale in (False, True):
            scores, per_dim = model.anomaly_scores(sd, target_dims=0, gamma=0.8, scale_scores=scale)
            p, r = preds.cpu().numpy(), recons.cpu().numpy()
            actual = series.numpy()[w:][:, [0]]
            ref = np.zeros_like(actual)
            for i in range(p.shape[1]):
                a = np.sqrt((p[:, i] - actual[:, i]) ** 2) + 0.8 * np.sqrt((r[:, i] - actual[:, i]) ** 2)
                if scale:
                    q75, q25 = np.percentile(a, [75, 25])
                    a = (a - np.median(a)) / (1 + (q75 - q25))
                ref[:, i] = a
            assert np.abs(per_dim.cpu().numpy() - ref).max() <= 1e-6
            assert np.abs(scores.cpu().numpy() - ref.mean(1)).max() <= 1e-6


def test_forward_series_on_a_wide_model(gpu_device):
    """More features than the LDS-staged convolution holds (F = 160): the straight-from-memory conv kernel must
    honour the window gather too, and the attention layers run the wide (K > 128) kernel."""
    from mtad_gat import MTAD_GAT
    torch.manual_seed(3)
    model = MTAD_GAT(n_features=160, window_size=20, out_dim=2, kernel_size=3, gru_hid_dim=24, recon_hid_dim=24).to(gpu_device).eval()
    series = torch.rand(20 + 30, 160, device=gpu_device)
    x = torch.stack([series[i:i + 20] for i in range(31)])
    with torch.no_grad():
        p0, r0 = model(x)
        p1, r1 = model.forward_series(series)
    assert torch.equal(p0, p1) and torch.equal(r0, r1)


def test_shared_convolution_rows_of_stride_one_windows_are_bit_equal(gpu_device):
    """From 1 024 stride-1 windows on, the series path computes every convolution row once (segment rows + the per-window
    edge rows) instead of once per window: outputs bit-equal to the materialised-window forward,
    in the default and the strict fp32 arithmetic, for a kernel size of 7 (MSL checkpoint) and of 3."""
    from mtad_gat import MTAD_GAT
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    w, F = case.kwargs["window_size"], case.kwargs["n_features"]
    g = torch.Generator().manual_seed(5)
    series = torch.rand(w + 1300, F, generator=g).to(gpu_device)
    torch.manual_seed(4)
    small = MTAD_GAT(n_features=11, window_size=24, out_dim=2, kernel_size=3, gru_hid_dim=40, recon_hid_dim=40).to(gpu_device).eval()
    series_s = torch.rand(24 + 2100, 11, device=gpu_device)
    with torch.no_grad():
        for m, sr, ww in ((model, series, w), (small, series_s, 24)):
            m.share_series_pair_scores = False          # (the shared pair scores have their own test below)
            n = sr.shape[0] - ww + 1
            x = torch.stack([sr[i:i + ww] for i in range(n)])
            for prec in ("fp32", "fp32_strict"):
                m.precision = prec
                p0, r0 = m(x)
                p1, r1 = m.forward_series(sr)
                assert torch.equal(p0, p1) and torch.equal(r0, r1), (prec, ww)
                p2, r2 = m.forward_series(sr, start=3, stride=1, count=1100)          # a segment that does not start at row 0
                p3, r3 = m(x[3:1103])           # (the same batch size: the recurrence kernels are chosen by it)
                assert torch.equal(p3, p2) and torch.equal(r3, r2), (prec, ww)
            m.precision = "auto"
        # score_series (windows 0 .. N - W, one forward each) over the same segment
        preds, last = model.score_series(series)
        xw = torch.stack([series[i:i + w] for i in range(series.shape[0] - w + 1)])
        pw, rw = model(xw)
        assert torch.equal(preds, pw[:-1]) and torch.equal(last, rw[1:, -1, :])


def test_shared_temporal_pair_scores_of_stride_one_windows(gpu_device):
    """SURVEY section 8f row 3, second half: from 1 024 stride-1 windows on the temporal layer's pair scores of interior rows are
    computed once per pair of series rows (k_tband_scores) and shared by the windows that contain both; the windows' own
    edge rows, the bias, softmax and aggregation stay per window (k_tband_edges / k_tband_att).  Against the per-window pair
    grid (same model, sharing off: itself bit-equal to forward() on the materialised windows) <= 1e-6 of the output scale --
    kernel sizes 7, 5, 3 (3, 2, 1 edge rows per side), a segment that does not start at row 0, the shipped MSL weights with
    their exploded attention biases, both fp32 arithmetics -- and the first windows against the oracle."""
    from mtad_gat import MTAD_GAT
    case = Case("msl")
    msl = case.build_model().to(gpu_device)
    g = torch.Generator().manual_seed(11)
    shapes = [(msl, case.kwargs["window_size"], case.kwargs["n_features"], 2300, case.state_dict(), case.kwargs["alpha"])]
    for (F, W, ks, n, od) in ((38, 100, 7, 1500, 38), (9, 30, 3, 3000, 1), (21, 64, 5, 1200, 3), (5, 17, 3, 1100, 5), (48, 128, 7, 1024, 2)):
        torch.manual_seed(F + W)
        m = MTAD_GAT(n_features=F, window_size=W, out_dim=od, kernel_size=ks, gru_hid_dim=48, recon_hid_dim=40).eval()
        with torch.no_grad():
            m.temporal_gat.bias.normal_()
            m.feature_gat.bias.normal_()
        shapes.append((m.to(gpu_device), W, F, n, {k: v.detach().cpu() for k, v in m.state_dict().items()}, 0.2))
    with torch.no_grad():
        for m, W, F, n, sd, alpha in shapes:
            series = (torch.rand(W + n + 6, F, generator=g) * 3 - 1).to(gpu_device)
            for prec in ("fp32", "fp32_strict"):
                m.precision = prec
                m.share_series_pair_scores = False
                p0, r0 = m.forward_series(series, start=5, stride=1, count=n)
                m.share_series_pair_scores = True
                p1, r1 = m.forward_series(series, start=5, stride=1, count=n)
                assert not (torch.equal(p0, p1) and torch.equal(r0, r1)), "the shared-score kernels did not run"
                tol = 1e-6 * max(1.0, r0.abs().max().item(), p0.abs().max().item())
                assert (p1 - p0).abs().max().item() <= tol and (r1 - r0).abs().max().item() <= tol, (W, F, prec)
            m.precision = "auto"
            x = torch.stack([series[5 + i:5 + i + W] for i in range(4)]).cpu()
            p_o, r_o = oracle.forward(x, sd, alpha=alpha)
            gate(p1[:4], p_o, what=f"shared pair scores, forecasts (W={W}, F={F})")
            gate(r1[:4], r_o, what=f"shared pair scores, reconstructions (W={W}, F={F})")
            # strided / listed windows and short calls do not qualify: they run the per-window kernels whatever the switch says
            pa, ra = m.forward_series(series, start=0, stride=2, count=40)
            m.share_series_pair_scores = False
            pb, rb = m.forward_series(series, start=0, stride=2, count=40)
            assert torch.equal(pa, pb) and torch.equal(ra, rb)
            m.share_series_pair_scores = "auto"
