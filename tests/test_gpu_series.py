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
