"""forward() walks a large call in pieces that alternate between the caller's stream and the handle's second lane
(mtadgat_capi.cpp forward_schedule): the results are those of forward() on each piece, whatever stream the caller is on,
call after call on the same workspace; option "lanes" = 1 keeps everything on the caller's stream."""
import pytest
import torch

from helpers import Case, gate
from oracle import mtad_gat_oracle as oracle

pytestmark = pytest.mark.gpu


def _pieces(n):
    if 8192 < n <= 16384:
        h = n // 2 // 32 * 32
        return [h, n - h]
    if n > 32768 and n % 32768:
        k = (n + 32767) // 32768
        return [32768] * (k - 1) + [n - 32768 * (k - 1)]
    return [n]


@pytest.mark.parametrize("n", [9000, 12300, 36900, 70000])
def test_two_lane_forward_equals_forward_on_the_pieces(n, gpu_device):
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    model.check_weight_contents = False
    W, F = case.kwargs["window_size"], case.kwargs["n_features"]
    g = torch.Generator().manual_seed(n)
    series = torch.rand(n + W - 1, F, generator=g).to(gpu_device)
    x = torch.stack([series[i:i + W] for i in range(n)]) if n < 20000 else torch.rand(n, W, F, generator=g).to(gpu_device)
    with torch.no_grad():
        p, r = model(x)
        eng = model._engine
        eng.set_option("lanes", 1)
        try:
            outs = [model(c) for c in torch.split(x, _pieces(n))]
            p1, r1 = model(x)
        finally:
            eng.set_option("lanes", 0)
        pp, rr = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        assert len(outs) > 1 and torch.equal(p, pp) and torch.equal(r, rr)
        tol = 2e-6 * max(1.0, r1.abs().max().item())
        assert (p - p1).abs().max().item() <= tol and (r - r1).abs().max().item() <= tol      # (one piece: other recurrence kernels)
        # on a side stream, three calls back to back on the same workspace, input produced on that stream right before
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            x2 = x * 1.0
            for _ in range(3):
                p2, r2 = model(x2)
            p2, r2 = p2.clone(), r2.clone()
        side.synchronize()
        assert torch.equal(p2, p) and torch.equal(r2, r)
        # the last windows (second lane's share) against the oracle
        p_o, r_o = oracle.forward(x[-3:].cpu(), case.state_dict(), alpha=case.kwargs["alpha"])
    gate(p[-3:], p_o, what="two-lane forward, forecasts of the last piece")
    gate(r[-3:], r_o, what="two-lane forward, reconstructions of the last piece")
    if n < 20000:
        with torch.no_grad():
            model.share_series_pair_scores = False
            ps, rs = model.forward_series(series, start=0, stride=1, count=n)
        assert torch.equal(ps, p) and torch.equal(rs, r)


def test_chunks_pieces_and_shared_scores_compose(gpu_device):
    """A call larger than the chunk size walks chunk by chunk, every chunk by its own piece schedule, and -- for a stride-1
    series -- every piece builds its own score band: whatever the combination, the outputs stay within 2e-6 of the single-chunk
    call (other recurrence kernels per size class) and the first / last windows match the oracle."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    model.check_weight_contents = False
    W, F = case.kwargs["window_size"], case.kwargs["n_features"]
    g = torch.Generator().manual_seed(77)
    n = 23000
    series = torch.rand(n + W - 1, F, generator=g).to(gpu_device)
    with torch.no_grad():
        model.share_series_pair_scores = True
        p0, r0 = model.forward_series(series, start=0, stride=1, count=n)
        eng = model._engine
        default_chunk = eng.chunk_windows()
        try:
            for chunk in (10000, 9000, 2048):
                eng.set_chunk_windows(chunk)
                p1, r1 = model.forward_series(series, start=0, stride=1, count=n)
                tol = 2e-6 * max(1.0, r0.abs().max().item())
                assert (p1 - p0).abs().max().item() <= tol and (r1 - r0).abs().max().item() <= tol, chunk
        finally:
            eng.set_chunk_windows(default_chunk)
            model.share_series_pair_scores = "auto"
        x = torch.stack([series[i:i + W] for i in (0, 1, n - 2, n - 1)]).cpu()
        p_o, r_o = oracle.forward(x, case.state_dict(), alpha=case.kwargs["alpha"])
    sel = torch.tensor([0, 1, n - 2, n - 1], device=gpu_device)
    gate(p1[sel], p_o, what="chunked series call, forecasts")
    gate(r1[sel], r_o, what="chunked series call, reconstructions")
