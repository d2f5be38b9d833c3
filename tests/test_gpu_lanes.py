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
        h = 8192 if n - 8192 <= 1536 else n // 2 // 32 * 32      # (a short tail rides beside a full 8 192-window piece)
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
            ps, rs = model.forward_series(series, start=0, stride=1, count=n)
        assert torch.equal(ps, p) and torch.equal(rs, r)


def test_chunks_and_pieces_compose(gpu_device):
    """A call larger than the chunk size walks chunk by chunk, every chunk by its own piece schedule -- here the stride-1 windows
    of a series, read out of it by the fused front end: whatever the combination, the outputs stay within 2e-6 of the single-chunk
    call (other recurrence kernels per size class) and the first / last windows match the oracle."""
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    model.check_weight_contents = False
    W, F = case.kwargs["window_size"], case.kwargs["n_features"]
    g = torch.Generator().manual_seed(77)
    n = 23000
    series = torch.rand(n + W - 1, F, generator=g).to(gpu_device)
    with torch.no_grad():
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
        x = torch.stack([series[i:i + W] for i in (0, 1, n - 2, n - 1)]).cpu()
        p_o, r_o = oracle.forward(x, case.state_dict(), alpha=case.kwargs["alpha"])
    sel = torch.tensor([0, 1, n - 2, n - 1], device=gpu_device)
    gate(p1[sel], p_o, what="chunked series call, forecasts")
    gate(r1[sel], r_o, what="chunked series call, reconstructions")


@pytest.mark.parametrize("name,n", [("msl", 256), ("smd_1_1", 300), ("smap", 1000), ("syn_v1_small", 64)])
def test_small_calls_run_independent_stages_side_by_side(name, n, gpu_device):
    """Calls of up to 1024 windows (the reference Predictor's 256, prediction.py:31) run the feature layer beside the temporal one
    and the forecasting head beside the decoder on two streams (mtadgat_capi.cpp forward_impl): same kernels, so the results must
    equal the one-stream schedule's bit for bit, call after call, on the caller's current stream whichever that is."""
    case = Case(name)
    model = case.build_model().to(gpu_device)
    model.check_weight_contents = False
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, case.kwargs["window_size"], case.kwargs["n_features"], generator=g).to(gpu_device)
    x[: case.x.shape[0]] = case.x.to(gpu_device)
    with torch.no_grad():
        p, r = model(x)
        eng = model._engine
        eng.set_option("lanes", 1)
        try:
            p1, r1 = model(x)
        finally:
            eng.set_option("lanes", 0)
        assert torch.equal(p, p1) and torch.equal(r, r1)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            x2 = x * 1.0
            for _ in range(4):
                p2, r2 = model(x2)
            p2, r2 = p2.clone(), r2.clone()
        side.synchronize()
        assert torch.equal(p2, p) and torch.equal(r2, r)
        ps, _, last = eng.forward_series(x[0].contiguous(), None, 0, 1, 1, want_recons=False, want_last=True)     # preds without recons
        assert (ps - p[:1]).abs().max().item() <= 2e-6 and (last - r[:1, -1]).abs().max().item() <= 2e-6     # (one window: other kernels)
    k = case.x.shape[0]
    gate(p[:k], case.preds, case.preds64, what=f"{name} predictions (two-stream small call)")
    gate(r[:k], case.recons, case.recons64, what=f"{name} recons (two-stream small call)")


def test_unchanged_weights_do_not_hold_the_call_up(gpu_device):
    """Default settings (check_weight_contents = True): the content check of the parameters is a kernel + an 8-byte copy whose
    result is read after the call's kernels are enqueued.  While the host is still inside forward() the stream must already hold
    the call's work: a long kernel queued in front of the call must not delay the host by its duration."""
    import time
    case = Case("msl")
    model = case.build_model().to(gpu_device)
    x = torch.rand(256, 100, 55).to(gpu_device)
    with torch.no_grad():
        for _ in range(3):
            p0, r0 = model(x)
        torch.cuda.synchronize()
        # edits through .data are still seen by the very next call (contents change, version counters do not)
        bias = model.forecasting_model.layers[3].bias
        v = bias._version
        bias.data.add_(0.5)
        assert bias._version == v
        p1, _ = model(x)
        assert torch.allclose(p1, p0 + 0.5, atol=1e-6)
        bias.data.sub_(0.5)
        p2, r2 = model(x)
        assert torch.equal(p2, p0) and torch.equal(r2, r0)
        # the unchanged-weights call: time on the host with an empty stream vs behind ~50 ms of queued work
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(x)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        big = torch.rand(8192, 8192, device=gpu_device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            big = big @ big * 1e-4
        t_queue = time.perf_counter() - t0                    # (enqueue only)
        model(x)
        t_behind = time.perf_counter() - t0 - t_queue
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"host time of forward(): empty stream {1e3 * t_host:.3f} ms, behind queued work {1e3 * t_behind:.3f} ms (queue drains in {1e3 * t_all:.1f} ms)")
        # the content check reads its 8 bytes through an event behind the queued work: the host waits for the QUEUE, once, not for
        # a stream synchronisation after its own kernels -- its own kernels are enqueued before it waits
        assert t_all > 5 * t_host
