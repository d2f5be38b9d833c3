"""Data-parallel sharding of sliding windows across the GPUs of a node.

Windows are independent units: rank r gets the contiguous block
[r*M/N, (r+1)*M/N) of window indices, runs the forward on it with its own replica of the
(1.7 MB) weights, and no collective touches the data path.  `torch.distributed` (backend "nccl"
= RCCL on ROCm, "gloo" in the CPU tests) is used only for the timing barrier / max-over-ranks and,
optionally, to gather per-window outputs.
"""
import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block of window indices owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _staged(t: torch.Tensor) -> bool:
    """gloo moves host memory: device tensors go through a host copy (two processes sharing one GPU in the tests, where RCCL
    refuses a second rank on the same device); RCCL ("nccl") takes device tensors as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_(t: torch.Tensor, op=None) -> torch.Tensor:
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time (the job's time); identity when not running distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_windows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank outputs (first dim = this rank's windows) back into window order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
    pad = max(sizes)          # all_gather wants equal shapes: pad the short shards, trim after
    mine = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    mine[: local.shape[0]] = local
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def _flat_gradient_buffer(model):
    """The HIP backward's flat gradient buffer when every parameter's `.grad` is (still) a view of it at its offset -- then the
    buffer IS the gradient bucket; None otherwise (torch-op backward, accumulated gradients, parameters without a gradient)."""
    eng = getattr(model, "_engine", None)
    held = getattr(eng, "_flat_grads", None) if eng is not None else None
    if held is None or getattr(model, "grad_path", None) != "hip":
        return None
    flat, offs = held
    try:
        import _hipgrad
        params = _hipgrad.param_order(model)
    except Exception:
        return None
    if len(params) != len(offs):
        return None
    base, esz = flat.data_ptr(), flat.element_size()
    for p, o in zip(params, offs):
        g = p.grad
        if g is None or g.dtype != flat.dtype or not g.is_contiguous() or g.data_ptr() != base + o * esz:
            return None
    return flat


def _bucket_params(model):
    """Parameters of the copy-path gradient bucket, in ONE order on every rank: the field order of mtadgat_params when the model
    has it (the order of the in-place bucket), model.parameters() order otherwise; parameters without a gradient get zeros so
    that the bucket has the same length everywhere."""
    try:
        import _hipgrad
        params = _hipgrad.param_order(model)
        if len(params) != len(list(model.parameters())):
            params = list(model.parameters())
    except Exception:
        params = list(model.parameters())
    params = [p for p in params if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    return params


def dp_training_step(model, x, y, optimizer, target_dims=None, timings=None):
    """One data-parallel optimisation step with the semantics of a single process seeing the
    global batch (reference training.py:106-127: loss = sqrt(MSE(y, preds)) + sqrt(MSE(x, recons))).

    sqrt(mean(.)) is not additive over shards, so averaging per-rank gradients would differ from
    the reference.  Instead: all-reduce the two squared-error sums and
    counts (4 scalars), form the global RMSEs, back-propagate the local surrogate
    SSE_f / (2 RMSE_f N_f) + SSE_r / (2 RMSE_r N_r) whose gradients sum over ranks to the global
    gradient, then sum-all-reduce one flat gradient bucket (RCCL over xGMI on the GPUs; the bucket
    is ~1.7 MB, latency-class).  Returns (forecast_rmse, recon_rmse) of the global batch.

    `timings`: optional dict; on a GPU the two exchanges are bracketed with events on the stream they run on and the pairs
    appended to timings["stats_events"] / timings["grad_events"] (no synchronisation here: read them after the step).
    """
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    optimizer.zero_grad()
    if distributed and x.device.type == "cuda":
        # one dropout stream for the logical batch: rank 0's seed, this shard's first global window index (equal
        # shards up to one window, see shard_range) -> the masks of the single-process step over the whole batch
        cdev = torch.device("cpu") if dist.get_backend() == "gloo" else x.device
        meta = torch.zeros(2, dtype=torch.int64, device=cdev)
        if dist.get_rank() == 0:
            meta[0] = int(torch.randint(0, 2 ** 62, (1,)).item())
        dist.broadcast(meta, src=0)
        counts = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(dist.get_world_size())]
        dist.all_gather(counts, torch.tensor([x.shape[0]], dtype=torch.int64, device=cdev))
        first = int(sum(int(c.item()) for c in counts[: dist.get_rank()]))
        object.__setattr__(model, "dropout_stream", (int(meta[0].item()), first))
    try:
        preds, recons = model(x)
    finally:
        if getattr(model, "dropout_stream", None) is not None:
            object.__setattr__(model, "dropout_stream", None)
    xt = x
    if target_dims is not None:
        xt = x[:, :, target_dims]
        y = y[:, :, target_dims].squeeze(-1)
    if preds.ndim == 3:
        preds = preds.squeeze(1)
    if y.ndim == 3:
        y = y.squeeze(1)
    sse_f = ((y - preds) ** 2).sum()
    sse_r = ((xt - recons) ** 2).sum()
    stats = torch.stack([sse_f.detach(), torch.tensor(float(preds.numel()), device=x.device),
                         sse_r.detach(), torch.tensor(float(recons.numel()), device=x.device)]).double()
    def _timed(key, fn):
        if timings is None or x.device.type != "cuda":
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        timings.setdefault(key, []).append((e0, e1))

    if distributed:
        _timed("stats_events", lambda: all_reduce_(stats))
    rmse_f = torch.sqrt(stats[0] / stats[1]).to(sse_f.dtype)
    rmse_r = torch.sqrt(stats[2] / stats[3]).to(sse_r.dtype)
    surrogate = sse_f / (2.0 * rmse_f * stats[1].to(sse_f.dtype)) + sse_r / (2.0 * rmse_r * stats[3].to(sse_r.dtype))
    surrogate.backward()
    if distributed:
        flat = _flat_gradient_buffer(model)
        # The in-place bucket (field order of mtadgat_params) and the copy path below lay the elements out differently: every
        # rank must take the same one, or a collective of matching size would silently sum misaligned gradients (a rank with
        # pre-existing .grad tensors, a torch-op fallback or a frozen parameter decides differently).  One MIN over a flag.
        agree = torch.tensor([1 if flat is not None else 0], dtype=torch.int32,
                             device=torch.device("cpu") if dist.get_backend() == "gloo" else x.device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if flat is not None and int(agree.item()) == 1:
            # HIP backward: every p.grad is a view of the backward's flat gradient buffer -- one all-reduce in place, no copies
            _timed("grad_events", lambda: all_reduce_(flat))
        else:
            params = _bucket_params(model)
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            _timed("grad_events", lambda: all_reduce_(flat))
            off = 0
            for p in params:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
    optimizer.step()
    return float(rmse_f), float(rmse_r)
