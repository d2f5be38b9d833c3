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
    has it (the order of the in-place bucket), model.parameters() order otherwise.  Parameters without a gradient contribute
    zeros to the bucket (same length everywhere) but keep `.grad = None` afterwards, so the optimizer skips them exactly as a
    single-process step would (Adam's weight decay / momentum would otherwise move them)."""
    try:
        import _hipgrad
        params = _hipgrad.param_order(model)
        if len(params) != len(list(model.parameters())):
            params = list(model.parameters())
    except Exception:
        params = list(model.parameters())
    return [p for p in params if p.requires_grad]


_MIX = 0x9E3779B97F4A7C15


def _step_seed(base: int, step: int) -> int:
    """Seed of step `step` of a run whose base seed rank 0 drew: the same on every rank without talking (splitmix-style)."""
    z = (base + (step + 1) * _MIX) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return int((z ^ (z >> 31)) & ((1 << 62) - 1))


def _settle_layout(model, n_local: int, device, shard_counts=None):
    """The data-parallel state every rank must agree on before a step's forward -- rank 0's base seed of the dropout streams and
    the shard sizes (first global window of this rank) -- settled ONCE per run with one all_gather and cached on the model
    (`model._dp_state`); later steps derive their seed from (base, step) and reuse the sizes.  The sizes ride along in every
    step's statistics exchange (slot r = rank r's count), so a change is noticed in the step it happens in, at no extra
    collective: dp_training_step then refreshes the cache from the exchanged sizes (that one step's dropout masks are valid but
    are not the single-process step's; `state["layout_refreshed"]` counts such steps)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    st = getattr(model, "_dp_state", None)
    if st is not None and st["world"] == world and st["rank"] == rank:
        return st
    cdev = torch.device("cpu") if dist.get_backend() == "gloo" else device
    mine = torch.zeros(2, dtype=torch.int64, device=cdev)
    mine[0] = int(torch.randint(0, 2 ** 62, (1,)).item()) if rank == 0 else 0
    mine[1] = n_local
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)                       # the only layout collective of the run
    got = torch.stack(got).cpu()
    counts = [int(v) for v in got[:, 1]] if shard_counts is None else [int(v) for v in shard_counts]
    st = {"world": world, "rank": rank, "seed": int(got[0, 0]), "counts": counts, "step": 0, "layout_refreshed": 0, "pinned": None}
    object.__setattr__(model, "_dp_state", st)
    return st


def dp_training_step(model, x, y, optimizer, target_dims=None, timings=None, shard_counts=None):
    """One data-parallel optimisation step with the semantics of a single process seeing the
    global batch (reference training.py:106-127: loss = sqrt(MSE(y, preds)) + sqrt(MSE(x, recons))).

    sqrt(mean(.)) is not additive over shards, so averaging per-rank gradients would differ from
    the reference.  Instead: all-reduce the two squared-error sums and
    counts, form the global RMSEs, back-propagate the local surrogate
    SSE_f / (2 RMSE_f N_f) + SSE_r / (2 RMSE_r N_r) whose gradients sum over ranks to the global
    gradient, then sum-all-reduce one flat gradient bucket (RCCL over xGMI on the GPUs; the bucket
    is ~1.7 MB, latency-class).  Returns (forecast_rmse, recon_rmse) of the global batch.

    Exactly TWO collectives per steady-state step and no host read before `backward()` is enqueued: the statistics exchange
    also carries what the ranks have to agree on -- whether every rank's gradients will sit in the HIP backward's flat buffer
    (the in-place bucket) and the shard sizes that key the dropout masks by global window index -- and its few words are
    copied to pinned memory behind the collective and read only after the backward has been enqueued.  The dropout seed is
    derived from a base seed settled once per run (`_settle_layout`: one all_gather in the first step).  `shard_counts`: the
    windows per rank when the caller knows them (shard_range), otherwise the first step's gathered sizes are kept.

    `timings`: optional dict; on a GPU the two exchanges are bracketed with events on the stream they run on and the pairs
    appended to timings["stats_events"] / timings["grad_events"] (no synchronisation here: read them after the step).
    """
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    optimizer.zero_grad()
    st = None
    if distributed:
        st = _settle_layout(model, int(x.shape[0]), x.device, shard_counts)
        if x.device.type == "cuda":
            # one dropout stream for the logical batch: the run's seed for this step, this shard's first global window index
            # -> the masks of the single-process step over the whole batch
            first = sum(st["counts"][: st["rank"]])
            object.__setattr__(model, "dropout_stream", (_step_seed(st["seed"], st["step"]), first))
        st["step"] += 1
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
    params = _bucket_params(model)
    # will this rank's gradients be views of the HIP backward's flat buffer?  Known before the backward: the step ran on the HIP
    # kernels, every parameter takes part and none holds a gradient tensor autograd would accumulate into
    n_all = sum(1 for _ in model.parameters())
    inplace_here = (getattr(model, "grad_path", None) == "hip" and getattr(model, "_engine", None) is not None and
                    len(params) == n_all and all(p.grad is None for p in params))
    world = dist.get_world_size() if distributed else 1
    head = [sse_f.detach().double().reshape(1), torch.tensor([float(preds.numel())], dtype=torch.float64, device=x.device),
            sse_r.detach().double().reshape(1), torch.tensor([float(recons.numel())], dtype=torch.float64, device=x.device)]
    if distributed:
        slots = [0.0] * world
        slots[st["rank"]] = float(x.shape[0])
        head.append(torch.tensor([0.0 if inplace_here else 1.0] + slots, dtype=torch.float64, device=x.device))
    stats = torch.cat(head)

    def _timed(key, fn):
        if timings is None or x.device.type != "cuda":
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        timings.setdefault(key, []).append((e0, e1))

    ready = None
    if distributed:
        _timed("stats_events", lambda: all_reduce_(stats))                       # collective 1 of 2
        if stats.is_cuda:
            # the agreement words go to pinned memory behind the collective; the host looks at them after backward() is enqueued
            if st["pinned"] is None or st["pinned"].numel() != 1 + world:
                st["pinned"] = torch.zeros(1 + world, dtype=torch.float64).pin_memory()
            st["pinned"].copy_(stats[4:], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
    rmse_f = torch.sqrt(stats[0] / stats[1]).to(sse_f.dtype)
    rmse_r = torch.sqrt(stats[2] / stats[3]).to(sse_r.dtype)
    surrogate = sse_f / (2.0 * rmse_f * stats[1].to(sse_f.dtype)) + sse_r / (2.0 * rmse_r * stats[3].to(sse_r.dtype))
    surrogate.backward()
    if distributed:
        if ready is not None:
            ready.synchronize()                      # (fired long ago: right behind the statistics exchange)
            agreed = st["pinned"]
        else:
            agreed = stats[4:]
        all_inplace = float(agreed[0]) == 0.0
        counts_now = [int(round(float(v))) for v in agreed[1:]]
        if shard_counts is None and counts_now != st["counts"]:
            st["counts"] = counts_now                # a shard changed size (last batch of an epoch): right from the next step on
            st["layout_refreshed"] += 1
        flat = _flat_gradient_buffer(model) if all_inplace else None
        if all_inplace and flat is None:
            raise RuntimeError("data-parallel step: this rank predicted the in-place gradient bucket but the HIP backward's flat "
                               "buffer does not hold every .grad (a hook replaced a gradient?)")
        if flat is not None:
            # HIP backward: every p.grad is a view of the backward's flat gradient buffer -- one all-reduce in place, no copies
            _timed("grad_events", lambda: all_reduce_(flat))                     # collective 2 of 2
        else:
            # copy path, same field order on every rank; parameters without a gradient ride as zeros and stay without one
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            _timed("grad_events", lambda: all_reduce_(flat))                     # collective 2 of 2
            off = 0
            for p in params:
                n = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
    optimizer.step()
    return float(rmse_f), float(rmse_r)
