#!/usr/bin/env python3
"""bench.py -- sliding windows/s through the MI355X HIP path of MTAD_GAT.forward.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full `MTAD_GAT.forward` (conv -> feature-GAT || temporal-GAT -> GRU ->
forecasting + reconstruction heads; reference mtad_gat.py:64-79) over one batch of
synthetic MSL-shaped windows (W=100, F=55, out_dim=1 -- the shape BASELINE.json's
metric is quoted on), float32, inputs resident in HBM before the timed region.
For N > 1 the driver launches one process per GPU (torch.distributed.run); the
windows shard across ranks with no data-path collective (weak scaling: every rank
runs its own batch), timing is bracketed by barrier + synchronize and the maximum
over ranks is used.  Rank 0 prints ONE JSON line.

Also in the line:
  roofline      -- the launch family that dominates the step, timed with HIP events on
                   the launch stream inside the timed region (C ABI: mtadgat_profile_*),
                   algorithmic FLOPs per launch / average duration vs the fp32 MFMA peak;
                   plus `hbm` with the algorithmic-bytes rate vs 8 TB/s that BASELINE.json
                   asks for (this path is compute-bound by ~100x).
  cpu_baseline  -- the oracle (the reference's formulation on CPU PyTorch, oracle/) timed on
                   this host's cores on a bounded sample, N=1 / rank 0 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mtad-gat-pytorch_amd"))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s


def load_msl_state_dict():
    """Shipped MSL checkpoint (reference output/MSL/27062021_111641/model.pt) as stored in the golden fixture."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "msl.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    sd = {k[len("sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    return sd, meta["kwargs"]


def algorithmic_flops(kw):
    """Per-window FLOPs of each launch family, minimal (re-associated) algebra, SURVEY.md section 8d."""
    F, W, k, H = kw["n_features"], kw["window_size"], kw["kernel_size"], kw["gru_hid_dim"]
    Hr, out = kw["recon_hid_dim"], kw["out_dim"]
    Ef, Et = 2 * W, 2 * F      # GATv2 default embed dims (modules.py:47-50, :148-151)
    nfc, fh = kw["forecast_n_layers"] + 1, kw["forecast_hid_dim"]
    fc = 2 * (H * fh + (nfc - 2) * fh * fh + fh * out)
    return {
        "conv": 2 * W * F * F * k,
        "proj": 2 * F * (2 * W) * Ef + 2 * W * (2 * F) * Et,
        # pairwise |L+R| terms (2 VALU ops each) + aggregation GEMMs
        "attend": 2 * (F * F * Ef + W * W * Et) + 2 * F * F * W + 2 * W * W * F,
        "gru": 2 * (3 * H * 3 * F + 3 * H * H) * W,
        "fc": fc,
        # decoder: recurrent GEMM + per-step Linear (+ the folded input term, <= 3 columns)
        "recon": 2 * (3 * Hr * Hr) * W + 2 * Hr * out * W + 2 * 3 * Hr * 3 * W,
    }


def cpu_baseline(sd, kw, budget_s=10.0):
    """Reference CPU path (oracle = the reference's formulation on CPU PyTorch) on this host's cores.

    The reference formulation is memory-bound (it materialises the (b,K,K,2D) pairwise tensors), so
    more threads is not faster on a many-core host: a one-iteration sweep picks the thread count,
    then that setting is timed for ~budget_s seconds."""
    from oracle import mtad_gat_oracle as oracle
    ncores = os.cpu_count() or 1
    b = 256   # the reference's own batch (args.py:47, prediction.py:31)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(b, kw["window_size"], kw["n_features"], generator=g)
    cands = sorted({min(n, ncores) for n in (8, 32, 128)})
    sweep = {}
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            oracle.forward(x[:32], sd, kw["alpha"], aten_gru=True)      # warm-up
            t0 = time.perf_counter()
            oracle.forward(x, sd, kw["alpha"], aten_gru=True)
            sweep[nt] = b / (time.perf_counter() - t0)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        iters, t0 = 0, time.perf_counter()
        while True:
            oracle.forward(x, sd, kw["alpha"], aten_gru=True)
            iters += 1
            el = time.perf_counter() - t0
            if el >= budget_s or iters >= 50:
                break
    return {"value": round(b * iters / el, 2), "unit": "windows/s", "cores": best, "kind": "port",
            "sample": f"{iters} x {b}-window batches (W={kw['window_size']},F={kw['n_features']}) in {el:.1f} s; "
                      f"oracle/mtad_gat_oracle.py forward (reference formulation, torch CPU fp32, aten::gru); "
                      f"threads picked by a 1-iteration sweep {{{', '.join(f'{k}: {v:.0f} w/s' for k, v in sweep.items())}}} "
                      f"on a host with {ncores} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="windows per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="windows per internal chunk (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from mtad_gat import MTAD_GAT
    sd, kw = load_msl_state_dict()
    model = MTAD_GAT(**kw)
    model.load_state_dict(sd)
    model = model.to(dev).eval()

    # this rank's shard of the job: contiguous block of windows, independent of the others
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, kw["window_size"], kw["n_features"], generator=g).to(dev)
    eng = model._sync_engine(dev)
    if args.chunk > 0:
        eng.set_chunk_windows(args.chunk)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            model(x)
        barrier()
        if not args.no_profile:
            eng.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            preds, recons = model(x)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        prof = eng.profile_read() if not args.no_profile else None
        eng.profile_enable(False)
    from sharding import max_over_ranks
    elapsed = max_over_ranks(elapsed, dev)      # the slowest rank's time is the job's time
    assert torch.isfinite(preds).all() and torch.isfinite(recons).all()

    if rank == 0:
        total_windows = world * B * args.steps
        value = total_windows / elapsed
        F, W, out = kw["n_features"], kw["window_size"], kw["out_dim"]
        alg_bytes = W * F * 4 + out * (1 + W) * 4         # read the window once, write preds + recons
        res = {
            "metric": "sliding windows/sec (W=100,F=55)", "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MSL-shaped sliding windows W=100 F=55 out_dim=1, full MTAD_GAT.forward "
                                   "(conv + feature-GAT + temporal-GAT + GRU + forecasting/reconstruction heads), "
                                   "weights = shipped MSL checkpoint, x ~ U[0,1) seed 1234+rank, eval mode",
                       "windows_per_gpu_per_step": B, "parallelism": f"dp{world} (windows sharded, no collective)"},
        }
        if prof:
            flops = algorithmic_flops(kw)
            # launch families by kernel template: the GRU layer and the reconstruction decoder are two
            # launches of the same kernel (k_gru), the two attention layers two launches of k_gat
            groups = {"k_conv": ["conv"], "k_gat": ["proj", "attend"], "k_gru": ["gru", "recon"], "k_rowgemm(fc)": ["fc"]}
            fams = {}
            tot = {}
            for fam, slots in groups.items():
                ms = sum(prof[s_][0] for s_ in slots)
                n = sum(prof[s_][1] for s_ in slots)
                fl = sum(flops[s_] for s_ in slots)
                if n:
                    tot[fam] = (ms, n, fl)
                    fams[fam] = {"ms_per_step": round(ms / args.steps, 3), "launches": int(n),
                                 "alg_gflop_per_launch": round(fl * B * args.steps / n / 1e9, 3),
                                 "tflops": round(fl * B * args.steps / (ms * 1e-3) / 1e12, 2)}
            fams["k_gat"]["note"] = "VALU-bound: 2 VALU ops per pairwise element counted as 2 flop"
            res["kernels"] = fams
            dom = max(fams, key=lambda k: fams[k]["ms_per_step"])
            ms_dom, n_dom, fl_dom = tot[dom]
            flops = dict(flops, **{dom: fl_dom})
            ach = fl_dom * B * args.steps / (ms_dom * 1e-3) / 1e12
            traffic = None   # HBM bytes per launch from the committed PMC pass (profiles/), scaled to this batch
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                if dom in tj["bytes_per_window"]:
                    traffic = int(tj["bytes_per_window"][dom] * B * args.steps / n_dom)
            except Exception:
                pass
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                               "avg_launch_ms": round(ms_dom / n_dom, 3),
                               "alg_flop_per_window": flops[dom],
                               "note": "v_mfma_f32_32x32x2_f32 (exact f32) peak; algorithmic FLOPs exclude tile padding"}
        gbs = value * alg_bytes / 1e9
        res["hbm"] = {"alg_bytes_per_window": alg_bytes, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(gbs / HBM_PEAK_GBS / world, 5),
                      "note": "whole-forward algorithmic bytes rate per GPU vs HBM peak; the path is compute-bound"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(sd, kw)
            except Exception as e:  # never lose the GPU number to a baseline problem
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
