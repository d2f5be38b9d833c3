#!/usr/bin/env python3
"""bench.py -- sliding windows/s through the MI355X HIP path of MTAD_GAT.forward.

    python bench.py --gpus N --steps K --warmup W [--mode infer|train]

A "step" is one full `MTAD_GAT.forward` (conv -> feature-GAT || temporal-GAT -> GRU ->
forecasting + reconstruction heads; reference mtad_gat.py:64-79) over one batch of
synthetic MSL-shaped windows (W=100, F=55, out_dim=1 -- the shape BASELINE.json's
metric is quoted on), float32, inputs resident in HBM before the timed region.
For N > 1 the driver launches one process per GPU (torch.distributed.run); the
windows shard across ranks with no data-path collective (weak scaling: every rank
runs its own batch), timing is bracketed by barrier + synchronize and the maximum
over ranks is used.  Rank 0 prints ONE JSON line.

Also in the line:
  roofline      -- the launch family that takes most of the step, timed with HIP events on
                   the launch stream inside the timed region (C ABI: mtadgat_profile_*): the
                   attention family (k_gath, pair grid on the vector ALU) against the VALU
                   lane-op rate, or the recurrences (k_gru_cm) against the guide's 2 500 TF dense
                   16-bit MFMA peak (`alg_frac` algorithmic, `issued_frac` incl. the three split
                   terms and padding); both families always under `roofline_valu` / `roofline_mfma`;
                   plus `hbm` with the algorithmic-bytes rate vs 8 TB/s that BASELINE.json
                   asks for (this path is compute-bound by ~100x).
  cpu_baseline  -- the unmodified reference module (imported from $MTADGAT_REFERENCE or /root/reference,
                   kind "reference") when that tree exists on this host, else the oracle port of it
                   (oracle/, kind "port"), timed on this host's cores on a bounded sample, N=1 / rank 0 only.
  sub           -- outside the timed region, N=1 only: `batch256` (the reference Predictor's fixed batch,
                   prediction.py:31: latency of one forward), `fp32_strict_b65536` (the flagship workload on the
                   fp32-MFMA kernels only), `train_step` (BASELINE config 3 shape: SMD F=38, out=38, batch 256 --
                   forward + loss + backward + Adam, training.py:106-127), `smap_bf16_b4096` (config 2, shipped
                   SMAP weights) and `config4_f512_w256` (config 4, batch 8192); each names the code path that ran.

--mode train times the data-parallel TRAINING step instead (BASELINE config 5's exchange step: sharding.dp_training_step =
forward + global-batch RMSE losses through one small statistics all-reduce + backward + one flat gradient all-reduce on RCCL + Adam
-- two collectives per step, no host read before the backward is enqueued;
reference training.py:106-127) at the MSL shape, `--batch` windows per GPU (default 8192); the two collectives are timed
separately with HIP events on the stream they run on.  Works at N = 1 (the collectives are skipped).
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Vector-ALU peak: MI355X_MICROARCH.md -- CDNA4 CUs have four SIMD-32 units, a wave64 VALU instruction issues over 2 cycles
# (v_fma_f32: 2 cyc): 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78.6 T fp32 lane-ops/s (= the 157.3 TFLOP/s fp32 vector peak / 2).
# Rounds 1-3 priced the attention family against half of that (16 lanes/clk, the CDNA3 SIMD width: 39.3 T); profiles/ubench_gat2.hip
# measures 2.5-2.9 cycles per wave64 instruction and SIMD for the pair-grid loop, i.e. 55-64 T lane-ops/s.  `frac` uses the
# guide's peak; `frac_vs_r03_peak` keeps the old denominator so that the rounds stay comparable.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
VALU_PEAK_TLANEOPS_R03 = 256 * 4 * 16 * 2.4e9 / 1e12


def load_msl_state_dict(name="msl"):
    """Shipped MSL (or SMAP / SMD) checkpoint of the reference (output/<DATASET>/<id>/model.pt) as stored in the golden fixture."""
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
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
        # aggregation GEMMs att (K x K) . V (K x D); the pairwise |L+R| term is VALU work (valu_lane_ops)
        "attend": 2 * F * F * W + 2 * W * W * F,
        "gru": 2 * (3 * H * 3 * F + 3 * H * H) * W,
        "fc": fc,
        # decoder: recurrent GEMM + per-step Linear (+ the folded input term, <= 3 columns)
        "recon": 2 * (3 * Hr * Hr) * W + 2 * Hr * out * W + 2 * 3 * Hr * 3 * W,
    }


def split_issue_factor(kw, two_piece_x=True):
    """The recurrence kernels in the default fp32 arithmetic: 16-bit MFMA MACs issued per algorithmic MAC (GRU layer +
    decoder together).  Every operand goes as two fp16 pieces (3 MFMA terms per product) -- the recurrent state, the
    attention outputs and the decoder's input by construction, the convolution's channels when their recorded range allows it
    (`two_piece_x`, read back from the engine: mtadgat_last_conv_max; three bf16 pieces = 6 terms for those chunks otherwise,
    on the tile-major kernel).  Chunk (16 features) and tile (32 hidden units) padding included; the chunk-major kernel
    (k_gru_cm) skips all-padding chunks."""
    F, W, H, Hr = kw["n_features"], kw["window_size"], kw["gru_hid_dim"], kw["recon_hid_dim"]
    up = lambda a, b: -(-a // b) * b      # noqa: E731
    qx = -(-(3 * F) // 16)                              # 16-feature input chunks with non-zero weights
    qb = 0 if two_piece_x else -(-F // 16)
    qh = lambda h: up(h, 32) // 16          # noqa: E731
    issued = (up(H, 32) * 16 * (6 * qb + 3 * (qx - qb) + 3 * qh(H)) + up(Hr, 32) * 16 * (3 * 1 + 3 * qh(Hr))) * 3 * W
    alg = (3 * H * (3 * F + H) + 3 * Hr * (3 + Hr)) * W
    return issued / alg


def kernel_sources_sha16():
    """Content hash of the kernel sources: profiles/collect.sh stamps the PMC traffic file with it, bench.py refuses a file
    whose stamp differs (the snapshot on the GPU box has no .git to ask)."""
    import hashlib
    d = os.path.join(ROOT, "mtad-gat-pytorch_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _reference_module():
    """The unmodified reference MTAD_GAT class, if its tree is on this host (never on the GPU box)."""
    ref = os.environ.get("MTADGAT_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "mtad_gat.py")):
        return None
    import importlib.util
    sys.path.append(ref)                      # its `from modules import ...`; appended: never shadows this package
    try:
        spec = importlib.util.spec_from_file_location("_reference_mtad_gat", os.path.join(ref, "mtad_gat.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.MTAD_GAT
    except Exception:
        return None
    finally:
        sys.path.remove(ref)


def valu_lane_ops(kw):
    """Per-window VALU lane-operations of the pairwise term: 2 per (i, j, k) element (add, |.|-accumulate)."""
    F, W = kw["n_features"], kw["window_size"]
    return {"attend": 2 * (F * F * 2 * W + W * W * 2 * F)}


def cpu_baseline(sd, kw, budget_s=10.0):
    """Reference CPU path on this host's cores: the reference module itself when its tree is present
    (kind "reference"), otherwise the oracle = the reference's formulation restated on CPU PyTorch
    (kind "port"; the two agree to 2e-6, tests/test_oracle_vs_reference.py).

    The reference formulation is memory-bound (it materialises the (b,K,K,2D) pairwise tensors), so
    more threads is not faster on a many-core host: a one-iteration sweep picks the thread count,
    then that setting is timed for ~budget_s seconds."""
    ncores = os.cpu_count() or 1
    b = 256   # the reference's own batch (args.py:47, prediction.py:31)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(b, kw["window_size"], kw["n_features"], generator=g)
    Ref = _reference_module()
    if Ref is not None:
        ref = Ref(**kw)
        ref.load_state_dict(sd)
        ref.eval()
        run, kind, what = (lambda xx: ref(xx)), "reference", "unmodified reference MTAD_GAT.forward (torch CPU fp32)"
    else:
        from oracle import mtad_gat_oracle as oracle
        run, kind = (lambda xx: oracle.forward(xx, sd, kw["alpha"], aten_gru=True)), "port"
        what = "oracle/mtad_gat_oracle.py forward (reference formulation, torch CPU fp32, aten::gru); reference tree absent on this host"
    cands = sorted({min(n, ncores) for n in (8, 32, 128)})
    sweep = {}
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            run(x[:32])      # warm-up
            t0 = time.perf_counter()
            run(x)
            sweep[nt] = b / (time.perf_counter() - t0)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        iters, t0 = 0, time.perf_counter()
        while True:
            run(x)
            iters += 1
            el = time.perf_counter() - t0
            if el >= budget_s or iters >= 50:
                break
    return {"value": round(b * iters / el, 2), "unit": "windows/s", "cores": best, "kind": kind,
            "sample": f"{iters} x {b}-window batches (W={kw['window_size']},F={kw['n_features']}) in {el:.1f} s; {what}; "
                      f"threads picked by a 1-iteration sweep {{{', '.join(f'{k}: {v:.0f} w/s' for k, v in sweep.items())}}} "
                      f"on a host with {ncores} logical cores"}


def _timed(fn, dev, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / iters


def sub_records(model, kw, dev, args_precision="fp32"):
    """Measurements of the other BASELINE configurations, outside the timed region (N=1 only)."""
    import torch.nn.functional as F
    from mtad_gat import MTAD_GAT
    out = {}
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        x256 = torch.rand(256, kw["window_size"], kw["n_features"], generator=g).to(dev)
        default_cwc = type(model)(**kw).check_weight_contents          # what an unchanged caller gets
        model.check_weight_contents = default_cwc
        t = _timed(lambda: model(x256), dev, 20)
        out["batch256_default"] = {"ms": round(1e3 * t, 3), "windows_per_s": round(256 / t, 1), "check_weight_contents": repr(default_cwc),
                                   "what": "one eval forward of the reference Predictor's fixed 256-window batch (prediction.py:31), MSL shape, "
                                           "module settings at their defaults (the per-call parameter-content check included)"}
        model.check_weight_contents = False
        t = _timed(lambda: model(x256), dev, 20)
        out["batch256"] = {"ms": round(1e3 * t, 3), "windows_per_s": round(256 / t, 1),
                           "what": "the same with check_weight_contents = False (parameter versions trusted: no content check)"}
        model.precision = "bf16"
        t = _timed(lambda: model(x256), dev, 20)
        model.precision = args_precision
        out["batch256_bf16"] = {"ms": round(1e3 * t, 3), "windows_per_s": round(256 / t, 1), "what": "the same with bf16 MFMA operands"}
        # the flagship workload with bf16 MFMA operands (fp32 accumulate / state; 2e-2 parity class), own roofline
        xb = torch.rand(65536, kw["window_size"], kw["n_features"], generator=g).to(dev)
        model.precision = "bf16"
        eng = model._sync_engine(dev)
        t = _timed(lambda: model(xb), dev, 5, warm=2)
        eng.profile_enable(True)
        for _ in range(3):
            model(xb)
        torch.cuda.synchronize(dev)
        prof = eng.profile_read()
        eng.profile_enable(False)
        model.precision = "fp32_strict"
        model._sync_engine(dev)
        ts = _timed(lambda: model(xb), dev, 3, warm=1)
        out["fp32_strict_b65536"] = {
            "ms": round(1e3 * ts, 3), "windows_per_s": round(65536 / ts, 1),
            "what": "the flagship workload with precision='fp32_strict': v_mfma_f32_32x32x2_f32 / fp32 VALU only (no split 16-bit "
                    "operands anywhere) -- the exact-fp32 figure next to the headline"}
        model.precision = "auto"
        fl = algorithmic_flops(kw)
        ms_gru = (prof["gru"][0] + prof["recon"][0]) / 3
        tf = (fl["gru"] + fl["recon"]) * 65536 / (ms_gru * 1e-3) / 1e12
        out["bf16_operands_b65536"] = {
            "ms": round(1e3 * t, 3), "windows_per_s": round(65536 / t, 1),
            "kernel_ms": {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1]},
            "roofline": {"kernel": "k_gru (bf16 build)", "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 4)},
            "what": "the flagship workload (MSL shape, 65536 windows, fp32 tensors in/out) with precision='bf16': bf16 MFMA operands "
                    "where a bf16 build of the kernel exists, fp32 accumulation, state, gates and softmax; <= 2e-2 of the fp32 reference"}
        del xb
        # the same 65 536 windows as stride-1 windows of one series (Predictor.get_score's access
        # pattern, prediction.py:51-63), gathered on the GPU (no (b, W, F) batch is materialised)
        try:
            series = torch.rand(65536 + kw["window_size"] - 1, kw["n_features"], generator=g).to(dev)
            tser = _timed(lambda: model.forward_series(series, start=0, stride=1, count=65536), dev, 3, warm=1)
            rec = {"ms": round(1e3 * tser, 3), "windows_per_s": round(65536 / tser, 1),
                   "what": "forward_series over 65 536 stride-1 windows of one (65 635, F) series: every window is read out of the series by the "
                           "fused front end (per-window pair grids; the shared-score band of rounds 4-5 is gone: it bought 0.5 %)"}
            out["series_b65536"] = rec
            del series
        except Exception as e:
            out["series_b65536"] = {"error": repr(e)}
    # BASELINE config 3 shape: SMD machine-1-1, F=38 -> out 38, batch 256 (args.py:47), dropout 0.3
    kw3 = dict(kw, n_features=38, out_dim=38, dropout=0.3)
    torch.manual_seed(0)
    m3 = MTAD_GAT(**kw3).to(dev).train()
    m3.check_weight_contents = False
    opt = torch.optim.Adam(m3.parameters(), lr=1e-3)
    x3 = torch.rand(256, kw3["window_size"], 38, generator=g).to(dev)
    y3 = torch.rand(256, 38, generator=g).to(dev)

    def step():
        opt.zero_grad()
        p, r = m3(x3)
        loss = torch.sqrt(F.mse_loss(y3, p)) + torch.sqrt(F.mse_loss(x3, r))
        loss.backward()
        opt.step()

    t = _timed(step, dev, 10)
    # the same model at a large batch, where the throughput kernels take over
    x4 = torch.rand(8192, kw3["window_size"], 38, generator=g).to(dev)
    y4 = torch.rand(8192, 38, generator=g).to(dev)

    def step_big():
        opt.zero_grad()
        p, r = m3(x4)
        loss = torch.sqrt(F.mse_loss(y4, p)) + torch.sqrt(F.mse_loss(x4, r))
        loss.backward()
        opt.step()

    t4 = _timed(step_big, dev, 3)
    out["train_step_b8192"] = {"ms": round(1e3 * t4, 3), "windows_per_s": round(8192 / t4, 1),
                               "what": "the same model, batch 8192 (fp32 step, split-operand recurrences; a bf16 request trains on this step too: "
                                       "BASELINE config 3's bf16 train loop -- the bf16 recurrence kernels of rounds 2-5 lost to it and were removed)"}
    del x4, y4
    out["train_step"] = {"ms": round(1e3 * t, 3), "windows_per_s": round(256 / t, 1),
                         "grad_path": getattr(m3, "grad_path", "hip"),
                         "what": "SMD shape (F=38, W=100, out=38), batch 256, dropout 0.3: forward + RMSE losses + backward + Adam "
                                 "(training.py:106-127), fp32"}
    # BASELINE config 2: SMAP (F=25), batch 4096, bf16 I/O, shipped SMAP weights
    try:
        sd2, kw2 = load_msl_state_dict("smap")
        m2 = MTAD_GAT(**kw2)
        m2.load_state_dict(sd2)
        w2 = "shipped SMAP checkpoint"
    except Exception:
        kw2 = dict(kw, n_features=25, out_dim=1)
        torch.manual_seed(0)
        m2 = MTAD_GAT(**kw2)
        w2 = "random-init weights (fixture tests/golden/smap.npz not found)"
    m2 = m2.to(dev).eval()
    m2.check_weight_contents = False
    x2 = torch.rand(4096, kw2["window_size"], 25, generator=g).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        t = _timed(lambda: m2(x2), dev, 10)
        e2 = m2._sync_engine(dev, bf16=True)
        e2.profile_enable(True)
        for _ in range(5):
            m2(x2)
        torch.cuda.synchronize(dev)
        p2 = e2.profile_read()
        e2.profile_enable(False)
    # config 2's roofline: the launch family that takes most of this (latency-class) call, priced on the pipe it runs on
    km2 = {k: round(v[0] / 5, 4) for k, v in p2.items() if v[1]}
    dom2 = max(km2, key=km2.get)
    if dom2 in ("attend", "proj"):
        fam_ms = km2.get("attend", 0.0) + km2.get("proj", 0.0)
        tl2 = valu_lane_ops(kw2)["attend"] * 4096 / (fam_ms * 1e-3) / 1e12
        roof2 = {"kernel": "attention family (k_gath: projection + pair grid + aggregation, both layers)", "bound": "valu",
                 "achieved": round(tl2, 2), "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-op/s", "frac": round(tl2 / VALU_PEAK_TLANEOPS, 4),
                 "ms": round(fam_ms, 4)}
    else:
        tf2 = algorithmic_flops(kw2)[dom2] * 4096 / (km2[dom2] * 1e-3) / 1e12
        roof2 = {"kernel": dom2, "bound": "mfma", "achieved": round(tf2, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(tf2 / BF16_MFMA_PEAK_TFLOPS, 4), "ms": km2[dom2],
                 "note": "algorithmic matrix FLOPs of the family / its time vs the dense 16-bit MFMA peak; at 4 096 windows the "
                         "recurrences are a 100-step latency chain on 128 workgroups, not a throughput kernel"}
    out["smap_bf16_b4096"] = {"ms": round(1e3 * t, 3), "windows_per_s": round(4096 / t, 1), "kernel_ms": km2, "roofline": roof2,
                              "what": f"SMAP shape (F=25), batch 4096, bf16 in / bf16 out, {w2}"}
    del m2, x2
    # BASELINE config 4: synthetic F=512, W=256, out=512, H=150, batch 8192 (chunked by the library), fp32, random init
    try:
        kw4 = dict(kw, n_features=512, window_size=256, out_dim=512)
        torch.manual_seed(0)
        m4 = MTAD_GAT(**kw4).to(dev).eval()
        m4.check_weight_contents = False
        x4 = torch.rand(8192, 256, 512, device=dev)
        with torch.no_grad():
            e4 = m4._sync_engine(dev)
            m4(x4[:256])
            t4 = _timed(lambda: m4(x4), dev, 2, warm=1)
            e4.profile_enable(True)
            m4(x4)
            torch.cuda.synchronize(dev)
            p4 = e4.profile_read()
            e4.profile_enable(False)
        fl4 = algorithmic_flops(kw4)
        km = {k: round(v[0], 2) for k, v in p4.items() if v[1]}
        dom4 = max(km, key=km.get)
        if dom4 == "attend":
            # the attention family is bound by the vector ALU (the pair grid), not by the matrix pipe: price it there
            tl4 = valu_lane_ops(kw4)["attend"] * 8192 / (km[dom4] * 1e-3) / 1e12
            roof4 = {"kernel": "attend (k_gat_wide, both attention layers)", "bound": "valu", "achieved": round(tl4, 2),
                     "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-op/s", "frac": round(tl4 / VALU_PEAK_TLANEOPS, 4),
                     "note": "largest launch family of this shape by time; 2 lane-operations per (query, key, embedding column) element "
                             "of the GATv2 scores / its time vs the vector-ALU peak (4 SIMD-32 per CU x 2.4 GHz)"}
        else:
            tf4 = fl4[dom4] * 8192 / (km[dom4] * 1e-3) / 1e12
            roof4 = {"kernel": dom4, "bound": "mfma", "achieved": round(tf4, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf4 / FP32_MFMA_PEAK_TFLOPS, 4),
                     "note": "largest launch family of this shape by time; algorithmic matrix FLOPs of that family / its time vs the fp32 MFMA peak"}
        out["config4_f512_w256"] = {
            "ms": round(1e3 * t4, 2), "windows_per_s": round(8192 / t4, 1), "chunk_windows": int(e4.chunk_windows()),
            "kernel_ms": km,
            "roofline": roof4,
            "what": "BASELINE config 4: F=512, W=256, out_dim=512, H=150, 8192 windows per call (processed in chunks of chunk_windows), "
                    "fp32, random-init weights"}
        # ... and its training step (forward + RMSE losses + HIP backward + Adam) on the wide attention kernels (round 5)
        try:
            m4.train()
            opt4 = torch.optim.Adam(m4.parameters(), lr=1e-4)
            xt4, yt4 = x4[:256].contiguous(), torch.rand(256, 512, device=dev)

            def step4():
                opt4.zero_grad()
                p_, r_ = m4(xt4)
                (torch.sqrt(F.mse_loss(yt4, p_)) + torch.sqrt(F.mse_loss(xt4, r_))).backward()
                opt4.step()

            # (three warm-up steps: the first steps of a shape this size grow PyTorch's caching allocator by several GB -- with one
            # warm-up the record read 106 ms for a step whose kernels take 51)
            tt4 = _timed(step4, dev, 3, warm=3)
            out["config4_f512_w256"]["train_step_b256"] = {"ms": round(1e3 * tt4, 2), "windows_per_s": round(256 / tt4, 1),
                                                           "grad_path": getattr(m4, "grad_path", None)}
        except Exception as e:
            out["config4_f512_w256"]["train_step_b256"] = {"error": repr(e)}
        del m4, x4
    except Exception as e:
        out["config4_f512_w256"] = {"error": repr(e)}
    return out


def train_mode(args, model, kw, dev, world, rank):
    """The exchange step of the path (BASELINE config 5): sharding.dp_training_step over this rank's shard of the global batch."""
    import torch.distributed as dist
    from sharding import dp_training_step, max_over_ranks
    B = args.batch
    model.train()
    model.precision = "auto"
    model.check_weight_contents = False          # optimizer steps bump the parameters' version counters
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(4321 + rank)
    x = torch.rand(B, kw["window_size"], kw["n_features"], generator=g).to(dev)
    y = torch.rand(B, 1, kw["n_features"], generator=g).to(dev)
    tdims = [0] if kw["out_dim"] == 1 else None     # MSL / SMAP: target dimension 0 (reference utils.py:46-49)
    timings = {}

    dist_on = dist.is_available() and dist.is_initialized()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        dp_training_step(model, x, y, opt, target_dims=tdims)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rm = dp_training_step(model, x, y, opt, target_dims=tdims, timings=timings)
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0                 # this rank's own time, before it waits for the others
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = max_over_ranks(time.perf_counter() - t0, dev)
    per_rank = per_rank_rates(mine, B * args.steps, dev)
    ar = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in timings.items()}
    if rank == 0:
        n_par = sum(p.numel() for p in model.parameters())
        res = {
            "metric": "training windows/sec (W=100,F=55), data-parallel step", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MSL-shaped windows W=100 F=55 out_dim=1, training step = MTAD_GAT.forward (train mode, dropout 0.3) + "
                                   "global-batch RMSE losses + backward + Adam (reference training.py:106-127), shipped MSL checkpoint as the start point",
                       "windows_per_gpu_per_step": B, "parallelism": f"dp{world}: windows sharded; 4-scalar SSE all-reduce + one flat "
                                                                      f"{4 * n_par} B gradient all-reduce per step (RCCL)"},
            "exchange_ms_per_step": {"stats_allreduce": round(ar.get("stats_events", 0.0), 4), "grad_allreduce": round(ar.get("grad_events", 0.0), 4),
                                     "note": "HIP events around the two collectives on the stream they are enqueued on; 0 at one GPU (skipped)"},
            "grad_path": getattr(model, "grad_path", None), "loss_rmse": [round(float(v), 6) for v in rm],
            "rccl_ranks": dist.get_world_size() if (dist_on and dist.get_backend() == "nccl") else 0, "per_rank_windows_per_s": per_rank,
            "backend": dist.get_backend() if dist_on else None,
        }
        print(json.dumps(res))


def per_rank_rates(seconds, windows, dev):
    """Every rank's own windows/s (its time to its own last kernel, before the closing barrier), gathered on all ranks;
    one entry when not distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [round(windows / seconds, 1)]
    t = torch.tensor([seconds], dtype=torch.float64, device=None if dist.get_backend() == "gloo" else dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [round(windows / float(o.item()), 1) for o in out]


def self_launch(n, share_devices=False):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line under torch.distributed.run
    (one process per GPU of this node, rendezvous on 127.0.0.1 at a free port).  Fails loudly when the node has fewer
    than N devices -- a silent one-GPU run reported as an N-GPU number is the failure mode this replaces."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not (share_devices and have >= 1):
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to run fewer ranks than asked for", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: MTAD_GAT.forward (the headline metric); train: the data-parallel training step (BASELINE config 5's exchange step)")
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU per step (default 65536, train mode 8192)")
    ap.add_argument("--chunk", type=int, default=0, help="windows per internal chunk (0 = library default)")
    ap.add_argument("--spawn", action="store_true", help="go through the self-launch path (torch.distributed.run + an RCCL process group) even at --gpus 1")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend.  nccl (= RCCL) is the default and the only one whose numbers count; gloo is a TEST-ONLY escape "
                         "hatch that lets N ranks share the GPUs that exist (rank r on cuda:(r mod devices), collectives staged through host "
                         "memory) so that the N-rank code path can be exercised on a one-GPU box -- the line then says backend: gloo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="skip the batch256 / train_step / bf16 sub-records")
    ap.add_argument("--precision", choices=["fp32", "fp32_strict", "bf16"], default="fp32",
                    help="arithmetic of the timed region: fp32 (the headline, <= 1e-5 parity: fp32 accumulation, products of the "
                         "large-batch kernels from split-bf16 operands), fp32_strict (fp32 MFMA only) or bf16 MFMA operands (<= 2e-2)")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 65536 if args.mode == "infer" else 8192

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # `python bench.py --gpus N` with no launcher around it: become the launcher -- one rank per GPU under
        # torch.distributed.run on this node, RCCL rendezvous on 127.0.0.1 -- and hand its exit code back
        sys.exit(self_launch(args.gpus, share_devices=args.backend == "gloo"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts its own ranks)")
    if args.backend == "gloo":
        if torch.cuda.device_count() < 1:
            raise SystemExit("bench.py: no GPU on this node")
        local_rank = local_rank % torch.cuda.device_count()          # test-only: ranks share the devices that exist
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local_rank} but this node shows {torch.cuda.device_count()} device(s)")
    distributed = "WORLD_SIZE" in os.environ and (world > 1 or args.spawn)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus or dist.get_backend() != args.backend:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks on {dist.get_backend()}, wanted {args.gpus} on {args.backend}")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from mtad_gat import MTAD_GAT
    sd, kw = load_msl_state_dict()
    model = MTAD_GAT(**kw)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.precision = args.precision

    if args.mode == "train":
        train_mode(args, model, kw, dev, world, rank)
        if distributed:
            dist.destroy_process_group()
        return

    # this rank's shard of the job: contiguous block of windows, independent of the others
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, kw["window_size"], kw["n_features"], generator=g).to(dev)
    eng = model._sync_engine(dev)
    if args.chunk > 0:
        eng.set_chunk_windows(args.chunk)

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
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
        mine = time.perf_counter() - t0             # this rank's own time, before it waits for the others
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        prof = eng.profile_read() if not args.no_profile else None
        eng.profile_enable(False)
    from sharding import max_over_ranks
    elapsed = max_over_ranks(elapsed, dev)      # the slowest rank's time is the job's time
    per_rank = per_rank_rates(mine, B * args.steps, dev)
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
            "vs_baseline": None,
            "dtype": {"fp32": "f32 (2xf16 split operands on the 16-bit MFMA pipe in the large-batch recurrences and attention projections, "
                              "f32 accumulate / state / gates / softmax; small batches: f32 MFMA)",
                      "fp32_strict": "f32 (f32 MFMA / f32 VALU only)", "bf16": "bf16 operands, f32 accumulate"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "MSL-shaped sliding windows W=100 F=55 out_dim=1, full MTAD_GAT.forward "
                                   "(conv + feature-GAT + temporal-GAT + GRU + forecasting/reconstruction heads), "
                                   "weights = shipped MSL checkpoint, x ~ U[0,1) seed 1234+rank, eval mode",
                       "windows_per_gpu_per_step": B, "parallelism": f"dp{world} (windows sharded, no collective)"},
            "rccl_ranks": dist.get_world_size() if (distributed and args.backend == "nccl") else 0, "per_rank_windows_per_s": per_rank,
            "backend": args.backend if distributed else None,
        }
        if prof:
            flops = algorithmic_flops(kw)
            valu = valu_lane_ops(kw)
            # which arithmetic the range guard chose on the device (read back, not assumed)
            conv_max = eng.last_conv_max(B, dev)
            two_piece = 0.0 < conv_max < 32768.0
            sf = split_issue_factor(kw, two_piece)
            # launch families by kernel template: the GRU layer and the reconstruction decoder are two launches of the same
            # kernel (k_gru_cm above 8 192 windows per chunk; the hidden-tile-split kernel on split operands from 2 561), the two attention layers two launches of k_gat
            groups = {"k_conv": ["conv"], "k_gat": ["proj", "attend"], "k_gru": ["gru", "recon"], "k_rowgemm(fc)": ["fc"]}
            fams, tot = {}, {}
            for fam, slots in groups.items():
                ms = sum(prof[s_][0] for s_ in slots)
                n = sum(prof[s_][1] for s_ in slots)
                fl = sum(flops[s_] for s_ in slots)
                vl = sum(valu.get(s_, 0) for s_ in slots)
                if not n:
                    continue
                tot[fam] = (ms, n, fl)
                tf = fl * B * args.steps / (ms * 1e-3) / 1e12
                split = args.precision == "fp32" and fam == "k_gru"
                # from 4 096 windows per chunk the convolution (k_conv_win) and, below the range guard, the attention layers' projection
                # and aggregation (k_gath) also form their products from two fp16 pieces on the 16-bit pipe
                chunk_w = min(B, eng.chunk_windows())
                pieces16 = args.precision == "fp32" and chunk_w >= 4096 and (fam == "k_conv" or (fam == "k_gat" and two_piece))
                pk = BF16_MFMA_PEAK_TFLOPS if (split or pieces16 or (args.precision == "bf16" and fam != "k_rowgemm(fc)")) else FP32_MFMA_PEAK_TFLOPS
                fams[fam] = {"ms_per_step": round(ms / args.steps, 3), "launches": int(n),
                             "alg_mfma_gflop_per_launch": round(fl * B * args.steps / n / 1e9, 3),
                             "mfma_tflops": round(tf, 2), "mfma_peak_tflops": round(pk, 1), "mfma_alg_frac": round(tf / pk, 4)}
                if pieces16 and fam == "k_conv":
                    cf = 3.0 * (((kw["n_features"] + 15) // 16 * 16) / kw["n_features"]) * (((kw["n_features"] + 31) // 32 * 32) / kw["n_features"])
                    fams[fam]["mfma_issued_per_alg_mac"] = round(cf, 3)       # three fp16 terms, input channels padded to 16, outputs to 32
                    fams[fam]["mfma_issued_frac"] = round(tf * cf / pk, 4)
                if split:
                    fams[fam]["mfma_issued_per_alg_mac"] = round(sf, 3)
                    fams[fam]["mfma_issued_frac"] = round(tf * sf / pk, 4)
                if vl:
                    tl = vl * B * args.steps / (ms * 1e-3) / 1e12
                    fams[fam]["alg_valu_glaneops_per_launch"] = round(vl * B * args.steps / n / 1e9, 3)
                    fams[fam]["valu_tlaneops"] = round(tl, 2)
                    fams[fam]["valu_frac"] = round(tl / VALU_PEAK_TLANEOPS, 4)
            res["kernels"] = fams
            res["range_guard"] = {"conv_max": conv_max, "two_fp16_pieces_everywhere": bool(two_piece),
                                  "note": "largest convolution output of the last step, read back from the engine: below 2^15 every operand of the "
                                          "recurrences / attention projections went as two fp16 pieces"}
            # HBM bytes per window and family: NOT measured in this run (PMC counters need their own rocprofv3 passes,
            # profiles/collect.sh); taken from the newest committed PMC summary -- only if it was collected on these very
            # kernel sources (content hash), otherwise reported as null
            traffic_pw, tsrc = {}, "no PMC summary collected on these kernel sources (profiles/*_traffic.json stamp differs or is absent)"
            try:
                cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
                tj = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))
                if tj.get("csrc_sha16") == kernel_sources_sha16():
                    traffic_pw = tj["bytes_per_window"]
                    tsrc = f"profiles/{cands[-1]} (rocprofv3 --pmc passes on the same kernel sources, FETCH_SIZE doubled per MI355X_MICROARCH.md), scaled to this batch; not measured in this run"
            except Exception:
                pass

            def traffic_of(fam, n_launch):
                key = {"k_conv": "k_conv", "k_gat": "k_gat", "k_gru": "k_gru", "k_rowgemm(fc)": "k_rowgemm"}[fam]
                return int(traffic_pw[key] * B * args.steps / n_launch) if key in traffic_pw else None

            dom = max(fams, key=lambda k: fams[k]["ms_per_step"])
            ms_g, n_g, fl_g = tot["k_gru"]
            ach = fl_g * B * args.steps / (ms_g * 1e-3) / 1e12
            pk_g = fams["k_gru"]["mfma_peak_tflops"]
            roof_gru = {"kernel": "k_gru (GRU layer + reconstruction decoder: k_gru_cm above 8 192 windows per chunk)", "bound": "mfma",
                        "achieved": round(ach, 2), "peak": pk_g, "unit": "TFLOP/s", "frac": round(ach / pk_g, 4),
                        "alg_frac": round(ach / pk_g, 4),
                        "issued_frac": round(ach * (sf if args.precision == "fp32" else 1.0) / pk_g, 4),
                        "traffic": traffic_of("k_gru", n_g), "traffic_source": tsrc, "avg_launch_ms": round(ms_g / n_g, 3),
                        "alg_flop_per_window": fl_g,
                        "note": ("peak = dense 16-bit MFMA peak of MI355X_MICROARCH.md; alg_frac = algorithmic FLOPs (no padding, one MAC per "
                                 f"weight and window) / time / peak; issued_frac = alg_frac x {sf:.2f} MFMA MACs issued per algorithmic MAC (three "
                                 "16-bit terms per fp32 product, 16-feature chunk and 32-unit tile padding) = share of the matrix pipe's time the "
                                 "kernel keeps it busy") if args.precision == "fp32" else "algorithmic FLOPs / time vs the peak of the MFMA type used"}
            ms_a, n_a, _ = tot["k_gat"]
            roof_valu = {"kernel": "k_gat (temporal + feature attention layer: k_gath, the fp16-piece build, on this workload)", "bound": "valu",
                         "achieved": fams["k_gat"]["valu_tlaneops"], "peak": round(VALU_PEAK_TLANEOPS, 1), "unit": "T lane-op/s",
                         "frac": fams["k_gat"]["valu_frac"],
                         "frac_vs_r03_peak": round(fams["k_gat"]["valu_tlaneops"] / VALU_PEAK_TLANEOPS_R03, 4),
                         "traffic": traffic_of("k_gat", n_a), "traffic_source": tsrc, "avg_launch_ms": round(ms_a / n_a, 3),
                         "note": "bound by neither HBM nor the matrix pipe but by the vector ALU: 2 lane-operations per (query, key, "
                                 "embedding column) element of the GATv2 score (add, |.|-accumulate); peak = 4 SIMD-32 per CU x 2.4 GHz "
                                 "(MI355X_MICROARCH.md), frac_vs_r03_peak = against the 16-lanes/clk/SIMD figure rounds 1-3 used"}
            # `roofline` = the launch family that takes most of the step; both families are also in the line under their own names
            roof_gru["largest_family_by_time"] = dom
            # (the two families are within a few per cent of each other at this shape: which one `roofline` names can flip from box
            # to box -- roofline_mfma and roofline_valu always carry both)
            res["roofline"] = dict(roof_valu if dom == "k_gat" else roof_gru, largest_family_by_time=dom,
                                   families_ms_per_step={k: v["ms_per_step"] for k, v in fams.items()})
            res["roofline_mfma"] = roof_gru
            res["roofline_valu"] = roof_valu
        gbs = value * alg_bytes / 1e9
        res["hbm"] = {"alg_bytes_per_window": alg_bytes, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(gbs / HBM_PEAK_GBS / world, 5),
                      "note": "whole-forward algorithmic bytes rate per GPU vs HBM peak; the path is compute-bound"}
        if world == 1 and not args.no_sub:
            try:
                res["sub"] = sub_records(model, kw, dev, args.precision)
            except Exception as e:
                res["sub"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(sd, kw)
            except Exception as e:  # never lose the GPU number to a baseline problem
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
