#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference itself.

Run in the build container (where /root/reference exists):

    python -B tests/golden/make_golden.py            # the small per-stage fixtures
    python -B tests/golden/make_golden.py --wide     # >= 256-window fixtures (outputs only) + C1 statistics
    python -B tests/golden/make_golden.py --grads    # gradients of the training loss, held by the reference

It imports the *unmodified* reference modules (`/root/reference/mtad_gat.py`,
`modules.py`), runs `MTAD_GAT.forward` on CPU in eval mode on seeded inputs and
writes one `<case>.npz` per case with: the input, the reference-format
state_dict (shipped checkpoints and small synthetic models) or the init seed
(large synthetic models), the float32 outputs + per-stage intermediates, and
the outputs of the same reference model run in float64 (used to judge
rounding noise).

The GPU box has no /root/reference: tests read only the .npz files.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("MTAD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from mtad_gat import MTAD_GAT as RefMTAD  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SHIPPED = {
    "msl": ("output/MSL/27062021_111641/model.pt", dict(n_features=55, out_dim=1)),
    "smap": ("output/SMAP/27062021_112545/model.pt", dict(n_features=25, out_dim=1)),
    "smd_1_1": ("output/SMD/1-1/27062021_114402/model.pt", dict(n_features=38, out_dim=38)),
}
# ctor kwargs of the shipped runs (output/*/config.txt): fc_n_layers=3, rest default
SHIPPED_KW = dict(window_size=100, kernel_size=7, use_gatv2=True, gru_n_layers=1, gru_hid_dim=150,
                  forecast_n_layers=3, forecast_hid_dim=150, recon_n_layers=1, recon_hid_dim=150,
                  dropout=0.3, alpha=0.2)

# small / odd-shaped synthetic models: (name, ctor kwargs, batch, store_state_dict)
SYNTH = [
    ("syn_v1_small", dict(n_features=7, window_size=12, out_dim=7, kernel_size=3, use_gatv2=False,
                          feat_gat_embed_dim=5, time_gat_embed_dim=6, gru_n_layers=2, gru_hid_dim=20,
                          forecast_n_layers=2, forecast_hid_dim=24, recon_n_layers=2, recon_hid_dim=18,
                          dropout=0.2, alpha=0.2), 5, True),
    ("syn_v2_embed", dict(n_features=9, window_size=16, out_dim=3, kernel_size=5, use_gatv2=True,
                          feat_gat_embed_dim=5, time_gat_embed_dim=3, gru_n_layers=1, gru_hid_dim=33,
                          forecast_n_layers=1, forecast_hid_dim=40, recon_n_layers=1, recon_hid_dim=35,
                          dropout=0.2, alpha=0.1), 37, True),
    ("syn_v2_wide", dict(n_features=70, window_size=130, out_dim=70, kernel_size=7, use_gatv2=True,
                         gru_n_layers=1, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
                         recon_n_layers=1, recon_hid_dim=150, dropout=0.3, alpha=0.2), 3, False),
    ("syn_v1_default", dict(n_features=25, window_size=100, out_dim=1, kernel_size=7, use_gatv2=False,
                            gru_n_layers=1, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
                            recon_n_layers=1, recon_hid_dim=150, dropout=0.3, alpha=0.2), 4, False),
    # BASELINE.json configs[3] shape at oracle-able batch
    ("syn_c4", dict(n_features=512, window_size=256, out_dim=512, kernel_size=7, use_gatv2=True,
                    gru_n_layers=1, gru_hid_dim=150, forecast_n_layers=3, forecast_hid_dim=150,
                    recon_n_layers=1, recon_hid_dim=150, dropout=0.3, alpha=0.2), 1, False),
]
INIT_SEED = 0
X_SEED = 1234


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def stages(model, x):
    """Intermediates, re-using the reference's own sub-modules (mtad_gat.py:64-79)."""
    xc = model.conv(x)
    h_feat = model.feature_gat(xc)
    h_temp = model.temporal_gat(xc)
    h_cat = torch.cat([xc, h_feat, h_temp], dim=2)
    _, h_end = model.gru(h_cat)
    h_end = h_end.view(x.shape[0], -1)
    return dict(xc=xc, h_feat=h_feat, h_temp=h_temp, h_end=h_end)


def run_case(name, model, kwargs, batch, store_sd, meta_extra):
    model.eval()
    g = torch.Generator().manual_seed(X_SEED)
    x = torch.rand(batch, kwargs["window_size"], kwargs["n_features"], generator=g)
    with torch.no_grad():
        preds, recons = model(x)
        st = stages(model, x)
        m64 = RefMTAD(**{k: v for k, v in kwargs.items()}).double()
        m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        m64.eval()
        p64, r64 = m64(x.double())
        st64 = stages(m64, x.double())
    out = dict(x=x.numpy(), preds=preds.numpy(), recons=recons.numpy(),
               preds64=p64.numpy(), recons64=r64.numpy(), h_end64=st64["h_end"].numpy())
    for k, v in st.items():
        out["stage_" + k] = v.numpy()
    sd = model.state_dict()
    if store_sd:
        for k, v in sd.items():
            out["sd/" + k] = v.numpy()
    meta = dict(name=name, kwargs=kwargs, batch=batch, x_seed=X_SEED, init_seed=INIT_SEED,
                sd_sha256=sd_digest(sd), store_sd=store_sd, torch=torch.__version__, **meta_extra)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: b={batch} preds{tuple(preds.shape)} recons{tuple(recons.shape)} "
          f"|p32-p64|={float((preds.double()-p64).abs().max()):.2e} "
          f"|r32-r64|={float((recons.double()-r64).abs().max()):.2e} "
          f"-> {os.path.getsize(path)/1e6:.2f} MB")


WIDE_BATCH = 300      # >= 256 with a ragged tail (256 + 44): SURVEY.md section 8d's parity gate
WIDE_X_SEED = 4321


def c1_series(n_rows, n_features, seed=0):
    """BASELINE config 1 / SURVEY 8d input statistics: column 0 = 0.5 sin(2 pi t / 97) + 0.05 N(0,1)
    (negative values: outside the [0,1] a MinMax-scaled training split would give), the other
    columns Bernoulli(0.05) on/off telemetry."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_rows)
    s = np.zeros((n_rows, n_features), dtype=np.float32)
    s[:, 0] = 0.5 * np.sin(2 * np.pi * t / 97.0) + 0.05 * rng.standard_normal(n_rows)
    s[:, 1:] = (rng.random((n_rows, n_features - 1)) < 0.05).astype(np.float32)
    s[17, 3] = 1.7          # a few out-of-range values in the binary columns as well
    s[201, 9] = -0.4
    return s


def run_wide(name, model, kwargs, x, extra):
    """Outputs only (x is regenerated from the seed / stored series by the tests); the float64 reference
    is stored rounded to float32 -- 6e-8 relative, far below the 1e-5 gate it serves."""
    model.eval()
    with torch.no_grad():
        preds, recons = model(x)
        st = stages(model, x)
        m64 = RefMTAD(**kwargs).double()
        m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        m64.eval()
        p64, r64 = m64(x.double())
    out = dict(preds=preds.numpy(), recons=recons.numpy(), preds64_f32=p64.float().numpy(),
               recons64_f32=r64.float().numpy(), stage_h_end=st["h_end"].numpy(), **extra)
    meta = dict(name=name, kwargs=kwargs, batch=int(x.shape[0]), x_sha256=hashlib.sha256(x.numpy().tobytes()).hexdigest(),
                sd_sha256=sd_digest(model.state_dict()), torch=torch.__version__)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: b={x.shape[0]} |p32-p64|={float((preds.double()-p64).abs().max()):.2e} "
          f"|r32-r64|={float((recons.double()-r64).abs().max()):.2e} -> {os.path.getsize(path)/1e6:.2f} MB")


def main_wide():
    """>= 256-window fixtures for the three shipped checkpoints (weights come from <case>.npz at test time)
    and the C1-statistics series through the MSL checkpoint."""
    torch.set_num_threads(os.cpu_count())
    for name, (rel, dims) in SHIPPED.items():
        kwargs = dict(SHIPPED_KW, **dims)
        model = RefMTAD(**kwargs)
        model.load_state_dict(torch.load(os.path.join(REF, rel), map_location="cpu"))
        g = torch.Generator().manual_seed(WIDE_X_SEED)
        x = torch.rand(WIDE_BATCH, kwargs["window_size"], kwargs["n_features"], generator=g)
        run_wide(name + "_wide", model, kwargs, x, {})
        if name == "msl":
            series = c1_series(100 + 320, 55)
            xs = torch.from_numpy(np.stack([series[i:i + 100] for i in range(320)]))
            run_wide("msl_c1", model, kwargs, xs, dict(series=series))


def main():
    torch.set_num_threads(os.cpu_count())
    for name, (rel, dims) in SHIPPED.items():
        kwargs = dict(SHIPPED_KW, **dims)
        model = RefMTAD(**kwargs)
        sd = torch.load(os.path.join(REF, rel), map_location="cpu")
        model.load_state_dict(sd)           # strict
        run_case(name, model, kwargs, 6, True, dict(source=rel))
    for name, kwargs, batch, store in SYNTH:
        torch.manual_seed(INIT_SEED)
        model = RefMTAD(**kwargs)
        # exercise the attention-bias path (reference initialises it to zeros, modules.py:60,161)
        g = torch.Generator().manual_seed(INIT_SEED + 1)
        with torch.no_grad():
            model.feature_gat.bias.copy_(torch.randn(model.feature_gat.bias.shape, generator=g))
            model.temporal_gat.bias.copy_(torch.randn(model.temporal_gat.bias.shape, generator=g))
        run_case(name, model, kwargs, batch, store,
                 dict(source="torch.manual_seed(%d) ctor init + randn GAT biases (seed %d)"
                      % (INIT_SEED, INIT_SEED + 1)))


# ---- gradients of the training step, held by the reference (training.py:106-127) ------------------------------------
# (case name, shipped checkpoint, batch, variant, target_dims): batches chosen so that the library's training step runs
# its window-per-workgroup recurrence kernels (<= 1792 windows), its 16-window-group kernels (<= 4096) and its
# hidden-tile-split kernels (above).  Variant "eval": model.eval() with autograd on -- the deterministic function.
# Variant "masks": model.train(), the library's counter-based dropout masks (tests/helpers.py restates the hash) injected
# into the reference's own dropout calls (torch.dropout in the attention layers, F.dropout behind nn.Dropout).
GRAD_CASES = [
    ("grads_msl_eval", "msl", 32, "eval", [0]),
    ("grads_msl_masks", "msl", 32, "masks", [0]),
    ("grads_smd_eval", "smd_1_1", 33, "eval", None),
    ("grads_smd_masks", "smd_1_1", 33, "masks", None),
    ("grads_msl_b2000_masks", "msl", 2000, "masks", [0]),
    ("grads_msl_b4100_eval", "msl", 4100, "eval", [0]),
]
GRAD_XY_SEED = 777
GRAD_CHUNK = 250
GRAD_DROP_SEED = (0x5EED << 32) | 0x1234ABCD      # both seed words in use


class _InjectedDropout:
    """While active, the reference's dropout calls take their keep-masks from a queue (in call order: feature attention,
    temporal attention, the forecasting head's hidden layers) instead of torch's generator."""

    def __init__(self, masks, p):
        self.queue = [masks["feat"], masks["temp"]] + list(masks["fc"])
        self.scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        self.p = p

    def _apply(self, inp, p, train):
        assert train and abs(p - self.p) < 1e-12 and self.queue, "unexpected dropout call in the reference forward"
        m = self.queue.pop(0).to(inp.dtype)
        assert m.shape == inp.shape, (tuple(m.shape), tuple(inp.shape))
        return inp * (m * float(self.scale))

    def __enter__(self):
        import torch.nn.functional as Fn
        self._saved = (torch.dropout, Fn.dropout)
        torch.dropout = lambda inp, p, train: self._apply(inp, p, train)
        Fn.dropout = lambda inp, p=0.5, training=True, inplace=False: self._apply(inp, p, training)
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as Fn
        torch.dropout, Fn.dropout = self._saved
        assert exc[0] is not None or not self.queue, "the reference made fewer dropout calls than there are masks"


def main_grads():
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import dropout_masks_like_the_library, grad_case_inputs, training_loss
    torch.set_num_threads(os.cpu_count())
    for name, ckpt, batch, variant, target_dims in GRAD_CASES:
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--force" not in sys.argv:
            print(f"{name}: exists (use --force to regenerate)")
            continue
        rel, dims = SHIPPED[ckpt]
        kwargs = dict(SHIPPED_KW, **dims)
        sd = torch.load(os.path.join(REF, rel), map_location="cpu")
        x, y = grad_case_inputs(kwargs, batch, GRAD_XY_SEED)
        masks = dropout_masks_like_the_library(kwargs, batch, GRAD_DROP_SEED) if variant == "masks" else None
        res = {}
        for dt in (torch.float32, torch.float64):
            model = RefMTAD(**kwargs).to(dt)
            model.load_state_dict({k: v.to(dt) for k, v in sd.items()})
            model.train() if variant == "masks" else model.eval()
            if batch <= GRAD_CHUNK:
                if masks is not None:
                    with _InjectedDropout(masks, kwargs["dropout"]):
                        preds, recons = model(x.to(dt))
                else:
                    preds, recons = model(x.to(dt))
                inj = None
            else:
                # the (b, W, W, 2F) attention tensors of the reference do not fit the host at this batch: the same model
                # and the same loss over the whole batch, with the forward of GRAD_CHUNK-window slices re-computed in
                # backward (torch.utils.checkpoint) -- no algebra of ours between the reference's forward and its loss
                from torch.utils.checkpoint import checkpoint
                inj = _InjectedDropout(masks, kwargs["dropout"]).__enter__() if masks is not None else None

                def run(xi, lo, hi):
                    if inj is not None:
                        inj.queue = [masks["feat"][lo:hi], masks["temp"][lo:hi]] + [m[lo:hi] for m in masks["fc"]]
                    return model(xi)

                outs = [checkpoint(run, x[lo:lo + GRAD_CHUNK].to(dt), lo, min(lo + GRAD_CHUNK, batch), use_reentrant=False)
                        for lo in range(0, batch, GRAD_CHUNK)]
                preds, recons = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            fl, rl = training_loss(preds, recons, x.to(dt), y.to(dt), target_dims)
            (fl + rl).backward()          # training.py:124-126
            if inj is not None:
                inj.queue = []
                inj.__exit__(None, None, None)
            res[dt] = dict(grads={n: p.grad.detach() for n, p in model.named_parameters()}, preds=preds.detach(),
                           recons=recons.detach(), loss=(float(fl), float(rl)))
        r32, r64 = res[torch.float32], res[torch.float64]
        out = {"g/" + n: g.numpy() for n, g in r32["grads"].items()}
        worst = 0.0
        for n, g in r32["grads"].items():
            noise = float((g.double() - r64["grads"][n]).abs().max())
            out["n/" + n] = np.float64(noise)
            worst = max(worst, noise / (1e-5 + 1e-4 * float(g.abs().max())))
        out["preds_head"] = r32["preds"][:64].numpy()
        out["recons_head"] = r32["recons"][:8].numpy()
        out["loss"] = np.array(r64["loss"], np.float64)
        meta = dict(name=name, kwargs=kwargs, batch=batch, variant=variant, target_dims=target_dims, weights_from=ckpt,
                    xy_seed=GRAD_XY_SEED, drop_seed=GRAD_DROP_SEED if variant == "masks" else None,
                    xy_sha256=hashlib.sha256(x.numpy().tobytes() + y.numpy().tobytes()).hexdigest(),
                    sd_sha256=sd_digest(sd), torch=torch.__version__)
        out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: b={batch} {variant} loss64=({r64['loss'][0]:.6f}, {r64['loss'][1]:.6f}) loss32=({r32['loss'][0]:.6f}, "
              f"{r32['loss'][1]:.6f}) reference's own fp32 noise = {worst:.3f} x the test gate -> {os.path.getsize(path)/1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    if "--grads" in sys.argv:
        main_grads()
    elif "--wide" in sys.argv:
        main_wide()
    else:
        main()
