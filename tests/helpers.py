"""Shared test helpers: golden fixtures (tests/golden/*.npz, generated from the reference by
tests/golden/make_golden.py) and tolerance gates."""
import hashlib
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_CASES = ["msl", "smap", "smd_1_1", "syn_v1_small", "syn_v2_embed", "syn_v2_wide", "syn_v1_default", "syn_c4"]
SHIPPED_CASES = ["msl", "smap", "smd_1_1"]
FP32_TOL = 1e-5   # BASELINE.json north_star: outputs within 1e-5 of the reference fp32 forward


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.kwargs = self.meta["kwargs"]
        self.x = torch.from_numpy(z["x"])
        self.preds = torch.from_numpy(z["preds"])
        self.recons = torch.from_numpy(z["recons"])
        self.preds64 = torch.from_numpy(z["preds64"])
        self.recons64 = torch.from_numpy(z["recons64"])
        self.h_end64 = torch.from_numpy(z["h_end64"])
        self.stages = {k[len("stage_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("stage_")}
        self._sd = {k[len("sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")} or None

    def build_model(self):
        """Our MTAD_GAT with this case's parameters (CPU).  Cases that do not store the state_dict
        re-create it from the recorded seeds; the digest check proves it equals the reference's."""
        from mtad_gat import MTAD_GAT
        if self._sd is not None:
            model = MTAD_GAT(**self.kwargs)
            model.load_state_dict(self._sd)   # strict
        else:
            torch.manual_seed(self.meta["init_seed"])
            model = MTAD_GAT(**self.kwargs)
            g = torch.Generator().manual_seed(self.meta["init_seed"] + 1)
            with torch.no_grad():
                model.feature_gat.bias.copy_(torch.randn(model.feature_gat.bias.shape, generator=g))
                model.temporal_gat.bias.copy_(torch.randn(model.temporal_gat.bias.shape, generator=g))
        assert sd_digest(model.state_dict()) == self.meta["sd_sha256"], "parameters differ from the reference's"
        return model.eval()

    def state_dict(self):
        return self._sd if self._sd is not None else self.build_model().state_dict()


def gate(ours, ref32, ref64=None, tol=FP32_TOL, what=""):
    """|ours - ref32| <= tol, or -- so a kernel *more* accurate than the reference's own float32
    rounding is not failed -- |ours - ref64| <= |ref32 - ref64| + tol."""
    ours = ours.detach().cpu().double()
    d32 = (ours - ref32.double()).abs().max().item()
    if d32 <= tol:
        return d32
    if ref64 is not None:
        d64 = (ours - ref64.double()).abs().max().item()
        noise = (ref32.double() - ref64.double()).abs().max().item()
        assert d64 <= noise + tol, f"{what}: |ours-ref32|={d32:.3e}, |ours-ref64|={d64:.3e} > ref noise {noise:.3e} + {tol}"
        return d32
    raise AssertionError(f"{what}: max abs diff {d32:.3e} > {tol}")


WIDE_CASES = ["msl_wide", "smap_wide", "smd_1_1_wide", "msl_c1"]


class WideCase:
    """>= 256-window fixture of a shipped checkpoint (tests/golden/make_golden.py --wide): reference
    outputs only; the input is regenerated from the recorded seed (or the stored C1 series) and
    checked against the recorded sha-256, the weights come from the small fixture of the same checkpoint."""

    def __init__(self, name):
        import hashlib
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.kwargs = self.meta["kwargs"]
        self.base = Case({"msl_wide": "msl", "smap_wide": "smap", "smd_1_1_wide": "smd_1_1", "msl_c1": "msl"}[name])
        self.preds = torch.from_numpy(z["preds"])
        self.recons = torch.from_numpy(z["recons"])
        self.preds64 = torch.from_numpy(z["preds64_f32"])
        self.recons64 = torch.from_numpy(z["recons64_f32"])
        self.h_end = torch.from_numpy(z["stage_h_end"])
        if "series" in z.files:
            self.series = torch.from_numpy(z["series"])
            w = self.kwargs["window_size"]
            self.x = torch.stack([self.series[i:i + w] for i in range(self.meta["batch"])])
        else:
            self.series = None
            g = torch.Generator().manual_seed(4321)
            self.x = torch.rand(self.meta["batch"], self.kwargs["window_size"], self.kwargs["n_features"], generator=g)
        assert hashlib.sha256(self.x.numpy().tobytes()).hexdigest() == self.meta["x_sha256"], "regenerated input differs"

    def build_model(self):
        return self.base.build_model()

    def state_dict(self):
        return self.base.state_dict()


# ---- the library's counter-based dropout, restated (csrc/mtadgat_device.h: mix32 / drop_window_key / drop_keep;
# stream numbers csrc/mtadgat_kernels.h: DROP_FEAT = 1, DROP_TEMP = 2, DROP_FC0 = 16).  The reference draws its masks from
# torch's generator (modules.py:90, :189, :310); to hold the reference's gradients WITH dropout as a fixture, the fixture
# generator injects these masks into the unmodified reference and the GPU test runs the kernels with the same seed.
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D); x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B); x ^= x >> np.uint32(16)
    return x


def dropout_keep_mask(seed, stream, p, n_windows, n_per_window, window0=0):
    """float32 (n_windows, n_per_window) of 0/1: element e of window w is kept iff the library's hash says so."""
    if p <= 0.0:
        return np.ones((n_windows, n_per_window), np.float32)
    t = p * 4294967296.0
    thresh = np.uint32(4294967295 if t >= 4294967295.0 else max(int(t), 1))
    w = np.arange(window0, window0 + n_windows, dtype=np.uint64)
    lo, hi = (w & np.uint64(0xFFFFFFFF)).astype(np.uint32), (w >> np.uint64(32)).astype(np.uint32)
    seed_lo, seed_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _mix32(lo ^ seed_lo)
        h = _mix32(h + hi * np.uint32(0x9E3779B9) + seed_hi + np.uint32((stream * 0x85EBCA6B) & 0xFFFFFFFF))
        e = np.arange(n_per_window, dtype=np.uint32) * np.uint32(0x9E3779B1) + np.uint32(0x7F4A7C15)
        keep = _mix32(h[:, None] ^ e[None, :]) >= thresh
    return keep.astype(np.float32)


def dropout_masks_like_the_library(kwargs, batch, seed, window0=0):
    """{"feat": (b,F,F), "temp": (b,W,W), "fc": [(b,hid)] * hidden layers} -- the layout of Engine.dropout_masks."""
    p, F_, W_ = kwargs["dropout"], kwargs["n_features"], kwargs["window_size"]
    hid, nh = kwargs["forecast_hid_dim"], kwargs["forecast_n_layers"]
    return {"feat": torch.from_numpy(dropout_keep_mask(seed, 1, p, batch, F_ * F_, window0)).reshape(batch, F_, F_),
            "temp": torch.from_numpy(dropout_keep_mask(seed, 2, p, batch, W_ * W_, window0)).reshape(batch, W_, W_),
            "fc": [torch.from_numpy(dropout_keep_mask(seed, 16 + i, p, batch, hid, window0)) for i in range(nh)]}


GRAD_CASES = ["grads_msl_eval", "grads_msl_masks", "grads_smd_eval", "grads_smd_masks", "grads_msl_b2000_masks", "grads_msl_b4100_eval"]


class GradCase:
    """Reference-held gradients of the training loss (tests/golden/make_golden.py --grads): every parameter's gradient from
    the unmodified reference model on CPU (float32), the per-parameter rounding noise of that computation (max |g32 - g64|
    against the same model in float64), the outputs at a few windows.  Inputs are regenerated from the recorded seeds and
    checked against their sha-256; weights come from the shipped checkpoint's small fixture."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.kwargs = self.meta["kwargs"]
        self.base = Case(self.meta["weights_from"])
        self.grads = {k[len("g/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g/")}
        self.noise = {k[len("n/"):]: float(z[k]) for k in z.files if k.startswith("n/")}
        self.preds_head = torch.from_numpy(z["preds_head"])
        self.recons_head = torch.from_numpy(z["recons_head"])
        self.loss = [float(v) for v in z["loss"]]
        self.x, self.y = grad_case_inputs(self.kwargs, self.meta["batch"], self.meta["xy_seed"])
        assert hashlib.sha256(self.x.numpy().tobytes() + self.y.numpy().tobytes()).hexdigest() == self.meta["xy_sha256"], \
            "torch.rand no longer reproduces the fixture's inputs"


def grad_case_inputs(kwargs, batch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, kwargs["window_size"], kwargs["n_features"], generator=g)
    y = torch.rand(batch, 1, kwargs["n_features"], generator=g)
    return x, y


def training_loss(preds, recons, x, y, target_dims):
    """reference training.py:113-124, verbatim in meaning: returns (forecast_loss, recon_loss)."""
    if target_dims is not None:
        x = x[:, :, target_dims]
        y = y[:, :, target_dims].squeeze(-1)
    if preds.ndim == 3:
        preds = preds.squeeze(1)
    if y.ndim == 3:
        y = y.squeeze(1)
    mse = torch.nn.MSELoss()
    return torch.sqrt(mse(y, preds)), torch.sqrt(mse(x, recons))
