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
