"""Device evaluation kernels (csrc/mtadgat_eval.hip via evaluation.py) against the known answers of the shipped MSL
run (its summary.txt) and against the CPU oracle on adversarial layouts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    z = np.load(os.path.join(HERE, "golden", "msl_eval.npz"))
    return z, json.loads(bytes(z["summary"]).decode())


def test_known_answers_of_the_shipped_msl_run(gpu_device):
    import evaluation as ev
    z, summary = _fixture()
    tr = torch.from_numpy(z["train_scores"]).to(gpu_device)
    te = torch.from_numpy(z["test_scores"]).to(gpu_device)
    lab = torch.from_numpy(z["test_labels"]).to(gpu_device).bool()
    e = ev.epsilon_eval(tr, te, lab, reg_level=0)
    ref = summary["epsilon_result"]
    for k in ("TP", "TN", "FP", "FN"):
        assert e[k] == ref[k], (k, e[k], ref[k])
    assert abs(e["threshold"] - ref["threshold"]) <= 1e-6 * ref["threshold"]
    assert abs(e["f1"] - ref["f1"]) <= 1e-9 and abs(e["latency"] - ref["latency"]) <= 1e-9 and e["reg_level"] == 0
    b = ev.bf_search(te, lab, start=0.01, end=2, step_num=100)
    ref = summary["bf_result"]
    assert b["threshold"] == ref["threshold"]
    for k in ("TP", "TN", "FP", "FN"):
        assert b[k] == ref[k], (k, b[k], ref[k])
    assert abs(b["f1"] - ref["f1"]) <= 1e-12 and abs(b["latency"] - ref["latency"]) <= 1e-9
    # per-feature threshold of predict_anomalies (prediction.py:141-144: reg_level 2)
    t0 = ev.find_epsilon(torch.from_numpy(z["train_score_0"]).to(gpu_device), reg_level=2)
    assert abs(t0 - float(z["thresh_0"])) <= 1e-6 * float(z["thresh_0"])


def test_score_arithmetic_of_get_score(gpu_device):
    import evaluation as ev
    z, _ = _fixture()
    n, W = 20000, 100
    values = torch.zeros(W + n, 3, device=gpu_device)
    values[W:, 1] = torch.from_numpy(z["true"]).to(gpu_device)
    preds = torch.from_numpy(z["forecast"]).to(gpu_device)[:, None]
    recons = torch.from_numpy(z["recon"]).to(gpu_device)[:, None]
    glob, per_dim = ev.anomaly_scores(preds, recons, values, W, target_dims=[1], gamma=1.0)
    assert (per_dim[:, 0].cpu() - torch.from_numpy(z["a_score_0"])).abs().max().item() <= 1e-6
    assert torch.equal(glob, per_dim[:, 0])


def test_point_adjust_edge_cases_against_the_oracle(gpu_device):
    import evaluation as ev
    rng = np.random.default_rng(1)
    for trial in range(12):
        n = int(rng.integers(50, 3000))
        label = (rng.random(n) < 0.2).astype(np.uint8)
        if trial % 3 == 0:
            label[: int(rng.integers(1, 5))] = 1             # segment starting at index 0: never back-filled to 0
        if trial == 4:
            label[:] = 0
        if trial == 5:
            label[:] = 1
        score = rng.random(n).astype(np.float32)
        thrs = rng.random(7)
        got = ev.point_adjust_counts(torch.from_numpy(score).to(gpu_device), torch.from_numpy(label).to(gpu_device).bool(), thrs)
        for t, c in zip(thrs, got):
            pred, lat = eo.point_adjust(score, label, t)
            f = eo.confusion(pred, label)
            assert tuple(c[:4]) == (f[3], f[4], f[5], f[6]), (trial, t)
            assert abs(c[4] / (c[5] + 1e-4) - lat) <= 1e-9
        e = score * 0.1 + (rng.random(n) < 0.01) * rng.random(n).astype(np.float32) * 3
        for reg in (0, 1, 2):
            a, b = ev.find_epsilon(torch.from_numpy(e.astype(np.float32)).to(gpu_device), reg), eo.find_epsilon(e.astype(np.float32), reg)
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (trial, reg, a, b)


def test_find_epsilon_degenerate_scores(gpu_device):
    """sd == 0 / mean == 0: numpy semantics of the reference (nan / inf scores), not ZeroDivisionError."""
    import warnings
    import evaluation as ev
    e = np.zeros(4000, np.float32)
    e[100], e[2000] = 3.0, -3.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cases = [np.full(500, 0.25, np.float32), np.zeros(300, np.float32), e]
        for c in cases:
            for reg in (0, 1, 2):
                got = ev.find_epsilon(torch.from_numpy(c).to(gpu_device), reg_level=reg)
                ref = eo.find_epsilon(c, reg_level=reg)
                assert abs(got - ref) <= 1e-6 * max(1.0, abs(ref)), (got, ref, reg)


def test_anomaly_scores_rejects_mismatched_inputs(gpu_device):
    import evaluation as ev
    preds = torch.zeros(10, 1, device=gpu_device)
    recons = torch.zeros(10, 1, device=gpu_device)
    with pytest.raises(ValueError):
        ev.anomaly_scores(preds, recons, torch.zeros(105, 3, device=gpu_device), 100, target_dims=[1])     # series too short
    with pytest.raises(ValueError):
        ev.anomaly_scores(preds, recons, torch.zeros(110, 3, device=gpu_device), 100, target_dims=[3])     # column 3 of 3
    with pytest.raises(ValueError):
        ev.anomaly_scores(preds, torch.zeros(9, 1, device=gpu_device), torch.zeros(110, 3, device=gpu_device), 100, target_dims=[1])
