"""The CPU restatement of the threshold evaluation (oracle/eval_oracle.py) pinned against the shipped MSL run's
scores and summary.txt (tests/golden/msl_eval.npz), and -- where the reference tree exists -- against the
reference's own eval_methods functions on adversarial label layouts."""
import json
import os
import sys
import types

import numpy as np
import pytest

from oracle import eval_oracle as eo

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    z = np.load(os.path.join(HERE, "golden", "msl_eval.npz"))
    return z, json.loads(bytes(z["summary"]).decode())


def test_epsilon_and_best_f1_reproduce_the_shipped_summary():
    z, summary = _fixture()
    e = eo.epsilon_eval(z["train_scores"], z["test_scores"], z["test_labels"], reg_level=0)       # MSL: reg_level 0 (predict.py:143)
    ref = summary["epsilon_result"]
    for k in ("TP", "TN", "FP", "FN"):
        assert e[k] == ref[k], k
    assert abs(e["threshold"] - ref["threshold"]) <= 1e-6 * ref["threshold"]
    assert abs(e["f1"] - ref["f1"]) <= 1e-9 and abs(e["latency"] - ref["latency"]) <= 1e-9
    b = eo.bf_search(z["test_scores"], z["test_labels"], 0.01, 2, 100)
    ref = summary["bf_result"]
    assert b["threshold"] == ref["threshold"]
    for k in ("TP", "TN", "FP", "FN"):
        assert b[k] == ref[k], k
    assert abs(b["f1"] - ref["f1"]) <= 1e-12 and abs(b["latency"] - ref["latency"]) <= 1e-9
    assert abs(eo.find_epsilon(z["train_score_0"], reg_level=2) - float(z["thresh_0"])) <= 1e-6 * float(z["thresh_0"])


@pytest.mark.skipif(not os.path.isfile("/root/reference/eval_methods.py"), reason="reference tree not present")
def test_point_adjust_equals_the_reference_state_machine():
    if "more_itertools" not in sys.modules:
        try:
            import more_itertools  # noqa: F401
        except ModuleNotFoundError:
            sys.modules["more_itertools"] = types.ModuleType("more_itertools")
    sys.path.append("/root/reference")
    sys.dont_write_bytecode = True
    try:
        import eval_methods as ref
    finally:
        sys.path.remove("/root/reference")
    rng = np.random.default_rng(0)
    for trial in range(30):
        n = int(rng.integers(5, 400))
        label = (rng.random(n) < 0.3).astype(np.float32)
        if trial % 3 == 0:
            label[: int(rng.integers(1, 4))] = 1            # an anomaly segment that starts at index 0
        if trial % 7 == 0:
            label[:] = 0
        score = rng.random(n).astype(np.float32)
        thr = float(rng.random())
        p_ref, l_ref = ref.adjust_predicts(score, label, thr, calc_latency=True)
        p, l = eo.point_adjust(score, label, thr)
        assert np.array_equal(p, p_ref) and abs(l - l_ref) <= 1e-12, trial
        assert np.allclose(eo.confusion(p, label), ref.calc_point2point(p_ref, label.astype(np.int64)), rtol=0, atol=1e-12)


def test_find_epsilon_degenerate_scores_follow_numpy_semantics():
    """Constant training scores: sd = 0, every z gives epsilon = mean, every point is 'anomalous', the pruned set is empty and
    the reference's score is nan (eval_methods.py:197-231) -> no z is accepted and the threshold is max(e_s)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert eo.find_epsilon(np.full(500, 0.25), reg_level=1) == 0.25
        assert eo.find_epsilon(np.zeros(300), reg_level=0) == 0.0
        e = np.zeros(4000)
        e[100], e[2000] = 3.0, -3.0                      # zero mean, non-zero sd: a division by zero mean (inf), not an exception
        assert np.isfinite(eo.find_epsilon(e, reg_level=1))
