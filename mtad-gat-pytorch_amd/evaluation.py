"""Anomaly-score post-processing on the GPU, under the reference's own function names.

Mirrors `eval_methods.py` (find_epsilon :189-236, adjust_predicts :6-55, calc_point2point :58-72,
epsilon_eval :164-186, bf_search :117-158) and the score arithmetic of `Predictor.get_score`
(prediction.py:72-91) on device-resident arrays: the O(N) passes -- moments, the 19-z epsilon table with its
+-49-sample dilation, point-adjusted confusion counts for one threshold or a whole sweep -- are HIP kernels
(csrc/mtadgat_eval.hip behind the `mtadgat_eval_*` C entry points); the scalar bookkeeping on top (which z /
threshold wins, precision / recall / F1 from the counts) is done here in float64 exactly as the reference writes
it.  Results equal the reference's dictionaries (tests/test_gpu_eval.py: the shipped MSL run's summary.txt).

POT (`pot_eval`: SPOT's Grimshaw fit, spot.py) is not ported: it is a sequential scalar algorithm over the
peaks only; the reference's implementation runs unchanged on `scores.cpu().numpy()`.
"""
import ctypes
import math

import numpy as np
import torch

import _native

_c_double_p = ctypes.POINTER(ctypes.c_double)


def _lib():
    lib = _native.load_library()
    if not getattr(lib, "_eval_bound", False):
        vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
        lib.mtadgat_eval_scores.argtypes = [vp, vp, vp, i64, ci, i64, vp, f32, vp, vp, vp]
        lib.mtadgat_eval_moments.argtypes = [vp, i64, vp, _c_double_p, vp]
        lib.mtadgat_eval_epsilon_table.argtypes = [vp, i64, _c_double_p, ci, ci, vp, _c_double_p, vp]
        lib.mtadgat_eval_point_adjust.argtypes = [vp, vp, i64, _c_double_p, ci, ci, ci, vp, _c_double_p, vp]
        lib._eval_bound = True
    return lib


def _dev1d(t, dtype, name):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise RuntimeError(f"{name} must be a tensor on the GPU (the evaluation kernels are HIP only)")
    t = t.detach().reshape(-1)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"mtadgat {what} failed (status {rc})")


def anomaly_scores(preds, recons, values, window_size, target_dims=None, gamma=1.0):
    """a_i[d] = |y_hat_i[d] - x_{i+W}[d]| + gamma |recon_i[d] - x_{i+W}[d]| and its mean over d
    (prediction.py:72-91, before the optional per-dimension scaling).  Returns (global (n,), per_dim (n, d))."""
    lib = _lib()
    n, d = preds.shape
    if values.ndim != 2 or values.shape[0] < window_size + n:
        raise ValueError(f"values must hold at least window_size + n = {window_size + n} rows (the series the {n} windows were cut "
                         f"from), got shape {tuple(values.shape)}")
    if recons.shape[0] != n or recons.shape[-1] != d:
        raise ValueError(f"recons {tuple(recons.shape)} does not match preds {tuple(preds.shape)}")
    actual = values[window_size:window_size + n].float().contiguous()
    dims = None
    if target_dims is not None:
        dims = torch.tensor([target_dims] if isinstance(target_dims, int) else list(target_dims), dtype=torch.int32, device=preds.device)
        if dims.numel() != d:
            raise RuntimeError("target_dims do not match the model's out_dim")
        tl = [target_dims] if isinstance(target_dims, int) else list(target_dims)
        if min(tl) < 0 or max(tl) >= actual.shape[1]:
            raise ValueError(f"target_dims {tl} outside the series' {actual.shape[1]} columns")
    elif actual.shape[1] != d:
        raise RuntimeError("out_dim differs from the number of features: pass target_dims")
    p, r = preds.float().contiguous(), recons.float().contiguous()
    per_dim = torch.empty((n, d), dtype=torch.float32, device=preds.device)
    glob = torch.empty((n,), dtype=torch.float32, device=preds.device)
    with torch.cuda.device(preds.device):
        _check(lib.mtadgat_eval_scores(p.data_ptr(), r.data_ptr(), actual.data_ptr(), n, d, actual.shape[1],
                                       dims.data_ptr() if dims is not None else None, float(gamma), per_dim.data_ptr(),
                                       glob.data_ptr(), _stream(preds)), "eval_scores")
    return glob, per_dim


def find_epsilon(errors, reg_level=1):
    """Threshold of Hundman et al. as the reference computes it (eval_methods.py:189-236)."""
    lib = _lib()
    e = _dev1d(errors, torch.float32, "errors")
    n = e.numel()
    scratch = torch.empty(64 + 4 * 64, dtype=torch.float64, device=e.device)
    mom = (ctypes.c_double * 2)()
    with torch.cuda.device(e.device):
        _check(lib.mtadgat_eval_moments(e.data_ptr(), n, scratch.data_ptr(), mom, _stream(e)), "eval_moments")
    mean = mom[0] / n
    sd = math.sqrt(max(mom[1] / n - mean * mean, 0.0))
    zs = np.arange(2.5, 12, 0.5)
    eps = mean + sd * zs
    tab = (ctypes.c_double * (4 * len(zs)))()
    with torch.cuda.device(e.device):
        _check(lib.mtadgat_eval_epsilon_table(e.data_ptr(), n, eps.ctypes.data_as(_c_double_p), len(zs), 49, scratch.data_ptr(), tab,
                                              _stream(e)), "eval_epsilon_table")
    best, max_score = None, -10000000
    mean64, sd64 = np.float64(mean), np.float64(sd)
    for k in range(len(zs)):
        ps, ps2, pc, dil = tab[4 * k], tab[4 * k + 1], tab[4 * k + 2], tab[4 * k + 3]
        if dil > 0:
            if pc > 0:
                pm = ps / pc
                psd = math.sqrt(max(ps2 / pc - pm * pm, 0.0))
            else:
                pm = psd = float("nan")         # np.mean / np.std of an empty pruned set
            # numpy float64 arithmetic as in the reference: constant scores (sd == 0) or a zero mean give nan / inf here,
            # not ZeroDivisionError; a nan score fails the comparison below and the z is skipped (eval_methods.py:220-231)
            with np.errstate(divide="ignore", invalid="ignore"):
                mean_perc_decrease = (mean64 - np.float64(pm)) / mean64
                sd_perc_decrease = (sd64 - np.float64(psd)) / sd64
                denom = 1 if reg_level == 0 else (dil if reg_level == 1 else dil ** 2)
                score = float((mean_perc_decrease + sd_perc_decrease) / denom)
            # `>=`: among equal scores the reference keeps the last z.  Two z with the same pruned set have the same
            # score exactly in the reference; here their float64 sums come from atomics in varying order, so
            # "equal" is taken with a 1e-9 relative margin.
            at_least = score >= max_score or (math.isfinite(max_score) and score >= max_score - 1e-9 * abs(max_score))
            if at_least and dil < n * 0.5:
                max_score, best = max(score, max_score), float(eps[k])
    if best is None:
        best = float(e.max().item())
    return best


def point_adjust_counts(score, label, thresholds, compare_f32=False, max_segments=65536):
    """adjust_predicts + calc_point2point for every threshold in one launch.
    Returns an (n_thr, 6) float64 array: TP, TN, FP, FN, latency sum, detected segments."""
    lib = _lib()
    s = _dev1d(score, torch.float32, "score")
    if label.dtype == torch.bool:
        lab = _dev1d(label, torch.uint8, "label")
    else:
        lab = _dev1d((label > 0.1), torch.uint8, "label")
    if lab.numel() != s.numel():
        raise ValueError("score and label must have the same length")
    thr = np.ascontiguousarray(np.asarray(thresholds, dtype=np.float64).reshape(-1))
    nt = thr.size
    scratch = torch.empty(7 * nt + max_segments + 2, dtype=torch.float64, device=s.device)
    out = np.zeros((nt, 6), dtype=np.float64)
    with torch.cuda.device(s.device):
        _check(lib.mtadgat_eval_point_adjust(s.data_ptr(), lab.data_ptr(), s.numel(), thr.ctypes.data_as(_c_double_p), nt,
                                             1 if compare_f32 else 0, max_segments, scratch.data_ptr(),
                                             out.ctypes.data_as(_c_double_p), _stream(s)), "eval_point_adjust")
    return out


def _scores_from_counts(tp, tn, fp, fn):
    """F1 / precision / recall with the reference's 1e-5 guards (eval_methods.py:6-21)."""
    prec = tp / (tp + fp + 0.00001)
    rec = tp / (tp + fn + 0.00001)
    return 2 * prec * rec / (prec + rec + 0.00001), prec, rec


def _result(counts, threshold, **extra):
    f1, prec, rec = _scores_from_counts(counts[0], counts[1], counts[2], counts[3])
    out = {"f1": f1, "precision": prec, "recall": rec, "TP": counts[0], "TN": counts[1], "FP": counts[2], "FN": counts[3],
           "threshold": threshold, "latency": counts[4] / (counts[5] + 1e-4)}
    out.update(extra)
    return out


def epsilon_eval(train_scores, test_scores, test_labels, reg_level=1):
    """Threshold from the training scores (find_epsilon), point-adjusted metrics on the test scores (eval_methods.py:164-186)."""
    eps = find_epsilon(train_scores, reg_level)
    if test_labels is None:
        return {"threshold": eps, "reg_level": reg_level}
    return _result(point_adjust_counts(test_scores, test_labels, [eps])[0], eps, reg_level=reg_level)


def bf_search(score, label, start, end=None, step_num=1, display_freq=1, verbose=False):
    """Best-F1 threshold sweep (eval_methods.py:117-158): all thresholds evaluated by one kernel launch."""
    if step_num is None or end is None:
        end, step_num = start, 1
    # the reference walks the thresholds by repeated addition in a Python float: the same sums, so the same thresholds
    increment = (end - start) / float(step_num)
    grid, at = [], start
    for _ in range(step_num):
        at += increment
        grid.append(at)
    table = point_adjust_counts(score, label, grid, compare_f32=True)      # float32 array > Python float: a float32 comparison
    best_row, best_thr, best_f1 = None, 0.0, -1.0
    for thr, row in zip(grid, table):
        f1 = _scores_from_counts(row[0], row[1], row[2], row[3])[0]
        if f1 > best_f1:                               # strict: the first of equal F1 values wins, as in the reference
            best_row, best_thr, best_f1 = row, thr, f1
    if best_row is None:
        return {"f1": -1.0, "precision": -1.0, "recall": -1.0, "TP": -1.0, "TN": -1.0, "FP": -1.0, "FN": -1.0, "threshold": 0.0, "latency": 0}
    return _result(best_row, best_thr)
