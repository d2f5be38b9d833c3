"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's threshold evaluation, the checker for the
device kernels in csrc/mtadgat_eval.hip.  Only tests/ may import this file; nothing in the product path does.

Restates, with the reference's semantics but written segment-wise instead of as its per-sample state machine:
  * find_epsilon        eval_methods.py:189-236 (Hundman et al. threshold; +-49-sample dilation of the exceedances)
  * adjust_predicts     eval_methods.py:6-55    (point adjust: a detected anomaly segment counts in full)
  * calc_point2point    eval_methods.py:58-72
  * bf_search           eval_methods.py:117-158
Pinned by tests/test_oracle_eval.py against the shipped MSL run (tests/golden/msl_eval.npz: its scores, labels and
the numbers of its summary.txt) and, where the reference tree exists, against the reference functions themselves.
"""
import numpy as np


def segments(label):
    """[(first, last)] of the runs of label > 0.1."""
    a = np.asarray(label) > 0.1
    d = np.diff(np.concatenate(([0], a.astype(np.int8), [0])))
    return list(zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1) - 1))


def point_adjust(score, label, threshold, compare_f32=False):
    """(predict, latency): what adjust_predicts(score, label, threshold, calc_latency=True) returns."""
    s = np.asarray(score)
    predict = (s > np.float32(threshold)) if compare_f32 else (s.astype(np.float64) > float(threshold))
    predict = predict.copy()
    lat, det = 0, 0
    for s0, s1 in segments(label):
        hit = np.flatnonzero(predict[s0:s1 + 1])
        if hit.size == 0:
            continue
        first = s0 + hit[0]
        det += 1
        b0 = max(s0, 1)                      # the reference's back-fill loop `range(i, 0, -1)` never reaches index 0
        if first > b0:
            lat += first - b0
        keep0 = predict[0]
        predict[s0:s1 + 1] = True
        if s0 == 0 and first > 0:
            predict[0] = keep0
    return predict, lat / (det + 1e-4)


def confusion(predict, label):
    actual = np.asarray(label) > 0.1
    tp = float(np.sum(predict & actual)); tn = float(np.sum(~predict & ~actual))
    fp = float(np.sum(predict & ~actual)); fn = float(np.sum(~predict & actual))
    precision = tp / (tp + fp + 0.00001)
    recall = tp / (tp + fn + 0.00001)
    f1 = 2 * precision * recall / (precision + recall + 0.00001)
    return f1, precision, recall, tp, tn, fp, fn


def find_epsilon(errors, reg_level=1):
    e = np.asarray(errors, dtype=np.float64)
    n = e.size
    mean, sd = e.mean(), e.std()
    best, max_score = None, -10000000
    for z in np.arange(2.5, 12, 0.5):
        eps = mean + sd * z
        above = e >= eps
        if not above.any():
            continue
        # dilation by +-49 samples through a prefix sum
        c = np.concatenate(([0], np.cumsum(above)))
        idx = np.arange(n)
        dil = int(np.sum(c[np.minimum(idx + 49, n - 1) + 1] - c[np.maximum(idx - 49, 0)] > 0))
        pruned = e[~above]
        score = ((mean - pruned.mean()) / mean + (sd - pruned.std()) / sd) / (1 if reg_level == 0 else dil if reg_level == 1 else dil ** 2)
        if score >= max_score and dil < n * 0.5:
            max_score, best = score, eps
    return float(e.max()) if best is None else float(best)


def epsilon_eval(train_scores, test_scores, test_labels, reg_level=1):
    thr = find_epsilon(train_scores, reg_level)
    pred, lat = point_adjust(test_scores, test_labels, thr)
    f = confusion(pred, test_labels)
    return dict(f1=f[0], precision=f[1], recall=f[2], TP=f[3], TN=f[4], FP=f[5], FN=f[6], threshold=thr, latency=lat, reg_level=reg_level)


def bf_search(score, label, start, end, step_num):
    thr, best, best_t, best_l = start, (-1.0,) * 7, 0.0, 0
    for _ in range(step_num):
        thr += (end - start) / float(step_num)
        pred, lat = point_adjust(score, label, thr, compare_f32=True)
        f = confusion(pred, label)
        if f[0] > best[0]:
            best, best_t, best_l = f, thr, lat
    return dict(f1=best[0], precision=best[1], recall=best[2], TP=best[3], TN=best[4], FP=best[5], FN=best[6], threshold=best_t, latency=best_l)
