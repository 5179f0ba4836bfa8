"""PCK as the reference evaluates it (mmpose 0.29 `keypoint_pck_accuracy`, restated in SURVEY
Appendix F; call site EdgeCape/datasets/datasets/mp100/test_base_dataset.py:100-133)."""
import numpy as np

PCK_THRESHOLDS = (0.05, 0.1, 0.15, 0.2, 0.25)


def pck_counts(pred, gt, mask, normalize, thresholds=PCK_THRESHOLDS):
    """Per-pair PCK sums for a shard. pred/gt [N,K,2] px, mask [N,K] bool, normalize [N,2].
    Returns float64 [len(thr) + 1]: sum over pairs of per-pair PCK at each threshold, then N —
    a fixed-size payload that ranks can all-gather/sum."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    N = pred.shape[0]
    out = np.zeros(len(thresholds) + 1, np.float64)
    for n in range(N):
        nrm = np.asarray(normalize[n], np.float64).copy()
        nrm[nrm <= 0] = 1e6
        dist = np.linalg.norm((pred[n] - gt[n]) / nrm[None], axis=-1)
        valid = np.asarray(mask[n], bool)
        for i, thr in enumerate(thresholds):
            out[i] += float((dist[valid] < thr).mean()) if valid.any() else 0.0
    out[-1] = N
    return out


def pck_from_counts(counts, thresholds=PCK_THRESHOLDS):
    n = max(counts[-1], 1.0)
    res = {f"PCK@{t}": counts[i] / n for i, t in enumerate(thresholds)}
    res["mPCK"] = float(np.mean([res[f"PCK@{t}"] for t in thresholds]))
    return res
