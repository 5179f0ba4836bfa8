"""Evaluation / report format of the reference.

* `keypoint_pck_accuracy`, `keypoint_auc`, `keypoint_nme`, `keypoint_epe`: mmpose 0.29 `core/evaluation/top_down_eval.py`
  (third-party, absent from the reference tree => restated from the published source, "parity unpinned");
  call sites EdgeCape/datasets/datasets/mp100/test_base_dataset.py:126,140,147,154.
* `report_metric`: TestBaseDataset._report_metric (test_base_dataset.py:71-155) on arrays instead of the COCO db.
* `evaluate`: TestPoseDataset.evaluate (test_dataset.py:254-319): `result_keypoints.json` records + metrics.
* `append_testing_log`: test.py:152-161.
* `pck_counts` / `pck_from_counts`: the fixed-size per-rank payload of the multi-GPU job.
"""
import json
import os
from collections import OrderedDict

import numpy as np

PCK_THRESHOLDS = (0.05, 0.1, 0.15, 0.2, 0.25)   # TestBaseDataset.PCK_threshold_list


def _calc_distances(preds, targets, mask, normalize):
    """[N,K,2] x2, mask [N,K] bool, normalize [N,2] -> distances [K,N] (-1 where masked out)."""
    N, K, _ = preds.shape
    _mask = mask.copy()
    _mask[np.where((normalize == 0).sum(1))[0], :] = False
    distances = np.full((N, K), -1, dtype=np.float32)
    normalize = normalize.copy()
    normalize[np.where(normalize <= 0)] = 1e6
    distances[_mask] = np.linalg.norm(((preds - targets) / normalize[:, None, :])[_mask], axis=-1)
    return distances.T


def _distance_acc(distances, thr=0.5):
    distance_valid = distances != -1
    num_distance_valid = distance_valid.sum()
    if num_distance_valid > 0:
        return (distances[distance_valid] < thr).sum() / num_distance_valid
    return -1


def keypoint_pck_accuracy(pred, gt, mask, thr, normalize):
    distances = _calc_distances(pred, gt, mask, normalize)
    acc = np.array([_distance_acc(d, thr) for d in distances])
    valid_acc = acc[acc >= 0]
    cnt = len(valid_acc)
    avg_acc = valid_acc.mean() if cnt > 0 else 0
    return acc, avg_acc, cnt


def keypoint_auc(pred, gt, mask, normalize, num_step=20):
    nor = np.tile(np.array([[normalize, normalize]]), (pred.shape[0], 1))
    x = [1.0 * i / num_step for i in range(num_step)]
    y = [keypoint_pck_accuracy(pred, gt, mask, thr, nor)[1] for thr in x]
    return sum(1.0 / num_step * yi for yi in y)


def keypoint_nme(pred, gt, mask, normalize_factor):
    distances = _calc_distances(pred, gt, mask, normalize_factor)
    distance_valid = distances[distances != -1]
    return distance_valid.sum() / max(1, len(distance_valid))


def keypoint_epe(pred, gt, mask):
    distances = _calc_distances(pred, gt, mask, np.ones((pred.shape[0], pred.shape[2]), dtype=np.float32))
    distance_valid = distances[distances != -1]
    return distance_valid.sum() / max(1, len(distance_valid))


def report_metric(preds, gts, masks, bbox_thr, metrics=("PCK",), thresholds=PCK_THRESHOLDS):
    """preds/gts: sequences of [K,2] pixel arrays, masks: [K] bool (query visible AND every support visible),
    bbox_thr: per pair max(bbox_w, bbox_h).  Returns the reference's info_str list of [name, value]."""
    info_str = []
    one = lambda a: np.expand_dims(np.asarray(a), 0)
    thr2 = [np.array([t, t], np.float64) for t in bbox_thr]
    if "PCK" in metrics:
        res = {t: [] for t in thresholds}
        for o, g, m, tb in zip(preds, gts, masks, thr2):
            for t in thresholds:
                res[t].append(keypoint_pck_accuracy(one(o), one(g), one(m), t, one(tb))[1])
        mpck = 0
        for t in thresholds:
            info_str.append(["PCK@" + str(t), np.mean(res[t])])
            mpck += np.mean(res[t])
        info_str.append(["mPCK", mpck / len(thresholds)])
    if "NME" in metrics:
        info_str.append(["NME", np.mean([keypoint_nme(one(o), one(g), one(m), one(tb)) for o, g, m, tb in zip(preds, gts, masks, thr2)])])
    if "AUC" in metrics:
        info_str.append(["AUC", np.mean([keypoint_auc(one(o), one(g), one(m), tb[0]) for o, g, m, tb in zip(preds, gts, masks, thr2)])])
    if "EPE" in metrics:
        info_str.append(["EPE", np.mean([keypoint_epe(one(o), one(g), one(m)) for o, g, m in zip(preds, gts, masks)])])
    return info_str


def sort_and_unique_bboxes(kpts, key="bbox_id"):
    kpts = sorted(kpts, key=lambda x: x[key])
    for i in range(len(kpts) - 1, 0, -1):
        if kpts[i][key] == kpts[i - 1][key]:
            del kpts[i]
    return kpts


def gt_from_db(joints_3d, joints_3d_visible, bbox, paired_samples):
    """The per-pair ground truth of TestBaseDataset._report_metric (test_base_dataset.py:96-113) from database arrays:
    joints_3d / joints_3d_visible [n_obj, K, 3], bbox [n_obj, 4] (x, y, w, h), paired_samples [n_pairs, shots + 1] (support ids, then
    the query id).  Pair p is scored on the query's joints where the query AND every support annotation mark the keypoint
    visible, normalised by the longer side of the query's box.  Returns {p: dict(joints, mask, bbox_thr)} for `evaluate`."""
    J, V, B = np.asarray(joints_3d), np.asarray(joints_3d_visible), np.asarray(bbox)
    gt = {}
    for p, pair in enumerate(np.asarray(paired_samples)):
        q = int(pair[-1])
        mask = V[q][:, 0] > 0
        for s in pair[:-1]:
            mask &= V[int(s)][:, 0] > 0
        gt[p] = dict(joints=J[q][:, :-1], mask=mask, bbox_thr=float(np.max(B[q][2:])))
    return gt


def evaluate(outputs, gt, res_folder, metric="PCK", image_id_of=None):
    """TestPoseDataset.evaluate: `outputs` = list of per-sample dicts as returned by `single_gpu_test`
    (preds [1,K,3], boxes [1,6], image_paths, bbox_ids); `gt` maps bbox_id -> dict(joints [K,2], mask [K], bbox_thr).
    Writes `${res_folder}/result_keypoints.json` (same record keys as the reference) and returns an OrderedDict."""
    metrics = metric if isinstance(metric, (list, tuple)) else [metric]
    for m in metrics:
        if m not in ("PCK", "AUC", "EPE", "NME"):
            raise KeyError(f"metric {m} is not supported")
    kpts = []
    for output in outputs:
        preds, boxes = output["preds"], output["boxes"]
        for i in range(len(output["image_paths"])):
            path = output["image_paths"][i]
            kpts.append({"keypoints": np.asarray(preds[i]).tolist(), "center": np.asarray(boxes[i][0:2]).tolist(),
                         "scale": np.asarray(boxes[i][2:4]).tolist(), "area": float(boxes[i][4]), "score": float(boxes[i][5]),
                         "image_id": image_id_of(path) if image_id_of else path, "bbox_id": int(output["bbox_ids"][i])})
    kpts = sort_and_unique_bboxes(kpts)
    os.makedirs(res_folder, exist_ok=True)
    res_file = os.path.join(res_folder, "result_keypoints.json")
    with open(res_file, "w") as f:
        json.dump(kpts, f, sort_keys=True, indent=4)
    with open(res_file) as f:
        loaded = json.load(f)
    assert len(loaded) == len(gt), "every pair must be predicted exactly once (test_base_dataset.py:96)"
    P = [np.array(k["keypoints"])[:, :-1] for k in loaded]
    G = [np.asarray(gt[k["bbox_id"]]["joints"], np.float64) for k in loaded]
    M = [np.asarray(gt[k["bbox_id"]]["mask"], bool) for k in loaded]
    T = [float(gt[k["bbox_id"]]["bbox_thr"]) for k in loaded]
    return OrderedDict(report_metric(P, G, M, T, metrics))


def append_testing_log(log_dir, config_file, checkpoint, results, log_file="testing_log.txt"):
    with open(os.path.join(log_dir, log_file), "a") as f:
        f.write("**  config_file: " + config_file + "\t checkpoint: " + checkpoint + "\t \n")
        for k, v in sorted(results.items()):
            f.write(f"\t {k}: {v}" + "\n")
        f.write("********************************************************************\n")


# ---- fixed-size per-rank payload (multi-GPU) -------------------------------------------------------
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
