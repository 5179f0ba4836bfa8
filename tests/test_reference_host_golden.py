"""CPU: host-side pieces of the path against fixtures produced by the REAL reference (oracle/make_golden.py --round2, which
imports EdgeCape/datasets/... and EdgeCape/models/utils/post_processing/... from /root/reference with the documented stand-ins):
crop-warp geometry, episode pairing, evaluate / result_keypoints.json / _report_metric plumbing."""
import json
import os

import numpy as np

from conftest import load_golden
from edgecape_amd import episodes, evaluation, preprocess


def test_affine_transform_vs_reference():
    """preprocess.get_affine_transform vs post_transforms.py:197-252 run in the build container (three float32 points + the float64
    3-point solve standing in for cv2.getAffineTransform): the same arithmetic, so the matrices agree to the rounding of the solve."""
    g, meta = load_golden("pre_geometry")
    n = len(g["rot"])
    for i in range(n):
        for inv, key in ((False, "fwd"), (True, "inv")):
            M = preprocess.get_affine_transform(g["center"][i], g["scale"][i], g["rot"][i], g["out_size"][i], shift=tuple(g["shift"][i]), inv=inv)
            ref = g[key][i]
            assert np.abs(M - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (i, key, np.abs(M - ref).max())
        w = preprocess.warp_points(g["pts"][i], g["fwd"][i])
        assert np.abs(w - g["warped"][i]).max() < 1e-9                      # same matrix -> same points (vectorised affine_transform)
        w2 = preprocess.warp_points(g["pts"][i], preprocess.get_affine_transform(g["center"][i], g["scale"][i], g["rot"][i], g["out_size"][i],
                                                                                 shift=tuple(g["shift"][i])))
        assert np.abs(w2 - g["warped"][i]).max() < 1e-8                     # pixels: a joint lands in the reference's heatmap cell
    # forward and inverse are inverses of each other
    M = preprocess.get_affine_transform([100., 80.], [1.5, 1.5], 25., (256, 256))
    Mi = preprocess.get_affine_transform([100., 80.], [1.5, 1.5], 25., (256, 256), inv=True)
    p = np.array([[3., 4.], [250., 17.]])
    assert np.abs(preprocess.warp_points(preprocess.warp_points(p, M), Mi) - p).max() < 1e-3   # (each from its own fp32 triangle)


def test_episode_pairs_vs_reference():
    """episodes.make_paired_samples vs TestPoseDataset.make_paired_samples (test_dataset.py:86-99): identical pair list."""
    g, meta = load_golden("eval_dataset")
    cat2obj = {int(k): v for k, v in meta["cat2obj"].items()}
    pairs = episodes.make_paired_samples(cat2obj, meta["valid_class_ids"], meta["num_shots"], meta["num_queries"], meta["num_episodes"])
    assert np.array_equal(pairs, g["pairs"])
    sup, ep = episodes.group_episodes(pairs)
    assert len(sup) == len(meta["valid_class_ids"]) * meta["num_episodes"] and ep.max() == len(sup) - 1
    assert np.array_equal(sup[ep], pairs[:, :-1]) and np.all(np.bincount(ep) == meta["num_queries"])


def test_evaluate_vs_reference(tmp_path):
    """evaluation.evaluate vs TestPoseDataset.evaluate -> _report_metric: same result_keypoints.json (byte for byte) and the same
    metric values.  (The mmpose metric functions inside the reference run were this package's restatement - see the fixture's
    meta: the plumbing is pinned, the mmpose arithmetic is covered by the known-answer tests in test_eval_checkpoint.py.)"""
    g, meta = load_golden("eval_dataset")
    pairs, preds, boxes = g["pairs"], g["preds"], g["boxes"]
    files = meta["image_files"]
    outputs = []
    for s in range(0, len(pairs), 3):
        idx = list(range(s, min(s + 3, len(pairs))))
        outputs.append(dict(preds=preds[idx], boxes=boxes[idx], image_paths=[files[pairs[i][-1]] for i in idx], bbox_ids=idx))
    outputs.append(dict(preds=preds[:1], boxes=boxes[:1], image_paths=[files[pairs[0][-1]]], bbox_ids=[0]))   # sampler padding duplicate
    name2id = {f[len(meta["img_prefix"]):]: meta["image_id_offset"] + i for i, f in enumerate(files)}
    gt = evaluation.gt_from_db(g["joints_3d"], g["joints_3d_visible"], g["bbox"], pairs)
    nv = evaluation.evaluate(outputs, gt, str(tmp_path), metric=["PCK", "AUC", "EPE", "NME"],
                             image_id_of=lambda p: name2id[p[len(meta["img_prefix"]):]])
    assert list(nv.keys()) == meta["metric_names"]
    assert np.allclose(np.array(list(nv.values()), np.float64), g["metric_values"], rtol=0, atol=1e-12)
    ours = open(os.path.join(str(tmp_path), "result_keypoints.json")).read()
    assert ours == bytes(g["result_json"]).decode()
    assert len(json.loads(ours)) == len(pairs)


def test_stream_schedule_covers_the_pair_order():
    """episodes.stream_schedule cuts the reference's pair order (every support set followed by its queries, test_dataset.py:86-99) into
    the calls of ec_forward_episodes: every pair exactly once and in order, an episode is encoded by the call that holds its first
    query, its slot is not handed to another episode before the call after its last query, and too small a cache is an error."""
    import pytest
    from edgecape_amd.episodes import stream_schedule
    rng = np.random.default_rng(3)
    for n_ep, batch, cap in ((32, 60, 6), (7, 4, 6), (5, 100, 5), (9, 15, 17), (4, 1, 2)):
        sizes = rng.integers(1, 16, n_ep) if batch != 60 else np.full(n_ep, 15)
        ep = np.repeat(np.arange(n_ep), sizes)
        calls = stream_schedule(ep, batch, cap)
        assert np.array_equal(np.concatenate([c["queries"] for c in calls]), np.arange(len(ep)))
        owner, encoded = {}, set()
        for c in calls:
            for e, s in zip(c["new_episodes"], c["new_slots"]):
                assert e not in encoded and 0 <= s < cap
                assert s not in owner or ep.tolist().index(e) > max(np.nonzero(ep == owner[s])[0])   # the old episode's queries are all behind us
                owner[int(s)] = e
                encoded.add(e)
            assert all(owner[int(s)] == e for s, e in zip(c["slot_of_query"], ep[c["queries"]]))
            assert sorted(set(c["new_episodes"])) == sorted(e for e in set(ep[c["queries"]].tolist()) if min(np.nonzero(ep == e)[0]) >= c["queries"][0])
        assert encoded == set(range(n_ep))
    with pytest.raises(ValueError):
        stream_schedule(np.repeat(np.arange(6), 2), 12, 3)          # six episodes alive in one call, three slots
    with pytest.raises(ValueError):
        stream_schedule(np.array([0, 1, 0]), 2, 4)                  # not the reference's order


def test_detector_support_key_tells_support_sets_apart():
    """EdgeCape.enable_episode_cache recognises a support set by its annotations in img_metas: same files + crops + keypoints = same
    set; another annotation in the same image file, another skeleton or another shot order is a different one."""
    from edgecape_amd.detector import EdgeCape
    base = dict(sample_image_file=["a.jpg", "b.jpg"], sample_center=[np.array([10., 20.]), np.array([5., 5.])],
                sample_scale=[np.array([1., 1.]), np.array([2., 2.])], sample_skeleton=[[[0, 1], [1, 2]]] * 2)
    k = EdgeCape._support_key
    assert k(base) == k(dict(base, query_image_file="q.jpg"))
    assert k(base) != k(dict(base, sample_center=[np.array([11., 20.]), np.array([5., 5.])]))
    assert k(base) != k(dict(base, sample_skeleton=[[[0, 1]]] * 2))
    assert k(base) != k(dict(base, sample_image_file=["b.jpg", "a.jpg"]))
    # visibility and rotation are part of the identity (they determine mask_s / target_s); tensors are accepted (demo.py hands
    # CUDA tensors over; here a CPU tensor); metas with nothing but file names give NO key - the detector then takes the plain path
    import torch
    vis = [np.ones((3, 3), np.float32), np.ones((3, 3), np.float32)]
    assert k(dict(base, sample_joints_3d_visible=vis)) != k(dict(base, sample_joints_3d_visible=[vis[0], vis[1] * 0]))
    assert k(dict(base, sample_rotation=[0, 0])) != k(dict(base, sample_rotation=[0, 30]))
    assert k(dict(base, sample_joints_3d=[torch.ones(3, 3), torch.zeros(3, 3)])) == k(dict(base, sample_joints_3d=[np.ones((3, 3), np.float32), np.zeros((3, 3), np.float32)]))
    assert k(dict(sample_image_file=["a.jpg"], sample_skeleton=[[]])) is None
    assert k(dict(sample_image_file=[""], sample_skeleton=[[]], sample_rotation=[0])) is None
