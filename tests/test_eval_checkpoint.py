"""CPU tests of the caller-side rows next to the hot path: checkpoint key handling and the
evaluation / report format.  Known answers are computed by hand from the definitions cited in the modules."""
import json
import os

import numpy as np
import pytest
import torch

from edgecape_amd import checkpoint, evaluation, synth
from edgecape_amd.engine import normalize_state_dict


def test_pck_known_answers():
    gt = np.array([[[10., 10.], [20., 20.], [30., 30.], [40., 40.]]])
    pred = gt + np.array([[[3., 4.], [0., 0.], [30., 40.], [6., 8.]]])        # distances 5, 0, 50, 10 px
    mask = np.array([[True, True, True, False]])
    norm = np.array([[100., 100.]])                                           # bbox side 100 -> 0.05, 0, 0.5
    acc, avg, cnt = evaluation.keypoint_pck_accuracy(pred, gt, mask, 0.2, norm)
    assert cnt == 4 or cnt == 3
    assert avg == pytest.approx(2 / 3)                                        # 0.05 and 0 < 0.2; 0.5 is not
    assert evaluation.keypoint_pck_accuracy(pred, gt, mask, 0.05, norm)[1] == pytest.approx(1 / 3)   # strict '<'
    assert evaluation.keypoint_epe(pred, gt, mask) == pytest.approx((5 + 0 + 50) / 3)
    assert evaluation.keypoint_nme(pred, gt, mask, norm) == pytest.approx((0.05 + 0 + 0.5) / 3)
    # AUC: mean over thr = i/20, i = 0..19 of PCK(thr): kp1 (0.05) counts for thr > 0.05 (18 steps), kp2 (0) for thr > 0
    # (19 steps), kp3 (0.5) for thr > 0.5 (9 steps: 0.55..0.95)
    assert evaluation.keypoint_auc(pred, gt, mask, 100.0) == pytest.approx((18 + 19 + 9) / 3 / 20)
    # all keypoints masked out -> PCK 0 (avg over no valid keypoints)
    assert evaluation.keypoint_pck_accuracy(pred, gt, np.zeros((1, 4), bool), 0.2, norm)[1] == 0


def test_report_and_counts_agree(tmp_path):
    rng = np.random.default_rng(3)
    N, K = 9, 12
    gts = rng.uniform(0, 200, (N, K, 2))
    preds = gts + rng.normal(0, 12, (N, K, 2))
    masks = rng.random((N, K)) > 0.3
    thr = rng.uniform(80, 160, N)
    info = dict(evaluation.report_metric(list(preds), list(gts), list(masks), list(thr), ["PCK", "NME", "AUC", "EPE"]))
    counts = evaluation.pck_counts(preds, gts, masks, np.stack([thr, thr], 1))
    from_counts = evaluation.pck_from_counts(counts)
    for t in evaluation.PCK_THRESHOLDS:
        assert info[f"PCK@{t}"] == pytest.approx(from_counts[f"PCK@{t}"], abs=1e-7)   # the all-reduced payload gives the same PCK
    assert info["mPCK"] == pytest.approx(from_counts["mPCK"], abs=1e-7)
    assert 0 <= info["AUC"] <= 1 and info["EPE"] > 0 and info["NME"] > 0

    # evaluate(): json records, sorting + de-duplication by bbox_id, testing log
    outputs = []
    order = [3, 1, 0, 2, 2, 4, 5, 6, 7, 8]                                   # out of order, one duplicate (DistributedSampler padding)
    for i in order:
        p3 = np.concatenate([preds[i], np.ones((K, 1))], 1)[None].astype(np.float32)
        outputs.append(dict(preds=p3, boxes=np.array([[1, 2, 0.5, 0.5, 10000, 1]], np.float32), image_paths=[f"img{i}.jpg"], bbox_ids=[i]))
    gt = {i: dict(joints=gts[i], mask=masks[i], bbox_thr=thr[i]) for i in range(N)}
    res = evaluation.evaluate(outputs, gt, str(tmp_path), metric=["PCK", "EPE"])
    recs = json.load(open(tmp_path / "result_keypoints.json"))
    assert [r["bbox_id"] for r in recs] == list(range(N))
    assert set(recs[0]) == {"keypoints", "center", "scale", "area", "score", "image_id", "bbox_id"}
    assert res["PCK@0.2"] == pytest.approx(info["PCK@0.2"], abs=1e-6)
    evaluation.append_testing_log(str(tmp_path), "configs/test/1shot_split1.py", "ckpt.pth", res)
    log = open(tmp_path / "testing_log.txt").read()
    assert "config_file: configs/test/1shot_split1.py" in log and "mPCK" in log
    with pytest.raises(KeyError):
        evaluation.evaluate(outputs, gt, str(tmp_path), metric="mAP")


def test_checkpoint_key_handling(tmp_path):
    arch = "dinov2_vits14"
    sd = synth.make_weights(arch, seed=2)
    # a stage-2 style mmcv checkpoint: wrapped in 'state_dict', backbone stored twice (encoder_sample is encoder_query,
    # EdgeCape.py:36), decoder self-attention with FUSED in_proj (bias_attn.py:236-265)
    ck = {}
    for k, v in sd.items():
        ck[k] = torch.from_numpy(np.asarray(v))
        if k.startswith("encoder_query."):
            ck["encoder_sample." + k[len("encoder_query."):]] = ck[k]
    for l in range(3):
        p = f"keypoint_head_module.transformer.decoder.layers.{l}.self_attn."
        w = torch.cat([ck.pop(p + f"{n}_proj.weight") for n in "qkv"], 0)
        b = torch.cat([ck.pop(p + f"{n}_proj.bias") for n in "qkv"], 0)
        ck[p + "in_proj_weight"], ck[p + "in_proj_bias"] = w, b
    path = str(tmp_path / "epoch_x.pth")
    torch.save({"meta": {"epoch": 1}, "state_dict": ck}, path)

    class Sink:
        def load_state_dict(self, sd, strict=True):
            self.sd = normalize_state_dict(sd)

    sink = Sink()
    checkpoint.load_checkpoint(sink, path)
    got = sink.sd
    assert not any(k.startswith("encoder_sample.") for k in got)
    assert set(got) == set(sd)
    for k in sd:
        assert np.array_equal(np.asarray(got[k].numpy() if isinstance(got[k], torch.Tensor) else got[k]), np.asarray(sd[k])), k

    # hub backbone import + pack round trip
    hub = {k[len("encoder_query."):]: v for k, v in sd.items() if k.startswith("encoder_query.")}
    head_only = {k: v for k, v in sd.items() if k.startswith("keypoint_head_module.")}
    merged = checkpoint.merge_state_dicts(hub, {"state_dict": head_only})
    assert set(merged) == set(sd)
    with pytest.raises(KeyError):
        checkpoint.import_dinov2_hub_state_dict({"head.weight": np.zeros(3)})
    pack = str(tmp_path / "model.safetensors")
    names = checkpoint.export_pack({"state_dict": ck}, pack)
    back = checkpoint.load_pack(pack)
    assert names == sorted(sd) and all(np.array_equal(back[k], np.asarray(sd[k], np.float32)) for k in sd)
