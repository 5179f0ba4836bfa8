"""TEST INFRASTRUCTURE (container-only): generate tests/golden/*.npz from the REAL reference.

Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

What it does:
  * imports the reference's own `TwoStageHead` / `SkeletonPredictor` /
    `TwoStageSupportRefineTransformer` / `EdgeCape` detector from /root/reference through
    `oracle/ref_stubs.py`, builds them from `configs/test/1shot_split1.py` unchanged (plus the
    documented in_channels / dim_feedforward overrides for ViT-B);
  * loads the seeded weights of `edgecape_amd.synth` by reference key names;
  * runs the reference fp32 CPU forward on seeded inputs and stores OUTPUTS ONLY (inputs and weights
    are regenerated from the seeds recorded in each fixture's `meta` json);
  * for the third-party backbone (not importable: torch.hub) stores outputs of HF
    `transformers` Dinov2Model with the same weights as an independent cross-check.

Fixtures are data (arrays), never reference source text.
"""
import copy
import json
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

from edgecape_amd import synth
from oracle import edgecape_oracle as orc
from oracle import ref_stubs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

HEAD_CASES = [
    # name, C, g, shots, n_kps, skeletons, seed
    ("head_s1_c384_g16_kp17", 384, 16, 1, [17, 17], "auto", 101),
    ("head_s5_c384_g16_mixed", 384, 16, 5, [100, 1], "auto", 102),
    ("head_s1_c768_g18_edge", 768, 18, 1, [0, 17], "empty", 103),
    ("head_s5_c768_g18_kp17", 768, 18, 5, [17, 30], "auto", 104),
]


def build_ref_head(ns, C):
    cfg = ref_stubs.load_reference_config()
    hc = copy.deepcopy(cfg["model"]["keypoint_head"])
    hc.pop("type")
    hc["in_channels"] = C
    hc["skeleton_head"]["dim_feedforward"] = C  # image_project = Conv2d(dim_feedforward, d_model)
    head = ns.HEADS.get("TwoStageHead")(**hc)
    head.eval()
    return head


def load_head_weights(head, sd, prefix="keypoint_head_module."):
    sub = {k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = head.load_state_dict(sub, strict=True)
    assert not missing and not unexpected


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: np.ascontiguousarray(v, dtype=np.float32) if v.dtype.kind == "f" else v for k, v in arrays.items()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KB")


def head_fixture(ns, name, C, g, shots, n_kps, skeletons, seed):
    wseed = 7
    sd = synth.make_head_weights(C=C, seed=wseed)
    head = build_ref_head(ns, C)
    load_head_weights(head, sd)
    inp = synth.make_head_inputs(len(n_kps), shots, C, g, seed, n_kps, skeletons)
    taps = {}
    hooks = []
    enc = head.transformer.encoder
    hooks.append(enc.register_forward_hook(lambda m, i, o: taps.__setitem__("enc", o)))
    hooks.append(head.query_proj.register_forward_hook(lambda m, i, o: taps.__setitem__("support_keypoints", o)))
    sk = head.skeleton_head
    hooks.append(sk.register_forward_hook(lambda m, i, o: taps.__setitem__("skel", o)))
    dec = head.transformer.decoder
    hooks.append(dec.register_forward_hook(lambda m, i, o: taps.__setitem__("dec", o)))
    with torch.no_grad():
        out, init_prop, sim, _, adj = head(torch.from_numpy(inp["feature_q"]),
                                           [torch.from_numpy(f) for f in inp["feature_s"]],
                                           [torch.from_numpy(t) for t in inp["target_s"]],
                                           torch.from_numpy(inp["mask_s"]), inp["skeleton"])
    for h in hooks:
        h.remove()
    enc_img, enc_kp = taps["enc"]
    _, attn_adj, unnorm = taps["skel"]
    hs, points = taps["dec"][0], taps["dec"][1]
    arrays = dict(
        output_kpts=out.numpy(), initial_proposals=init_prop.numpy(), similarity_map=sim.numpy(),
        adj=adj.numpy(), attn_adj=attn_adj.numpy(), unnormalized_adj=unnorm.numpy(),
        support_keypoints=taps["support_keypoints"].numpy(),
        enc_kp=enc_kp.numpy(), enc_img_first8=enc_img[:8].numpy(), enc_img_last8=enc_img[-8:].numpy(),
        out_points=torch.stack(points).numpy(), hs_last=hs[-1].numpy(),
    )
    meta = dict(kind="head", C=C, g=g, shots=shots, n_kps=n_kps, skeletons=skeletons, input_seed=seed,
                weight_seed=wseed, generator="oracle/make_golden.py", reference="orhir/EdgeCape TwoStageHead.forward")
    save(name, arrays, meta)


class _OracleBackbone(torch.nn.Module):
    """Stand-in for torch.hub DINOv2 inside the reference detector (EdgeCape.py:35-36)."""

    def __init__(self, sd, heads):
        super().__init__()
        self.sd, self.heads = sd, heads

    def get_intermediate_layers(self, x, n=1, reshape=True):
        return [orc.dinov2_features(self.sd, x, self.heads)]


def detector_fixture(name, arch, image_size, shots, seed):
    wseed = 11
    sd = synth.make_weights(arch, seed=wseed)
    a = synth.ARCHS[arch]
    ns = ref_stubs.import_detector(lambda nm: _OracleBackbone(sd, a["heads"]))
    cfg = ref_stubs.load_reference_config()
    mc = copy.deepcopy(cfg["model"])
    mc.pop("type")
    mc["keypoint_head"]["in_channels"] = a["C"]
    mc["keypoint_head"]["skeleton_head"]["dim_feedforward"] = a["C"]
    mc["pretrained"] = arch
    model = ns.POSENETS.get("EdgeCape")(**mc)
    model.eval()
    load_head_weights(model.keypoint_head_module, sd)
    batch = synth.make_pairs(2, shots, image_size, seed=seed)
    with torch.no_grad():
        res = model(img_s=[torch.from_numpy(x) for x in batch["img_s"]], img_q=torch.from_numpy(batch["img_q"]),
                    target_s=[torch.from_numpy(x) for x in batch["target_s"]],
                    target_weight_s=[torch.from_numpy(x) for x in batch["target_weight_s"]],
                    target_q=torch.from_numpy(batch["target_q"]), target_weight_q=torch.from_numpy(batch["target_weight_q"]),
                    img_metas=batch["img_metas"], return_loss=False)
    arrays = dict(preds=res["preds"], boxes=res["boxes"], points=res["points"], skeleton=res["skeleton"],
                  bbox_ids=np.array(res["bbox_ids"], np.int64))
    meta = dict(kind="detector", arch=arch, image_size=list(image_size) if isinstance(image_size, tuple) else image_size, shots=shots, input_seed=seed, weight_seed=wseed,
                bs=2, reference="orhir/EdgeCape EdgeCape.forward_test with the oracle DINOv2 as torch.hub stand-in")
    save(name, arrays, meta)


def hf_backbone_fixture(name, arch, image_size, seed):
    """HF transformers Dinov2Model with identical weights; native grid == g so no interpolation on
    the HF side; the table fed to both sides is the upstream-convention one.
    image_size (H, W), round 4: a NON-SQUARE input.  This transformers version interpolates with size=(h, w) - upstream's
    interpolate_offset = 0 convention - while the torch.hub models the reference loads use the offset-0.1 scale factors (SURVEY
    App. C), so HF's interpolate_pos_encoding is replaced by the upstream-convention table here as in the square fixtures; the per-axis
    interpolation itself is pinned against torch's upsample_bicubic2d in tests/test_oracle_golden.py."""
    from transformers import Dinov2Config, Dinov2Model
    wseed = 13
    a = synth.ARCHS[arch]
    C, depth, heads = a["C"], a["depth"], a["heads"]
    sd = synth.make_backbone_weights(arch, seed=wseed, prefix="")
    nonsquare = isinstance(image_size, (tuple, list))
    if nonsquare:
        pos = orc.interpolate_pos_embed(sd["pos_embed"], (image_size[0] // 14, image_size[1] // 14))
        hf_size = [image_size[0] // 14 * 14, image_size[1] // 14 * 14]
    else:
        g = image_size // 14
        pos = orc.interpolate_pos_embed(sd["pos_embed"], g)
        hf_size = g * 14
    cfg = Dinov2Config(hidden_size=C, num_hidden_layers=depth, num_attention_heads=heads, image_size=hf_size,
                       patch_size=14, mlp_ratio=4, qkv_bias=True, layer_norm_eps=1e-6, layerscale_value=1.0,
                       hidden_act="gelu", use_swiglu_ffn=False, attn_implementation="eager")
    m = Dinov2Model(cfg).eval()
    t = lambda k: torch.from_numpy(sd[k])
    hsd = {}
    hsd["embeddings.cls_token"] = t("cls_token")
    hsd["embeddings.mask_token"] = torch.zeros(1, C)
    hsd["embeddings.position_embeddings"] = pos[None]
    hsd["embeddings.patch_embeddings.projection.weight"] = t("patch_embed.proj.weight")
    hsd["embeddings.patch_embeddings.projection.bias"] = t("patch_embed.proj.bias")
    for i in range(depth):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        hsd[q + "norm1.weight"], hsd[q + "norm1.bias"] = t(p + "norm1.weight"), t(p + "norm1.bias")
        Wq, bq = t(p + "attn.qkv.weight"), t(p + "attn.qkv.bias")
        for j, nm in enumerate(("query", "key", "value")):
            hsd[q + f"attention.attention.{nm}.weight"] = Wq[j * C:(j + 1) * C]
            hsd[q + f"attention.attention.{nm}.bias"] = bq[j * C:(j + 1) * C]
        hsd[q + "attention.output.dense.weight"], hsd[q + "attention.output.dense.bias"] = t(p + "attn.proj.weight"), t(p + "attn.proj.bias")
        hsd[q + "layer_scale1.lambda1"] = t(p + "ls1.gamma")
        hsd[q + "norm2.weight"], hsd[q + "norm2.bias"] = t(p + "norm2.weight"), t(p + "norm2.bias")
        hsd[q + "mlp.fc1.weight"], hsd[q + "mlp.fc1.bias"] = t(p + "mlp.fc1.weight"), t(p + "mlp.fc1.bias")
        hsd[q + "mlp.fc2.weight"], hsd[q + "mlp.fc2.bias"] = t(p + "mlp.fc2.weight"), t(p + "mlp.fc2.bias")
        hsd[q + "layer_scale2.lambda1"] = t(p + "ls2.gamma")
    hsd["layernorm.weight"], hsd["layernorm.bias"] = t("norm.weight"), t("norm.bias")
    missing, unexpected = m.load_state_dict(hsd, strict=False)
    assert not unexpected, unexpected
    assert all("mask_token" in k for k in missing), missing
    if nonsquare:
        m.embeddings.interpolate_pos_encoding = lambda emb, h, w: pos[None]
    rng = np.random.default_rng(seed)
    img = np.stack([synth._smooth_image(rng, *image_size) if nonsquare else synth._smooth_image(rng, image_size)])
    with torch.no_grad():
        o = m(pixel_values=torch.from_numpy(img), output_hidden_states=True)
    hs = o.hidden_states
    feat = o.last_hidden_state[0, 1:]  # final layernorm applied, cls dropped -> [HW, C]
    arrays = dict(tokens0_first4=hs[0][0, :4].numpy(), block0_first4=hs[1][0, :4].numpy(),
                  feat_tokens_first8=feat[:8].numpy(), feat_tokens_last8=feat[-8:].numpy(),
                  feat_mean=feat.mean(0).numpy(), feat_abs_mean=np.array([feat.abs().mean().item()], np.float32))
    meta = dict(kind="backbone_hf", arch=arch, image_size=image_size, input_seed=seed, weight_seed=wseed,
                reference="HF transformers Dinov2Model (independent cross-check; upstream dinov2 is not importable)",
                transformers=__import__("transformers").__version__)
    save(name, arrays, meta)


def fuse_self_attn_in_proj(sub):
    """q/k/v_proj of every decoder self-attention -> the fused in_proj_weight / in_proj_bias keys of a checkpoint written before
    the attention module was swapped (what BiasedMultiheadAttention._load_from_state_dict, bias_attn.py:236-265, accepts)."""
    out = dict(sub)
    for k in list(sub):
        if k.endswith("self_attn.q_proj.weight"):
            base = k[:-len("q_proj.weight")]
            for kind, fused in (("weight", "in_proj_weight"), ("bias", "in_proj_bias")):
                parts = [out.pop(base + f"{n}_proj.{kind}") for n in "qkv"]
                out[base + fused] = torch.cat(parts, 0)
    return out


def fused_checkpoint_fixture(ns, name, C, g, shots, n_kps, skeletons, seed):
    """the reference head loads a checkpoint with FUSED decoder in_proj keys through its own
    _load_from_state_dict; outputs of that head are the fixture (the product loads the same file through load_checkpoint)."""
    wseed = 7
    sd = synth.make_head_weights(C=C, seed=wseed)
    prefix = "keypoint_head_module."
    sub = {k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix)}
    fused = fuse_self_attn_in_proj(sub)
    n_fused = sum(k.endswith("in_proj_weight") and ".decoder." in k and "self_attn" in k for k in fused)
    assert n_fused >= 3 and not any(k.endswith("self_attn.q_proj.weight") for k in fused)
    head = build_ref_head(ns, C)
    missing, unexpected = head.load_state_dict(fused, strict=True)
    assert not missing and not unexpected
    head_plain = build_ref_head(ns, C)
    load_head_weights(head_plain, sd)
    inp = synth.make_head_inputs(len(n_kps), shots, C, g, seed, n_kps, skeletons)
    args = (torch.from_numpy(inp["feature_q"]), [torch.from_numpy(f) for f in inp["feature_s"]],
            [torch.from_numpy(t) for t in inp["target_s"]], torch.from_numpy(inp["mask_s"]), inp["skeleton"])
    with torch.no_grad():
        out, init_prop, sim, _, adj = head(*args)
        out_p = head_plain(*args)[0]
    arrays = dict(output_kpts=out.numpy(), initial_proposals=init_prop.numpy(), similarity_map=sim.numpy(), adj=adj.numpy())
    meta = dict(kind="head_fused_checkpoint", C=C, g=g, shots=shots, n_kps=n_kps, skeletons=skeletons, input_seed=seed, weight_seed=wseed,
                fused_layers=int(n_fused), max_abs_diff_vs_unfused_load=float((out - out_p).abs().max()),
                reference="orhir/EdgeCape TwoStageHead loaded from fused self_attn.in_proj_* keys via bias_attn.py:236-265")
    save(name, arrays, meta)


def _metric_fns():
    from edgecape_amd import evaluation as ev
    return dict(keypoint_pck_accuracy=ev.keypoint_pck_accuracy, keypoint_auc=ev.keypoint_auc, keypoint_nme=ev.keypoint_nme,
                keypoint_epe=ev.keypoint_epe)


def geometry_fixture(ns, name, seed=401):
    """SURVEY §8f rank 3 (host geometry): get_affine_transform / affine_transform of post_transforms.py:197-270 on seeded boxes."""
    rng = np.random.default_rng(seed)
    n = 48
    center = rng.uniform(20, 600, (n, 2)).astype(np.float32)
    scale = rng.uniform(0.25, 4.0, (n, 2)).astype(np.float32)
    scale[::2, 1] = scale[::2, 0]                       # the test pipeline's boxes are square; keep both kinds
    rot = np.where(np.arange(n) % 3 == 0, 0.0, rng.uniform(-80, 80, n))
    shift = np.where((np.arange(n) % 4 == 0)[:, None], rng.uniform(-0.15, 0.15, (n, 2)), 0.0)
    out_size = np.array([[256, 256], [224, 224], [384, 384], [64, 64]])[np.arange(n) % 4]
    fwd = np.stack([ns.post.get_affine_transform(center[i], scale[i], rot[i], out_size[i], shift=tuple(shift[i])) for i in range(n)])
    inv = np.stack([ns.post.get_affine_transform(center[i], scale[i], rot[i], out_size[i], shift=tuple(shift[i]), inv=True) for i in range(n)])
    pts = rng.uniform(0, 640, (n, 17, 2))
    warped = np.stack([[ns.post.affine_transform(pts[i, j], fwd[i]) for j in range(17)] for i in range(n)])
    arrays = dict(center=center, scale=scale, rot=rot.astype(np.float64), shift=shift.astype(np.float64), out_size=out_size.astype(np.int64),
                  fwd=fwd.astype(np.float64), inv=inv.astype(np.float64), pts=pts.astype(np.float64), warped=warped.astype(np.float64))
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    meta = dict(kind="geometry", seed=seed, cv2_getAffineTransform="float64 3-point solve stand-in (cv2 absent)",
                reference="EdgeCape/models/utils/post_processing/post_transforms.py:197-270 get_affine_transform / affine_transform")
    os.makedirs(OUT, exist_ok=True)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)      # (float64 kept: `save` would narrow to float32)
    print(name)


def msra_fixture(ns, name, seed=402):
    """Rows a33 / §8f rank 3: TopDownGenerateTargetFewShot._msra_generate_target (top_down_transform.py:113-199), sigma = 1,
    on seeded joints that include every border case of the 3-sigma patch."""
    gen = ns.pipe.TopDownGenerateTargetFewShot(sigma=1)
    rng = np.random.default_rng(seed)
    cases = []
    for image_size, hm in ((256, 64), (224, 64), (384, 64), (256, 32)):
        K = 40
        j = rng.uniform(0, image_size, (K, 3))
        j[:, 2] = 0
        border = np.array([[-30, 10], [-11.9, 50], [-2, -2], [0, 0], [1.9, 2.1], [image_size - 0.1, image_size - 0.1], [image_size + 9, 40],
                           [image_size + 14.1, 40], [image_size * 2, 5], [40, -13.9], [40, -14.1], [image_size / 2, image_size / 2],
                           [image_size / hm * 7.5, image_size / hm * 8.49999], [image_size / hm * 63.49, image_size / hm * 0.5]])
        j[:len(border), :2] = border
        v = (rng.uniform(0, 1, (K, 1)) > 0.25).astype(np.float32) * np.where(rng.uniform(0, 1, (K, 1)) > 0.5, 2.0, 1.0)
        v[:len(border)] = 1.0
        vis = np.concatenate([v, v, np.zeros_like(v)], 1).astype(np.float32)
        cfg = dict(image_size=np.array([image_size, image_size]), heatmap_size=np.array([hm, hm]), joint_weights=None,
                   use_different_joint_weights=False)
        target, weight = gen._msra_generate_target(cfg, j.astype(np.float32), vis, 1)
        cases.append((image_size, hm, j.astype(np.float32), vis, target, weight))
    arrays = {}
    for i, (isz, hm, j, vis, t, w) in enumerate(cases):
        arrays[f"joints_{i}"], arrays[f"visible_{i}"], arrays[f"target_{i}"], arrays[f"weight_{i}"] = j, vis, t, w
    meta = dict(kind="msra", seed=seed, cases=[[c[0], c[1]] for c in cases], sigma=1,
                reference="EdgeCape/datasets/pipelines/top_down_transform.py:113-199 _msra_generate_target (unbiased_encoding=False)")
    save(name, arrays, meta)


def dataset_fixture(ns, name, seed=403):
    """Rows a30-a32 / §8f ranks 1 and 4: the reference's episode pairing (test_dataset.py:86-99) and its evaluate ->
    result_keypoints.json -> _report_metric plumbing (test_dataset.py:254-319, test_base_dataset.py:71-155) on a synthetic db.
    The mmpose metric functions inside _report_metric are edgecape_amd.evaluation's restatement (mmpose is absent): this pins the
    plumbing - record assembly, sort/unique, masks, bbox normalisation, per-pair averaging, mPCK - not the mmpose arithmetic."""
    import tempfile
    TD = ns.test_dataset.TestPoseDataset
    rng = np.random.default_rng(seed)
    K = 100
    n_obj = 60
    db = []
    for i in range(n_obj):
        nk = int(rng.integers(3, 25))
        j3 = np.zeros((K, 3), np.float32)
        j3[:nk, :2] = rng.uniform(10, 500, (nk, 2))
        v3 = np.zeros((K, 3), np.float32)
        v3[:nk, :2] = (rng.uniform(0, 1, (nk, 1)) > 0.2).astype(np.float32)
        w, h = rng.uniform(40, 400, 2)
        db.append(dict(image_file=f"data/mp100/cat{i % 4}/img_{i:04d}.jpg", joints_3d=j3, joints_3d_visible=v3,
                       bbox=np.array([rng.uniform(0, 100), rng.uniform(0, 100), w, h], np.float32), bbox_id=i, head_size=None))
    ds = object.__new__(TD)
    ds.db = db
    ds.img_prefix = "data/mp100/"
    ds.name2id = {d["image_file"][len(ds.img_prefix):]: 1000 + i for i, d in enumerate(db)}
    ds.PCK_threshold_list = [0.05, 0.1, 0.15, 0.2, 0.25]
    ds.num_shots, ds.num_queries, ds.num_episodes = 2, 5, 3
    ds.valid_class_ids = [3, 7, 11, 12]
    ds.cat2obj = {c: [i for i in range(n_obj) if i % 4 == k] for k, c in enumerate(ds.valid_class_ids)}
    ds.make_paired_samples()
    pairs = np.array(ds.paired_samples)
    # synthetic model outputs per pair, batched in threes, with one duplicated record (sampler padding)
    preds = rng.uniform(0, 500, (len(pairs), K, 3)).astype(np.float32)
    for p, pair in enumerate(pairs):                  # make them plausible: query gt + noise on most keypoints
        gt = db[pair[-1]]["joints_3d"][:, :2]
        preds[p, :, :2] = gt + rng.normal(0, 12, (K, 2)).astype(np.float32)
    boxes = rng.uniform(0, 300, (len(pairs), 6)).astype(np.float32)
    outputs = []
    for s in range(0, len(pairs), 3):
        idx = list(range(s, min(s + 3, len(pairs))))
        outputs.append(dict(preds=preds[idx], boxes=boxes[idx], image_paths=[db[pairs[i][-1]]["image_file"] for i in idx],
                            bbox_ids=[int(i) for i in idx]))
    outputs.append(dict(preds=preds[:1], boxes=boxes[:1], image_paths=[db[pairs[0][-1]]["image_file"]], bbox_ids=[0]))
    with tempfile.TemporaryDirectory() as td:
        nv = ds.evaluate(outputs, td, metric=["PCK", "AUC", "EPE", "NME"])
        res_json = open(os.path.join(td, "result_keypoints.json")).read()
    arrays = dict(pairs=pairs.astype(np.int64), preds=preds, boxes=boxes,
                  joints_3d=np.stack([d["joints_3d"] for d in db]), joints_3d_visible=np.stack([d["joints_3d_visible"] for d in db]),
                  bbox=np.stack([d["bbox"] for d in db]), metric_values=np.array(list(nv.values()), np.float64),
                  result_json=np.frombuffer(res_json.encode(), np.uint8))
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    meta = dict(kind="dataset", seed=seed, metric_names=list(nv.keys()), num_shots=2, num_queries=5, num_episodes=3,
                valid_class_ids=ds.valid_class_ids, cat2obj={str(k): v for k, v in ds.cat2obj.items()},
                image_files=[d["image_file"] for d in db], img_prefix=ds.img_prefix, image_id_offset=1000,
                mmpose_metric_functions="edgecape_amd.evaluation restatement (mmpose absent): plumbing pinned, arithmetic not",
                reference="EdgeCape/datasets/datasets/mp100/test_dataset.py:86-99,254-319; test_base_dataset.py:71-155,218-226")
    os.makedirs(OUT, exist_ok=True)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name)


def round2_fixtures(ns):
    fused_checkpoint_fixture(ns, "head_s1_c384_g16_kp17_fusedckpt", 384, 16, 1, [17, 17], "auto", 101)
    dns = ref_stubs.install_datasets(_metric_fns())
    geometry_fixture(dns, "pre_geometry")
    msra_fixture(dns, "pre_msra")
    dataset_fixture(dns, "eval_dataset")


def round4_fixtures():
    """Non-square inputs (VERDICT r3 missing item 4; the reference takes any img.shape[-2:], EdgeCape.py:143): the REAL reference
    head on 14 x 20 / 21 x 16 feature maps (incl. the (w, h) reshape of the proposal generator's one-hot, encoder_decoder.py:93-97, which
    is only a spatial neighbourhood on square maps), and HF Dinov2Model on a 224 x 308 image."""
    hf_backbone_fixture("bb_hf_vits14_224x308", "dinov2_vits14", (224, 308), 304)
    ns = ref_stubs.install()
    head_fixture(ns, "head_s2_c384_g14x20_kp17", 384, (14, 20), 2, [17, 30], "auto", 105)
    head_fixture(ns, "head_s1_c768_g21x16_mixed", 768, (21, 16), 1, [60, 0], "auto", 106)
    # the reference DETECTOR on non-square images (forward_test -> decode with img_size = [width, height], head.py:324-387)
    detector_fixture("det_vits14_229x311_s2", "dinov2_vits14", (229, 311), 2, 203)


def variant_fixture(ns, name, C, g, shots, n_kps, skeletons, seed, learn_skeleton):
    """The reference head of an EARLIER training stage (run.py:44-88), built from configs/train/1shot_split1.py as run.py derives it:
    stage 1 = the file as it is (SkeletonPredictor(learn_skeleton=False): ground-truth adjacency, skeleton.py:70-74; decoder
    self-attention = nn.MultiheadAttention, no Markov bias, encoder_decoder.py:551-560), stage 2 = + learn_skeleton (run.py:67-72).
    Weights: the synthetic head weights with the decoder's q / k / v_proj fused into the in_proj keys nn.MultiheadAttention owns and
    the Markov-MLP keys (which that module does not have) dropped."""
    wseed = 7
    sd = synth.make_head_weights(C=C, seed=wseed)
    cfg = ref_stubs.load_reference_config("configs/train/1shot_split1.py")
    hc = copy.deepcopy(cfg["model"]["keypoint_head"])
    hc.pop("type")
    hc["in_channels"] = C
    hc["skeleton_head"]["dim_feedforward"] = C
    if learn_skeleton:
        hc["skeleton_head"]["learn_skeleton"] = True
        hc["learn_skeleton"] = True
        hc["masked_supervision"] = True
    head = ns.HEADS.get("TwoStageHead")(**hc)
    head.eval()
    prefix = "keypoint_head_module."
    sub = {k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix)}
    sub = {k: v for k, v in fuse_self_attn_in_proj(sub).items() if "markov_structural_mlp" not in k}
    missing, unexpected = head.load_state_dict(sub, strict=True)
    assert not missing and not unexpected
    inp = synth.make_head_inputs(len(n_kps), shots, C, g, seed, n_kps, skeletons)
    taps = {}
    hk = head.transformer.decoder.register_forward_hook(lambda m, i, o: taps.__setitem__("dec", o))
    with torch.no_grad():
        out, init_prop, sim, _, adj = head(torch.from_numpy(inp["feature_q"]), [torch.from_numpy(f) for f in inp["feature_s"]],
                                           [torch.from_numpy(t) for t in inp["target_s"]], torch.from_numpy(inp["mask_s"]), inp["skeleton"])
    hk.remove()
    arrays = dict(output_kpts=out.numpy(), initial_proposals=init_prop.numpy(), similarity_map=sim.numpy(), adj=adj.numpy(),
                  out_points=torch.stack(taps["dec"][1]).numpy())
    meta = dict(kind="head_variant", C=C, g=g, shots=shots, n_kps=n_kps, skeletons=skeletons, input_seed=seed, weight_seed=wseed,
                learn_skeleton=bool(learn_skeleton), attn_bias=False, config="configs/train/1shot_split1.py",
                reference="orhir/EdgeCape TwoStageHead.forward of training stage %d (run.py:44-88)" % (2 if learn_skeleton else 1))
    save(name, arrays, meta)


def round5_fixtures():
    """VERDICT r4 missing item 4: inference on the reference's other model variants."""
    ns = ref_stubs.install()
    variant_fixture(ns, "head_stage1_c384_g16", 384, 16, 1, [17, 40], "auto", 107, learn_skeleton=False)
    variant_fixture(ns, "head_stage2_c384_g16_s2", 384, 16, 2, [17, 0], "auto", 108, learn_skeleton=True)


def main():
    if "--round5" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        round5_fixtures()
        return
    if "--round4" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        round4_fixtures()
        return
    if "--round3" in sys.argv:          # HF cross-check of the 24-block / 16-head restatement (cfg5's backbone, ViT-L/14 @384)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        hf_backbone_fixture("bb_hf_vitl14_384", "dinov2_vitl14", 384, 303)
        return
    if "--round2" in sys.argv:          # only the fixtures added in round 2 (the others are unchanged)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        round2_fixtures(ref_stubs.install())
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # HF first: the torchvision stub installed for the reference import confuses transformers' import probes
    hf_backbone_fixture("bb_hf_vits14_224", "dinov2_vits14", 224, 301)
    hf_backbone_fixture("bb_hf_vitb14_256", "dinov2_vitb14", 256, 302)
    hf_backbone_fixture("bb_hf_vitl14_384", "dinov2_vitl14", 384, 303)
    ns = ref_stubs.install()
    for case in HEAD_CASES:
        head_fixture(ns, *case)
    detector_fixture("det_vits14_224_s1", "dinov2_vits14", 224, 1, 201)
    detector_fixture("det_vits14_224_s5", "dinov2_vits14", 224, 5, 202)
    round2_fixtures(ns)
    round4_fixtures()
    round5_fixtures()
    leaked = [os.path.join(d, x) for d, ds, _ in os.walk(ref_stubs.REF_ROOT) for x in ds if x == "__pycache__"]
    assert not leaked, f"bytecode leaked into the reference tree: {leaked}"


if __name__ == "__main__":
    main()
