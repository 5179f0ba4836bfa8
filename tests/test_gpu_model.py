"""-m gpu parity tests proper: the HIP path through the C ABI vs (i) the committed golden vectors made by
the REAL reference and (ii) the CPU oracle on the same seeded inputs.

Tolerances (fp32 parity mode): north star = 1e-3 abs on output keypoints of valid keypoints; the checks
here are much tighter on every intermediate so that a kernel bug cannot hide inside the 1e-3 budget.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from edgecape_amd import synth

pytestmark = pytest.mark.gpu

HEAD = ["head_s1_c384_g16_kp17", "head_s5_c384_g16_mixed", "head_s1_c768_g18_edge", "head_s5_c768_g18_kp17",
        "head_s2_c384_g14x20_kp17", "head_s1_c768_g21x16_mixed"]   # the last two (round 4): the real head on NON-SQUARE feature maps
ARCH_OF_C = {384: "dinov2_vits14", 768: "dinov2_vitb14", 1024: "dinov2_vitl14"}


def _engine(sd, arch, image_size, bs, shots, **kw):
    from edgecape_amd.engine import HipEngine
    return HipEngine(sd, arch=arch, image_size=image_size, max_batch=bs, max_shots=shots, **kw)


def _valid_mask(n_kps, K=100):
    m = np.zeros((len(n_kps), K), bool)
    for b, nk in enumerate(n_kps):
        m[b, :nk] = True
    return m


@pytest.mark.parametrize("name", HEAD)
def test_head_vs_reference_golden(name):
    gold, meta = load_golden(name)
    C, g = meta["C"], meta["g"]
    g = tuple(g) if isinstance(g, list) else g
    gh, gw = g if isinstance(g, tuple) else (g, g)
    arch = ARCH_OF_C[C]
    sd = synth.make_backbone_weights(arch, seed=3)       # backbone weights are not used by ec_head
    sd.update(synth.make_head_weights(C=C, seed=meta["weight_seed"]))
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], C, g, meta["input_seed"], meta["n_kps"], meta["skeletons"])
    eng = _engine(sd, arch, (gh * 14, gw * 14) if gh != gw else gh * 14, len(meta["n_kps"]), meta["shots"])
    o = eng.head(inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"], inp["skeleton"])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in o.items()}
    errs = {k: float(np.abs(got[k] - gold[k]).max()) for k in ("adj", "attn_adj", "similarity_map", "initial_proposals",
                                                               "out_points", "output_kpts")}
    sk = eng.debug("support_keypoints").reshape(gold["support_keypoints"].shape)
    errs["support_keypoints"] = float(np.abs(sk - gold["support_keypoints"]).max())
    enc = eng.debug("enc").reshape(len(meta["n_kps"]), gh * gw + 100, 256)
    errs["enc_kp"] = float(np.abs(enc[:, gh * gw:].transpose(1, 0, 2) - gold["enc_kp"]).max())
    print(name, errs)
    assert errs["support_keypoints"] < 2e-5
    assert errs["adj"] < 1e-5 and errs["attn_adj"] < 1e-5
    assert errs["enc_kp"] < 1e-4
    assert errs["similarity_map"] < 1e-3          # |similarity| up to ~50
    assert errs["initial_proposals"] < 2e-4
    v = _valid_mask(meta["n_kps"])
    e_valid = np.abs(got["output_kpts"] - gold["output_kpts"])[:, v].max() if v.any() else 0.0
    assert e_valid < 1e-4, e_valid                # north-star bound is 1e-3
    assert errs["output_kpts"] < 1e-3 and errs["out_points"] < 1e-3   # padded slots too


@pytest.mark.parametrize("name,head_precision", [("head_stage1_c384_g16", "fp32"), ("head_stage2_c384_g16_s2", "fp32"),
                                                 ("head_stage1_c384_g16", "mixed"), ("head_stage2_c384_g16_s2", "mixed")])
def test_head_variants_vs_reference_golden(name, head_precision):
    """VERDICT r4 missing item 4: the heads of the reference's earlier training stages (run.py:44-88) against outputs of the REAL
    reference head built from configs/train/1shot_split1.py: stage 1 = SkeletonPredictor(learn_skeleton=False) (ground-truth adjacency,
    skeleton.py:70-74) + decoder self-attention without the Markov bias (nn.MultiheadAttention, encoder_decoder.py:551-560); stage 2 =
    the learnt skeleton, still no bias.  The state dict carries the FUSED in_proj keys such a checkpoint has."""
    gold, meta = load_golden(name)
    C, g = meta["C"], meta["g"]
    arch = ARCH_OF_C[C]
    sd = synth.make_backbone_weights(arch, seed=3)
    sd.update(synth.as_stage_checkpoint(synth.make_head_weights(C=C, seed=meta["weight_seed"])))
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], C, g, meta["input_seed"], meta["n_kps"], meta["skeletons"])
    eng = _engine(sd, arch, g * 14, len(meta["n_kps"]), meta["shots"], learn_skeleton=meta["learn_skeleton"], attn_bias=meta["attn_bias"],
                  head_precision=head_precision)
    o = eng.head(inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"], inp["skeleton"])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in o.items()}
    errs = {k: float(np.abs(got[k] - gold[k]).max()) for k in ("adj", "similarity_map", "initial_proposals", "out_points", "output_kpts")}
    print(name, head_precision, errs)
    exact = head_precision == "fp32"
    assert errs["adj"] < (1e-6 if not meta["learn_skeleton"] else 1e-5 if exact else 1e-3)
    assert errs["similarity_map"] < 1e-3 and errs["initial_proposals"] < 2e-4
    v = _valid_mask(meta["n_kps"])
    assert np.abs(got["output_kpts"] - gold["output_kpts"])[:, v].max() < (1e-4 if exact else 5e-4)      # north-star bound is 1e-3
    assert errs["output_kpts"] < 1e-3 and errs["out_points"] < 1e-3   # padded slots too


@pytest.mark.parametrize("arch,image_size", [("dinov2_vits14", 224), ("dinov2_vitb14", 256)])
def test_backbone_vs_oracle(arch, image_size):
    from oracle import edgecape_oracle as orc
    sd = synth.make_weights(arch, seed=21)
    rng = np.random.default_rng(5)
    img = np.stack([synth._smooth_image(rng, image_size) for _ in range(3)])
    with torch.no_grad():
        ref = orc.dinov2_features(sd, img, synth.ARCHS[arch]["heads"]).numpy()
    eng = _engine(sd, arch, image_size, 3, 1)
    got = eng.backbone(img, nchw=True).cpu().numpy()
    tok = eng.backbone(img, nchw=False).cpu().numpy()
    err = np.abs(got - ref).max()
    print(arch, "feature err", err, "scale", np.abs(ref).max())
    assert err < 2e-4
    g = image_size // 14
    assert np.array_equal(tok.reshape(3, g, g, -1).transpose(0, 3, 1, 2), got)


@pytest.mark.parametrize("arch,image_size,n_img", [("dinov2_vits14", 224, 2), ("dinov2_vitb14", 256, 5), ("dinov2_vits14", (224, 308), 5)])
def test_backbone_bf16x3_kconcat_vs_oracle(arch, image_size, n_img, monkeypatch):
    """bf16x3 backbone in its K-CONCATENATED form (round 5: activations as bf16 [hi | lo | hi] planes written by LayerNorm / attention /
    the fc1 epilogue, weights [W_hi | W_hi | W_lo], one 16-bit GEMM of depth 3 K per Linear) against the CPU oracle and against the
    split-on-load kernels it replaces (EC_BB_X3=0).  2 images of ViT-S: M = 514 rows, the small-M fallback (2-barrier 16-bit kernel with
    the split-plane store); 5 images of ViT-B: M = 1625, the 8-phase kernel's fp32 / residual / GELU + split epilogues with a ragged last
    row tile; ViT-S on a non-square image: N = 1152 / 384 / 1536, half-filled column tiles."""
    from oracle import edgecape_oracle as orc
    sd = synth.make_weights(arch, seed=23)
    rng = np.random.default_rng(9)
    img = np.stack([synth._smooth_image(rng, *image_size) if isinstance(image_size, tuple) else synth._smooth_image(rng, image_size)
                    for _ in range(n_img)])
    with torch.no_grad():
        ref = orc.dinov2_features(sd, img, synth.ARCHS[arch]["heads"]).numpy()
    scale = float(np.abs(ref).max())
    errs = {}
    for x3 in ("1", "0"):
        monkeypatch.setenv("EC_BB_X3", x3)
        eng = _engine(sd, arch, image_size, n_img, 1, backbone_precision="bf16x3", head_precision="bf16x3")
        got = eng.backbone(img, nchw=True).cpu().numpy()
        assert np.isfinite(got).all()
        errs[x3] = float(np.abs(got - ref).max())
        del eng
    print(arch, image_size, n_img, "feature err: K-concatenated", errs["1"], "split on load", errs["0"], "scale", scale)
    assert errs["1"] < 1e-4 * max(scale, 1.0)            # fp32 kernels: < 2e-4 (test_backbone_vs_oracle); bf16x3 sits at ~1e-5
    assert errs["1"] < 3 * errs["0"] + 1e-5


@pytest.mark.parametrize("arch,image_size,n_img", [("dinov2_vits14", 224, 2), ("dinov2_vitb14", 256, 5), ("dinov2_vits14", (224, 308), 5), ("dinov2_vitl14", 384, 1)])
def test_backbone_fp16x2_vs_oracle(arch, image_size, n_img):
    """EC_F16X2 backbone (round 6: fp16 main product + both correction terms in one block-scaled FP8 pass, two MFMA units per product;
    rows travel as [fp16 | e5m2 | e5m2] written by LayerNorm / attention / the fc1 epilogue) against the CPU oracle: features within
    3e-4 of the feature scale (the emulation, oracle/x2_at_scale.py, puts the scheme at ~2e-5 relative; bf16x3 at ~5e-6, fp16 at ~4e-4).
    And BATCH INVARIANCE, which this mode has by construction (one GEMM kernel for every M): the features of image 0 computed alone
    equal, bit for bit, its features inside the batch."""
    from oracle import edgecape_oracle as orc
    sd = synth.make_weights(arch, seed=23)
    rng = np.random.default_rng(9)
    img = np.stack([synth._smooth_image(rng, *image_size) if isinstance(image_size, tuple) else synth._smooth_image(rng, image_size)
                    for _ in range(n_img)])
    with torch.no_grad():
        ref = orc.dinov2_features(sd, img, synth.ARCHS[arch]["heads"]).numpy()
    scale = float(np.abs(ref).max())
    eng = _engine(sd, arch, image_size, n_img, 1, backbone_precision="fp16x2", head_precision="bf16x3")
    got = eng.backbone(img, nchw=True).cpu().numpy()
    assert np.isfinite(got).all()
    err = float(np.abs(got - ref).max())
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(arch, image_size, n_img, "fp16x2 feature err max", err, "relative (Frobenius)", rel, "scale", scale)
    assert err < 3e-4 * max(scale, 1.0) and rel < 1e-4
    one = eng.backbone(img[:1], nchw=True).cpu().numpy()
    assert np.array_equal(one[0], got[0])


@pytest.mark.parametrize("arch,image_size", [("dinov2_vits14", 224), ("dinov2_vitb14", 256), ("dinov2_vitl14", 384),
                                             ("dinov2_vits14", (224, 308))])
def test_backbone_vs_hf_golden(arch, image_size):
    name = {"dinov2_vits14": "bb_hf_vits14_224", "dinov2_vitb14": "bb_hf_vitb14_256", "dinov2_vitl14": "bb_hf_vitl14_384"}[arch]
    if isinstance(image_size, tuple):
        name = "bb_hf_vits14_224x308"                    # round 4: a non-square image
    gold, meta = load_golden(name)
    sd = synth.make_backbone_weights(arch, seed=meta["weight_seed"])
    sd.update(synth.make_head_weights(C=synth.ARCHS[arch]["C"], seed=1))
    rng = np.random.default_rng(meta["input_seed"])
    img = np.stack([synth._smooth_image(rng, *image_size) if isinstance(image_size, tuple) else synth._smooth_image(rng, image_size)])
    eng = _engine(sd, arch, image_size, 1, 1)
    tok = eng.backbone(img, nchw=False).cpu().numpy()[0]
    assert np.abs(tok[:8] - gold["feat_tokens_first8"]).max() < 3e-4
    assert np.abs(tok[-8:] - gold["feat_tokens_last8"]).max() < 3e-4
    assert np.abs(tok.mean(0) - gold["feat_mean"]).max() < 3e-4


@pytest.mark.parametrize("prec", ["fp32/fp32", "fp16x2/bf16x3"])
@pytest.mark.parametrize("name", ["det_vits14_224_s1", "det_vits14_224_s5", "det_vits14_229x311_s2"])
def test_forward_test_vs_reference_golden(name, prec):
    """Whole `model(..., return_loss=False)` through the reference-shaped Python face, against the outputs of the REAL reference detector
    (oracle/make_golden.py): the exact engine, and the tolerance-conforming fast mode (fp16x2 backbone + bf16x3 head, round 6) at the same
    1e-3 tolerance."""
    from edgecape_amd import Config  # noqa: F401
    from edgecape_amd.detector import EdgeCape, hip_library_loaded
    gold, meta = load_golden(name)
    arch = meta["arch"]
    sd = synth.make_weights(arch, seed=meta["weight_seed"])
    size = tuple(meta["image_size"]) if isinstance(meta["image_size"], list) else meta["image_size"]   # (H, W): a non-square fixture (round 4)
    batch = synth.make_pairs(meta["bs"], meta["shots"], size, seed=meta["input_seed"])
    head_cfg = dict(type="TwoStageHead", in_channels=synth.ARCHS[arch]["C"],
                    transformer=dict(type="TwoStageSupportRefineTransformer", d_model=256, nhead=8, num_encoder_layers=3,
                                     num_decoder_layers=3, dim_feedforward=384, dropout=0.1, similarity_proj_dim=256,
                                     dynamic_proj_dim=128, activation="relu", normalize_before=False,
                                     return_intermediate_dec=True, use_bias_attn_module=True, attn_bias=True, max_hops=4),
                    share_kpt_branch=False, num_decoder_layer=3,
                    positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    skeleton_head=dict(type="SkeletonPredictor", learn_skeleton=True), learn_skeleton=True,
                    masked_supervision=True, masking_ratio=0.5, model_freeze="skeleton")
    model = EdgeCape(keypoint_head=head_cfg, encoder_config=dict(), train_cfg=dict(), test_cfg=dict(flip_test=False),
                     pretrained=arch, backbone_precision=prec.split("/")[0], head_precision=prec.split("/")[1])
    model.load_state_dict(sd)
    model.eval()
    t = lambda x: torch.from_numpy(x)
    res = model(img_s=[t(x) for x in batch["img_s"]], img_q=t(batch["img_q"]), target_s=[t(x) for x in batch["target_s"]],
                target_weight_s=[t(x) for x in batch["target_weight_s"]], target_q=t(batch["target_q"]),
                target_weight_q=t(batch["target_weight_q"]), img_metas=batch["img_metas"], return_loss=False)
    assert hip_library_loaded()
    valid = batch["target_weight_s"][0][:, :, 0] > 0
    ep = np.abs(res["points"] - gold["points"])
    print(name, prec, "points err valid", ep[:, valid].max(), "all", ep.max(), "skeleton", np.abs(res["skeleton"] - gold["skeleton"]).max())
    assert ep[:, valid].max() < 1e-3                      # north star: 1e-3 abs on normalised coordinates
    assert np.abs(res["skeleton"] - gold["skeleton"]).max() < 1e-4
    assert np.abs(res["preds"] - gold["preds"])[valid].max() < 0.4   # pixels: 1e-3 * 224 (311) * 1.25
    assert np.array_equal(res["boxes"], gold["boxes"])
    assert list(res["bbox_ids"]) == list(gold["bbox_ids"])
    assert np.all(res["preds"][..., 2] == 1)


def test_forward_vs_oracle_vitb_256():
    """BASELINE config 2 shape (ViT-B/14 @256, 18x18 grid) at a batch the CPU oracle finishes in seconds."""
    from oracle import edgecape_oracle as orc
    arch, H, bs = "dinov2_vitb14", 256, 2
    sd = synth.make_weights(arch, seed=31)
    batch = synth.make_pairs(bs, 1, H, seed=77, fixed_n_kp=False)
    res_ref, out_ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    eng = _engine(sd, arch, H, bs, 1)
    mask = batch["target_weight_s"][0]
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    valid = mask[:, :, 0] > 0
    got = o["output_kpts"].cpu().numpy()
    ref = out_ref["output_kpts"].numpy()
    sim_err = np.abs(o["similarity_map"].cpu().numpy() - out_ref["similarity_map"].numpy()).max()
    flips = (o["similarity_map"].cpu().numpy().reshape(bs, 100, -1).argmax(-1) !=
             out_ref["similarity_map"].numpy().reshape(bs, 100, -1).argmax(-1))[valid].sum()
    err = np.abs(got - ref)[:, valid].max()
    print("vitb256: kpt err", err, "sim err", sim_err, "argmax flips", flips)
    assert flips == 0
    assert err < 1e-3
    assert np.abs(o["adj"].cpu().numpy() - out_ref["adj"].numpy()).max() < 1e-4


@pytest.mark.parametrize("name", ["head_s1_c384_g16_kp17", "head_s5_c768_g18_kp17"])
def test_head_bf16x3_vs_reference_golden(name):
    """Head throughput mode (split-bf16 GEMMs, everything else fp32) still meets the fp32 parity gate against the
    REAL reference's outputs: 1e-3 abs on output keypoints (observed ~1e-5), no argmax flips."""
    gold, meta = load_golden(name)
    C, g = meta["C"], meta["g"]
    arch = ARCH_OF_C[C]
    sd = synth.make_backbone_weights(arch, seed=3)
    sd.update(synth.make_head_weights(C=C, seed=meta["weight_seed"]))
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], C, g, meta["input_seed"], meta["n_kps"], meta["skeletons"])
    eng = _engine(sd, arch, g * 14, len(meta["n_kps"]), meta["shots"], head_precision="bf16x3")
    o = eng.head(inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"], inp["skeleton"])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in o.items()}
    v = _valid_mask(meta["n_kps"])
    bsz = len(meta["n_kps"])
    flips = (got["similarity_map"].reshape(bsz, 100, -1).argmax(-1) != gold["similarity_map"].reshape(bsz, 100, -1).argmax(-1))[v].sum()
    e_valid = np.abs(got["output_kpts"] - gold["output_kpts"])[:, v].max()
    e_adj = np.abs(got["adj"] - gold["adj"]).max()
    e_sim = np.abs(got["similarity_map"] - gold["similarity_map"]).max()
    print(name, "bf16x3 head: kpt err", e_valid, "adj err", e_adj, "sim err", e_sim, "flips", flips)
    assert flips == 0
    assert e_valid < 1e-3
    assert e_adj < 1e-3


def test_forward_bf16x3_head_vs_oracle_vitb_256():
    """fp32 backbone + bf16x3 head end to end at the BASELINE config-2 shape: inside the 1e-3 coordinate gate."""
    from oracle import edgecape_oracle as orc
    arch, H, bs = "dinov2_vitb14", 256, 2
    sd = synth.make_weights(arch, seed=31)
    batch = synth.make_pairs(bs, 1, H, seed=77, fixed_n_kp=False)
    res_ref, out_ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    eng = _engine(sd, arch, H, bs, 1, head_precision="bf16x3")
    mask = batch["target_weight_s"][0]
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    valid = mask[:, :, 0] > 0
    err = np.abs(o["output_kpts"].cpu().numpy() - out_ref["output_kpts"].numpy())[:, valid].max()
    flips = (o["similarity_map"].cpu().numpy().reshape(bs, 100, -1).argmax(-1) !=
             out_ref["similarity_map"].numpy().reshape(bs, 100, -1).argmax(-1))[valid].sum()
    print("vitb256 fp32 backbone + bf16x3 head: kpt err", err, "flips", flips)
    assert flips == 0 and err < 1e-3


def test_bf16_backbone_mode_vitb_256():
    """Throughput mode: backbone GEMMs/attention on bf16 MFMA (fp32 accumulate), head fp32.  bf16 cannot
    meet the 1e-3 coordinate bound; it is judged on feature error and PCK agreement."""
    from oracle import edgecape_oracle as orc
    arch, H, bs = "dinov2_vitb14", 256, 4
    sd = synth.make_weights(arch, seed=31)
    batch = synth.make_pairs(bs, 1, H, seed=77, fixed_n_kp=False)
    res_ref, out_ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    eng = _engine(sd, arch, H, bs, 1, backbone_precision="bf16")
    feat = eng.backbone(batch["img_q"], nchw=True).cpu().numpy()
    fref = out_ref["feature_q"].numpy()
    rel = np.abs(feat - fref).max() / np.abs(fref).max()
    print("bf16 backbone: max feature err / max |feature| =", rel, " mean abs err", np.abs(feat - fref).mean())
    assert rel < 0.05 and np.abs(feat - fref).mean() < 0.02
    mask = batch["target_weight_s"][0]
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    valid = mask[:, :, 0] > 0
    got, ref = o["output_kpts"].cpu().numpy()[-1], out_ref["output_kpts"].numpy()[-1]
    d = np.linalg.norm((got - ref) * H, axis=-1)[valid]            # pixels
    thr = 0.2 * H
    gt = batch["gt_q"]
    pck_g = (np.linalg.norm(got * H - gt, axis=-1)[valid] < thr).mean()
    pck_r = (np.linalg.norm(ref * H - gt, axis=-1)[valid] < thr).mean()
    print(f"bf16 mode: median |d| {np.median(d):.3f}px p90 {np.percentile(d, 90):.3f}px max {d.max():.2f}px; PCK@0.2 {pck_g:.4f} vs {pck_r:.4f}")
    assert abs(pck_g - pck_r) <= 0.1          # north star: PCK@0.2 within +-0.1
    assert np.median(d) < 1.0


def _fwd(eng, batch):
    mask = batch["target_weight_s"][0].copy()
    for tw in batch["target_weight_s"]:
        mask = mask * tw
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items() if isinstance(v, torch.Tensor)}, mask[:, :, 0] > 0


@pytest.mark.parametrize("size", [(224, 308), (294, 196)])
def test_nonsquare_forward_vs_oracle(size):
    """The reference takes any img_q.shape[-2:] (EdgeCape.py:143; VERDICT r3 missing item 4): whole forward_test on non-square
    images - wider than high and higher than wide, one of them not a multiple of 14 (floor semantics) - through the registry-built
    detector, against the oracle (itself pinned on non-square maps by the real head and by HF: tests/test_oracle_golden.py); exact-fp32
    and the bench's default precision, plus the pipelined entry point bit-equal to ec_forward."""
    from oracle import edgecape_oracle as orc
    from edgecape_amd.engine import HipEngine
    arch, bs = "dinov2_vits14", 3
    H, W = size
    sd = synth.make_weights(arch, seed=31)
    batch = synth.make_pairs(bs, 2, (H + 5, W + 3), seed=77, fixed_n_kp=False)      # + 5 / + 3 pixels: the patch embedding floors
    res, out = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    ref = {k: out[k].numpy() for k in ("output_kpts", "similarity_map", "adj", "initial_proposals")}
    mask = batch["target_weight_s"][0] * batch["target_weight_s"][1]
    valid = mask[:, :, 0] > 0
    skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]
    for prec, tol in ((dict(), 1e-4), (dict(backbone_precision="fp16", head_precision="mixed"), 1e-3)):
        eng = HipEngine(sd, arch=arch, image_size=(H + 5, W + 3), max_batch=bs, max_shots=2, **prec)
        assert (eng.gh, eng.gw) == ((H + 5) // 14, (W + 3) // 14)
        o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, skel)
        torch.cuda.synchronize()
        got = {k: o[k].cpu().numpy() for k in ref}
        assert got["similarity_map"].shape == ref["similarity_map"].shape
        flips = (got["similarity_map"].reshape(bs, 100, -1).argmax(-1) != ref["similarity_map"].reshape(bs, 100, -1).argmax(-1))[valid].sum()
        err = np.abs(got["output_kpts"] - ref["output_kpts"])[:, valid].max()
        perr = np.abs(got["initial_proposals"] - ref["initial_proposals"])[valid].max()
        print(size, prec or "fp32", "kpt err", err, "proposal err", perr, "adj err", np.abs(got["adj"] - ref["adj"]).max(), "flips", flips)
        assert flips == 0 and err < tol and perr < (2e-4 if not prec else 2e-3)
        assert np.abs(got["adj"] - ref["adj"]).max() < 1e-4
        if prec:                                       # the pipelined entry point on the same engine: bit-equal
            dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
            e, off = eng._edges(skel, bs)
            outs = eng._outputs(bs)
            eng.forward_pipelined(dev(batch["img_q"]), [dev(x) for x in batch["img_s"]], [dev(x) for x in batch["target_s"]],
                                  dev(mask.reshape(bs, -1)), e, off, outs)
            eng.pipeline_flush()
            torch.cuda.synchronize()
            for k in ref:
                assert np.array_equal(outs[0][k].cpu().numpy(), got[k]), k
        del eng


@pytest.mark.parametrize("K,n_kp", [(7, 7), (150, 131), (256, 40)])
def test_dynamic_keypoint_count_vs_oracle(K, n_kp):
    """The reference takes any number of keypoint slots (head.py:175-184; the demo passes the number of clicked points,
    gradio_utils/utils.py:142-148); VERDICT r3 missing item 4: K up to 256 (was 128).  ViT-S/14 @ 224, 2 pairs, exact-fp32 engine
    against the oracle: a handful of slots, 150 (not a multiple of 4: the scalar tail of the attention-bias loads; 131 valid) and
    256 (the adjacency kernels' limit), plus the bench's default precision on the same pairs."""
    from oracle import edgecape_oracle as orc
    arch, H, bs = "dinov2_vits14", 224, 2
    sd = synth.make_weights(arch, seed=21)
    batch = synth.make_pairs(bs, 1, H, seed=1234 + K, n_kp=n_kp, K=K)
    _, out = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    ref = {k: out[k].numpy() for k in ("output_kpts", "similarity_map", "adj", "initial_proposals")}
    for prec, tol in ((dict(), 1e-4), (dict(backbone_precision="fp16", head_precision="mixed"), 1e-3)):
        eng = _engine(sd, arch, H, bs, 1, num_kpts=K, **prec)
        got, valid = _fwd(eng, batch)
        assert got["output_kpts"].shape == (3, bs, K, 2) and got["adj"].shape == (bs, 2, K, K)
        flips = (got["similarity_map"].reshape(bs, K, -1).argmax(-1) != ref["similarity_map"].reshape(bs, K, -1).argmax(-1))[valid].sum()
        err = np.abs(got["output_kpts"] - ref["output_kpts"])[:, valid].max()
        print("K", K, prec or "fp32", "kpt err", err, "adj err", np.abs(got["adj"] - ref["adj"]).max(), "flips", flips)
        assert flips == 0 and err < tol
        assert np.abs(got["adj"] - ref["adj"]).max() < 1e-4
        del eng


def test_forward_vs_oracle_5shot_vitb_256():
    """BASELINE config 4 shape (5 support images per query, ViT-B/14 @256) at a batch the CPU oracle finishes in seconds:
    support-stack pooling (mean over shots), per-shot skeleton refinement, mask = product of the shots' weights."""
    from oracle import edgecape_oracle as orc
    arch, H, bs, S = "dinov2_vitb14", 256, 2, 5
    sd = synth.make_weights(arch, seed=41)
    batch = synth.make_pairs(bs, S, H, seed=91, fixed_n_kp=False)
    res_ref, out_ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    got, valid = _fwd(_engine(sd, arch, H, bs, S), batch)
    err = np.abs(got["output_kpts"] - out_ref["output_kpts"].numpy())[:, valid].max()
    flips = (got["similarity_map"].reshape(bs, 100, -1).argmax(-1) !=
             out_ref["similarity_map"].numpy().reshape(bs, 100, -1).argmax(-1))[valid].sum()
    print("5-shot vitb256: kpt err", err, "flips", flips)
    assert flips == 0 and err < 1e-3
    assert np.abs(got["adj"] - out_ref["adj"].numpy()).max() < 1e-4


def test_forward_vs_oracle_vitl_384():
    """BASELINE config 5 shape: ViT-L/14 (24 blocks, C = 1024, 16 heads) at 384x384 -> 27x27 grid, 730 tokens (floor
    semantics); skeleton-head GCN width follows the backbone width (F4)."""
    from oracle import edgecape_oracle as orc
    arch, H, bs = "dinov2_vitl14", 384, 1
    sd = synth.make_weights(arch, seed=51)
    batch = synth.make_pairs(bs, 1, H, seed=17, fixed_n_kp=False)
    res_ref, out_ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    eng = _engine(sd, arch, H, bs, 1)
    feat = eng.backbone(batch["img_q"], nchw=True).cpu().numpy()
    ferr = np.abs(feat - out_ref["feature_q"].numpy()).max()
    got, valid = _fwd(eng, batch)
    err = np.abs(got["output_kpts"] - out_ref["output_kpts"].numpy())[:, valid].max()
    flips = (got["similarity_map"].reshape(bs, 100, -1).argmax(-1) !=
             out_ref["similarity_map"].numpy().reshape(bs, 100, -1).argmax(-1))[valid].sum()
    print("vitl384: feature err", ferr, "kpt err", err, "flips", flips)
    assert ferr < 5e-4
    assert flips == 0 and err < 1e-3


def test_full_size_properties_cfg2_bf16():
    """BASELINE config 2 at FULL size (32 pairs, ViT-B/14 @256, throughput precisions): size-independent properties.
      * shard invariance: the first 2 pairs of the 32-pair batch equal a 2-pair batch of the same pairs (pairs are independent,
        SURVEY §8e) — this is what makes data-parallel sharding exact;
      * structural invariants of the reference: adj[:,0] = diag(valid), adj[:,1] rows sum to 1 on valid rows and are
        zero on padded rows/columns, attn_adj[0] = I, all padded keypoint slots of a sample produce identical outputs,
        coordinates inside [0,1]."""
    arch, H, bs = "dinov2_vitb14", 256, 32
    sd = synth.make_weights(arch, seed=0)
    batch = synth.make_pairs(bs, 1, H, seed=1000, fixed_n_kp=False)
    eng = _engine(sd, arch, H, bs, 1, backbone_precision="bf16", head_precision="bf16x3")
    got, valid = _fwd(eng, batch)
    small = synth.make_pairs(2, 1, H, seed=1000, fixed_n_kp=False)
    got2, valid2 = _fwd(eng, small)
    for k in ("output_kpts", "similarity_map", "adj", "initial_proposals"):
        a = got[k][:, :2] if k == "output_kpts" else got[k][:2]
        d = np.abs(a - got2[k]).max()
        print("shard invariance", k, d)
        assert d < 1e-5, (k, d)
    adj, attn = got["adj"], got["attn_adj"]
    K = adj.shape[-1]
    for b in range(bs):
        v = valid[b]
        assert np.array_equal(adj[b, 0], np.diag(v.astype(np.float32)))
        rs = adj[b, 1].sum(-1)
        assert np.allclose(rs[v], 1.0, atol=1e-5) and np.all(adj[b, 1][~v] == 0) and np.all(adj[b, 1][:, ~v] == 0)
        assert np.array_equal(attn[0, b], np.eye(K, dtype=np.float32))
        pad = got["output_kpts"][:, b, ~v]                     # [layers, n_pad, 2]
        if pad.shape[1] > 1:
            assert np.abs(pad - pad[:, :1]).max() == 0.0       # padded slots are indistinguishable tokens
    assert np.all(np.isfinite(got["output_kpts"])) and got["output_kpts"].min() >= 0 and got["output_kpts"].max() <= 1


@pytest.mark.parametrize("shots", [1, 5])
def test_support_cache_matches_pairwise_forward(shots):
    """encode 3 support sets once, run 8 queries against them (episode map with repeats) and compare
    with the plain pairwise ec_forward on the expanded (support, query) batch — identical by construction."""
    arch, H = "dinov2_vits14", 224
    sd = synth.make_weights(arch, seed=61)
    sup = synth.make_pairs(3, shots, H, seed=300, fixed_n_kp=False)          # 3 episodes (their queries are not used)
    qry = synth.make_pairs(8, 1, H, seed=400)
    ep = np.array([0, 0, 1, 2, 2, 2, 1, 0], np.int32)
    eng = _engine(sd, arch, H, 8, shots)
    mask = sup["target_weight_s"][0].copy()
    for tw in sup["target_weight_s"]:
        mask = mask * tw
    skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
    cache = eng.support_encode(sup["img_s"], sup["target_s"], mask, skels)
    got = eng.forward_cached(qry["img_q"], cache, ep)
    torch.cuda.synchronize()
    ref = eng.forward(qry["img_q"], [x[ep] for x in sup["img_s"]], [x[ep] for x in sup["target_s"]], mask[ep], [skels[e] for e in ep])
    torch.cuda.synchronize()
    for k in ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points"):
        d = (got[k] - ref[k]).abs().max().item()
        print("cache vs pairwise", k, d)
        assert d < 1e-6, (k, d)
    with pytest.raises(Exception):
        eng.forward_cached(qry["img_q"], cache, np.full(8, 3, np.int32))     # episode index out of range


# Switches of different subsystems share a run (each alternative path is still exercised; a failing group is bisected by hand), and the
# four child processes run SIDE BY SIDE on the one GPU (round 5: ~35 s each - mostly interpreter start-up, engine builds and the host side of
# small tests - were 6 x 35 s in a row, a third of the suite).
SWITCHES = ["EC_CHAIN=0 EC_G8_DYN=1", "EC_OVERLAP=0 EC_COMPACT=2", "EC_OVERLAP=1 EC_PIPE_FULL=0 EC_COMPACT=0", "EC_G8_DYN=0 EC_GEMM8_OFF=1"]


@pytest.fixture(scope="module")
def switch_children():
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import test_gpu_precision_modes as tpm
    tpm._case("cfg2")          # the oracle's answer for the one oracle-checked test of the children: computed once, read from the cache by all
    sel = ("head_vs_reference_golden or forward_test_vs_reference_golden or (headline_mode_fp16_mixed_head and cfg2) or "
           "(forward_pipelined_bit_equal and 224-4) or support_cache_matches or (episodes_stream and fp16)")
    procs = {}
    for sw in SWITCHES:
        env = dict(os.environ, EC_SWITCH_CHILD="1", **dict(kv.split("=") for kv in sw.split()))
        out = tempfile.TemporaryFile(mode="w+")
        procs[sw] = (subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_model.py"), os.path.join(here, "test_gpu_precision_modes.py"),
                                       os.path.join(here, "test_gpu_next_rows.py"), "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                                      env=env, stdout=out, stderr=subprocess.STDOUT, text=True), out)
    yield procs
    for p, out in procs.values():
        if p.poll() is None:
            p.kill()
        out.close()


@pytest.mark.parametrize("switch", SWITCHES)
def test_runtime_switch_matrix(switch, switch_children):
    """Every A/B switch that keeps an alternative code path alive in the shipped library (README "Runtime switches") through the
    reference-generated golden vectors of the head and the detector, the cfg2 precision gate, the pipelined and the episode-cache
    paths: a switch is read once per process, so each group runs in a child process (all four started together by the fixture)."""
    p, out = switch_children[switch]
    p.wait(timeout=900)
    out.seek(0)
    text = out.read()
    assert p.returncode == 0, text[-3000:]
    assert " passed" in text and "failed" not in text, text[-500:]
