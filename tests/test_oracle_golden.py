"""Pin the CPU oracle (oracle/edgecape_oracle.py) to outputs of the REAL reference.

The fixtures under tests/golden/ were produced by oracle/make_golden.py importing
/root/reference (head, skeleton head, transformer, detector) and HF Dinov2Model (backbone
cross-check).  Tolerance: 1e-5 abs (fp32 CPU both sides).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from edgecape_amd import synth
from oracle import edgecape_oracle as orc

HEAD = ["head_s1_c384_g16_kp17", "head_s5_c384_g16_mixed", "head_s1_c768_g18_edge", "head_s5_c768_g18_kp17",
        "head_s2_c384_g14x20_kp17", "head_s1_c768_g21x16_mixed"]      # the last two (round 4): NON-SQUARE feature maps through the real head
TOL = 1e-5


def _run_head(meta):
    sd = synth.make_head_weights(C=meta["C"], seed=meta["weight_seed"])
    g = tuple(meta["g"]) if isinstance(meta["g"], list) else meta["g"]
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], meta["C"], g, meta["input_seed"],
                                 meta["n_kps"], meta["skeletons"])
    taps = {}
    with torch.no_grad():
        out = orc.head_forward(sd, inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"],
                               inp["skeleton"], taps=taps)
    return out, taps, inp


@pytest.mark.parametrize("name", HEAD)
def test_head_matches_reference(name):
    gold, meta = load_golden(name)
    out, taps, inp = _run_head(meta)
    # argmax flips would show up as O(1/g) jumps; none are expected at 1e-5 agreement of the maps
    checks = dict(
        support_keypoints=taps["support_keypoints"], adj=out["adj"], attn_adj=out["attn_adj"],
        unnormalized_adj=out["unnormalized_adj"], enc_kp=taps["enc_kp"], enc_img_first8=taps["enc_img"][:8],
        enc_img_last8=taps["enc_img"][-8:], initial_proposals=out["initial_proposals"],
        out_points=out["out_points"], output_kpts=out["output_kpts"], hs_last=out["hs"][-1].transpose(0, 1),
    )
    for k, v in checks.items():
        err = np.abs(v.numpy() - gold[k]).max()
        # initial_proposals is a soft-argmax over exp(similarity) with |similarity| up to ~50: fp32
        # rounding of the logits (1e-5 relative) is amplified by the softmax, hence 3e-5 there.
        tol = 3e-5 if k == "initial_proposals" else TOL
        assert err <= tol, f"{name}:{k} max abs err {err}"
    sim_err = np.abs(out["similarity_map"].numpy() - gold["similarity_map"]).max()
    scale = np.abs(gold["similarity_map"]).max()
    assert sim_err <= 1e-5 * max(1.0, scale), f"{name}: similarity_map err {sim_err} (scale {scale})"


@pytest.mark.parametrize("name", ["head_stage1_c384_g16", "head_stage2_c384_g16_s2"])
def test_head_variants_match_reference(name):
    """The reference head of training stages 1 and 2 (run.py:44-88; built by oracle/make_golden.py --round5 from
    configs/train/1shot_split1.py): SkeletonPredictor(learn_skeleton=False) -> the normalised ground-truth adjacency, and decoder
    layers whose self-attention is nn.MultiheadAttention without the Markov bias (encoder_decoder.py:551-560, 605-612)."""
    gold, meta = load_golden(name)
    sd = synth.as_stage_checkpoint(synth.make_head_weights(C=meta["C"], seed=meta["weight_seed"]))
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], meta["C"], meta["g"], meta["input_seed"], meta["n_kps"], meta["skeletons"])
    with torch.no_grad():
        out = orc.head_forward(sd, inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"], inp["skeleton"],
                               learn_skeleton=meta["learn_skeleton"], attn_bias=meta["attn_bias"])
    for k in ("adj", "initial_proposals", "out_points", "output_kpts"):
        err = np.abs(out[k].numpy() - gold[k]).max()
        assert err <= (3e-5 if k == "initial_proposals" else TOL), f"{name}:{k} max abs err {err}"
    assert np.abs(out["similarity_map"].numpy() - gold["similarity_map"]).max() <= 1e-5 * max(1.0, np.abs(gold["similarity_map"]).max())
    assert (out["attn_adj"] is None) == (not meta["learn_skeleton"])
    if not meta["learn_skeleton"]:      # ground-truth adjacency: rows of valid keypoints sum to one (or are empty), padded rows / columns zero
        for b, nk in enumerate(meta["n_kps"]):
            rs = gold["adj"][b, 1].sum(-1)
            assert np.all((np.abs(rs[:nk] - 1) < 1e-6) | (rs[:nk] == 0)) and np.all(gold["adj"][b, 1][nk:] == 0)


@pytest.mark.parametrize("name", HEAD)
def test_structural_invariants(name):
    """invariants derivable from the reference alone."""
    gold, meta = load_golden(name)
    adj, attn = gold["adj"], gold["attn_adj"]
    for b, nk in enumerate(meta["n_kps"]):
        valid = np.zeros(100, bool)
        valid[:nk] = True
        assert np.array_equal(adj[b, 0], np.diag(valid.astype(np.float32)))
        rs = adj[b, 1].sum(-1)
        if nk > 0:
            assert np.allclose(rs[:nk], 1.0, atol=1e-5)
        assert np.all(adj[b, 1][nk:] == 0) and np.all(adj[b, 1][:, nk:] == 0)
        assert np.array_equal(attn[0, b], np.eye(100, dtype=np.float32))


@pytest.mark.parametrize("name", ["det_vits14_224_s1", "det_vits14_224_s5", "det_vits14_229x311_s2"])
def test_detector_matches_reference(name):
    """(det_vits14_229x311_s2, round 4: the real reference detector on NON-SQUARE images - decode with img_size = [width, height])"""
    gold, meta = load_golden(name)
    sd = synth.make_weights(meta["arch"], seed=meta["weight_seed"])
    size = tuple(meta["image_size"]) if isinstance(meta["image_size"], list) else meta["image_size"]
    batch = synth.make_pairs(meta["bs"], meta["shots"], size, seed=meta["input_seed"])
    res, _ = orc.forward_test(sd, batch, synth.ARCHS[meta["arch"]]["heads"])
    assert np.abs(res["points"] - gold["points"]).max() <= TOL
    assert np.abs(res["skeleton"] - gold["skeleton"]).max() <= TOL
    assert np.abs(res["preds"] - gold["preds"]).max() <= 3e-3  # pixels (x224 .. x311 of 1e-5)
    assert np.array_equal(res["boxes"], gold["boxes"])
    assert list(res["bbox_ids"]) == list(gold["bbox_ids"])
    assert np.all(res["preds"][..., 2] == 1)  # head.py:374


@pytest.mark.parametrize("name", ["bb_hf_vits14_224", "bb_hf_vitb14_256", "bb_hf_vitl14_384", "bb_hf_vits14_224x308"])
def test_backbone_matches_hf(name):
    """(the 224 x 308 fixture, round 4: a non-square image, positional table interpolated by HF itself with per-axis scale factors)"""
    gold, meta = load_golden(name)
    arch = meta["arch"]
    sd = synth.make_backbone_weights(arch, seed=meta["weight_seed"])
    rng = np.random.default_rng(meta["input_seed"])
    size = meta["image_size"]
    img = np.stack([synth._smooth_image(rng, *size) if isinstance(size, list) else synth._smooth_image(rng, size)])
    taps = {}
    with torch.no_grad():
        orc.dinov2_features(sd, img, synth.ARCHS[arch]["heads"], taps=taps)
    assert np.abs(taps["tokens0"][0, :4].numpy() - gold["tokens0_first4"]).max() <= TOL
    assert np.abs(taps["block0"][0, :4].numpy() - gold["block0_first4"]).max() <= 1e-4
    f = taps["feat_tokens"][0]
    assert np.abs(f[:8].numpy() - gold["feat_tokens_first8"]).max() <= 2e-4
    assert np.abs(f[-8:].numpy() - gold["feat_tokens_last8"]).max() <= 2e-4
    assert np.abs(f.mean(0).numpy() - gold["feat_mean"]).max() <= 2e-4


def test_msra_target_invariants():
    """top_down_transform.py:165-194: <=49 non-zeros, peak 1.0 at int(x/stride+0.5)."""
    kp = np.array([[100.3, 57.9], [2.0, 220.0], [-50.0, -50.0]], np.float32)
    t, w = synth.msra_target(kp, np.ones(3), 224)
    assert (t[0] > 0).sum() == 49 and t[0].max() == 1.0
    my, mx = np.unravel_index(t[0].argmax(), t[0].shape)
    assert (mx, my) == (int(100.3 / 3.5 + 0.5), int(57.9 / 3.5 + 0.5))
    assert 0 < (t[1] > 0).sum() < 49
    assert w[2] == 0 and t[2].sum() == 0


def test_pipeline_oracle_matches_reference_msra():
    """oracle/pipeline_oracle.py (and the data synthesiser's synth.msra_target) vs the reference's own _msra_generate_target
    outputs (fixture pre_msra): bit-for-bit, including the border cases and visibility values other than 0 / 1."""
    from oracle.pipeline_oracle import msra_target_ref64
    g, meta = load_golden("pre_msra")
    for i, (image_size, hm) in enumerate(meta["cases"]):
        j, v = g[f"joints_{i}"], g[f"visible_{i}"]
        t, w = msra_target_ref64(j[:, :2], v[:, 0], image_size, heatmap_size=hm, sigma=meta["sigma"])
        assert np.array_equal(t, g[f"target_{i}"]) and np.array_equal(w, g[f"weight_{i}"]), i


@pytest.mark.parametrize("M,gh,gw", [(37, 16, 22), (37, 27, 18), (37, 37, 20), (16, 18, 37)])
def test_posembed_interpolation_nonsquare_vs_torch(M, gh, gw):
    """Per-axis scale factors for a non-square token grid (rows: (gh + 0.1) / M, columns: (gw + 0.1) / M), incl. one axis at the native
    size - upstream still resamples it (scale (M + 0.1) / M)."""
    import torch.nn.functional as F
    from edgecape_amd.posembed import interpolate_pos_embed
    rng = np.random.default_rng(M * 1000 + gh * 40 + gw)
    C = 24
    pe = rng.normal(0, 0.02, (1, 1 + M * M, C)).astype(np.float32)
    got = interpolate_pos_embed(pe, (gh, gw))
    patch = torch.from_numpy(pe[0, 1:]).reshape(1, M, M, C).permute(0, 3, 1, 2)
    ref = F.interpolate(patch, scale_factor=(float(gh + 0.1) / M, float(gw + 0.1) / M), mode="bicubic", antialias=False)
    assert ref.shape[-2:] == (gh, gw)
    ref = ref.permute(0, 2, 3, 1).reshape(gh * gw, C).numpy()
    assert got.shape == (1 + gh * gw, C) and np.array_equal(got[0], pe[0, 0])
    assert np.abs(got[1:] - ref).max() < 1e-6
    assert np.abs(got - orc.interpolate_pos_embed(pe, (gh, gw)).numpy()).max() < 1e-6


@pytest.mark.parametrize("M,g", [(37, 16), (37, 18), (37, 27), (16, 18), (37, 37), (10, 27)])
def test_posembed_interpolation_vs_torch(M, g):
    """The product's numpy bicubic (edgecape_amd/posembed.py, used once at load time) vs torch's upsample_bicubic2d called the
    way dinov2's interpolate_pos_encoding calls it (scale_factor = (g + 0.1) / M, align_corners False, antialias False)."""
    import torch.nn.functional as F
    from edgecape_amd.posembed import interpolate_pos_embed
    rng = np.random.default_rng(M * 100 + g)
    C = 24
    pe = rng.normal(0, 0.02, (1, 1 + M * M, C)).astype(np.float32)
    got = interpolate_pos_embed(pe, g)
    assert got.shape == (1 + g * g, C) and got.dtype == np.float32 and np.array_equal(got[0], pe[0, 0])
    if g == M:
        assert np.array_equal(got, pe[0])
        return
    patch = torch.from_numpy(pe[0, 1:]).reshape(1, M, M, C).permute(0, 3, 1, 2)
    s = float(g + 0.1) / M
    ref = F.interpolate(patch, scale_factor=(s, s), mode="bicubic", antialias=False)
    assert ref.shape[-2:] == (g, g)
    ref = ref.permute(0, 2, 3, 1).reshape(g * g, C).numpy()
    assert np.abs(got[1:] - ref).max() < 1e-6                                             # same 16 taps and fp32 weights; |table| ~ 0.08
    assert np.abs(got - orc.interpolate_pos_embed(pe, g).numpy()).max() < 1e-6            # and the oracle's call of the same op
