"""-m gpu: error behaviour of the boundary (return codes + ec_last_error, no exceptions across the ABI; the
Python face raises like the reference does) and edge inputs the reference accepts."""
import ctypes as C

import numpy as np
import pytest
import torch

from edgecape_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from edgecape_amd.engine import HipEngine
    sd = synth.make_weights("dinov2_vits14", seed=4)
    return HipEngine(sd, arch="dinov2_vits14", image_size=224, max_batch=2, max_shots=1), sd


def _batch(bs=2, seed=5):
    b = synth.make_pairs(bs, 1, 224, seed=seed)
    mask = b["target_weight_s"][0]
    return b, mask, [m["sample_skeleton"][0] for m in b["img_metas"]]


def test_out_of_range_edge_is_an_argument_error(eng):
    e, _ = eng
    b, mask, sk = _batch()
    sk = [list(sk[0]) + [(3, 100)], sk[1]]          # index 100 is outside [0, K): the reference raises IndexError (skeleton.py:178)
    with pytest.raises(_lib.EdgeCapeHipError, match="out of range"):
        e.forward(b["img_q"], b["img_s"], b["target_s"], mask, sk)


def test_pipelined_call_error_leaves_the_pipeline_usable(eng):
    """ec_forward_pipelined with a bad edge list fails before anything is enqueued, with a head still in flight from the call before;
    the next pipelined call and the flush still deliver both good calls' results, bit-equal to ec_forward."""
    e, _ = eng
    b, mask, sk = _batch()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    iq, is_, ts, ms = dev(b["img_q"]), [dev(x) for x in b["img_s"]], [dev(x) for x in b["target_s"]], dev(mask.reshape(2, -1))
    edges, off = e._edges(sk, 2)
    ref = e._outputs(2)
    e.forward_resident(iq, is_, ts, ms, edges, off, ref)
    torch.cuda.synchronize()
    o1, o2, o3 = e._outputs(2), e._outputs(2), e._outputs(2)
    e.forward_pipelined(iq, is_, ts, ms, edges, off, o1)
    bad = edges.copy()
    bad[0, 1] = 100
    with pytest.raises(_lib.EdgeCapeHipError, match="out of range"):
        e.forward_pipelined(iq, is_, ts, ms, bad, off, o2)
    e.pipeline_flush()
    torch.cuda.synchronize()
    for k in ("output_kpts", "similarity_map", "adj"):
        assert torch.equal(o1[0][k], ref[0][k]), k
    e.forward_pipelined(iq, is_, ts, ms, edges, off, o3)
    e.pipeline_flush()
    torch.cuda.synchronize()
    for k in ("output_kpts", "similarity_map", "adj"):
        assert torch.equal(o3[0][k], ref[0][k]), k


def test_batch_larger_than_configured(eng):
    e, _ = eng
    b, mask, sk = _batch(3)
    with pytest.raises(_lib.EdgeCapeHipError, match="maxima"):
        e.forward(b["img_q"], b["img_s"], b["target_s"], mask, sk)


def test_missing_weight_and_call_order():
    lib = _lib.load()
    cfg = _lib.EcConfig(embed_dim=384, depth=12, num_heads=6, image_size=224, patch=14, num_kpts=100, d_model=256, nhead=8, enc_layers=3,
                        dec_layers=3, skel_layers=3, ffn_dim=384, skel_ffn_dim=384, max_hops=4, heatmap_size=64, max_shots=1, max_batch=1,
                        backbone_precision=0, head_precision=0)
    h = C.c_void_p()
    assert lib.ec_create(C.byref(cfg), C.byref(h)) == 0
    x = torch.zeros(1, 3, 224, 224, device="cuda")
    y = torch.zeros(1, 256, 384, device="cuda")
    assert lib.ec_backbone(h, x.data_ptr(), 1, y.data_ptr(), 0, None) == -3 and b"finalized" in lib.ec_last_error()      # EC_ERR_STATE
    assert lib.ec_finalize(h) == -3 and b"missing tensor" in lib.ec_last_error()
    bad = _lib.EcConfig(**{**{f[0]: getattr(cfg, f[0]) for f in cfg._fields_}, "num_kpts": 257})
    h2 = C.c_void_p()
    assert lib.ec_create(C.byref(bad), C.byref(h2)) == -1 and b"num_kpts" in lib.ec_last_error()                           # EC_ERR_ARG
    assert lib.ec_destroy(h) == 0


def test_single_pair_and_single_keypoint(eng):
    """bs = 1 with ONE valid keypoint and an empty skeleton (the demo's degenerate input; the app substitutes [(0,0)])."""
    from oracle import edgecape_oracle as orc
    e, sd = eng
    b = synth.make_pairs(1, 1, 224, seed=11, n_kp=1, skeleton="empty")
    mask = b["target_weight_s"][0]
    o = e.forward(b["img_q"], b["img_s"], b["target_s"], mask, [[]])
    torch.cuda.synchronize()
    res_ref, out_ref = orc.forward_test(sd, b, synth.ARCHS["dinov2_vits14"]["heads"])
    valid = mask[:, :, 0] > 0
    err = np.abs(o["output_kpts"].cpu().numpy() - out_ref["output_kpts"].numpy())[:, valid].max()
    assert valid.sum() == 1 and err < 1e-3
    assert np.abs(o["adj"].cpu().numpy() - out_ref["adj"].numpy()).max() < 1e-4


def test_sample_with_no_visible_support_keypoint(eng):
    """One pair whose support keypoints are ALL invisible (target_weight_s = 0 everywhere): the reference un-masks key 0 of
    such a sample so the softmax over the keypoint tokens stays finite (encoder_decoder.py:359-360, skeleton.py:98-99).
    Every output of that pair must match the oracle (no 'valid' subset to hide behind) and the other pair must not notice."""
    from oracle import edgecape_oracle as orc
    e, sd = eng
    b, mask, skel = _batch(2, seed=21)
    b["target_weight_s"][0][0] = 0.0                      # sample 0: nothing visible
    mask = b["target_weight_s"][0]
    o = e.forward(b["img_q"], b["img_s"], b["target_s"], mask, skel)
    torch.cuda.synchronize()
    res_ref, out_ref = orc.forward_test(sd, b, synth.ARCHS["dinov2_vits14"]["heads"])
    got, ref = o["output_kpts"].cpu().numpy(), out_ref["output_kpts"].numpy()
    assert np.isfinite(got).all() and np.isfinite(ref).all()
    assert np.abs(got[:, 0] - ref[:, 0]).max() < 1e-3                                  # the all-invisible pair, all K slots
    valid1 = mask[1, :, 0] > 0
    assert np.abs(got[:, 1][:, valid1] - ref[:, 1][:, valid1]).max() < 1e-3            # its neighbour in the batch
    assert np.abs(o["adj"].cpu().numpy() - out_ref["adj"].numpy()).max() < 1e-4


def test_dense_support_heatmaps(eng):
    """Support heatmaps that are NOT Gaussian blobs (every pixel positive): the fused pooling kernel then visits all 324 token
    cells per keypoint instead of the usual ~20 - slower, but it must be the same weighted sum (head.py:175-186)."""
    from oracle import edgecape_oracle as orc
    e, sd = eng
    b, mask, skel = _batch(2, seed=33)
    rng = np.random.default_rng(0)
    b["target_s"][0] = (b["target_s"][0] + rng.uniform(0.05, 1.0, b["target_s"][0].shape)).astype(np.float32)
    o = e.forward(b["img_q"], b["img_s"], b["target_s"], mask, skel)
    torch.cuda.synchronize()
    res_ref, out_ref = orc.forward_test(sd, b, synth.ARCHS["dinov2_vits14"]["heads"])
    valid = mask[:, :, 0] > 0
    err = np.abs(o["output_kpts"].cpu().numpy() - out_ref["output_kpts"].numpy())[:, valid].max()
    assert err < 1e-3, err


def test_detector_rejects_what_the_reference_rejects():
    from edgecape_amd.detector import EdgeCape
    head = dict(type="TwoStageHead", in_channels=384,
                transformer=dict(type="TwoStageSupportRefineTransformer", d_model=256, nhead=8, num_encoder_layers=3, num_decoder_layers=3,
                                 dim_feedforward=384, similarity_proj_dim=256, dynamic_proj_dim=128, use_bias_attn_module=True,
                                 attn_bias=True, max_hops=4),
                positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                skeleton_head=dict(type="SkeletonPredictor", learn_skeleton=True), learn_skeleton=True)
    m = EdgeCape(keypoint_head=head, pretrained="dinov2_vits14")
    with pytest.raises(RuntimeError, match="no weights"):
        m(img_s=[torch.zeros(1, 3, 224, 224)], img_q=torch.zeros(1, 3, 224, 224), target_s=[torch.zeros(1, 100, 64, 64)],
          target_weight_s=[torch.ones(1, 100, 1)], img_metas=[dict(sample_skeleton=[[]], query_center=np.zeros(2), query_scale=np.ones(2),
                                                                  query_image_file="q", sample_image_file=["s"])], return_loss=False)
    with pytest.raises(NotImplementedError):
        m(img_s=[], img_q=torch.zeros(1, 3, 224, 224), return_loss=True)
    with pytest.raises(AssertionError):                     # head.py:101-103: embed_dims == 2 * num_feats
        EdgeCape(keypoint_head={**head, "positional_encoding": dict(type="SinePositionalEncoding", num_feats=64, normalize=True)},
                 pretrained="dinov2_vits14")
    with pytest.raises(KeyError):
        EdgeCape(keypoint_head=head, pretrained="resnet50")


@pytest.mark.parametrize("K", [17, 5, 1])
def test_dynamic_keypoint_count_like_the_demo(K):
    """The demos call the model with K = number of clicked points and all weights 1 (gradio_utils/utils.py:142-148) — K is
    whatever target_s[0].shape[1] says.  Checked against the oracle (the golden fixtures are K = 100)."""
    from oracle import edgecape_oracle as orc
    from edgecape_amd.engine import HipEngine
    arch, H, bs = "dinov2_vits14", 224, 2
    sd = synth.make_weights(arch, seed=8)
    b = synth.make_pairs(bs, 1, H, seed=21, n_kp=K, K=K)
    res_ref, out_ref = orc.forward_test(sd, b, synth.ARCHS[arch]["heads"])
    e = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=1, num_kpts=K)
    mask = b["target_weight_s"][0]
    o = e.forward(b["img_q"], b["img_s"], b["target_s"], mask, [m["sample_skeleton"][0] for m in b["img_metas"]])
    torch.cuda.synchronize()
    err = np.abs(o["output_kpts"].cpu().numpy() - out_ref["output_kpts"].numpy()).max()
    eadj = np.abs(o["adj"].cpu().numpy() - out_ref["adj"].numpy()).max()
    print("K", K, "kpt err", err, "adj err", eadj)
    assert o["output_kpts"].shape == (3, bs, K, 2) and err < 1e-3 and eadj < 1e-4
