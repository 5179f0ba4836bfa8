"""-m gpu: SURVEY §8f "next" rows and rows a30-a33 against the reference, not against this package's own restatements:
  * MSRA targets (a33 / f3)        ec_msra_targets vs targets produced by the reference's _msra_generate_target (fixture pre_msra)
  * fused-in_proj checkpoint (f2)  a checkpoint with fused decoder in_proj keys -> export_pack / load_pack -> HIP head, vs the
                                   REFERENCE head that loaded the same keys through bias_attn.py:236-265 (fixture ..._fusedckpt)
  * support-side cache (f1)        ec_support_encode + ec_forward_cached vs oracle.forward_test on the expanded (support, query) pairs
  * evaluation loop (a30 - a32)    the real HIP model through apis.single_gpu_test -> evaluation.evaluate, vs the same loop over
                                   the CPU oracle's forward_test
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from edgecape_amd import synth

pytestmark = pytest.mark.gpu


def test_msra_targets_vs_reference_golden():
    from edgecape_amd import preprocess as pp
    g, meta = load_golden("pre_msra")
    for i, (image_size, hm) in enumerate(meta["cases"]):
        joints, vis = g[f"joints_{i}"], g[f"visible_{i}"]
        t, w = pp.msra_targets(joints[None, :, :2], vis[None, :, 0], image_size, heatmap_size=hm, sigma=meta["sigma"])
        t, w = t.cpu().numpy()[0], w.cpu().numpy()[0]
        assert np.array_equal(w, g[f"weight_{i}"]), (i, np.abs(w - g[f"weight_{i}"]).max())
        assert np.array_equal(t, g[f"target_{i}"]), (i, np.abs(t - g[f"target_{i}"]).max())      # bit-exact: same fp32 gaussian patch


def _fuse(sd):
    """q/k/v_proj of every decoder self-attention -> fused in_proj_weight / in_proj_bias (an older-format checkpoint)."""
    out = dict(sd)
    for k in list(sd):
        if k.endswith("self_attn.q_proj.weight"):
            base = k[:-len("q_proj.weight")]
            for kind, fused in (("weight", "in_proj_weight"), ("bias", "in_proj_bias")):
                out[base + fused] = np.concatenate([out.pop(base + f"{n}_proj.{kind}") for n in "qkv"], 0)
    return out


def test_fused_checkpoint_vs_reference_golden(tmp_path):
    from edgecape_amd import checkpoint
    from edgecape_amd.engine import HipEngine
    gold, meta = load_golden("head_s1_c384_g16_kp17_fusedckpt")
    C, g = meta["C"], meta["g"]
    sd = synth.make_backbone_weights("dinov2_vits14", seed=3)
    sd.update(synth.make_head_weights(C=C, seed=meta["weight_seed"]))
    fused = _fuse(sd)
    assert sum(k.endswith("self_attn.in_proj_weight") and ".decoder." in k for k in fused) == meta["fused_layers"]
    assert not any(k.endswith("self_attn.q_proj.weight") for k in fused)
    # checkpoint file as the reference's tools write it, read back through the product's loader, packed for ec_load_tensor
    ck = tmp_path / "stage1_style.pth"
    torch.save({"state_dict": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in fused.items()}, "meta": {}}, ck)
    loaded = torch.load(ck, map_location="cpu", weights_only=False)
    names = checkpoint.export_pack(loaded, str(tmp_path / "pack.safetensors"))
    assert any(n.endswith("decoder.layers.0.self_attn.q_proj.weight") for n in names)       # split back by the product's key handling
    pack = checkpoint.load_pack(str(tmp_path / "pack.safetensors"))
    inp = synth.make_head_inputs(len(meta["n_kps"]), meta["shots"], C, g, meta["input_seed"], meta["n_kps"], meta["skeletons"])
    eng = HipEngine(pack, arch="dinov2_vits14", image_size=g * 14, max_batch=len(meta["n_kps"]), max_shots=meta["shots"])
    o = eng.head(inp["feature_q"], inp["feature_s"], inp["target_s"], inp["mask_s"], inp["skeleton"])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in o.items()}
    v = np.zeros((len(meta["n_kps"]), 100), bool)
    for b, nk in enumerate(meta["n_kps"]):
        v[b, :nk] = True
    assert np.abs(got["adj"] - gold["adj"]).max() < 1e-5
    assert np.abs(got["similarity_map"] - gold["similarity_map"]).max() < 1e-3
    assert np.abs(got["initial_proposals"] - gold["initial_proposals"]).max() < 2e-4
    assert np.abs(got["output_kpts"] - gold["output_kpts"])[:, v].max() < 1e-4


@pytest.mark.parametrize("shots", [1, 5])
def test_support_cache_vs_oracle(shots):
    """f1: 3 cached support sets, 8 queries with a repeating episode map, against the CPU oracle run on the 8 expanded pairs."""
    from oracle import edgecape_oracle as orc
    from edgecape_amd.engine import HipEngine
    arch, H = "dinov2_vits14", 224
    sd = synth.make_weights(arch, seed=61)
    sup = synth.make_pairs(3, shots, H, seed=300, fixed_n_kp=False)          # 3 episodes (their own queries are not used)
    qry = synth.make_pairs(8, 1, H, seed=400)
    ep = np.array([0, 0, 1, 2, 2, 2, 1, 0], np.int32)
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=8, max_shots=shots)
    mask = sup["target_weight_s"][0].copy()
    for tw in sup["target_weight_s"]:
        mask = mask * tw
    skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
    cache = eng.support_encode(sup["img_s"], sup["target_s"], mask, skels)
    got = eng.forward_cached(qry["img_q"], cache, ep)
    torch.cuda.synchronize()
    # the same 8 (support set, query) pairs as one plain batch for the oracle
    metas = [dict(qry["img_metas"][i], sample_skeleton=sup["img_metas"][e]["sample_skeleton"]) for i, e in enumerate(ep)]
    batch = dict(img_q=qry["img_q"], img_s=[x[ep] for x in sup["img_s"]], target_s=[x[ep] for x in sup["target_s"]],
                 target_weight_s=[x[ep] for x in sup["target_weight_s"]], img_metas=metas)
    res, ref = orc.forward_test(sd, batch, synth.ARCHS[arch]["heads"])
    valid = mask[ep][:, :, 0] > 0
    flips = (got["similarity_map"].cpu().numpy().reshape(8, 100, -1).argmax(-1) != ref["similarity_map"].numpy().reshape(8, 100, -1).argmax(-1)) & valid
    assert flips.sum() == 0
    d = np.abs(got["output_kpts"].cpu().numpy() - ref["output_kpts"].numpy())[:, valid]
    print("cache vs oracle: max |d kpt|", d.max(), "adj", np.abs(got["adj"].cpu().numpy() - ref["adj"].numpy()).max())
    assert d.max() < 1e-4                                                        # fp32 engine; north-star bound 1e-3
    assert np.abs(got["adj"].cpu().numpy() - ref["adj"].numpy()).max() < 1e-5
    assert np.abs(got["initial_proposals"].cpu().numpy() - ref["initial_proposals"].numpy())[valid].max() < 2e-4


def test_eval_loop_real_model_vs_oracle(tmp_path):
    """a30-a32: configs/test-style model built from the registry, run by apis.single_gpu_test over a loader of pair batches, scored
    by evaluation.evaluate; the same loop with the CPU oracle as the model gives the reference numbers."""
    from edgecape_amd import apis, evaluation
    from edgecape_amd.detector import EdgeCape, hip_library_loaded
    from oracle import edgecape_oracle as orc
    arch, H = "dinov2_vits14", 224
    sd = synth.make_weights(arch, seed=11)
    head_cfg = dict(type="TwoStageHead", in_channels=synth.ARCHS[arch]["C"],
                    transformer=dict(type="TwoStageSupportRefineTransformer", d_model=256, nhead=8, num_encoder_layers=3,
                                     num_decoder_layers=3, dim_feedforward=384, dropout=0.1, similarity_proj_dim=256,
                                     dynamic_proj_dim=128, activation="relu", normalize_before=False,
                                     return_intermediate_dec=True, use_bias_attn_module=True, attn_bias=True, max_hops=4),
                    share_kpt_branch=False, num_decoder_layer=3,
                    positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    skeleton_head=dict(type="SkeletonPredictor", learn_skeleton=True), learn_skeleton=True,
                    masked_supervision=True, masking_ratio=0.5, model_freeze="skeleton")
    model = EdgeCape(keypoint_head=head_cfg, encoder_config=dict(), train_cfg=dict(), test_cfg=dict(flip_test=False), pretrained=arch)
    model.load_state_dict(sd)
    batches = [synth.make_pairs(n, 1, H, seed=700 + i, first_index=10 * i, fixed_n_kp=False) for i, n in enumerate((3, 2, 3))]
    t = lambda x: torch.from_numpy(x)

    def loader():
        for b in batches:
            yield dict(img_s=[t(x) for x in b["img_s"]], img_q=t(b["img_q"]), target_s=[t(x) for x in b["target_s"]],
                       target_weight_s=[t(x) for x in b["target_weight_s"]], target_q=t(b["target_q"]),
                       target_weight_q=t(b["target_weight_q"]), img_metas=b["img_metas"])

    class OracleModel:                       # the reference detector's call signature over the CPU restatement
        def eval(self):
            return self

        def __call__(self, return_loss=False, **data):
            b = dict(img_q=data["img_q"].numpy(), img_s=[x.numpy() for x in data["img_s"]], target_s=[x.numpy() for x in data["target_s"]],
                     target_weight_s=[x.numpy() for x in data["target_weight_s"]], img_metas=data["img_metas"])
            return orc.forward_test(sd, b, synth.ARCHS[arch]["heads"])[0]

    res_hip = apis.single_gpu_test(model, loader())
    assert hip_library_loaded()
    # the pipelined loop (submit batch i+1 before collecting batch i, ec_forward_pipelined): identical records, same order
    res_pipe = apis.single_gpu_test(model, loader(), pipelined=True)
    assert len(res_pipe) == len(res_hip)
    for a, b in zip(res_pipe, res_hip):
        assert a["bbox_ids"] == b["bbox_ids"] and a["image_paths"] == b["image_paths"]
        assert np.array_equal(a["preds"], b["preds"]) and np.array_equal(a["boxes"], b["boxes"])
    res_ref = apis.single_gpu_test(OracleModel(), loader())
    assert len(res_hip) == len(res_ref) == 8 and all(r["preds"].shape == (1, 100, 3) and r["boxes"].shape == (1, 6) for r in res_hip)
    gt = {}
    for b in batches:
        mask = b["target_weight_s"][0][:, :, 0] > 0
        for i, m in enumerate(b["img_metas"]):
            gt[int(m["bbox_id"])] = dict(joints=b["gt_q"][i], mask=mask[i] & (b["target_weight_q"][i, :, 0] > 0), bbox_thr=float(H))
    nv_hip = evaluation.evaluate(res_hip, gt, str(tmp_path / "hip"), metric=["PCK", "AUC", "EPE", "NME"])
    nv_ref = evaluation.evaluate(res_ref, gt, str(tmp_path / "ref"), metric=["PCK", "AUC", "EPE", "NME"])
    print("hip", dict(nv_hip), "\nref", dict(nv_ref))
    for k in nv_ref:
        tol = 0.1 if k.startswith("PCK") or k in ("mPCK", "AUC") else 0.3 if k == "EPE" else 2e-3        # EPE in pixels, NME relative
        assert abs(nv_hip[k] - nv_ref[k]) <= tol, (k, nv_hip[k], nv_ref[k])
    for a, b in zip(res_hip, res_ref):                     # per-sample records: same ids, boxes bit-equal, keypoints within 0.3 px
        assert a["bbox_ids"] == b["bbox_ids"] and a["image_paths"] == b["image_paths"] and np.array_equal(a["boxes"], b["boxes"])
        m = gt[int(a["bbox_ids"][0])]["mask"]
        assert np.abs(a["preds"][0, m, :2] - b["preds"][0, m, :2]).max() < 0.3 if m.any() else True


_COMPACT_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine
arch, H, bs, S = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
d = np.load(sys.argv[2])
sd = synth.make_weights(arch, seed=7)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16", head_precision="mixed")
skel = [d["edges"][d["off"][b]:d["off"][b + 1]].tolist() for b in range(bs)]
o = eng.forward(d["img_q"], [d["img_s%d" % s] for s in range(S)], [d["target_s%d" % s] for s in range(S)], d["mask"], skel)
torch.cuda.synchronize()
np.savez(sys.argv[7], **{k: o[k].cpu().numpy() for k in ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")})
"""


@pytest.mark.parametrize("arch,H,bs,S", [("dinov2_vits14", 224, 6, 1), ("dinov2_vits14", 224, 5, 3)])
def test_row_compaction_bit_equal(tmp_path, arch, H, bs, S):
    """Row compaction of the token-row chains (round 4: the chains compute the valid keypoint tokens and ONE representative masked
    token per sample, the other masked rows are copies - exact in the reference, head.py:187, skeleton.py:186-189): every output of
    ec_forward (EC_COMPACT=2: compaction in every call; by default only pipelined calls compact, where it pays) bit-equal to a run with
    EC_COMPACT=0 (every row computed; a child process), on a batch with
    the awkward masks - no valid keypoint at all (token 0 then stays a row of its own: its key is un-masked,
    encoder_decoder.py:359-360), every keypoint valid (nothing to copy), one valid, one masked, scattered masks - and 1 / 3 shots."""
    import subprocess
    import sys
    from edgecape_amd.engine import HipEngine
    b = synth.make_pairs(bs, S, H, seed=4321, fixed_n_kp=False)
    mask = b["target_weight_s"][0].copy()
    for tw in b["target_weight_s"]:
        mask = mask * tw
    K = mask.shape[1]
    mask[0] = 0.0                                   # no valid keypoint
    mask[1] = 1.0                                   # all valid
    mask[2] = 0.0; mask[2, 37] = 1.0                # one valid, in the second ballot half's neighbourhood
    mask[3] = 1.0; mask[3, 70] = 0.0                # one masked (its own representative, nothing to copy)
    if bs > 4:
        rng = np.random.default_rng(5)
        mask[4] = (rng.random((K, 1)) < 0.4).astype(np.float32)   # scattered
    skels = [m["sample_skeleton"][0] for m in b["img_metas"]]
    old = os.environ.get("EC_COMPACT")
    os.environ["EC_COMPACT"] = "2"                  # compaction in EVERY call of this engine (default: pipelined calls only; read at ec_finalize)
    try:
        eng = HipEngine(synth.make_weights(arch, seed=7), arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16",
                        head_precision="mixed")
    finally:
        if old is None:
            del os.environ["EC_COMPACT"]
        else:
            os.environ["EC_COMPACT"] = old
    o = eng.forward(b["img_q"], b["img_s"], b["target_s"], mask, skels)
    torch.cuda.synchronize()
    edges, off = eng._edges(skels, bs)
    inp = dict(img_q=b["img_q"], mask=mask, edges=edges.reshape(-1, 2), off=off)
    for s in range(S):
        inp["img_s%d" % s] = b["img_s"][s]; inp["target_s%d" % s] = b["target_s"][s]
    np.savez(tmp_path / "in.npz", **inp)
    (tmp_path / "child.py").write_text(_COMPACT_CHILD)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, str(tmp_path / "child.py"), root, str(tmp_path / "in.npz"), arch, str(H), str(bs), str(S), str(tmp_path / "out.npz")],
                       env=dict(os.environ, EC_COMPACT="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(tmp_path / "out.npz")
    for k in ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points"):
        got = o[k].cpu().numpy()
        assert np.isfinite(got).all(), k
        assert np.array_equal(got, ref[k]), (k, float(np.abs(got - ref[k]).max()))
    # and the masked slots of a sample are indistinguishable tokens, as in the reference (sample 0: all but token 0)
    kp = o["output_kpts"].cpu().numpy()
    for bi in range(bs):
        pad = np.nonzero(mask[bi, :, 0] == 0)[0]
        if bi == 0:
            pad = pad[1:]
        if len(pad) > 1:
            assert np.abs(kp[:, bi, pad] - kp[:, bi, pad[:1]]).max() == 0.0


def test_submit_returns_before_its_head_has_finished():
    """detector.submit() must not block on the batch's head (its device -> host copies go into PINNED buffers; a copy into
    pageable memory would hold the calling thread until the head has run, and the next submit() could not overlap it).  At cfg2
    size the head alone takes > 1.5 ms of device time behind a ~5 ms backbone, submit()'s host part well under a millisecond:
    the ticket's `done` event must still be pending when submit() returns, and collect() must deliver forward_test's result."""
    from edgecape_amd.detector import EdgeCape
    arch, H, bs = "dinov2_vitb14", 256, 32
    sd = synth.make_weights(arch, seed=3)
    head_cfg = dict(type="TwoStageHead", in_channels=synth.ARCHS[arch]["C"],
                    transformer=dict(type="TwoStageSupportRefineTransformer", d_model=256, nhead=8, num_encoder_layers=3,
                                     num_decoder_layers=3, dim_feedforward=384, dropout=0.1, similarity_proj_dim=256,
                                     dynamic_proj_dim=128, activation="relu", normalize_before=False,
                                     return_intermediate_dec=True, use_bias_attn_module=True, attn_bias=True, max_hops=4),
                    share_kpt_branch=False, num_decoder_layer=3,
                    positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    skeleton_head=dict(type="SkeletonPredictor", learn_skeleton=True, dim_feedforward=synth.ARCHS[arch]["C"]), learn_skeleton=True,
                    masked_supervision=True, masking_ratio=0.5, model_freeze="skeleton")
    model = EdgeCape(keypoint_head=head_cfg, encoder_config=dict(), train_cfg=dict(), test_cfg=dict(flip_test=False), pretrained=arch,
                     backbone_precision="fp16", head_precision="mixed")
    model.load_state_dict(sd)
    b = synth.make_pairs(bs, 1, H, seed=900, fixed_n_kp=False)
    t = lambda x: torch.from_numpy(x).cuda()
    data = dict(img_s=[t(x) for x in b["img_s"]], img_q=t(b["img_q"]), target_s=[t(x) for x in b["target_s"]],
                target_weight_s=[torch.from_numpy(x) for x in b["target_weight_s"]], img_metas=b["img_metas"])
    ref = model(return_loss=False, **data)                   # also the warm-up (engine build, first launches)
    torch.cuda.synchronize()
    pending = []
    for _ in range(3):
        ticket = model.submit(**data)
        pending.append(ticket["done"].query())               # True = the copies (hence the head) had finished inside submit()
        res = model.collect(ticket)
        assert np.array_equal(res["preds"], ref["preds"]) and np.array_equal(res["points"], ref["points"]) and np.array_equal(res["skeleton"], ref["skeleton"])
    assert not all(pending), pending     # (a blocking copy makes EVERY submit wait for its head; one slow host moment must not fail the test)
    assert all(h.is_pinned() for free in model._pin_pool.values() for h in free) and sum(len(f) for f in model._pin_pool.values()) == 3


def test_forward_pipelined_stress():
    """A long random sequence of pipelined / plain calls and flushes on one handle (the head of call i beside the backbone of call
    i + 1, heads queued behind one another, plain calls and flushes cutting in): every call's outputs bit-equal to ec_forward's."""
    from edgecape_amd.engine import HipEngine
    arch, H, bs, S = "dinov2_vits14", 224, 6, 1
    sd = synth.make_weights(arch, seed=4)
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16", head_precision="mixed")
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    keys = ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")
    batches, ref = [], []
    for i in range(6):
        b = synth.make_pairs(bs, S, H, seed=7000 + 50 * i, fixed_n_kp=False)
        e, o = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
        batches.append(dict(iq=dev(b["img_q"]), is_=[dev(x) for x in b["img_s"]], ts=[dev(x) for x in b["target_s"]],
                            ms=dev(b["target_weight_s"][0].reshape(bs, -1)), edges=e, off=o))
        outs = eng._outputs(bs)
        eng.forward_resident(batches[-1]["iq"], batches[-1]["is_"], batches[-1]["ts"], batches[-1]["ms"], e, o, outs)
        torch.cuda.synchronize()
        ref.append({k: outs[0][k].clone() for k in keys})
    rng = np.random.default_rng(11)
    copy_stream = torch.cuda.Stream()
    pending = []                                                   # (batch index, host copies, event)
    n_pipe = 0
    for step in range(60):
        bi = int(rng.integers(0, len(batches)))
        b = batches[bi]
        outs = eng._outputs(bs)                                    # fresh output set per call: results may be fetched late
        if rng.random() < 0.75:
            eng.forward_pipelined(b["iq"], b["is_"], b["ts"], b["ms"], b["edges"], b["off"], outs)
            n_pipe += 1
        else:
            eng.forward_resident(b["iq"], b["is_"], b["ts"], b["ms"], b["edges"], b["off"], outs)
        copy_stream.wait_stream(torch.cuda.current_stream())
        eng.pipeline_flush(copy_stream)                            # (after a plain call: nothing pending, a no-op wait)
        with torch.cuda.stream(copy_stream):
            host = {k: outs[0][k].to("cpu", non_blocking=True) for k in keys}
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        pending.append((bi, host, ev, outs))
        if rng.random() < 0.2:
            eng.pipeline_flush()                                   # a flush on the compute stream cuts the overlap now and then
    torch.cuda.synchronize()
    assert n_pipe > 30
    for step, (bi, host, ev, _) in enumerate(pending):
        for k in keys:
            assert torch.equal(host[k], ref[bi][k].cpu()), (step, bi, k)


@pytest.mark.parametrize("arch,H,bs,S,nb,prec", [("dinov2_vits14", 224, 4, 1, 5, "fp16/mixed"), ("dinov2_vits14", 224, 4, 5, 5, "fp16/mixed"),
                                                    ("dinov2_vitb14", 256, 32, 1, 3, "fp16/mixed"), ("dinov2_vits14", 224, 32, 1, 3, "fp16/mixed"),
                                                    ("dinov2_vitl14", 384, 8, 1, 3, "fp16/mixed"), ("dinov2_vitb14", 256, 16, 5, 3, "fp16/mixed"),
                                                    ("dinov2_vitb14", 256, 32, 1, 3, "fp16x2/bf16x3"), ("dinov2_vits14", 224, 4, 1, 3, "fp16x2/bf16x3")])
def test_forward_pipelined_bit_equal(arch, H, bs, S, nb, prec):
    """ec_forward_pipelined (the head of call i beside the backbone of call i+1) against ec_forward on the same batches: every
    output of every batch bit-equal, with alternating output sets, results fetched through a copy stream behind ec_pipeline_flush,
    and with plain ec_forward calls mixed into the sequence (every other entry point waits for a pending head).  The cfg2-sized case
    also runs the backbone's QKV / fc1 GEMMs on their DYNAMIC tile schedule in the pipelined calls (static in ec_forward) - in the
    headline precision and in the conforming one (fp16x2 backbone / bf16x3 head, round 6)."""
    from edgecape_amd.engine import HipEngine
    sd = synth.make_weights(arch, seed=3)
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision=prec.split("/")[0], head_precision=prec.split("/")[1])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    batches = []
    for i in range(nb):
        b = synth.make_pairs(bs, S, H, seed=900 + 100 * i, fixed_n_kp=False)
        mask = b["target_weight_s"][0].copy()
        for tw in b["target_weight_s"]:
            mask = mask * tw
        e, o = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
        batches.append(dict(iq=dev(b["img_q"]), is_=[dev(x) for x in b["img_s"]], ts=[dev(x) for x in b["target_s"]],
                            ms=dev(mask.reshape(bs, -1)), edges=e, off=o))
    keys = ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")
    ref = []
    for b in batches:                                           # the unpipelined answers
        o, _ = outs = eng._outputs(bs)
        eng.forward_resident(b["iq"], b["is_"], b["ts"], b["ms"], b["edges"], b["off"], outs)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in keys})
    copy_stream = torch.cuda.Stream()
    sets = [eng._outputs(bs) for _ in range(2)]
    got, events = [], []
    for i, b in enumerate(batches):
        outs = sets[i & 1]
        if i >= 2:                                              # the set is about to be reused: its previous results were copied out
            events[i - 2].synchronize()
        eng.forward_pipelined(b["iq"], b["is_"], b["ts"], b["ms"], b["edges"], b["off"], outs)
        copy_stream.wait_stream(torch.cuda.current_stream())
        eng.pipeline_flush(copy_stream)
        with torch.cuda.stream(copy_stream):
            got.append({k: outs[0][k].to("cpu", non_blocking=True) for k in keys})
            ev = torch.cuda.Event()
            ev.record(copy_stream)
            events.append(ev)
    torch.cuda.synchronize()
    for i in range(len(batches)):
        for k in keys:
            assert torch.equal(got[i][k], ref[i][k].cpu()), (i, k)
    # mixed sequence: pipelined, plain, pipelined, flush on the compute stream
    o0, o1, o2 = eng._outputs(bs), eng._outputs(bs), eng._outputs(bs)
    b0, b1, b2 = batches[0], batches[1], batches[2]
    eng.forward_pipelined(b0["iq"], b0["is_"], b0["ts"], b0["ms"], b0["edges"], b0["off"], o0)
    eng.forward_resident(b1["iq"], b1["is_"], b1["ts"], b1["ms"], b1["edges"], b1["off"], o1)
    eng.forward_pipelined(b2["iq"], b2["is_"], b2["ts"], b2["ms"], b2["edges"], b2["off"], o2)
    eng.pipeline_flush()
    torch.cuda.current_stream().synchronize()
    for o, r in ((o0, ref[0]), (o1, ref[1]), (o2, ref[2])):
        for k in keys:
            assert torch.equal(o[0][k], r[k]), k


def _episode_stream_data(n_ep, qpe, S, H):
    """n_ep episodes of qpe queries each, in the reference's pair order (test_dataset.py:86-99): supports from one synthetic set,
    queries from another; returns (sup batch, mask [n_ep,K,1], skeletons, query images [n_ep*qpe,3,H,H], episode_of_pair)."""
    sup = synth.make_pairs(n_ep, S, H, seed=300, fixed_n_kp=False)
    qry = synth.make_pairs(n_ep * qpe, 1, H, seed=400)
    mask = sup["target_weight_s"][0].copy()
    for tw in sup["target_weight_s"]:
        mask = mask * tw
    skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
    return sup, mask, skels, qry["img_q"], np.repeat(np.arange(n_ep, dtype=np.int32), qpe)


def _run_episode_stream(eng, cache, calls, sup, mask, skels, img_q, pipelined):
    keys = ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")
    res = []
    for c in calls:
        new = None
        if len(c["new_episodes"]):
            e = np.asarray(c["new_episodes"])
            new = dict(img_s=[x[e] for x in sup["img_s"]], target_s=[x[e] for x in sup["target_s"]], mask_s=mask[e],
                       skeletons=[skels[i] for i in e], slots=c["new_slots"])
        o = eng.forward_episodes(cache, img_q[c["queries"]], c["slot_of_query"], new=new, pipelined=pipelined)   # (own output set per call)
        res.append(o)
    if pipelined:
        eng.pipeline_flush()
    torch.cuda.synchronize()
    return [{k: o[k].cpu().numpy() for k in keys} for o in res]


@pytest.mark.parametrize("shots,precision", [(1, "fp32"), (5, "fp32"), (1, "fp16"), (3, "fp16")])
def test_forward_episodes_stream(shots, precision):
    """f1, streaming form (ec_forward_episodes): 5 episodes x 5 queries in the reference's pair order, cut into calls of 6 queries
    with a 3-slot cache (slots are re-used; calls with two, one and NO new episode occur; the supports of the new episodes ride in
    the queries' backbone pass).  (a) every call equals ec_forward on its expanded (support set, query) pairs - which the oracle and
    the golden fixtures pin - to 1e-6 (fp32) / the precision gates' bound (fp16 / mixed: the backbone batch composition differs, the
    arithmetic per image does not); (b) the same stream through the PIPELINED form, head of call i beside the backbone of call
    i + 1, row compaction on, is bit-equal to the plain one; (c) and so is ec_support_encode + ec_forward_cached; (d) every call against
    the CPU oracle on its expanded pairs."""
    from edgecape_amd.engine import HipEngine, SupportCache
    from edgecape_amd.episodes import stream_schedule
    arch, H, n_ep, qpe, bs = "dinov2_vits14", 224, 5, 5, 6
    sd = synth.make_weights(arch, seed=61)
    sup, mask, skels, img_q, ep = _episode_stream_data(n_ep, qpe, shots, H)
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=shots, backbone_precision=precision,
                    head_precision="mixed" if precision == "fp16" else "fp32")
    calls = stream_schedule(ep, bs, 3)
    assert [len(c["new_episodes"]) for c in calls] == [2, 1, 1, 1, 0]
    plain = _run_episode_stream(eng, SupportCache(eng, 3), calls, sup, mask, skels, img_q, False)
    piped = _run_episode_stream(eng, SupportCache(eng, 3), calls, sup, mask, skels, img_q, True)
    for i, c in enumerate(calls):
        e = ep[c["queries"]]
        ref = eng.forward(img_q[c["queries"]], [x[e] for x in sup["img_s"]], [x[e] for x in sup["target_s"]], mask[e], [skels[j] for j in e])
        torch.cuda.synchronize()
        for k, got in plain[i].items():
            d = float(np.abs(got - ref[k].cpu().numpy()).max())
            # fp16 / mixed: the skeleton head's image projections switch kernels with the number of support images in the call (the
            # 8-phase 16-bit GEMM from 1024 rows on, ec_model.hip project_image_kv) - same precision class, not the same bits
            tol = 1e-6 if precision == "fp32" else {"output_kpts": 5e-4, "out_points": 5e-4, "adj": 1e-3, "attn_adj": 1e-3}.get(k, 1e-6)
            if precision != "fp32" and len(c["queries"]) < bs:
                # the last call has ONE query: below 1024 token rows the backbone's GEMMs leave the 8-phase kernel for the 2-barrier one,
                # whose tile shape follows M (1 image here, 1 + S in ec_forward) - fp16 roundings of the same class, not the same bits
                tol = {"similarity_map": 0.5, "adj": 1e-3, "attn_adj": 1e-3}.get(k, 5e-3)
            assert d < tol, (i, k, d)
            assert np.array_equal(got, piped[i][k]), (i, k, float(np.abs(got - piped[i][k]).max()))
    # (d) against the ORACLE (the streaming form was only ever compared with ec_forward): every expanded (support set, query)
    # pair of every call through the CPU restatement of the reference, which recomputes the support side per pair (EdgeCape.py:131-163)
    from oracle import edgecape_oracle as orc   # the checker
    import os
    import test_gpu_precision_modes as tpm
    heads = synth.ARCHS[arch]["heads"]
    # (the children of test_runtime_switch_matrix run four at a time beside the suite: they keep the HIP-vs-HIP parts only; the GPU boxes
    #  show 256 CPUs and grant 16 - an eager CPU forward on torch's default thread count thrashes: tpm.effective_cpus)
    torch.set_num_threads(max(1, min(16, tpm.effective_cpus())))
    for i, c in enumerate(calls if not os.environ.get("EC_SWITCH_CHILD") else []):
        e = ep[c["queries"]]
        with torch.no_grad():
            fq = orc.dinov2_features(sd, img_q[c["queries"]], heads)
            fs = [orc.dinov2_features(sd, x[e], heads) for x in sup["img_s"]]
            ref = orc.head_forward(sd, fq, fs, [x[e] for x in sup["target_s"]], orc._t(mask[e]), [skels[j] for j in e])
        valid = mask[e][:, :, 0] > 0
        am_g = plain[i]["similarity_map"].reshape(len(e), valid.shape[1], -1).argmax(-1)
        am_r = ref["similarity_map"].numpy().reshape(len(e), valid.shape[1], -1).argmax(-1)
        flip = (am_g != am_r) & valid
        clean = ~flip.any(1)
        d = np.abs(plain[i]["output_kpts"] - ref["output_kpts"].numpy())
        dmax = float(d[:, clean][:, valid[clean]].max()) if clean.any() else 0.0
        # fp32 engine: exact-fp32 MFMAs, 1e-6 observed; fp16 / mixed: the headline precision's single-batch gates (max 3e-4 on flip-free samples, <= 1 flip)
        assert int(flip.sum()) <= (0 if precision == "fp32" else 1), (i, int(flip.sum()))
        assert dmax < (5e-5 if precision == "fp32" else 3e-4), (i, dmax)
        assert np.abs(plain[i]["adj"] - ref["adj"].numpy()).max() < (1e-5 if precision == "fp32" else 1e-3), i
    # (c) the two-call form on the same queries: all episodes encoded at once, then the queries of call 1 (episodes 1 and 2)
    cache = eng.support_encode(sup["img_s"], sup["target_s"], mask, skels)
    got = eng.forward_cached(img_q[calls[1]["queries"]], cache, ep[calls[1]["queries"]])
    torch.cuda.synchronize()
    for k in plain[1]:
        d = float(np.abs(got[k].cpu().numpy() - plain[1][k]).max())
        tol = 1e-6 if precision == "fp32" else {"output_kpts": 5e-4, "out_points": 5e-4, "adj": 1e-3, "attn_adj": 1e-3}.get(k, 1e-6)
        assert d < tol, (k, d)
    # error behaviour: an empty slot, a slot number past the cache, two new episodes in one slot
    c2 = SupportCache(eng, 3)
    with pytest.raises(Exception, match="empty"):
        eng.forward_episodes(c2, img_q[:2], np.array([0, 0], np.int32))
    with pytest.raises(Exception, match="out of range"):
        eng.forward_episodes(c2, img_q[:2], np.array([3, 0], np.int32))
    with pytest.raises(Exception, match="share a cache slot"):
        eng.forward_episodes(c2, None, None, new=dict(img_s=[x[:2] for x in sup["img_s"]], target_s=[x[:2] for x in sup["target_s"]],
                                                      mask_s=mask[:2], skeletons=skels[:2], slots=[1, 1]))
    torch.cuda.synchronize()


def test_one_shot_call_on_a_multi_shot_model_pipelined():
    """a model built for max_shots = 3 and called with S = 1 through ec_forward_pipelined (row compaction on by
    default there) ran its skeleton head on the row plan of an EARLIER S = 3 call - or on an empty one.  A 3-shot call, then a 1-shot
    call with a different batch mask, both pipelined: the 1-shot outputs must be bit-equal to plain ec_forward (no compaction)."""
    from edgecape_amd.engine import HipEngine
    arch, H, bs = "dinov2_vits14", 224, 4
    eng = HipEngine(synth.make_weights(arch, seed=9), arch=arch, image_size=H, max_batch=bs, max_shots=3, backbone_precision="fp16",
                    head_precision="mixed")
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    prepared = []
    for S, seed in ((3, 50), (1, 77)):
        b = synth.make_pairs(bs, S, H, seed=seed, fixed_n_kp=False)
        mask = b["target_weight_s"][0].copy()
        for tw in b["target_weight_s"]:
            mask = mask * tw
        e, o = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
        prepared.append((dev(b["img_q"]), [dev(x) for x in b["img_s"]], [dev(x) for x in b["target_s"]], dev(mask.reshape(bs, -1)), e, o))
    keys = ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")
    ref = eng._outputs(bs)
    eng.forward_resident(*prepared[1], ref)
    torch.cuda.synchronize()
    o3, o1 = eng._outputs(bs), eng._outputs(bs)
    eng.forward_pipelined(*prepared[0], o3)
    eng.forward_pipelined(*prepared[1], o1)
    eng.pipeline_flush()
    torch.cuda.synchronize()
    for k in keys:
        assert torch.isfinite(o1[0][k]).all(), k
        assert torch.equal(o1[0][k], ref[0][k]), (k, float((o1[0][k] - ref[0][k]).abs().max()))


@pytest.mark.parametrize("shots", [1, 2])
def test_detector_episode_cache_is_a_drop_in(shots):
    """SURVEY 8f rank 1 at the reference's OWN call boundary: the evaluation loop hands the detector batches of (support set, query)
    pairs in which 15 consecutive pairs carry the same support set (test_dataset.py:86-99).  With enable_episode_cache() forward_test
    (and the pipelined submit / collect loop) recognise a support set by its annotations in img_metas, encode it once - in the backbone
    pass of the batch that first shows it - and return the records of the plain path: 3 support sets x 4 queries in batches of 5 (the
    batches cut through the episodes, one batch meets a support set the 2-slot cache has already dropped)."""
    from edgecape_amd import apis
    from edgecape_amd.detector import EdgeCape
    arch, H, n_ep, qpe, bs = "dinov2_vits14", 224, 3, 4, 5
    sd = synth.make_weights(arch, seed=11)
    head_cfg = dict(type="TwoStageHead", in_channels=synth.ARCHS[arch]["C"],
                    transformer=dict(type="TwoStageSupportRefineTransformer", d_model=256, nhead=8, num_encoder_layers=3,
                                     num_decoder_layers=3, dim_feedforward=384, dropout=0.1, similarity_proj_dim=256,
                                     dynamic_proj_dim=128, activation="relu", normalize_before=False,
                                     return_intermediate_dec=True, use_bias_attn_module=True, attn_bias=True, max_hops=4),
                    share_kpt_branch=False, num_decoder_layer=3,
                    positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    skeleton_head=dict(type="SkeletonPredictor", learn_skeleton=True), learn_skeleton=True)
    sup = synth.make_pairs(n_ep, shots, H, seed=300, fixed_n_kp=False)
    qry = synth.make_pairs(n_ep * qpe, 1, H, seed=400)
    order = [0] * qpe + [1] * qpe + [2] * qpe
    order[-1] = 0                                      # the last pair returns to support set 0, long after it left a 2-slot cache
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x))

    def loader(files_only=False):
        for b0 in range(0, len(order), bs):
            e = np.array(order[b0:b0 + bs])
            q = np.arange(b0, b0 + len(e))
            metas = []
            for qi, ei in zip(q, e):
                m = dict(qry["img_metas"][qi])
                for k in [k for k in m if k.startswith("sample_")]:
                    del m[k]
                for k in (("sample_skeleton", "sample_image_file") if files_only else [k for k in sup["img_metas"][ei] if k.startswith("sample_")]):
                    m[k] = sup["img_metas"][ei][k]
                m["bbox_id"] = int(qi)
                metas.append(m)
            yield dict(img_s=[t(x[e]) for x in sup["img_s"]], img_q=t(qry["img_q"][q]), target_s=[t(x[e]) for x in sup["target_s"]],
                       target_weight_s=[t(x[e]) for x in sup["target_weight_s"]], target_q=t(qry["target_q"][q]),
                       target_weight_q=t(qry["target_weight_q"][q]), img_metas=metas)

    def run(cache_slots, pipelined, files_only=False):
        model = EdgeCape(keypoint_head=head_cfg, encoder_config=dict(), train_cfg=dict(), test_cfg=dict(flip_test=False), pretrained=arch)
        model.load_state_dict(sd)
        if cache_slots:
            model.enable_episode_cache(cache_slots)
        return apis.single_gpu_test(model, loader(files_only), pipelined=pipelined), model

    ref, _ = run(0, False)
    for slots, pipelined in ((2, False), (8, False), (2, True)):
        got, model = run(slots, pipelined)
        assert len(got) == len(ref) == n_ep * qpe
        for a, b in zip(got, ref):
            assert a["bbox_ids"] == b["bbox_ids"] and a["image_paths"] == b["image_paths"] and np.array_equal(a["boxes"], b["boxes"])
            assert np.abs(a["preds"] - b["preds"]).max() < 1e-3, (slots, pipelined, float(np.abs(a["preds"] - b["preds"]).max()))   # pixels of a 224 px box
        st = next(iter(model._episodes.values()))
        assert len(st["slot_of"]) == min(slots, n_ep)   # three support sets seen, at most `slots` kept
    # metas that carry nothing but the file names cannot tell two annotations of one image apart - such batches take the
    # plain path (no cache state is built) and return the plain path's records
    got, model = run(8, False, files_only=True)
    assert not model._episodes
    for a, b in zip(got, ref):
        assert np.array_equal(a["preds"], b["preds"]) and a["bbox_ids"] == b["bbox_ids"]


def test_forward_episodes_stress_ragged_stream():
    """ec_forward_episodes on a RAGGED stream, pipelined with two alternating output sets and a copy stream behind ec_pipeline_flush (the
    way detector.submit uses it): 11 episodes of 1..7 queries in calls of 5 with a 4-slot cache - slots re-used many times, calls with
    three new episodes and with none, plain ec_forward calls mixed in between - every output of every call bit-equal to the same stream
    issued plain (complete at the end of each call), fp16 / mixed with row compaction on in the pipelined calls only."""
    from edgecape_amd.engine import HipEngine, SupportCache
    from edgecape_amd.episodes import stream_schedule
    arch, H, bs, S = "dinov2_vits14", 224, 5, 2
    rng = np.random.default_rng(11)
    sizes = rng.integers(1, 8, 11)
    ep = np.repeat(np.arange(len(sizes), dtype=np.int32), sizes)
    sup = synth.make_pairs(len(sizes), S, H, seed=310, fixed_n_kp=False)
    qry = synth.make_pairs(len(ep), 1, H, seed=410)
    mask = sup["target_weight_s"][0].copy()
    for tw in sup["target_weight_s"]:
        mask = mask * tw
    mask[3] = 0.0                                   # an episode without any valid keypoint
    skels = [m["sample_skeleton"][0] for m in sup["img_metas"]]
    eng = HipEngine(synth.make_weights(arch, seed=23), arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16",
                    head_precision="mixed")
    calls = stream_schedule(ep, bs, 4)
    assert max(len(c["new_episodes"]) for c in calls) >= 3 and min(len(c["new_episodes"]) for c in calls) == 0
    prepared = []
    for c in calls:
        new = None
        if len(c["new_episodes"]):
            e = np.asarray(c["new_episodes"])
            new = dict(img_s=[x[e] for x in sup["img_s"]], target_s=[x[e] for x in sup["target_s"]], mask_s=mask[e],
                       skeletons=[skels[i] for i in e], slots=c["new_slots"])
        prepared.append(eng.prepare_episode_call(qry["img_q"][c["queries"]], c["slot_of_query"], new))
    keys = ("output_kpts", "initial_proposals", "similarity_map", "adj", "attn_adj", "out_points")
    cache = SupportCache(eng, 4)
    ref = []
    for p in prepared:
        o = eng.forward_episodes(cache, prepared=p)
        torch.cuda.synchronize()
        ref.append({k: o[k].cpu() for k in keys})
    # a plain pairwise batch between the pipelined calls (every other entry point waits for a pending head)
    pb = synth.make_pairs(bs, S, H, seed=77, fixed_n_kp=False)
    pmask = pb["target_weight_s"][0].copy()
    for tw in pb["target_weight_s"]:
        pmask = pmask * tw
    pskel = [m["sample_skeleton"][0] for m in pb["img_metas"]]
    plain_ref = eng.forward(pb["img_q"], pb["img_s"], pb["target_s"], pmask, pskel)
    torch.cuda.synchronize()
    cache = SupportCache(eng, 4)
    copy_stream = torch.cuda.Stream()
    sets = {}
    got, events, plain_got = [], [], []
    for i, p in enumerate(prepared):
        key = (p["bs"], i & 1)
        if key not in sets:
            sets[key] = eng._outputs(p["bs"])
        if i >= 2:
            events[i - 2].synchronize()             # the set's previous results (call i - 2 or earlier) have been copied out
        eng.forward_episodes(cache, prepared=p, outputs=sets[key], pipelined=True)
        copy_stream.wait_stream(torch.cuda.current_stream())
        eng.pipeline_flush(copy_stream)
        with torch.cuda.stream(copy_stream):
            got.append({k: sets[key][0][k].to("cpu", non_blocking=True) for k in keys})
            ev = torch.cuda.Event()
            ev.record(copy_stream)
            events.append(ev)
        if i in (2, 6):
            plain_got.append(eng.forward(pb["img_q"], pb["img_s"], pb["target_s"], pmask, pskel))
    eng.pipeline_flush()
    torch.cuda.synchronize()
    for i in range(len(prepared)):
        for k in keys:
            assert torch.isfinite(got[i][k]).all(), (i, k)
            assert torch.equal(got[i][k], ref[i][k]), (i, k, float((got[i][k] - ref[i][k]).abs().max()))
    for o in plain_got:
        for k in keys:
            assert torch.equal(o[k], plain_ref[k]), k
