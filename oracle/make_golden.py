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
    meta = dict(kind="detector", arch=arch, image_size=image_size, shots=shots, input_seed=seed, weight_seed=wseed,
                bs=2, reference="orhir/EdgeCape EdgeCape.forward_test with the oracle DINOv2 as torch.hub stand-in")
    save(name, arrays, meta)


def hf_backbone_fixture(name, arch, image_size, seed):
    """HF transformers Dinov2Model with identical weights; native grid == g so no interpolation on
    the HF side; the table fed to both sides is the upstream-convention one."""
    from transformers import Dinov2Config, Dinov2Model
    wseed = 13
    a = synth.ARCHS[arch]
    C, depth, heads = a["C"], a["depth"], a["heads"]
    sd = synth.make_backbone_weights(arch, seed=wseed, prefix="")
    g = image_size // 14
    pos = orc.interpolate_pos_embed(sd["pos_embed"], g)
    cfg = Dinov2Config(hidden_size=C, num_hidden_layers=depth, num_attention_heads=heads, image_size=g * 14,
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
    rng = np.random.default_rng(seed)
    img = np.stack([synth._smooth_image(rng, image_size)])
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


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # HF first: the torchvision stub installed for the reference import confuses transformers' import probes
    hf_backbone_fixture("bb_hf_vits14_224", "dinov2_vits14", 224, 301)
    hf_backbone_fixture("bb_hf_vitb14_256", "dinov2_vitb14", 256, 302)
    ns = ref_stubs.install()
    for case in HEAD_CASES:
        head_fixture(ns, *case)
    detector_fixture("det_vits14_224_s1", "dinov2_vits14", 224, 1, 201)
    detector_fixture("det_vits14_224_s5", "dinov2_vits14", 224, 5, 202)
    leaked = [os.path.join(d, x) for d, ds, _ in os.walk(ref_stubs.REF_ROOT) for x in ds if x == "__pycache__"]
    assert not leaked, f"bytecode leaked into the reference tree: {leaked}"


if __name__ == "__main__":
    main()
