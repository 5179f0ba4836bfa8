"""Seeded synthetic weights and (support, query) pairs for the EdgeCape hot path.

No MP-100 data and no released checkpoints are reachable offline (reference README.md:87-96),
so parity tests and bench.py run on inputs regenerated from seeds.  Weight
tensors carry the *reference's* state-dict names (SURVEY Appendix B for the head,
`keypoint_head_module.*`; upstream facebookresearch/dinov2 names for the backbone,
`encoder_query.*`), so the same dict loads into the real reference head (oracle/make_golden.py),
into the CPU oracle and into the HIP library.

Everything here is host-side numpy plumbing; nothing is timed.
"""
import math

import numpy as np

# torch.hub entry -> (width C, depth, heads); facebookresearch/dinov2 hub/backbones.py
ARCHS = {
    "dinov2_vits14": dict(C=384, depth=12, heads=6),
    "dinov2_vitb14": dict(C=768, depth=12, heads=12),
    "dinov2_vitl14": dict(C=1024, depth=24, heads=16),
}
PATCH = 14
POS_GRID = 37  # 518 / 14

# COCO-17 skeleton, 0-based (MP-100 'person' category)
COCO17_EDGES = [(15, 13), (13, 11), (16, 14), (14, 12), (11, 12), (5, 11), (6, 12), (5, 6), (5, 7),
                (6, 8), (7, 9), (8, 10), (1, 2), (0, 1), (0, 2), (1, 3), (2, 4), (3, 5), (4, 6)]


def _xavier(rng, shape, gain=1.0):
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def _bias(rng, n, s=0.02):
    return (rng.standard_normal(n) * s).astype(np.float32)


def _ln(rng, n, prefix, out):
    out[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    out[prefix + ".bias"] = _bias(rng, n, 0.05)


def make_backbone_weights(arch="dinov2_vits14", seed=0, prefix="encoder_query."):
    """Random-init DINOv2 ViT weights under upstream state-dict names."""
    a = ARCHS[arch]
    C, depth = a["C"], a["depth"]
    rng = np.random.default_rng(seed)
    w = {}
    w["cls_token"] = (rng.standard_normal((1, 1, C)) * 0.02).astype(np.float32)
    w["pos_embed"] = (rng.standard_normal((1, 1 + POS_GRID * POS_GRID, C)) * 0.02).astype(np.float32)
    w["mask_token"] = np.zeros((1, C), np.float32)
    w["patch_embed.proj.weight"] = _xavier(rng, (C, 3, PATCH, PATCH))
    w["patch_embed.proj.bias"] = _bias(rng, C)
    for i in range(depth):
        p = f"blocks.{i}."
        _ln(rng, C, p + "norm1", w)
        w[p + "attn.qkv.weight"] = _xavier(rng, (3 * C, C))
        w[p + "attn.qkv.bias"] = _bias(rng, 3 * C)
        w[p + "attn.proj.weight"] = _xavier(rng, (C, C))
        w[p + "attn.proj.bias"] = _bias(rng, C)
        w[p + "ls1.gamma"] = rng.uniform(0.2, 0.6, C).astype(np.float32)
        _ln(rng, C, p + "norm2", w)
        w[p + "mlp.fc1.weight"] = _xavier(rng, (4 * C, C))
        w[p + "mlp.fc1.bias"] = _bias(rng, 4 * C)
        w[p + "mlp.fc2.weight"] = _xavier(rng, (C, 4 * C))
        w[p + "mlp.fc2.bias"] = _bias(rng, C)
        w[p + "ls2.gamma"] = rng.uniform(0.2, 0.6, C).astype(np.float32)
    _ln(rng, C, "norm", w)
    return {prefix + k: v for k, v in w.items()}


def _mha_fused(rng, p, E, out):
    out[p + "in_proj_weight"] = _xavier(rng, (3 * E, E))
    out[p + "in_proj_bias"] = _bias(rng, 3 * E)
    out[p + "out_proj.weight"] = _xavier(rng, (E, E))
    out[p + "out_proj.bias"] = _bias(rng, E)


def _mha_cross(rng, p, d, out):
    # nn.MultiheadAttention(2d, nhead, vdim=d): encoder_decoder.py:561,573
    E = 2 * d
    out[p + "q_proj_weight"] = _xavier(rng, (E, E))
    out[p + "k_proj_weight"] = _xavier(rng, (E, E))
    out[p + "v_proj_weight"] = _xavier(rng, (E, d))
    out[p + "in_proj_bias"] = _bias(rng, 3 * E)
    out[p + "out_proj.weight"] = _xavier(rng, (E, E))
    out[p + "out_proj.bias"] = _bias(rng, E)


def _decoder_layer(rng, p, d, F, out, biased, two_way, max_hops=4, nhead=8):
    if biased:  # BiasedMultiheadAttention: bias_attn.py:66-83
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"] = _xavier(rng, (d, d))
            out[p + f"self_attn.{n}.bias"] = _bias(rng, d)
        h = max_hops + nhead
        out[p + "self_attn.markov_structural_mlp.0.weight"] = _xavier(rng, (h, max_hops + 1), gain=2.0)
        out[p + "self_attn.markov_structural_mlp.0.bias"] = _bias(rng, h, 0.1)
        out[p + "self_attn.markov_structural_mlp.3.weight"] = _xavier(rng, (nhead, h), gain=2.0)
        out[p + "self_attn.markov_structural_mlp.3.bias"] = _bias(rng, nhead, 0.1)
    else:
        _mha_fused(rng, p + "self_attn.", d, out)
    _mha_cross(rng, p + "multihead_attn.", d, out)
    out[p + "choker.weight"] = _xavier(rng, (d, 2 * d))
    out[p + "choker.bias"] = _bias(rng, d)
    out[p + "ffn1.conv.weight"] = _xavier(rng, (2 * F, d, 1))
    out[p + "ffn1.conv.bias"] = _bias(rng, 2 * F)
    out[p + "ffn2.weight"] = _xavier(rng, (d, F))
    out[p + "ffn2.bias"] = _bias(rng, d)
    for n in ("norm1", "norm2", "norm3"):
        _ln(rng, d, p + n, out)
    if two_way:
        _mha_cross(rng, p + "cross_attn_image_to_token.", d, out)
        out[p + "cross_attn_image_to_token_choker.weight"] = _xavier(rng, (d, 2 * d))
        out[p + "cross_attn_image_to_token_choker.bias"] = _bias(rng, d)
        _ln(rng, d, p + "norm4", out)


def make_head_weights(C=384, F_s=None, seed=1, d=256, F_d=384, prefix="keypoint_head_module."):
    """Random head weights under the reference's names.

    `kpt_branch.*.mlp.6` and `skeleton_head.zero_conv` are zero at construction in the reference
    (head.py:151-153,158-159); they are made non-zero here so the refinement and the predicted-
    adjacency paths are exercised.  C = backbone width, F_s = skeleton GCN width (= C).
    """
    F_s = C if F_s is None else F_s
    rng = np.random.default_rng(seed)
    w = {}
    w["transformer.mask_token"] = np.zeros((1, d), np.float32)
    for i in range(3):
        p = f"transformer.encoder.layers.{i}."
        _mha_fused(rng, p + "self_attn.", d, w)
        w[p + "linear1.weight"] = _xavier(rng, (F_d, d))
        w[p + "linear1.bias"] = _bias(rng, F_d)
        w[p + "linear2.weight"] = _xavier(rng, (d, F_d))
        w[p + "linear2.bias"] = _bias(rng, d)
        _ln(rng, d, p + "norm1", w)
        _ln(rng, d, p + "norm2", w)
    for i in range(3):
        _decoder_layer(rng, f"transformer.decoder.layers.{i}.", d, F_d, w, biased=True, two_way=False)
    _ln(rng, d, "transformer.decoder.norm", w)
    for i in range(2):
        w[f"transformer.decoder.ref_point_head.layers.{i}.weight"] = _xavier(rng, (d, d))
        w[f"transformer.decoder.ref_point_head.layers.{i}.bias"] = _bias(rng, d)
    pg = "transformer.proposal_generator."
    w[pg + "support_proj.weight"] = _xavier(rng, (d, d))
    w[pg + "support_proj.bias"] = _bias(rng, d)
    w[pg + "query_proj.weight"] = _xavier(rng, (d, d))
    w[pg + "query_proj.bias"] = _bias(rng, d)
    w[pg + "dynamic_proj.0.weight"] = _xavier(rng, (128, d))
    w[pg + "dynamic_proj.0.bias"] = _bias(rng, 128)
    w[pg + "dynamic_proj.2.weight"] = _xavier(rng, (d, 128))
    w[pg + "dynamic_proj.2.bias"] = _bias(rng, d)
    w["input_proj.weight"] = _xavier(rng, (d, C, 1, 1))
    w["input_proj.bias"] = _bias(rng, d)
    w["query_proj.weight"] = _xavier(rng, (d, C))
    w["query_proj.bias"] = _bias(rng, d)
    for i in range(3):
        for j in (0, 2, 4):
            w[f"kpt_branch.{i}.mlp.{j}.weight"] = _xavier(rng, (d, d))
            w[f"kpt_branch.{i}.mlp.{j}.bias"] = _bias(rng, d)
        w[f"kpt_branch.{i}.mlp.6.weight"] = (rng.standard_normal((2, d)) * 0.02).astype(np.float32)
        w[f"kpt_branch.{i}.mlp.6.bias"] = _bias(rng, 2)
    for i in range(3):
        _decoder_layer(rng, f"skeleton_head.skeleton_predictor.{i}.", d, F_s, w, biased=False, two_way=True)
    w["skeleton_head.image_project.weight"] = _xavier(rng, (d, F_s, 1, 1))
    w["skeleton_head.image_project.bias"] = _bias(rng, d)
    for n in ("k_proj", "q_proj"):  # unused in forward (skeleton.py:49-50)
        w[f"skeleton_head.{n}.weight"] = _xavier(rng, (d, d))
        w[f"skeleton_head.{n}.bias"] = _bias(rng, d)
    w["skeleton_head.mh_linear.weight"] = _xavier(rng, (1, 8, 1, 1))
    w["skeleton_head.mh_linear.bias"] = _bias(rng, 1)
    w["skeleton_head.zero_conv.weight"] = np.full((1, 1, 1, 1), 0.5, np.float32)
    w["skeleton_head.zero_conv.bias"] = np.full((1,), -0.1, np.float32)
    return {prefix + k: v for k, v in w.items()}


def make_weights(arch="dinov2_vits14", seed=0, outliers=False):
    C = ARCHS[arch]["C"]
    sd = make_backbone_weights(arch, seed)
    sd.update(make_head_weights(C=C, seed=seed + 1))
    if outliers:
        add_activation_outliers(sd, arch, seed)
    return sd


def add_activation_outliers(sd, arch, seed=0, prefix="encoder_query."):
    """Give random-init backbone weights the activation statistics released DINOv2 checkpoints are known for (the checkpoints themselves
    are unreachable offline, README.md:96 of the reference), so that the 16-bit modes are exercised on them at MODEL level:
      * "massive activations": from block 2 on, four channels of the fp32 residual stream carry values ~ 100-200 x the typical ones
        (a handful of fc2 rows scaled up), which every later LayerNorm then has to normalise against - the other channels of a token
        come out an order of magnitude smaller than usual;
      * outlier neurons in the MLP hidden layer: a few fc1 rows of every block scaled so that their pre-activations reach the
        hundreds (fp16 range 65504: far inside; what is stressed is the uniform RELATIVE rounding across five orders of magnitude);
      * a few LayerNorm gains of 10-15 (outlier channels of the normalised operand of the QKV / fc1 GEMMs).
    Returns the dict (modified in place)."""
    a = ARCHS[arch]
    C, depth = a["C"], a["depth"]
    rng = np.random.default_rng(seed + 7777)
    big = rng.choice(C, 4, replace=False)
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        if i == 2:
            sd[p + "mlp.fc2.weight"][big] *= 150.0
            sd[p + "ls2.gamma"][big] = 1.0
        hot = rng.choice(4 * C, 6, replace=False)
        sd[p + "mlp.fc1.weight"][hot] *= 40.0
        sd[p + "mlp.fc2.weight"][:, hot] *= 0.05          # their contribution to the output stays ordinary
        g = rng.choice(C, 3, replace=False)
        sd[p + "norm1.weight"][g] *= rng.uniform(10.0, 15.0, 3).astype(np.float32)
        sd[p + "norm2.weight"][g] *= rng.uniform(10.0, 15.0, 3).astype(np.float32)
    return sd


def msra_target(joints_xy, visible, image_size, heatmap_size=64, sigma=1):
    """Synthetic-input helper: MSRA gaussian heatmaps [K, hm, hm] + weights [K, 1] of the keypoints (the kind of target the
    reference's pipeline feeds the model; oracle/pipeline_oracle.py is the line-by-line restatement the tests pin to the
    reference).  Vectorised: every heatmap cell looks its value up in the (2*3*sigma+1)^2 float32 patch centred on the cell
    int(x / stride + 0.5) of its keypoint; cells outside the patch, invisible keypoints and keypoints whose whole patch lies
    outside the map are zero."""
    from .preprocess import gaussian_7x7
    joints_xy = np.asarray(joints_xy)
    K, hm, r = len(joints_xy), heatmap_size, 3 * sigma
    patch = gaussian_7x7(sigma)
    if isinstance(image_size, (tuple, list)):    # (H, W): feat_stride = image_size / heatmap_size per axis (x: W, y: H)
        stride = np.array([image_size[1] / hm, image_size[0] / hm], np.float32)
    else:
        stride = np.float32(image_size / hm)
    mu = (joints_xy[:, :2] / stride + 0.5).astype(np.int64)                      # int(): truncation toward zero
    inside = (mu[:, 0] - r < hm) & (mu[:, 1] - r < hm) & (mu[:, 0] + r + 1 >= 0) & (mu[:, 1] + r + 1 >= 0)
    weight = (np.asarray(visible, np.float32).reshape(K) * inside).astype(np.float32)
    cells = np.arange(hm)
    dx = cells[None, :] - mu[:, 0:1] + r                                         # patch column of every map column, per keypoint
    dy = cells[None, :] - mu[:, 1:2] + r
    okx, oky = (dx >= 0) & (dx <= 2 * r), (dy >= 0) & (dy <= 2 * r)
    vals = patch[np.clip(dy, 0, 2 * r)[:, :, None], np.clip(dx, 0, 2 * r)[:, None, :]]
    target = np.where(oky[:, :, None] & okx[:, None, :] & (weight > 0.5)[:, None, None], vals, np.float32(0)).astype(np.float32)
    return target, weight[:, None]


def random_skeleton(rng, n_kp):
    """Random spanning tree + n_kp//4 extra edges, 0-based."""
    if n_kp <= 1:
        return []
    edges = [(int(rng.integers(0, i)), i) for i in range(1, n_kp)]
    for _ in range(n_kp // 4):
        a, b = rng.choice(n_kp, 2, replace=False)
        edges.append((int(a), int(b)))
    return edges


def _smooth_image(rng, H, W=None):
    """N(0,1) pixels plus a low-frequency field so backbone features are not degenerate."""
    W = H if W is None else W
    img = rng.standard_normal((3, H, W)).astype(np.float32) * 0.5
    yy, xx = np.meshgrid(np.linspace(0, 1, H, dtype=np.float32), np.linspace(0, 1, W, dtype=np.float32), indexing="ij")
    for c in range(3):
        for _ in range(4):
            fx, fy = rng.uniform(0.5, 6.0, 2)
            ph = rng.uniform(0, 2 * np.pi)
            img[c] += (0.6 * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)).astype(np.float32)
    return img


def make_pairs(bs, shots=1, image_size=224, seed=0, n_kp=17, K=100, fixed_n_kp=True,
               skeleton="auto", first_index=0):
    """Synthetic batch in the reference's batch-dict layout (test_base_dataset.py:157-184).

    Pair i is seeded by `seed + first_index + i`, so any rank can regenerate its shard
    (SURVEY §8d/§8e).  Returns dict(img_s, target_s, target_weight_s, img_q, target_q,
    target_weight_q, img_metas) of numpy arrays / python lists, plus 'gt_q' [bs,K,2] pixel
    keypoints for a PCK figure.
    """
    # image_size: an int (square) or (H, W); the square path draws exactly the random numbers it always did
    square = not isinstance(image_size, (tuple, list))
    H, Wd = (image_size, image_size) if square else (int(image_size[0]), int(image_size[1]))
    size = H if square else (H, Wd)
    hi = H if square else np.array([Wd, H], np.float32)          # per-coordinate (x, y) upper bounds
    img_q = np.zeros((bs, 3, H, Wd), np.float32)
    img_s = [np.zeros((bs, 3, H, Wd), np.float32) for _ in range(shots)]
    target_s = [np.zeros((bs, K, 64, 64), np.float32) for _ in range(shots)]
    tw_s = [np.zeros((bs, K, 1), np.float32) for _ in range(shots)]
    target_q = np.zeros((bs, K, 64, 64), np.float32)
    tw_q = np.zeros((bs, K, 1), np.float32)
    gt_q = np.zeros((bs, K, 2), np.float32)
    metas = []
    for i in range(bs):
        rng = np.random.default_rng(seed + first_index + i)
        nk = n_kp if fixed_n_kp else int(rng.integers(8, 69))
        nk = min(nk, K)
        vis = np.zeros(K, np.float32)
        vis[:nk] = 1
        img_q[i] = _smooth_image(rng, H, Wd)
        base = rng.uniform(8, hi - 8, size=(K, 2)).astype(np.float32)
        kps = []
        for s in range(shots):
            img_s[s][i] = _smooth_image(rng, H, Wd)
            kp = np.clip(base + rng.normal(0, 2.0, base.shape), 0, hi - 1).astype(np.float32)
            kps.append(kp)
            t, tw = msra_target(kp, vis, size)
            target_s[s][i], tw_s[s][i] = t, tw
        gq = np.clip(base + rng.normal(0, 4.0, base.shape), 0, hi - 1).astype(np.float32)
        gt_q[i] = gq
        target_q[i], tw_q[i] = msra_target(gq, vis, size)
        if skeleton == "auto":
            edges = list(COCO17_EDGES) if nk == 17 else random_skeleton(rng, nk)
        elif skeleton == "empty":
            edges = []
        else:
            edges = list(skeleton)
        metas.append({
            "sample_skeleton": [edges for _ in range(shots)],
            "query_skeleton": edges,
            "query_center": np.array([Wd / 2, H / 2], np.float32),
            "query_scale": np.array([Wd / 200 * 1.25, H / 200 * 1.25], np.float32),
            "query_image_file": f"synthetic/q{seed + first_index + i}.png",
            "sample_image_file": [f"synthetic/s{seed + first_index + i}_{s}.png" for s in range(shots)],
            # the support annotations as the reference's Collect hands them over (test_base_dataset.py:171-184): what the detector's
            # episode cache recognises a support set by (nothing above is derived from these: no random number is drawn for them)
            "sample_center": [np.array([Wd / 2, H / 2], np.float32) for _ in range(shots)],
            "sample_scale": [np.array([Wd / 200 * 1.25, H / 200 * 1.25], np.float32) for _ in range(shots)],
            "sample_rotation": [0 for _ in range(shots)],
            "sample_joints_3d": [np.concatenate([kp, np.zeros((K, 1), np.float32)], 1) for kp in kps],
            "sample_joints_3d_visible": [np.repeat(vis[:, None], 3, 1) for _ in range(shots)],
            "sample_bbox_id": [first_index + i for _ in range(shots)],
            "query_bbox_score": 1.0,
            "bbox_id": first_index + i,
            "query_bbox": np.array([0, 0, Wd, H], np.float32),
        })
    return dict(img_s=img_s, target_s=target_s, target_weight_s=tw_s, img_q=img_q, target_q=target_q,
                target_weight_q=tw_q, img_metas=metas, gt_q=gt_q)


def make_head_inputs(bs, shots, C, g, seed, n_kps, skeletons="auto", K=100, image_size=None):
    """Seeded inputs for the *head alone* (feature maps instead of images): used by the golden
    fixtures, because the reference backbone cannot be imported.

    n_kps: list of valid-keypoint counts per sample.  The support descriptor of each valid keypoint
    is planted at a random query cell so similarity maps are peaky (SURVEY §7 "Discontinuities").
    Returns dict(feature_q [bs,C,g,g], feature_s list[shots], target_s, mask_s [bs,K,1], skeleton list).
    """
    # g: an int (square grid) or (gh, gw); the square path draws exactly the random numbers it always did
    square = not isinstance(g, (tuple, list))
    gh, gw = (g, g) if square else (int(g[0]), int(g[1]))
    if square:
        image_size = image_size or g * PATCH
        hi = image_size
        cell = np.float32(image_size)
    else:
        image_size = (gh * PATCH, gw * PATCH)                       # (H, W)
        hi = np.array([gw * PATCH, gh * PATCH], np.float32)         # per-coordinate (x, y) upper bounds
    rng = np.random.default_rng(seed)
    feature_q = rng.standard_normal((bs, C, gh, gw)).astype(np.float32)
    feature_s = [rng.standard_normal((bs, C, gh, gw)).astype(np.float32) for _ in range(shots)]
    target_s = [np.zeros((bs, K, 64, 64), np.float32) for _ in range(shots)]
    mask_s = np.zeros((bs, K, 1), np.float32)
    skel = []
    for i in range(bs):
        nk = n_kps[i]
        vis = np.zeros(K, np.float32)
        vis[:nk] = 1
        base = rng.uniform(8, hi - 8, size=(K, 2)).astype(np.float32)
        m = np.ones((K, 1), np.float32)
        for s in range(shots):
            kp = np.clip(base + rng.normal(0, 2.0, base.shape), 0, hi - 1).astype(np.float32)
            t, tw = msra_target(kp, vis, image_size)
            target_s[s][i] = t
            m = m * tw
        mask_s[i] = m
        for k in range(nk):
            if square:
                cx = min(int(base[k, 0] / image_size * g), g - 1)
                cy = min(int(base[k, 1] / image_size * g), g - 1)
                qy, qx = rng.integers(0, g, 2)
            else:
                cx = min(int(base[k, 0] / hi[0] * gw), gw - 1)
                cy = min(int(base[k, 1] / hi[1] * gh), gh - 1)
                qy, qx = int(rng.integers(0, gh)), int(rng.integers(0, gw))
            feature_q[i, :, qy, qx] += 1.5 * feature_s[0][i, :, cy, cx]
        if skeletons == "auto":
            skel.append(list(COCO17_EDGES) if nk == 17 else random_skeleton(rng, nk))
        elif skeletons == "empty":
            skel.append([])
        else:
            skel.append(list(skeletons[i]))
    return dict(feature_q=feature_q, feature_s=feature_s, target_s=target_s, mask_s=mask_s, skeleton=skel)


def as_stage_checkpoint(sd):
    """The keys a checkpoint of an EARLIER training stage of the reference carries (run.py:44-88: decoder self-attention =
    nn.MultiheadAttention): every decoder self_attn.{q,k,v}_proj fused into in_proj_{weight,bias}, no markov_structural_mlp."""
    out = {k: v for k, v in sd.items() if "markov_structural_mlp" not in k}
    for k in list(out):
        if ".transformer.decoder.layers." in k and k.endswith("self_attn.q_proj.weight"):
            base = k[:-len("q_proj.weight")]
            for kind, fused in (("weight", "in_proj_weight"), ("bias", "in_proj_bias")):
                out[base + fused] = np.concatenate([out.pop(base + f"{n}_proj.{kind}") for n in "qkv"], 0)
    return out
