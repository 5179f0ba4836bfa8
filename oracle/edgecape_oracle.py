"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A PyTorch-CPU fp32 restatement of the reference's `EdgeCape.forward_test` hot path
(orhir/EdgeCape @ /root/reference).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this file; the product path (`edgecape_amd/`) never
does and fails loudly when the HIP library is missing.

Pinning: the reference ships no tests or golden vectors (F2), so this oracle is pinned
against *outputs of the reference itself run in the build container*:
  * head/skeleton/transformer: `oracle/make_golden.py` imports the real reference modules
    (`oracle/ref_stubs.py`) and stores their outputs under `tests/golden/`; `tests/test_oracle_golden.py`
    checks this file against them to 1e-5 abs.
  * backbone: `facebookresearch/dinov2` is an un-vendored, un-pinned torch.hub dependency
    (EdgeCape.py:35-36) => "parity unpinned by the reference"; this restatement follows the published
    architecture and is cross-checked against HF `transformers` Dinov2Model with
    identical weights (fixtures under tests/golden/ made by the same script).

Every function cites the reference file:line it restates.  Eval mode: all Dropout = identity.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PATCH = 14


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.float()
    return torch.from_numpy(np.ascontiguousarray(x)).float()


class W:
    """state-dict view with a key prefix; values converted to torch fp32 lazily."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return _t(self.sd[self.prefix + name])

    def sub(self, p):
        return W(self.sd, self.prefix + p)

    def has(self, name):
        return (self.prefix + name) in self.sd


# --------------------------------------------------------------------------------------
# Backbone: facebookresearch/dinov2 DinoVisionTransformer (third-party)
# call sites: EdgeCape/models/detectors/EdgeCape.py:35-36,188-189
# --------------------------------------------------------------------------------------
def interpolate_pos_embed(pos_embed, g):
    """dinov2 `interpolate_pos_encoding`: bicubic, align_corners=False, antialias=False,
    scale_factor=(g+0.1)/M per axis (interpolate_offset=0.1).  pos_embed [1,1+M*M,C] -> [1+gh*gw, C]; g: int (square) or
    (gh, gw) = (rows, columns) - upstream scales the first spatial axis of the table with x.shape[2] // 14 (it calls it `w`)."""
    gh, gw = (g, g) if isinstance(g, int) else (int(g[0]), int(g[1]))
    pos_embed = _t(pos_embed)
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    C = pos_embed.shape[-1]
    cls_pos = pos_embed[0, :1]
    if gh == M and gw == M:
        return pos_embed[0]
    patch = pos_embed[0, 1:].reshape(1, M, M, C).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=(float(gh + 0.1) / M, float(gw + 0.1) / M), mode="bicubic", antialias=False)
    assert patch.shape[-2:] == (gh, gw)
    patch = patch.permute(0, 2, 3, 1).reshape(gh * gw, C)
    return torch.cat([cls_pos, patch], 0)


def dinov2_features(sd, img, heads, prefix="encoder_query.", taps=None, pos_table=None):
    """`get_intermediate_layers(img, n=1, reshape=True)[0]` -> [B, C, g, g].

    Floor semantics for H % 14 != 0: conv stride-14 VALID gives g = H // 14.
    """
    w = W(sd, prefix)
    img = _t(img)
    B, _, H, Wd = img.shape
    gh, gw = H // PATCH, Wd // PATCH
    pw = w("patch_embed.proj.weight")
    C = pw.shape[0]
    x = F.conv2d(img, pw, w("patch_embed.proj.bias"), stride=PATCH)  # PatchEmbed.forward
    x = x[:, :, :gh, :gw].flatten(2).transpose(1, 2)  # [B, HW, C]
    pos = interpolate_pos_embed(w("pos_embed"), (gh, gw)) if pos_table is None else _t(pos_table)
    x = torch.cat([w("cls_token").expand(B, -1, -1), x], 1) + pos[None]  # prepare_tokens_with_masks
    if taps is not None:
        taps["tokens0"] = x.clone()
    hd = C // heads
    depth = 0
    while w.has(f"blocks.{depth}.norm1.weight"):
        depth += 1
    for i in range(depth):
        b = w.sub(f"blocks.{i}.")
        # Block.forward: x = x + ls1(attn(norm1(x))); x = x + ls2(mlp(norm2(x)))
        y = F.layer_norm(x, (C,), b("norm1.weight"), b("norm1.bias"), 1e-6)
        qkv = F.linear(y, b("attn.qkv.weight"), b("attn.qkv.bias"))
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)).softmax(-1)
        y = (attn @ v).transpose(1, 2).reshape(B, T, C)
        y = F.linear(y, b("attn.proj.weight"), b("attn.proj.bias"))
        x = x + b("ls1.gamma") * y
        y = F.layer_norm(x, (C,), b("norm2.weight"), b("norm2.bias"), 1e-6)
        y = F.linear(y, b("mlp.fc1.weight"), b("mlp.fc1.bias"))
        if taps is not None:   # activation statistics (tests/test_gpu_precision_modes.py: planted outliers must really be there)
            taps["hidden_absmax"] = max(float(taps.get("hidden_absmax", 0.0)), float(y.abs().max()))
        y = F.gelu(y)
        y = F.linear(y, b("mlp.fc2.weight"), b("mlp.fc2.bias"))
        x = x + b("ls2.gamma") * y
        if taps is not None:
            taps["resid_absmax"] = max(float(taps.get("resid_absmax", 0.0)), float(x.abs().max()))
            taps["resid_absmedian"] = float(x.abs().median())
        if taps is not None and i == 0:
            taps["block0"] = x.clone()
    x = F.layer_norm(x, (C,), w("norm.weight"), w("norm.bias"), 1e-6)
    x = x[:, 1:]  # drop cls
    if taps is not None:
        taps["feat_tokens"] = x.clone()
    return x.reshape(B, gh, gw, C).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# Positional encodings: EdgeCape/models/utils/positional_encoding.py
# --------------------------------------------------------------------------------------
def _dim_t(num_feats=128, temperature=10000):
    d = torch.arange(num_feats, dtype=torch.float32)
    return temperature ** (2 * (d // 2) / num_feats)


def sine_pos_embed_image(bs, g, num_feats=128, scale=2 * math.pi, eps=1e-6, gw=None):
    """positional_encoding.py:57-94 with an all-False mask -> [bs, 2*num_feats, g, gw] (gw defaults to g: square)."""
    gw = g if gw is None else gw
    not_mask = torch.ones(bs, g, gw, dtype=torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = _dim_t(num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(bs, g, gw, -1)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(bs, g, gw, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def sine_pos_embed_coords(coord, num_feats=128, scale=2 * math.pi):
    """positional_encoding.py:96-122: [bs, K, 2] normalised coords -> [bs, K, 2*num_feats]."""
    x_embed, y_embed = coord[:, :, 0] * scale, coord[:, :, 1] * scale
    dim_t = _dim_t(num_feats)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    bs, kpt, _ = pos_x.shape
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).view(bs, kpt, -1)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).view(bs, kpt, -1)
    return torch.cat((pos_y, pos_x), dim=2)


def inverse_sigmoid(x, eps=1e-3):
    """head.py:27-31 / encoder_decoder.py:14-18."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# --------------------------------------------------------------------------------------
# Attention primitives (torch.nn.MultiheadAttention eval semantics, written out)
# --------------------------------------------------------------------------------------
def _attend(q, k, v, nhead, key_padding_mask=None, bias=None):
    """q [Lq,bs,E], k [Lk,bs,E], v [Lk,bs,Ev] (already projected; q NOT yet scaled)."""
    Lq, bs, E = q.shape
    Lk = k.shape[0]
    hd = E // nhead
    hdv = v.shape[-1] // nhead
    qh = q.reshape(Lq, bs, nhead, hd).permute(1, 2, 0, 3) * hd ** -0.5
    kh = k.reshape(Lk, bs, nhead, hd).permute(1, 2, 0, 3)
    vh = v.reshape(Lk, bs, nhead, hdv).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-2, -1)  # [bs,h,Lq,Lk]
    if bias is not None:
        s = s + bias
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = s.softmax(-1)
    o = p @ vh  # [bs,h,Lq,hdv]
    return o.permute(2, 0, 1, 3).reshape(Lq, bs, nhead * hdv)


def mha_fused(w, x_q, x_k, x_v, nhead, key_padding_mask=None):
    """nn.MultiheadAttention(E, nhead) with fused in_proj (encoder_decoder.py:444,558)."""
    Wi, bi = w("in_proj_weight"), w("in_proj_bias")
    E = Wi.shape[1]
    q = F.linear(x_q, Wi[:E], bi[:E])
    k = F.linear(x_k, Wi[E:2 * E], bi[E:2 * E])
    v = F.linear(x_v, Wi[2 * E:], bi[2 * E:])
    o = _attend(q, k, v, nhead, key_padding_mask)
    return F.linear(o, w("out_proj.weight"), w("out_proj.bias"))


def mha_cross(w, x_q, x_k, x_v, nhead, key_padding_mask=None):
    """nn.MultiheadAttention(2d, nhead, vdim=d) with separate q/k/v weights
    (encoder_decoder.py:561,573)."""
    bi = w("in_proj_bias")
    E = w("q_proj_weight").shape[0]
    q = F.linear(x_q, w("q_proj_weight"), bi[:E])
    k = F.linear(x_k, w("k_proj_weight"), bi[E:2 * E])
    v = F.linear(x_v, w("v_proj_weight"), bi[2 * E:])
    o = _attend(q, k, v, nhead, key_padding_mask)
    return F.linear(o, w("out_proj.weight"), w("out_proj.bias"))


def biased_self_attn(w, x, attn_bias, key_padding_mask, nhead):
    """BiasedMultiheadAttention.forward (bias_attn.py:106-231), self-attention, bias_attn=True.
    q,k,v are all projections of `query` (:149-151); q scaled after bias add (:152);
    bias = MLP(5->12 ReLU ->8) over attn_bias.permute(1,2,3,0) (:188-191); fp32 softmax."""
    q = F.linear(x, w("q_proj.weight"), w("q_proj.bias"))
    k = F.linear(x, w("k_proj.weight"), w("k_proj.bias"))
    v = F.linear(x, w("v_proj.weight"), w("v_proj.bias"))
    b = attn_bias.permute(1, 2, 3, 0)  # [bs, K, K, hops+1]
    b = F.linear(b, w("markov_structural_mlp.0.weight"), w("markov_structural_mlp.0.bias")).relu()
    b = F.linear(b, w("markov_structural_mlp.3.weight"), w("markov_structural_mlp.3.bias"))
    b = b.permute(0, 3, 1, 2)  # [bs, h, K, K]
    o = _attend(q, k, v, nhead, key_padding_mask, bias=b)
    return F.linear(o, w("out_proj.weight"), w("out_proj.bias"))


def gcn_layer(w, x, adj):
    """GCNLayer.forward, batch_first=False (encoder_decoder.py:508-524). x [K,bs,d], adj [bs,2,K,K]."""
    x = x.permute(1, 2, 0)  # [bs, d, K]
    x = F.conv1d(x, w("conv.weight"), w("conv.bias"))
    b, kc, v = x.shape
    x = x.view(b, 2, kc // 2, v)
    x = torch.einsum("bkcv,bkwv->bcw", x, adj)
    return x.relu().permute(2, 0, 1)


def decoder_layer(w, x, mem, tgt_mask, mem_mask, pos_cat, init_pos, adj, attn_adj, nhead=8,
                  biased=False, two_way=False):
    """TransformerDecoderLayer.forward (encoder_decoder.py:584-651)."""
    HW = mem.shape[0]
    d = x.shape[-1]
    if biased:
        sa = biased_self_attn(w.sub("self_attn."), x, attn_adj, tgt_mask, nhead)
    else:
        sa = mha_fused(w.sub("self_attn."), x, x, x, nhead, tgt_mask)
    x = F.layer_norm(x + sa, (d,), w("norm1.weight"), w("norm1.bias"))
    cq = torch.cat((x, init_pos + pos_cat[HW:]), dim=-1)
    ck = torch.cat((mem, pos_cat[:HW]), dim=-1)
    ca = mha_cross(w.sub("multihead_attn."), cq, ck, mem, nhead, mem_mask)
    x = F.layer_norm(x + F.linear(ca, w("choker.weight"), w("choker.bias")), (d,), w("norm2.weight"), w("norm2.bias"))
    z = gcn_layer(w.sub("ffn1."), x, adj).relu()
    x = F.layer_norm(x + F.linear(z, w("ffn2.weight"), w("ffn2.bias")), (d,), w("norm3.weight"), w("norm3.bias"))
    if two_way:
        q = torch.cat((mem, pos_cat[:HW]), dim=-1)
        k = torch.cat((x, init_pos + pos_cat[HW:]), dim=-1)
        t4 = mha_cross(w.sub("cross_attn_image_to_token."), q, k, x, nhead, None)
        mem = F.layer_norm(mem + F.linear(t4, w("cross_attn_image_to_token_choker.weight"),
                                          w("cross_attn_image_to_token_choker.bias")),
                           (d,), w("norm4.weight"), w("norm4.bias"))
    return x, mem


# --------------------------------------------------------------------------------------
# Skeleton head: EdgeCape/models/keypoint_heads/skeleton.py
# --------------------------------------------------------------------------------------
def adj_from_edges(skeleton, K, kp_mask):
    """adj_mx_from_edges + normalize_adj (skeleton.py:171-194). kp_mask [bs,K] True = padded."""
    bs = len(skeleton)
    A = torch.zeros(bs, K, K)
    for b in range(bs):
        e = torch.tensor(skeleton[b])
        if e.dim() > 1:
            A[b, e[:, 0], e[:, 1]] = 1
            A[b, e[:, 1], e[:, 0]] = 1
    At = A.transpose(1, 2)
    cond = (At > A).float()
    U = A + At * cond - A * cond
    valid = ~kp_mask
    adj = U * valid[..., None] * valid[:, None]
    adj = torch.nan_to_num(adj / adj.sum(dim=-1, keepdim=True))
    return torch.stack((torch.diag_embed(valid).float(), adj), dim=1)


def soft_normalize_adj(adj_mx, kp_mask):
    """skeleton.py:196-205 (adj_normalization=True, gcn_norm=False, mask_res=False)."""
    valid = ~kp_mask
    adj = adj_mx * (valid[..., None] * valid[:, None])
    adj = adj / (adj.sum(dim=-1, keepdim=True) + 1e-8)
    return torch.stack((torch.diag_embed(valid).float(), adj), dim=1)


def remove_all_true(kp_mask):
    """skeleton.py:98-99 / encoder_decoder.py:359-360."""
    m = kp_mask.clone()
    m[kp_mask.logical_not().sum(dim=-1) == 0, 0] = False
    return m


def skeleton_head(w, skeleton, kp_features, image_features, kp_mask, pos_img, nhead=8, max_hop=4, taps=None):
    """SkeletonPredictor.forward with learn_skeleton=True (skeleton.py:58-161)."""
    bs, K, d = kp_features.shape
    gt_adj = adj_from_edges(skeleton, K, kp_mask)
    binary = gt_adj[:, 1] > 0
    # refine_features (:82-115)
    adj_r = soft_normalize_adj(binary.float(), kp_mask)
    imgs = [F.conv2d(f, w("image_project.weight"), w("image_project.bias")) for f in image_features]
    zero_pos = torch.zeros(K, bs, d)
    pos_tok = pos_img.flatten(2).permute(2, 0, 1)
    pos_cat = torch.cat((pos_tok, zero_pos))
    x0 = kp_features.permute(1, 0, 2)
    imgs = [f.flatten(2).permute(2, 0, 1) for f in imgs]
    m = remove_all_true(kp_mask)
    outs = []
    for mem in imgs:
        x = x0.clone()
        for i in range(3):
            x, mem = decoder_layer(w.sub(f"skeleton_predictor.{i}."), x, mem, m, None, pos_cat, zero_pos,
                                   adj_r, None, nhead, biased=False, two_way=True)
        outs.append(x.permute(1, 0, 2))
    kp = torch.mean(torch.stack(outs, 0), 0)
    if taps is not None:
        taps["skel_kp_refined"] = kp.clone()
    # predict_skeleton (:134-150)
    kn = kp / (kp.norm(dim=-1, keepdim=True) + 1e-8)
    P = torch.bmm(kn, kn.transpose(1, 2))
    P = (P + P.transpose(1, 2)) / 2
    P = P * w("zero_conv.weight").reshape(()) + w("zero_conv.bias").reshape(())
    U = (binary.float() + P).relu()
    adj = soft_normalize_adj(U, kp_mask)
    valid = ~kp_mask
    unnorm = U * valid.unsqueeze(-1) * valid.unsqueeze(-2)
    # markov_transition_matrix (:152-161)
    A = adj[:, 1]
    A = A / (A.sum(dim=-1, keepdim=True) + 1e-8)
    attn_adj = torch.stack([torch.matrix_power(A, p) for p in range(max_hop + 1)])
    return adj, attn_adj, unnorm


# --------------------------------------------------------------------------------------
# Transformer: EdgeCape/models/keypoint_heads/encoder_decoder.py
# --------------------------------------------------------------------------------------
def encoder(w, src, query, kp_mask, pos_cat, nhead=8):
    """TransformerEncoder.forward + TransformerEncoderLayer.forward (:276-310, :461-483)."""
    n, bs, d = src.shape
    x = torch.cat((src, query), 0)
    mask_cat = torch.cat((torch.zeros(bs, n, dtype=torch.bool), kp_mask), 1)
    for i in range(3):
        l = w.sub(f"layers.{i}.")
        x = x + pos_cat  # :467 — cumulative, feeds q, k AND v
        x = F.layer_norm(x + mha_fused(l.sub("self_attn."), x, x, x, nhead, mask_cat), (d,),
                         l("norm1.weight"), l("norm1.bias"))
        y = F.linear(F.linear(x, l("linear1.weight"), l("linear1.bias")).relu(), l("linear2.weight"), l("linear2.bias"))
        x = F.layer_norm(x + y, (d,), l("norm2.weight"), l("norm2.bias"))
    return x[:n], x[n:]


def proposal_generator(w, query_feat, support_feat, h, wd):
    """ProposalGenerator.forward (:49-112)."""
    _, bs, c = query_feat.shape
    qf = query_feat.transpose(0, 1)
    sf = support_feat.transpose(0, 1)
    nq = sf.shape[1]
    fs = F.linear(sf, w("support_proj.weight"), w("support_proj.bias"))
    fq = F.linear(qf, w("query_proj.weight"), w("query_proj.bias"))
    pa = torch.tanh(F.linear(F.linear(fs, w("dynamic_proj.0.weight"), w("dynamic_proj.0.bias")).relu(),
                             w("dynamic_proj.2.weight"), w("dynamic_proj.2.bias")))
    fs = (pa + 1) * fs
    sim = torch.bmm(fq, fs.transpose(1, 2)).transpose(1, 2).reshape(bs, nq, h, wd)
    gy, gx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h), torch.linspace(0.5, wd - 0.5, wd), indexing="ij")
    grid = torch.stack([gx, gy], 0).permute(1, 2, 0).reshape(1, 1, h * wd, 2)
    norm = torch.tensor([wd, h], dtype=torch.float32)[None, None]
    p = sim.flatten(2).softmax(-1)
    prop_loss = (p[..., None] * grid).sum(2) / norm
    amax = torch.argmax(sim.reshape(bs, nq, -1), dim=-1, keepdim=True)
    onehot = F.one_hot(amax, num_classes=wd * h).reshape(bs, nq, wd, h).float()
    local = F.max_pool2d(onehot, 3, 1, 1).reshape(bs, nq, wd * h, 1)
    pl = p[..., None] * local
    pl = pl / (pl.sum(dim=-2, keepdim=True) + 1e-10)
    prop = (pl * grid).sum(2) / norm
    return prop_loss, sim, prop


def token_mlp(w, x):
    """TokenDecodeMLP (head.py:34-58): 3x(Linear,GELU) + Linear 256->2."""
    for j in (0, 2, 4):
        x = F.gelu(F.linear(x, w(f"mlp.{j}.weight"), w(f"mlp.{j}.bias")))
    return F.linear(x, w("mlp.6.weight"), w("mlp.6.bias"))


def decoder(w, kpt_w, support, mem, pos_cat, kp_mask, initial_proposals, adj, attn_adj, nhead=8, taps=None, attn_bias=True):
    """TransformerDecoder.forward (:330-425).  attn_bias=False: the layers' self-attention is nn.MultiheadAttention (fused in_proj keys,
    encoder_decoder.py:551-560, 605-612: the stage-1 / stage-2 models of run.py:44-88); with BiasedMultiheadAttention and attn_adj = None
    (learn_skeleton=False) no bias is added either (bias_attn.py:188)."""
    d = support.shape[-1]
    x = support
    bi = initial_proposals
    points = [bi]
    inter = []
    m = remove_all_true(kp_mask)
    bs, HW = mem.shape[1], mem.shape[0]
    mem_mask = torch.zeros(bs, HW, dtype=torch.bool)
    for li in range(3):
        qpe = sine_pos_embed_coords(bi).transpose(0, 1)
        rp = w.sub("ref_point_head.")
        qpe = F.linear(F.gelu(F.linear(qpe, rp("layers.0.weight"), rp("layers.0.bias"))),
                       rp("layers.1.weight"), rp("layers.1.bias"))
        lw = w.sub(f"layers.{li}.")
        if attn_bias and attn_adj is not None:
            x, mem = decoder_layer(lw, x, mem, m, mem_mask, pos_cat, qpe, adj, attn_adj, nhead, biased=True, two_way=False)
        elif attn_bias:   # BiasedMultiheadAttention without a Markov stack: q/k/v_proj keys, no bias term
            x, mem = decoder_layer(_FusedView(lw), x, mem, m, mem_mask, pos_cat, qpe, adj, None, nhead, biased=False, two_way=False)
        else:
            x, mem = decoder_layer(lw, x, mem, m, mem_mask, pos_cat, qpe, adj, None, nhead, biased=False, two_way=False)
        inter.append(F.layer_norm(x, (d,), w("norm.weight"), w("norm.bias")))
        delta = token_mlp(kpt_w[li], x.transpose(0, 1))
        bi = (inverse_sigmoid(bi) + delta).sigmoid()
        points.append(bi)
    return torch.stack(inter), points


# --------------------------------------------------------------------------------------
# Head: EdgeCape/models/keypoint_heads/head.py:161-222
# --------------------------------------------------------------------------------------
class _FusedView:
    """A decoder layer's weights with self_attn.{q,k,v}_proj presented as the fused in_proj_{weight,bias} that mha_fused reads."""

    def __init__(self, w):
        self.w = w

    def __call__(self, name):
        return self.w(name)

    def sub(self, p):
        inner = self.w.sub(p)
        if p != "self_attn.":
            return inner

        def get(name):
            if name == "in_proj_weight":
                return torch.cat([inner("q_proj.weight"), inner("k_proj.weight"), inner("v_proj.weight")], 0)
            if name == "in_proj_bias":
                return torch.cat([inner("q_proj.bias"), inner("k_proj.bias"), inner("v_proj.bias")], 0)
            return inner(name)
        return get


def head_forward(sd, feature_q, feature_s, target_s, mask_s, skeleton, prefix="keypoint_head_module.", taps=None,
                 learn_skeleton=True, attn_bias=True):
    """learn_skeleton=False: SkeletonPredictor returns the normalised ground-truth adjacency and no Markov stack (skeleton.py:73-74);
    attn_bias: see decoder()."""
    w = W(sd, prefix)
    feature_q, mask_s = _t(feature_q), _t(mask_s)
    feature_s = [_t(f) for f in feature_s]
    target_s = [_t(t) for t in target_s]
    fq = F.conv2d(feature_q, w("input_proj.weight"), w("input_proj.bias"))
    bs, d, h, wd = fq.shape
    pos_img = sine_pos_embed_image(bs, h, gw=wd)
    embeds = []
    for feat, tgt in zip(feature_s, target_s):
        rf = F.interpolate(feat, size=tgt.shape[-2:], mode="bilinear", align_corners=False)
        tgt = tgt / (tgt.sum(dim=-1).sum(dim=-1)[:, :, None, None] + 1e-8)
        embeds.append(tgt.flatten(2) @ rf.flatten(2).permute(0, 2, 1))
    sk = torch.mean(torch.stack(embeds, 0), 0)
    if taps is not None:
        taps["pooled"] = sk.clone()
    sk = sk * mask_s
    sk = F.linear(sk, w("query_proj.weight"), w("query_proj.bias"))
    kp_mask = (~mask_s.to(torch.bool)).squeeze(-1)
    K = sk.shape[1]
    if taps is not None:
        taps["support_keypoints"] = sk.clone()
    if learn_skeleton:
        adj, attn_adj, unnorm = skeleton_head(w.sub("skeleton_head."), skeleton, sk, feature_s, kp_mask, pos_img, taps=taps)
    else:
        adj = adj_from_edges(skeleton, K, kp_mask)
        attn_adj, unnorm = None, adj[:, 1] > 0
    # TwoStageSupportRefineTransformer.forward (encoder_decoder.py:183-260)
    tw = w.sub("transformer.")
    src = fq.flatten(2).permute(2, 0, 1)
    pos_cat = torch.cat((pos_img.flatten(2).permute(2, 0, 1), torch.zeros(K, bs, d)))
    mem, kp = encoder(tw.sub("encoder."), src, sk.transpose(0, 1), kp_mask, pos_cat)
    if taps is not None:
        taps["enc_img"], taps["enc_kp"] = mem.clone(), kp.clone()
    prop_loss, sim, prop = proposal_generator(tw.sub("proposal_generator."), mem, kp, h, wd)
    kpt_w = [w.sub(f"kpt_branch.{i}.") for i in range(3)]
    hs, points = decoder(tw.sub("decoder."), kpt_w, kp, mem, pos_cat, kp_mask, prop, adj, attn_adj, taps=taps, attn_bias=attn_bias)
    hs = hs.transpose(1, 2)  # [3, bs, K, d]
    outs = []
    for i in range(3):
        outs.append((token_mlp(kpt_w[i], hs[i]) + inverse_sigmoid(points[i])).sigmoid())
    out = dict(output_kpts=torch.stack(outs, 0), initial_proposals=prop_loss, similarity_map=sim, adj=adj,
               attn_adj=attn_adj, unnormalized_adj=unnorm, out_points=torch.stack(points, 0), hs=hs,
               decoder_proposals=prop)
    return out


# --------------------------------------------------------------------------------------
# Detector: EdgeCape/models/detectors/EdgeCape.py:131-191 ; decode: head.py:324-387
# --------------------------------------------------------------------------------------
def transform_preds(coords, center, scale, output_size):
    """post_transforms.py:150-194 (use_udp=False)."""
    scale = scale * 200.0
    sx, sy = scale[0] / output_size[0], scale[1] / output_size[1]
    out = coords.copy()
    out[:, 0] = coords[:, 0] * sx + center[0] - scale[0] * 0.5
    out[:, 1] = coords[:, 1] * sy + center[1] - scale[1] * 0.5
    return out


def decode(img_metas, output, img_size):
    """TwoStageHead.decode (head.py:324-387). output numpy [bs,K,2] normalised."""
    bs = len(img_metas)
    Wd, H = img_size
    output = output * np.array([Wd, H])[None, None, :]
    c = np.zeros((bs, 2), np.float32)
    s = np.zeros((bs, 2), np.float32)
    score = np.ones(bs)
    paths, ids = [], []
    for i in range(bs):
        c[i] = img_metas[i]["query_center"]
        s[i] = img_metas[i]["query_scale"]
        paths.append(img_metas[i]["query_image_file"])
        if "query_bbox_score" in img_metas[i]:
            score[i] = float(np.array(img_metas[i]["query_bbox_score"]).reshape(-1)[0])
        if "bbox_id" in img_metas[i]:
            ids.append(img_metas[i]["bbox_id"])
        elif "query_bbox_id" in img_metas[i]:
            ids.append(img_metas[i]["query_bbox_id"])
    preds = np.zeros(output.shape)
    for i in range(bs):
        preds[i] = transform_preds(output[i], c[i], s[i], [Wd, H])
    all_preds = np.zeros((bs, preds.shape[1], 3), np.float32)
    all_boxes = np.zeros((bs, 6), np.float32)
    all_preds[:, :, 0:2] = preds[:, :, 0:2]
    all_preds[:, :, 2:3] = 1.0
    all_boxes[:, 0:2] = c
    all_boxes[:, 2:4] = s
    all_boxes[:, 4] = np.prod(s * 200.0, axis=1)
    all_boxes[:, 5] = score
    return dict(preds=all_preds, boxes=all_boxes, image_paths=paths, bbox_ids=ids)


def forward_test(sd, batch, heads, taps=None, pos_table=None, learn_skeleton=True, attn_bias=True):
    """EdgeCape.forward_test (EdgeCape.py:131-163) on a `synth.make_pairs`-style batch."""
    with torch.no_grad():
        img_q = _t(batch["img_q"])
        mask_s = _t(batch["target_weight_s"][0])
        for tw in batch["target_weight_s"]:
            mask_s = mask_s * _t(tw)  # EdgeCape.py:175-177 (first weight is squared; weights are 0/1)
        fq = dinov2_features(sd, img_q, heads, taps=taps, pos_table=pos_table)
        fs = [dinov2_features(sd, im, heads, pos_table=pos_table) for im in batch["img_s"]]
        skeleton = [m["sample_skeleton"][0] for m in batch["img_metas"]]
        out = head_forward(sd, fq, fs, batch["target_s"], mask_s, skeleton, taps=taps, learn_skeleton=learn_skeleton, attn_bias=attn_bias)
        out["feature_q"], out["feature_s"] = fq, fs
    H, Wd = img_q.shape[-2:]
    res = decode(batch["img_metas"], out["output_kpts"][-1].numpy(), [Wd, H])
    res["points"] = torch.cat((out["initial_proposals"][None], out["output_kpts"])).numpy()
    res["sample_image_file"] = [m["sample_image_file"] for m in batch["img_metas"]]
    res["skeleton"] = out["adj"][0].numpy()
    return res, out


def keypoint_pck(pred, gt, mask, thr, normalize):
    """mmpose 0.29 keypoint_pck_accuracy restated, N pairs at once.
    pred/gt [N,K,2] pixels, mask [N,K] bool, normalize [N,2]. Returns mean-over-pairs PCK."""
    N, K, _ = pred.shape
    pcks = []
    for n in range(N):
        nrm = normalize[n].astype(np.float64)
        nrm = np.where(nrm <= 0, 1e6, nrm)
        dist = np.linalg.norm((pred[n] - gt[n]) / nrm[None], axis=-1)
        valid = mask[n]
        if valid.sum() == 0:
            pcks.append(0.0)
        else:
            pcks.append(float((dist[valid] < thr).mean()))
    return float(np.mean(pcks)), pcks
