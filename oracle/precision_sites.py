"""CPU ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Which rounding SITE of the fp16 backbone carries the error that flips proposal argmaxes?  The emulation of precision_study.py with
the fp16 rounding switched on at ONE site at a time (everything else fp32): relative error of the backbone features and the mean /
max error of the similarity map (scale ~45) on a few cfg2 pairs.

    python oracle/precision_sites.py [--pairs 8]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgecape_amd import synth  # noqa: E402
from oracle import edgecape_oracle as orc  # noqa: E402

SITES = ["patch", "ln1", "w_qkv", "q", "k", "v", "p", "att", "w_proj", "y1", "ln2", "w_fc1", "hid", "w_fc2", "y2"]


def backbone(sd, img, heads, on, prefix="encoder_query."):
    r = lambda x, s: x.half().float() if (s in on) else x
    w = orc.W(sd, prefix)
    img = orc._t(img)
    B, _, H, _ = img.shape
    g = H // 14
    pw = w("patch_embed.proj.weight")
    C = pw.shape[0]
    x = F.conv2d(r(img, "patch"), r(pw, "patch"), w("patch_embed.proj.bias"), stride=14)
    x = x[:, :, :g, :g].flatten(2).transpose(1, 2)
    pos = orc.interpolate_pos_embed(w("pos_embed"), g)
    x = torch.cat([w("cls_token").expand(B, -1, -1), x], 1) + pos[None]
    hd = C // heads
    depth = 0
    while w.has(f"blocks.{depth}.norm1.weight"):
        depth += 1
    for i in range(depth):
        b = w.sub(f"blocks.{i}.")
        y = r(F.layer_norm(x, (C,), b("norm1.weight"), b("norm1.bias"), 1e-6), "ln1")
        qkv = F.linear(y, r(b("attn.qkv.weight"), "w_qkv"), b("attn.qkv.bias"))
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = r(qkv[0], "q"), r(qkv[1], "k"), r(qkv[2], "v")
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        p = torch.exp(s - s.amax(-1, keepdim=True))
        y = (r(p, "p") @ v) / p.sum(-1, keepdim=True)
        y = r(y.transpose(1, 2).reshape(B, T, C), "att")
        y = r(b("ls1.gamma") * F.linear(y, r(b("attn.proj.weight"), "w_proj"), b("attn.proj.bias")), "y1")
        x = x + y
        y = r(F.layer_norm(x, (C,), b("norm2.weight"), b("norm2.bias"), 1e-6), "ln2")
        y = r(F.gelu(F.linear(y, r(b("mlp.fc1.weight"), "w_fc1"), b("mlp.fc1.bias"))), "hid")
        y = r(b("ls2.gamma") * F.linear(y, r(b("mlp.fc2.weight"), "w_fc2"), b("mlp.fc2.bias")), "y2")
        x = x + y
    x = F.layer_norm(x, (C,), w("norm.weight"), w("norm.bias"), 1e-6)[:, 1:]
    return x.reshape(B, g, g, C).permute(0, 3, 1, 2).contiguous()


def run(sd, batch, heads, on):
    with torch.no_grad():
        mask_s = orc._t(batch["target_weight_s"][0])
        fq = backbone(sd, batch["img_q"], heads, on)
        fs = [backbone(sd, im, heads, on) for im in batch["img_s"]]
        skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]
        return fq, orc.head_forward(sd, fq, fs, batch["target_s"], mask_s, skel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--arch", default="dinov2_vitb14")
    ap.add_argument("--wseed", type=int, default=0)
    args = ap.parse_args()
    sd = synth.make_weights(args.arch, seed=args.wseed)
    heads = synth.ARCHS[args.arch]["heads"]
    batch = synth.make_pairs(args.pairs, 1, 256, seed=1000, fixed_n_kp=False)
    valid = batch["target_weight_s"][0][:, :, 0] > 0
    f0, o0 = run(sd, batch, heads, set())
    s0 = o0["similarity_map"].reshape(args.pairs, 100, -1)[valid]
    rows = []
    for name, on in [(s, {s}) for s in SITES] + [("all", set(SITES)), ("all but q,k", set(SITES) - {"q", "k"}),
                                                  ("all but weights", set(SITES) - {"w_qkv", "w_proj", "w_fc1", "w_fc2", "patch"}),
                                                  ("all but y1,y2", set(SITES) - {"y1", "y2"}), ("all but hid", set(SITES) - {"hid"})]:
        f, o = run(sd, batch, heads, on)
        ds = (o["similarity_map"].reshape(args.pairs, 100, -1)[valid] - s0).abs()
        rows.append((name, float((f - f0).norm() / f0.norm()), float(ds.mean()), float(ds.max())))
        print(f"{name:16s} feature rel err {rows[-1][1]:.3e}   sim err mean {rows[-1][2]:.3e} max {rows[-1][3]:.3e}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
