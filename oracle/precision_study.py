"""CPU ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Operand-rounding emulation of the backbone on the CPU oracle: which 16-bit operand format keeps `output_kpts` of the
full forward_test inside the 1e-3 gate of BASELINE.json's north_star?  Every MFMA operand of the HIP backbone
(im2col patches, LN outputs, q/k/v, softmax probabilities, attention output, GELU output, all weights) and every 16-bit
branch output (proj / fc2 results before the residual add) is rounded to the format under study; accumulation, softmax
statistics, LayerNorm and the residual stream stay fp32 exactly as in the kernels.

    python oracle/precision_study.py [--pairs 32] [--arch dinov2_vitb14] [--size 256]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgecape_amd import synth  # noqa: E402
from oracle import edgecape_oracle as orc  # noqa: E402


def rounder(fmt):
    if fmt == "fp32":
        return lambda x: x
    if fmt == "bf16":
        return lambda x: x.bfloat16().float()
    if fmt == "fp16":
        return lambda x: x.half().float()
    if fmt == "bf16x2":   # hi + lo bf16: what bf16x3 keeps of an operand (the lo*lo product is dropped: ~2^-17 relative)
        def r(x):
            hi = x.bfloat16().float()
            return hi + (x - hi).bfloat16().float()
        return r
    raise ValueError(fmt)


def backbone_emul(sd, img, heads, rnd, prefix="encoder_query."):
    w = orc.W(sd, prefix)
    img = orc._t(img)
    B, _, H, _ = img.shape
    g = H // 14
    pw = w("patch_embed.proj.weight")
    C = pw.shape[0]
    x = F.conv2d(rnd(img), rnd(pw), w("patch_embed.proj.bias"), stride=14)
    x = x[:, :, :g, :g].flatten(2).transpose(1, 2)
    pos = orc.interpolate_pos_embed(w("pos_embed"), g)
    x = torch.cat([w("cls_token").expand(B, -1, -1), x], 1) + pos[None]
    hd = C // heads
    depth = 0
    while w.has(f"blocks.{depth}.norm1.weight"):
        depth += 1
    for i in range(depth):
        b = w.sub(f"blocks.{i}.")
        y = rnd(F.layer_norm(x, (C,), b("norm1.weight"), b("norm1.bias"), 1e-6))
        qkv = rnd(F.linear(y, rnd(b("attn.qkv.weight")), b("attn.qkv.bias")))
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        p = torch.exp(s - s.amax(-1, keepdim=True))
        y = (rnd(p) @ v) / p.sum(-1, keepdim=True)          # the kernel rounds the un-normalised P, sums in fp32
        y = rnd(y.transpose(1, 2).reshape(B, T, C))
        y = rnd(b("ls1.gamma") * F.linear(y, rnd(b("attn.proj.weight")), b("attn.proj.bias")))
        x = x + y
        y = rnd(F.layer_norm(x, (C,), b("norm2.weight"), b("norm2.bias"), 1e-6))
        y = rnd(F.gelu(F.linear(y, rnd(b("mlp.fc1.weight")), b("mlp.fc1.bias"))))
        y = rnd(b("ls2.gamma") * F.linear(y, rnd(b("mlp.fc2.weight")), b("mlp.fc2.bias")))
        x = x + y
    x = F.layer_norm(x, (C,), w("norm.weight"), w("norm.bias"), 1e-6)[:, 1:]
    return x.reshape(B, g, g, C).permute(0, 3, 1, 2).contiguous()


def run(sd, batch, heads, rnd):
    with torch.no_grad():
        mask_s = orc._t(batch["target_weight_s"][0])
        for tw in batch["target_weight_s"]:
            mask_s = mask_s * orc._t(tw)
        fq = backbone_emul(sd, batch["img_q"], heads, rnd)
        fs = [backbone_emul(sd, im, heads, rnd) for im in batch["img_s"]]
        skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]
        out = orc.head_forward(sd, fq, fs, batch["target_s"], mask_s, skel)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--arch", default="dinov2_vitb14")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--shots", type=int, default=1)
    ap.add_argument("--formats", default="bf16,fp16,bf16x2")
    args = ap.parse_args()
    sd = synth.make_weights(args.arch, seed=0)
    heads = synth.ARCHS[args.arch]["heads"]
    fmts = args.formats.split(",")
    errs = {f: [] for f in fmts}
    flips = {f: 0 for f in fmts}
    t0 = time.time()
    for c0 in range(0, args.pairs, args.chunk):
        n = min(args.chunk, args.pairs - c0)
        batch = synth.make_pairs(n, args.shots, args.size, seed=1000, first_index=c0, fixed_n_kp=False)
        mask = batch["target_weight_s"][0].copy()
        for tw in batch["target_weight_s"]:
            mask = mask * tw
        valid = mask[:, :, 0] > 0
        ref = run(sd, batch, heads, rounder("fp32"))
        am_ref = ref["similarity_map"].reshape(n, 100, -1).argmax(-1).numpy()
        for f in fmts:
            got = run(sd, batch, heads, rounder(f))
            d = (got["output_kpts"] - ref["output_kpts"]).abs().numpy()[:, valid]
            errs[f].append(d.reshape(-1))
            am = got["similarity_map"].reshape(n, 100, -1).argmax(-1).numpy()
            flips[f] += int((am != am_ref)[valid].sum())
        print(f"[{c0 + n}/{args.pairs}] {time.time() - t0:.0f}s", flush=True)
    for f in fmts:
        e = np.concatenate(errs[f])
        print(f"{f:7s} max {e.max():.3e}  p99.9 {np.quantile(e, 0.999):.3e}  p99 {np.quantile(e, 0.99):.3e}  median {np.median(e):.3e}  "
              f"frac>1e-3 {np.mean(e > 1e-3):.5f}  argmax flips {flips[f]}")


if __name__ == "__main__":
    main()
