"""CPU ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

How many significand bits do the CORRECTION terms of a split-precision product need?  The conforming mode (bf16x3) computes every
backbone product as  a_hi W_hi + a_lo W_hi + a_hi W_lo  with three 16-bit MFMAs.  The two correction terms are
~2^-9 (bf16 hi) / ~2^-12 (fp16 hi) of the product, so a few bits of THEIR operands already put the total below the bf16x3 error - and
gfx950 has block-scaled FP8 MFMAs at twice the 16-bit rate (v_mfma_scale_f32_16x16x128_f8f6f4: E8M0 scale per 32 K-elements of each
operand, which is what the small magnitude of the lo parts needs).  A product would then cost 1 + 2 * 0.5 = 2 units instead of 3.

This script EMULATES such schemes in the four Linear layers of every backbone block (everything else fp32, head fp32) on a few cfg2
pairs and reports, against the fp32 oracle: relative error of the backbone features, mean / max error of the similarity map (scale
~45: the quantity whose near-ties flip the proposal argmax, encoder_decoder.py:91-110) and the argmax flips on these pairs.  The
similarity-map error is the continuous proxy of the flip RATE (12 of 20 293 at the fp16 error level, 0-1 at the bf16x3 level:
profiles/r05_conformance_*.json).  Nothing here runs in the product; it prices a lead before any kernel is written.

    python oracle/correction_terms_study.py [--pairs 8] [--arch dinov2_vitb14]

Second use (round 5): a given batch of the at-scale conformance sets (tests/test_gpu_precision_modes.py conformance_at_scale: its
configuration, weight seed and batch index) under the same schemes, with one keypoint singled out - does the emulated scheme flip the
near-tie that the GPU mode flips?
    python oracle/correction_terms_study.py --config cfg4 --wseed 1 --batch-index 14 --kpt 9,23 --schemes bf16x3,fp16x3,fp16+e4m3
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

F8 = {"e4m3": (torch.float8_e4m3fn, 8, 448.0), "e5m2": (torch.float8_e5m2, 15, 57344.0)}   # dtype, exponent of the largest binade, largest finite


def q_mx(x, fmt, block=32):
    """Block-scaled FP8 along the last (K) axis, as the hardware's MX operands: one power-of-two scale per `block` elements that puts
    the block's largest magnitude into the format's top binade, elements rounded to nearest even; returned de-quantised (exact in fp32)."""
    dt, emax, fmax = F8[fmt]
    K = x.shape[-1]
    pad = (-K) % block
    xp = F.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*xp.shape[:-1], -1, block)
    amax = xb.abs().amax(-1, keepdim=True)
    scale = torch.exp2((torch.floor(torch.log2(amax.clamp_min(2.0 ** -100))) - emax).clamp_min(-126.0))   # (an all-zero block keeps a finite scale)
    y = ((xb / scale).clamp(-fmax, fmax).to(dt).float() * scale).reshape(xp.shape)   # (OCP MX: magnitudes above the largest finite saturate)
    return y[..., :K] if pad else y


def hi_lo(x, hi):
    h = x.half().float() if hi == "fp16" else x.bfloat16().float()
    return h, x - h


def linear(a, w, b, scheme):
    """scheme: fp32 | fp16 | bf16x3 | fp16x3 | '<hi>+<fmt>' (hi = fp16 / bf16 main product, corrections in block-scaled <fmt>) |
    '<hi>+<fmt>:w' (only the weight side of the corrections in FP8) | '<hi>+none' (no correction terms: hi x hi only)"""
    if scheme == "fp32":
        return F.linear(a, w, b)
    if scheme == "fp16":
        return F.linear(a.half().float(), w.half().float(), b)
    if scheme in ("bf16x3", "fp16x3"):
        hi = "bf16" if scheme == "bf16x3" else "fp16"
        ah, al = hi_lo(a, hi)
        wh, wl = hi_lo(w, hi)
        al = hi_lo(al, hi)[0]
        wl = hi_lo(wl, hi)[0]
        return F.linear(ah, wh) + F.linear(al, wh) + F.linear(ah, wl) + b
    hi, fmt = scheme.split("+")
    w_only = fmt.endswith(":w")
    fmt = fmt[:-2] if w_only else fmt
    ah, al = hi_lo(a, hi)
    wh, wl = hi_lo(w, hi)
    y = F.linear(ah, wh)
    if fmt != "none":
        qa = (lambda t: hi_lo(t, hi)[0]) if w_only else (lambda t: q_mx(t, fmt))
        y = y + F.linear(qa(al), q_mx(wh, fmt)) + F.linear(qa(ah), q_mx(wl, fmt))
    return y + b


def backbone(sd, img, heads, scheme, prefix="encoder_query."):
    """facebookresearch/dinov2 forward as oracle/precision_sites.py restates it; the four Linear layers of a block under `scheme`,
    attention in fp32 except for scheme 'fp16' (q, k, v, p rounded to fp16 as the headline mode does)."""
    r = (lambda x: x.half().float()) if scheme == "fp16" else (lambda x: x)
    w = orc.W(sd, prefix)
    img = orc._t(img)
    B, _, H, _ = img.shape
    g = H // 14
    pw = w("patch_embed.proj.weight")
    C = pw.shape[0]
    x = F.conv2d(img, pw, w("patch_embed.proj.bias"), stride=14)          # (split precision in every fast mode)
    x = x[:, :, :g, :g].flatten(2).transpose(1, 2)
    pos = orc.interpolate_pos_embed(w("pos_embed"), g)
    x = torch.cat([w("cls_token").expand(B, -1, -1), x], 1) + pos[None]
    hd = C // heads
    depth = 0
    while w.has(f"blocks.{depth}.norm1.weight"):
        depth += 1
    for i in range(depth):
        b = w.sub(f"blocks.{i}.")
        y = F.layer_norm(x, (C,), b("norm1.weight"), b("norm1.bias"), 1e-6)
        qkv = linear(y, b("attn.qkv.weight"), b("attn.qkv.bias"), scheme)
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = r(qkv[0]), r(qkv[1]), r(qkv[2])
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        p = torch.exp(s - s.amax(-1, keepdim=True))
        y = (r(p) @ v) / p.sum(-1, keepdim=True)
        y = r(y.transpose(1, 2).reshape(B, T, C))
        x = x + r(b("ls1.gamma") * linear(y, b("attn.proj.weight"), b("attn.proj.bias"), scheme))
        y = F.layer_norm(x, (C,), b("norm2.weight"), b("norm2.bias"), 1e-6)
        y = r(F.gelu(linear(y, b("mlp.fc1.weight"), b("mlp.fc1.bias"), scheme)))
        x = x + r(b("ls2.gamma") * linear(y, b("mlp.fc2.weight"), b("mlp.fc2.bias"), scheme))
    x = F.layer_norm(x, (C,), w("norm.weight"), w("norm.bias"), 1e-6)[:, 1:]
    return x.reshape(B, g, g, C).permute(0, 3, 1, 2).contiguous()


def run(sd, batch, heads, scheme):
    with torch.no_grad():
        mask_s = orc._t(batch["target_weight_s"][0])
        fq = backbone(sd, batch["img_q"], heads, scheme)
        fs = [backbone(sd, im, heads, scheme) for im in batch["img_s"]]
        skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]
        return fq, orc.head_forward(sd, fq, fs, batch["target_s"], mask_s, skel)


def one_conformance_batch(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_precision_modes as T
    c = T.CFG[args.config]
    sd = synth.make_weights(c["arch"], seed=args.wseed)
    heads = synth.ARCHS[c["arch"]]["heads"]
    batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"] + 100000 * (1 + args.wseed), first_index=args.batch_index * c["bs"], fixed_n_kp=False)
    mask = batch["target_weight_s"][0].copy()
    for tw in batch["target_weight_s"]:
        mask = mask * tw
    valid = mask[:, :, 0] > 0
    skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]

    def sim(scheme):
        with torch.no_grad():
            fq = backbone(sd, batch["img_q"], heads, scheme)
            fs = [backbone(sd, im, heads, scheme) for im in batch["img_s"]]
            return orc.head_forward(sd, fq, fs, batch["target_s"], orc._t(mask), skel)["similarity_map"].reshape(c["bs"], 100, -1).numpy()

    so = orc.forward_test(sd, batch, heads)[1]["similarity_map"].reshape(c["bs"], 100, -1).numpy()
    s0 = sim("fp32")
    gap = lambda m: np.where(valid, np.sort(m, -1)[:, :, -1] - np.sort(m, -1)[:, :, -2], np.inf)
    i = tuple(int(x) for x in args.kpt.split(",")) if args.kpt else tuple(int(x) for x in np.unravel_index(np.argmin(gap(so)), valid.shape))
    print(f"{args.config}, weight seed {args.wseed}, batch {args.batch_index}: {int(valid.sum())} valid keypoints; keypoint {i}: top-2 gap {gap(so)[i]:.3e} in the oracle's map "
          f"(argmax {int(so[i].argmax())}), {gap(s0)[i]:.3e} in this script's fp32 backbone (argmax {int(s0[i].argmax())})")
    for scheme in args.schemes.split(","):
        s = sim(scheme)
        print(f"{scheme:12s} flips in the batch {int(((s.argmax(-1) != s0.argmax(-1)) & valid).sum())}   keypoint {i}: argmax {int(s[i].argmax())} "
              f"{'FLIPPED' if s[i].argmax() != s0[i].argmax() else 'kept'}, map err at it {float(np.abs(s[i] - s0[i]).max()):.2e}   "
              f"batch map err mean {float(np.abs(s - s0)[valid].mean()):.2e} max {float(np.abs(s - s0)[valid].max()):.2e}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--arch", default="dinov2_vitb14")
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--outliers", action="store_true", help="weights with planted DINOv2-like activation outliers")
    ap.add_argument("--schemes", default="fp16,bf16x3,fp16x3,fp16+none,fp16+e4m3,fp16+e5m2,fp16+e4m3:w,bf16+e4m3")
    ap.add_argument("--config", default=None, help="cfg1 | cfg2 | cfg4 | cfg5: one batch of the at-scale conformance set instead of --pairs")
    ap.add_argument("--batch-index", type=int, default=0)
    ap.add_argument("--kpt", default=None, help="sample,keypoint to single out")
    args = ap.parse_args()
    if args.config:
        return one_conformance_batch(args)
    sd = synth.make_weights(args.arch, seed=args.wseed)
    if args.outliers:
        sd = synth.add_activation_outliers(sd, args.arch)
    heads = synth.ARCHS[args.arch]["heads"]
    H = 224 if args.arch == "dinov2_vits14" else 256
    batch = synth.make_pairs(args.pairs, 1, H, seed=1000, fixed_n_kp=False)
    valid = batch["target_weight_s"][0][:, :, 0] > 0
    f0, o0 = run(sd, batch, heads, "fp32")
    s0 = o0["similarity_map"].reshape(args.pairs, 100, -1)[valid]
    k0 = o0["output_kpts"][-1][valid] if o0["output_kpts"].dim() == 4 else o0["output_kpts"][valid]
    print(f"{args.arch}, {args.pairs} pairs, {int(valid.sum())} valid keypoints, weight seed {args.wseed}{', planted outliers' if args.outliers else ''}")
    print("MFMA units per product: fp16 1 | bf16x3 / fp16x3 3 | <hi>+none 1 | <hi>+<fp8> 2 (FP8 MFMAs at twice the 16-bit rate)")
    for scheme in args.schemes.split(","):
        f, o = run(sd, batch, heads, scheme)
        s = o["similarity_map"].reshape(args.pairs, 100, -1)[valid]
        ds = (s - s0).abs()
        flips = int((s.argmax(-1) != s0.argmax(-1)).sum())
        k = o["output_kpts"][-1][valid] if o["output_kpts"].dim() == 4 else o["output_kpts"][valid]
        print(f"{scheme:14s} feature rel err {float((f - f0).norm() / f0.norm()):.3e}   sim err mean {float(ds.mean()):.3e} max {float(ds.max()):.3e}"
              f"   argmax flips {flips}   max |d kpt| {float((k - k0).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
