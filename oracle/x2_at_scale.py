"""CPU ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

AT-SCALE emulation of candidate 2-MFMA-unit conforming backbones ("extend the emulation to the 512-pair scale of one
config so the expected flip count is a number, not a proportionality argument"), run BEFORE any kernel was written.

The same disjoint pairs and weight seeds as tools/conformance.py / tests/test_gpu_precision_modes.py::conformance_at_scale; per batch
  reference   oracle.forward_test (fp32)                                   -> similarity-map argmax, output_kpts
  scheme X    this file's backbone with every block Linear under X, head = oracle.head_forward (fp32)
and per scheme: proposal-argmax flips among valid keypoints against the oracle, max |d kpt| on the flip-free samples, similarity-map
error.  The GPU mode adds a bf16x3 head on top (its own error is what sets the continuous part) - this script prices the
BACKBONE scheme alone, exactly as profiles/r05_correction_terms_study.txt did for 8 pairs.

Schemes (oracle/correction_terms_study.py has the others):
  fp16x2     a W ~ a_hi W_hi  [fp16 x fp16 MFMA]  +  q5(a_lo) q4(W_hi) + q5(a_hi) q4(W_lo)  [one FP8 MFMA pass of depth 2K at twice the rate]
             activations e5m2 with FIXED exponents (a_lo * 2^11 has the exponent range of a itself, e5m2 has fp16's exponent range: no
             data-dependent scale, no reduction in any producer, cannot overflow while a fits fp16), weights e4m3 with one static
             power-of-two scale per tensor and plane (known at ec_finalize).  2 MFMA units per product.
  fp16x2e4   the same with e4m3 activations under a per-row dynamic scale (what a block-scaled producer could do at best)
  fp16x25    a_hi W_hi + q5(a_lo) q4(W_hi)  [FP8]  + a_hi W_lo16 [fp16 MFMA]: 2.5 units
  fp16x15e2m3 / fp16x15e3m2   (round 6, a LEAD, nothing built) both correction terms in MX FP6 under block-32 scales: 1.5 units
    python oracle/x2_at_scale.py --config cfg2 --batches 8 --seeds 0,1 --schemes bf16x3,fp16x2 --out profiles/r06_x2_emulation_cfg2.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from edgecape_amd import synth  # noqa: E402
from oracle import edgecape_oracle as orc  # noqa: E402
from oracle import correction_terms_study as cts  # noqa: E402

E4MAX, E5MAX = 448.0, 57344.0


def q5(x):
    """e5m2, round to nearest even, saturating at the largest finite value"""
    return x.clamp(-E5MAX, E5MAX).to(torch.float8_e5m2).float()


def q4(x):
    return x.clamp(-E4MAX, E4MAX).to(torch.float8_e4m3fn).float()


def w_planes(w):
    """static per-tensor power-of-two scales: the plane's largest magnitude lands in e4m3's top binade"""
    wh = w.half().float()
    wl = w - wh
    out = []
    for t in (w, wl):
        amax = float(t.abs().max())
        e = int(np.floor(np.log2(E4MAX / amax))) if amax > 0 else 0
        out.append(q4(t * 2.0 ** e) * 2.0 ** -e)
    return wh, out[0], out[1], wl


def q6(x, fmt, block=32):
    """MX FP6 along the last (K) axis: one power-of-two scale per `block` elements that puts the block's largest magnitude into the format's top
    binade, elements rounded to nearest even on the format's grid, saturating.
    e2m3: 1 + 2 + 3 bits, bias 1, values {0, 0.125 .. 0.875 (subnormal), 1 .. 7.5}; e3m2: 1 + 3 + 2 bits, bias 3, max 28.  De-quantised (exact in fp32)."""
    mant, emax, fmax, emin = {"e2m3": (3, 2, 7.5, 0), "e3m2": (2, 4, 28.0, -2)}[fmt]
    K = x.shape[-1]
    pad = (-K) % block
    xp = F.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*xp.shape[:-1], -1, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -100)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    v = (xb / scale).clamp(-fmax, fmax)
    e = torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -40))).clamp_min(float(emin))      # binade of the element (subnormals share the lowest one)
    step = torch.exp2(e - mant)
    y = (torch.round(v / step) * step).clamp(-fmax, fmax) * scale                          # (round half to even: torch.round)
    y = y.reshape(xp.shape)
    return y[..., :K] if pad else y


_wcache = {}


def linear_x2(a, w, b, scheme, key):
    if key not in _wcache:
        _wcache[key] = w_planes(w)
    wh, wh8, wl8, wl = _wcache[key]
    ah = a.half().float()
    al = a - ah
    y = F.linear(ah, wh)
    if scheme == "fp16x2":
        y = y + F.linear(q5(al * 2048.0) * (1.0 / 2048.0), wh8) + F.linear(q5(a), wl8)
    elif scheme == "fp16x2e4":
        def rowq(t):
            amax = t.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -100)
            s = torch.exp2(torch.floor(torch.log2(E4MAX / amax)))
            return q4(t * s) / s
        y = y + F.linear(rowq(al), wh8) + F.linear(rowq(a), wl8)
    elif scheme in ("fp16x15e2m3", "fp16x15e3m2"):
        # the lead of DESIGN section 10: both correction terms in MX FP6 (block-32 scales on activations AND weights) - the f8f6f4 MFMA runs FP6
        # at twice the FP8 rate: 1.5 units per product
        f = scheme[-4:]
        y = y + F.linear(q6(al, f), q6(w, f)) + F.linear(q6(a, f), q6(wl, f))
    elif scheme == "fp16x25":
        y = y + F.linear(q5(al * 2048.0) * (1.0 / 2048.0), wh8) + F.linear(ah, wl.half().float())
    else:
        raise ValueError(scheme)
    return y + b


def backbone(sd, img, heads, scheme, wkey, prefix="encoder_query."):
    if scheme in ("fp32", "fp16", "bf16x3", "fp16x3") or "+" in scheme:
        return cts.backbone(sd, img, heads, scheme, prefix)
    w = orc.W(sd, prefix)
    img = orc._t(img)
    B, _, H, _ = img.shape
    g = H // 14
    pw = w("patch_embed.proj.weight")
    C = pw.shape[0]
    x = F.conv2d(img, pw, w("patch_embed.proj.bias"), stride=14)
    x = x[:, :, :g, :g].flatten(2).transpose(1, 2)
    pos = orc.interpolate_pos_embed(w("pos_embed"), g)
    x = torch.cat([w("cls_token").expand(B, -1, -1), x], 1) + pos[None]
    hd = C // heads
    depth = 0
    while w.has(f"blocks.{depth}.norm1.weight"):
        depth += 1
    for i in range(depth):
        b = w.sub(f"blocks.{i}.")
        lin = lambda t, n: linear_x2(t, b(n + ".weight"), b(n + ".bias"), scheme, (wkey, i, n))
        y = F.layer_norm(x, (C,), b("norm1.weight"), b("norm1.bias"), 1e-6)
        qkv = lin(y, "attn.qkv")
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        y = ((q @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, T, C)
        x = x + b("ls1.gamma") * lin(y, "attn.proj")
        y = F.layer_norm(x, (C,), b("norm2.weight"), b("norm2.bias"), 1e-6)
        y = F.gelu(lin(y, "mlp.fc1"))
        x = x + b("ls2.gamma") * lin(y, "mlp.fc2")
    x = F.layer_norm(x, (C,), w("norm.weight"), w("norm.bias"), 1e-6)[:, 1:]
    return x.reshape(B, g, g, C).permute(0, 3, 1, 2).contiguous()


def main():
    import test_gpu_precision_modes as T
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--seeds", default="0,1")
    ap.add_argument("--schemes", default="bf16x3,fp16x2")
    ap.add_argument("--outliers", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    c = T.CFG[args.config]
    heads = synth.ARCHS[c["arch"]]["heads"]
    schemes = args.schemes.split(",")
    acc = {s: dict(flips=0, flipped=[], max_clean=0.0, max_all=0.0, map_err_sum=0.0, map_err_max=0.0, clean_samples=0) for s in schemes}
    n_valid = pairs = 0
    t0 = time.time()
    for ws in (int(x) for x in args.seeds.split(",")):
        sd = synth.make_weights(c["arch"], seed=ws, outliers=args.outliers)
        for bi in range(args.batches):
            batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"] + 100000 * (1 + ws), first_index=bi * c["bs"], fixed_n_kp=False)
            mask = batch["target_weight_s"][0].copy()
            for tw in batch["target_weight_s"]:
                mask = mask * tw
            valid = mask[:, :, 0] > 0
            skel = [m["sample_skeleton"][0] for m in batch["img_metas"]]
            with torch.no_grad():
                ref = orc.forward_test(sd, batch, heads)[1]
                rs = ref["similarity_map"].reshape(c["bs"], valid.shape[1], -1).numpy()
                rk = ref["output_kpts"].numpy()
                n_valid += int(valid.sum())
                pairs += c["bs"]
                for s in schemes:
                    imgs = torch.cat([orc._t(batch["img_q"])] + [orc._t(im) for im in batch["img_s"]], 0)
                    f = backbone(sd, imgs, heads, s, (args.config, ws, args.outliers))
                    fq, fs = f[:c["bs"]], [f[c["bs"] * (1 + j):c["bs"] * (2 + j)] for j in range(c["S"])]
                    o = orc.head_forward(sd, fq, fs, batch["target_s"], orc._t(mask), skel)
                    ss = o["similarity_map"].reshape(c["bs"], valid.shape[1], -1).numpy()
                    flip = (ss.argmax(-1) != rs.argmax(-1)) & valid
                    d = np.abs(o["output_kpts"].numpy() - rk)
                    clean = ~flip.any(1)
                    a = acc[s]
                    a["flips"] += int(flip.sum())
                    a["flipped"] += [(ws, bi, int(i), int(k), float(np.sort(rs[i, k])[-1] - np.sort(rs[i, k])[-2])) for i, k in zip(*np.nonzero(flip))]
                    a["clean_samples"] += int(clean.sum())
                    if clean.any():
                        a["max_clean"] = max(a["max_clean"], float(d[:, clean][:, valid[clean]].max()))
                    a["max_all"] = max(a["max_all"], float(d[:, valid].max()))
                    e = np.abs(ss - rs)[valid]
                    a["map_err_sum"] += float(e.mean())
                    a["map_err_max"] = max(a["map_err_max"], float(e.max()))
            print(f"[{time.time() - t0:6.0f}s] seed {ws} batch {bi}: " + "  ".join(f"{s}: {acc[s]['flips']} flips, clean max {acc[s]['max_clean']:.2e}" for s in schemes), flush=True)
    nb = args.batches * len(args.seeds.split(","))
    rec = dict(what="CPU emulation of the backbone's block Linears under each scheme, head fp32 (oracle.head_forward), against oracle.forward_test",
               config=args.config, outliers=args.outliers, pairs=pairs, n_valid=n_valid,
               schemes={s: dict(flips=a["flips"], flipped_kpts_ws_batch_sample_kpt_gap=a["flipped"], clean_samples=a["clean_samples"], max_clean=a["max_clean"],
                                max_all=a["max_all"], map_err_mean=a["map_err_sum"] / nb, map_err_max=a["map_err_max"]) for s, a in acc.items()})
    print(json.dumps(rec, indent=1))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
