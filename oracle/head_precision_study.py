"""CPU ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Which parts of the head tolerate single-pass fp16 MFMA operands?  The Linear / 1x1-conv GEMMs of the selected head parts run with
both operands rounded to fp16 (fp32 accumulation), everything else exact; the backbone is the fp16 emulation of precision_study.py
(the product's default).  Error of `output_kpts` against the all-fp32 oracle, argmax flips of the proposal generator.

    python oracle/head_precision_study.py [--pairs 32] [--parts skeleton,decoder,encoder,proposal,input]
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
from oracle import precision_study as ps  # noqa: E402


class FShim:
    """torch.nn.functional with fp16-rounded GEMM operands while `active`."""
    def __init__(self):
        self.active = False

    def __getattr__(self, name):
        return getattr(F, name)

    def _r(self, x):
        return x.half().float() if self.active else x

    def linear(self, x, w, b=None):
        return F.linear(self._r(x), self._r(w), b)

    def conv2d(self, x, w, b=None, **kw):
        return F.conv2d(self._r(x), self._r(w), b, **kw)

    def conv1d(self, x, w, b=None, **kw):
        return F.conv1d(self._r(x), self._r(w), b, **kw)


def install(parts):
    shim = FShim()
    orc.F = shim

    def wrap(name):
        fn = getattr(orc, name)

        def w(*a, **k):
            prev = shim.active
            shim.active = True
            try:
                return fn(*a, **k)
            finally:
                shim.active = prev
        setattr(orc, name, w)
    table = {"skeleton": "skeleton_head", "decoder": "decoder", "encoder": "encoder", "proposal": "proposal_generator"}
    for p in parts:
        if p in table:
            wrap(table[p])
    return shim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--parts", default="skeleton,decoder")
    ap.add_argument("--backbone", default="fp16")
    args = ap.parse_args()
    arch, size = "dinov2_vitb14", 256
    sd = synth.make_weights(arch, seed=0)
    heads = synth.ARCHS[arch]["heads"]
    parts = [p for p in args.parts.split(",") if p]
    errs, errs0, flips, flips0 = [], [], 0, 0
    t0 = time.time()
    real = {n: getattr(orc, n) for n in ("skeleton_head", "decoder", "encoder", "proposal_generator")}
    for c0 in range(0, args.pairs, args.chunk):
        n = min(args.chunk, args.pairs - c0)
        batch = synth.make_pairs(n, 1, size, seed=1000, first_index=c0, fixed_n_kp=False)
        valid = batch["target_weight_s"][0][:, :, 0] > 0
        orc.F = F
        for k, v in real.items():
            setattr(orc, k, v)
        ref = ps.run(sd, batch, heads, ps.rounder("fp32"))
        base = ps.run(sd, batch, heads, ps.rounder(args.backbone))
        install(parts)
        got = ps.run(sd, batch, heads, ps.rounder(args.backbone))
        am_ref = ref["similarity_map"].reshape(n, 100, -1).argmax(-1).numpy()
        for out, e, tag in ((base, errs0, 0), (got, errs, 1)):
            e.append((out["output_kpts"] - ref["output_kpts"]).abs().numpy()[:, valid].reshape(-1))
            am = out["similarity_map"].reshape(n, 100, -1).argmax(-1).numpy()
            f = int((am != am_ref)[valid].sum())
            if tag:
                flips += f
            else:
                flips0 += f
        print(f"[{c0 + n}/{args.pairs}] {time.time() - t0:.0f}s", flush=True)
    for tag, e, f in (("backbone only", errs0, flips0), ("+ head parts " + ",".join(parts), errs, flips)):
        e = np.concatenate(e)
        print(f"{tag:40s} max {e.max():.3e}  p99.9 {np.quantile(e, 0.999):.3e}  p99 {np.quantile(e, 0.99):.3e}  median {np.median(e):.3e}  "
              f"frac>1e-3 {np.mean(e > 1e-3):.5f}  argmax flips {f}")


if __name__ == "__main__":
    main()
