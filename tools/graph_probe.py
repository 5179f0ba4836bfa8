#!/usr/bin/env python
"""Probe (tools/, not product code): what would a HIP graph buy the reference's call contract?  One plain ec_forward call (cfg2, fp16 / mixed, resident
inputs: backbone on the caller's stream, the head's lanes forked off it and joined back by events) is captured into a graph through torch's stream
capture (relaxed mode: the library's helper streams join the capture through its own event dependencies) and replayed; timed against the same calls
issued normally, outputs compared bit for bit.  DESIGN.md section 10 said "expected: little; unmeasured" - this measures it.
    python tools/graph_probe.py [--precision fp16 --head-precision mixed --steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from edgecape_amd import synth
from edgecape_amd.engine import HipEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--head-precision", default="mixed")
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    arch, H, bs, S = "dinov2_vitb14", 256, 32, 1
    sd = synth.make_weights(arch, seed=0)
    eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision=a.precision, head_precision=a.head_precision)
    b = synth.make_pairs(bs, S, H, seed=1000, fixed_n_kp=False)
    mask = b["target_weight_s"][0].copy()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    iq, is_, ts, ms = dev(b["img_q"]), [dev(x) for x in b["img_s"]], [dev(x) for x in b["target_s"]], dev(mask.reshape(bs, -1))
    edges, off = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
    outs = eng._outputs(bs)
    st = torch.cuda.Stream()
    keys = ("output_kpts", "similarity_map", "adj")

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.cuda.stream(st):
        call = lambda: eng.forward_resident(iq, is_, ts, ms, edges, off, outs)
        ms_plain = timed(call, a.steps)
        ref = {k: outs[0][k].clone() for k in keys}
        print(f"ec_forward issued normally: {ms_plain:.3f} ms per call = {bs / ms_plain * 1e3:.0f} pairs/s", flush=True)
        g = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
                call()
        except Exception as e:  # noqa: BLE001
            print("capture failed:", repr(e)[:600])
            return
        for k in keys:
            outs[0][k].zero_()
        ms_graph = timed(g.replay, a.steps)
        same = all(torch.equal(outs[0][k], ref[k]) for k in keys)
        print(f"the same call replayed from a captured graph: {ms_graph:.3f} ms per call = {bs / ms_graph * 1e3:.0f} pairs/s ({ms_graph / ms_plain - 1:+.1%}); outputs bit-equal: {same}")


if __name__ == "__main__":
    main()
