#!/usr/bin/env python
"""A/B: the same resident forward on the legacy default stream vs a user-created stream (does the null stream serialise the
helper streams?).  python tools/stream_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine

bs, S, H, arch = 32, 1, 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=0)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="bf16", head_precision="bf16x3")
batch = synth.make_pairs(bs, S, H, seed=1000, fixed_n_kp=False)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
iq = dev(batch["img_q"]); is_ = [dev(x) for x in batch["img_s"]]; ts = [dev(x) for x in batch["target_s"]]
ms = dev(batch["target_weight_s"][0].reshape(bs, -1))
edges, off = eng._edges([m["sample_skeleton"][0] for m in batch["img_metas"]], bs)
outs = eng._outputs(bs)


def run(n):
    for _ in range(5):
        eng.forward_resident(iq, is_, ts, ms, edges, off, outs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_resident(iq, is_, ts, ms, edges, off, outs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"default (null) stream: {run(30):.3f} ms/step")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print(f"user stream          : {run(30):.3f} ms/step")
print(f"default (null) stream: {run(30):.3f} ms/step")
