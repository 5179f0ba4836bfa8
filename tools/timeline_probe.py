#!/usr/bin/env python
"""EC_TIMELINE=1 python tools/timeline_probe.py : unprofiled per-lane milestones of the head (us from the head's start)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EC_TIMELINE", "1")
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine

bs, S, H, arch = 32, int(os.environ.get("SHOTS", 1)), 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=0)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision=os.environ.get("BB", "fp16"), head_precision=os.environ.get("HEADP", "bf16x3"))
b = synth.make_pairs(bs, S, H, seed=1000, fixed_n_kp=False)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
iq = dev(b["img_q"]); is_ = [dev(x) for x in b["img_s"]]; ts = [dev(x) for x in b["target_s"]]
ms = dev(b["target_weight_s"][0].reshape(bs, -1))
edges, off = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
outs = eng._outputs(bs)
for _ in range(4):
    eng.forward_resident(iq, is_, ts, ms, edges, off, outs)
torch.cuda.synchronize()
if os.environ.get("SUPPORT_ONLY"):   # the support lanes alone (episode-cache path): how much does the query lane cost them?
    skel = [m["sample_skeleton"][0] for m in b["img_metas"]]
    cache = None
    for _ in range(3):
        cache = eng.support_encode(b["img_s"], b["target_s"], b["target_weight_s"][0].reshape(bs, -1), skel, cache)
    torch.cuda.synchronize()
