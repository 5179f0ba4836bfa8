#!/usr/bin/env python
"""EC_TIMELINE=2 python tools/pipe_timeline.py : milestones of N pipelined cfg2 steps on every lane (us from the first mark), no sync
inside the loop.  BB / BBend: backbone of a call on the caller's stream; head: the head starts (after the wait for the previous call's
decoder); Q.prop: the query lane leaves the caller's stream; S.end / Q.end: support lane / decoder of that call done."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EC_TIMELINE"] = "2"
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine

bs, S, H, arch = 32, 1, 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=0)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16", head_precision="mixed")
b = synth.make_pairs(bs, S, H, seed=1000, fixed_n_kp=False)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
iq = dev(b["img_q"]); is_ = [dev(x) for x in b["img_s"]]; ts = [dev(x) for x in b["target_s"]]
ms = dev(b["target_weight_s"][0].reshape(bs, -1))
edges, off = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
sets = [eng._outputs(bs), eng._outputs(bs)]
for rep in range(2):
    for i in range(int(os.environ.get("STEPS", 4))):
        eng.forward_pipelined(iq, is_, ts, ms, edges, off, sets[i & 1])
    eng.pipeline_flush()
    torch.cuda.synchronize()
