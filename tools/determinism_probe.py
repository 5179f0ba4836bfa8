#!/usr/bin/env python
"""Run-to-run determinism of ec_forward on cfg2 batches (a race shows up as outputs that differ between two runs of the same call)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine

bs, S, H, arch = 32, 1, 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=int(os.environ.get("WSEED", 1)))
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16", head_precision="mixed")
keys = ("similarity_map", "initial_proposals", "output_kpts", "adj")
bad = 0
for bi in range(int(os.environ.get("NB", 8))):
    b = synth.make_pairs(bs, S, H, seed=1000, first_index=bi * bs, fixed_n_kp=False)
    mask = b["target_weight_s"][0]
    runs = []
    for r in range(3):
        o = eng.forward(b["img_q"], b["img_s"], b["target_s"], mask, [m["sample_skeleton"][0] for m in b["img_metas"]])
        torch.cuda.synchronize()
        runs.append({k: o[k].cpu().numpy().copy() for k in keys})
    for r in (1, 2):
        for k in keys:
            n = int((runs[0][k] != runs[r][k]).sum())
            if n:
                bad += 1
                print(f"batch {bi} run {r} {k}: {n} elements differ, max {np.abs(runs[0][k] - runs[r][k]).max():.3e}")
print("ENC_CHAIN", os.environ.get("EC_ENC_CHAIN", "1"), "nondeterministic outputs:", bad)
