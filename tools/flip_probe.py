#!/usr/bin/env python
"""Anatomy of the proposal-argmax flips of a mode on the conformance data of one weight seed: per flipped keypoint the oracle's top-2
gap and the HIP path's values at the two candidates."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_precision_modes as T
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine
from oracle import edgecape_oracle as orc

c = T.CFG["cfg2"]
ws = int(os.environ.get("WSEED", 1))
torch.set_num_threads(32)
w = synth.make_weights(c["arch"], seed=ws)
eng = HipEngine(w, arch=c["arch"], image_size=c["H"], max_batch=c["bs"], max_shots=c["S"], backbone_precision="fp16", head_precision="mixed")
nf = 0
for b in range(int(os.environ.get("NB", 8))):
    batch = synth.make_pairs(c["bs"], c["S"], c["H"], seed=c["iseed"] + 17 * ws + b, fixed_n_kp=False)
    mask = batch["target_weight_s"][0].copy()
    _, out = orc.forward_test(w, batch, synth.ARCHS[c["arch"]]["heads"])
    o = eng.forward(batch["img_q"], batch["img_s"], batch["target_s"], mask, [m["sample_skeleton"][0] for m in batch["img_metas"]])
    torch.cuda.synchronize()
    sg = o["similarity_map"].cpu().numpy().reshape(c["bs"], 100, -1)
    sr = out["similarity_map"].numpy().reshape(c["bs"], 100, -1)
    valid = mask[:, :, 0] > 0
    ag, ar = sg.argmax(-1), sr.argmax(-1)
    for s, k in zip(*np.nonzero((ag != ar) & valid)):
        nf += 1
        r = np.sort(sr[s, k])[::-1]
        print(f"batch {b} sample {s} kp {k}: oracle best {ar[s,k]} ({sr[s,k,ar[s,k]]:.5f}) second gap {r[0]-r[1]:.2e}; oracle at hip's {ag[s,k]}: {sr[s,k,ag[s,k]]:.5f}; "
              f"hip at oracle's best {sg[s,k,ar[s,k]]:.5f} at its own {sg[s,k,ag[s,k]]:.5f}; n identical rows in this sample: "
              f"{int((np.abs(sr[s] - sr[s, k]).max(-1) < 1e-6).sum())}")
print("flips", nf)
