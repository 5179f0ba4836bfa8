#!/usr/bin/env python
"""Error of the head's intermediate outputs against the CPU oracle on one cfg2 batch (weight seed / first pair index from the
environment): similarity map, proposals, encoder tap.  For A/B runs of a head switch (EC_ENC_CHAIN=0 / 1 ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine
from oracle import edgecape_oracle as orc

bs, S, H, arch = 32, 1, 256, "dinov2_vitb14"
seed, first = int(os.environ.get("WSEED", 1)), int(os.environ.get("FIRST", 0))
bb = os.environ.get("BB", "fp16")
sd = synth.make_weights(arch, seed=seed)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision=bb, head_precision="mixed")
b = synth.make_pairs(bs, S, H, seed=1000, first_index=first, fixed_n_kp=False)
mask = b["target_weight_s"][0]
o = eng.forward(b["img_q"], b["img_s"], b["target_s"], mask, [m["sample_skeleton"][0] for m in b["img_metas"]])
torch.cuda.synchronize()
torch.set_num_threads(16)
res, out = orc.forward_test(sd, b, synth.ARCHS[arch]["heads"])
valid = mask[:, :, 0] > 0
sg = o["similarity_map"].cpu().numpy().reshape(bs, 100, -1)
sr = out["similarity_map"].numpy().reshape(bs, 100, -1)
d = np.abs(sg - sr)[valid]
top2 = np.sort(sr[valid], axis=-1)[:, -2:]
gap = top2[:, 1] - top2[:, 0]
flips = (sg.argmax(-1) != sr.argmax(-1)) & valid
print(f"wseed {seed} first {first} bb {bb} ENC_CHAIN={os.environ.get('EC_ENC_CHAIN', '1')}: sim max err {d.max():.3e} mean {d.mean():.3e} (scale {np.abs(sr).max():.1f}); "
      f"min top-2 gap {gap.min():.3e}; flips {int(flips.sum())}; kpt err max {np.abs(o['output_kpts'].cpu().numpy() - out['output_kpts'].numpy())[:, valid].max():.3e}")
