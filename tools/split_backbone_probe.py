#!/usr/bin/env python
"""Would two HALF-batch backbones on two streams beat one full-batch backbone?  (LayerNorm / attention of one half beside the GEMMs of
the other.)  Two engines (own workspace), 32 images each, against one engine on 64 images; backbone only (ec_backbone)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import synth, _lib
from edgecape_amd.engine import HipEngine

H, arch, n = 256, "dinov2_vitb14", 64
sd = synth.make_weights(arch, seed=0)
e1 = HipEngine(sd, arch=arch, image_size=H, max_batch=n // 2, max_shots=1, backbone_precision="fp16", head_precision="mixed")
ea = HipEngine(sd, arch=arch, image_size=H, max_batch=n // 4, max_shots=1, backbone_precision="fp16", head_precision="mixed")
eb = HipEngine(sd, arch=arch, image_size=H, max_batch=n // 4, max_shots=1, backbone_precision="fp16", head_precision="mixed")
img = torch.randn(n, 3, H, H, device="cuda")
out = torch.empty(n, e1.HW, e1.C, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def full(reps):
    for _ in range(reps):
        _lib.check(e1.lib.ec_backbone(e1.h, img.data_ptr(), n, out.data_ptr(), _lib.EC_LAYOUT_TOKENS, _lib.current_stream()))


def halves(reps):
    for _ in range(reps):
        _lib.check(ea.lib.ec_backbone(ea.h, img[: n // 2].data_ptr(), n // 2, out[: n // 2].data_ptr(), _lib.EC_LAYOUT_TOKENS, sa.cuda_stream))
        _lib.check(eb.lib.ec_backbone(eb.h, img[n // 2:].data_ptr(), n // 2, out[n // 2:].data_ptr(), _lib.EC_LAYOUT_TOKENS, sb.cuda_stream))


for fn, name in ((full, "one 64-image backbone"), (halves, "two 32-image backbones on two streams"), (full, "one 64-image backbone"),
                 (halves, "two 32-image backbones on two streams")):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(20)
    torch.cuda.synchronize()
    print(f"{name:42s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per 64 images")
