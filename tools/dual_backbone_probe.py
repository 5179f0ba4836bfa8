#!/usr/bin/env python
"""Two FULL-batch backbones (64 images each, the cfg2 step's) on two streams against the same work on one stream: what would running
the backbones of consecutive pipelined calls side by side buy (LayerNorm - HBM-bound - and attention - VALU-bound - of one beside the
GEMMs of the other; the GEMM workgroups hold 480 of a SIMD's 512 registers, so co-residency is what limits it)?
    python tools/dual_backbone_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import synth, _lib
from edgecape_amd.engine import HipEngine

H, arch, n = 256, "dinov2_vitb14", 64
sd = synth.make_weights(arch, seed=0)
ea = HipEngine(sd, arch=arch, image_size=H, max_batch=n // 2, max_shots=1, backbone_precision="fp16", head_precision="mixed")
eb = HipEngine(sd, arch=arch, image_size=H, max_batch=n // 2, max_shots=1, backbone_precision="fp16", head_precision="mixed")
img = torch.randn(2, n, 3, H, H, device="cuda")
out = torch.empty(2, n, ea.HW, ea.C, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def bb(e, i, stream):
    _lib.check(e.lib.ec_backbone(e.h, img[i].data_ptr(), n, out[i].data_ptr(), _lib.EC_LAYOUT_TOKENS, stream))


def serial(reps):
    for _ in range(reps):
        bb(ea, 0, sa.cuda_stream)
        bb(ea, 1, sa.cuda_stream)


def side_by_side(reps):
    for _ in range(reps):
        bb(ea, 0, sa.cuda_stream)
        bb(eb, 1, sb.cuda_stream)


for fn, name in ((serial, "two backbones, one stream"), (side_by_side, "two backbones, two streams"), (serial, "two backbones, one stream"),
                 (side_by_side, "two backbones, two streams")):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(15)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 15
    print(f"{name}: {dt * 1e3:.3f} ms per pair of backbones ({2 * n / dt:.0f} images/s)", flush=True)
