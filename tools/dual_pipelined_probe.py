#!/usr/bin/env python
"""What would TWO backbone lanes inside ec_forward_pipelined buy?  Emulation with two engines (own weights / workspaces) driven from one
host thread: pipelined calls alternate between the engines, each on its own caller stream, so that the backbone of call i + 1 runs beside
the backbone of call i (and beside the deferred head of call i - 1) - against ONE engine running the same calls.
    python tools/dual_pipelined_probe.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import synth
from edgecape_amd.engine import HipEngine

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S, H, arch = 1, 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=0)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
b = synth.make_pairs(bs, S, H, seed=1000, fixed_n_kp=False)
iq = dev(b["img_q"]); is_ = [dev(x) for x in b["img_s"]]; ts = [dev(x) for x in b["target_s"]]
ms = dev(b["target_weight_s"][0].reshape(bs, -1))
engs = [HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="fp16", head_precision="mixed") for _ in range(2)]
edges, off = engs[0]._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
outs = [[e._outputs(bs), e._outputs(bs)] for e in engs]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one(n):
    with torch.cuda.stream(streams[0]):
        for i in range(n):
            engs[0].forward_pipelined(iq, is_, ts, ms, edges, off, outs[0][i & 1])
        engs[0].pipeline_flush()
    torch.cuda.synchronize()


def two(n):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            engs[k].forward_pipelined(iq, is_, ts, ms, edges, off, outs[k][(i >> 1) & 1])
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            engs[k].pipeline_flush()
    torch.cuda.synchronize()


n = 40
for fn, name in ((one, "one engine, pipelined"), (two, "two engines alternating, pipelined"), (one, "one engine, pipelined"),
                 (two, "two engines alternating, pipelined"), (one, "one engine, pipelined"), (two, "two engines alternating, pipelined")):
    fn(6)
    t0 = time.perf_counter()
    fn(n)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.3f} ms per step ({bs / dt:.0f} pairs/s)", flush=True)
