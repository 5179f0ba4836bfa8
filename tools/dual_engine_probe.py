#!/usr/bin/env python
"""Aggregate throughput of two independent engines (own weights/workspace, own stream, own host thread) on ONE GPU vs one engine:
how much of the head's latency-bound time can hide under another batch's backbone?"""
import os
import sys
import threading
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


class Worker:
    def __init__(self, seed):
        self.eng = HipEngine(sd, arch=arch, image_size=H, max_batch=bs, max_shots=S, backbone_precision="bf16", head_precision="bf16x3")
        b = synth.make_pairs(bs, S, H, seed=seed, fixed_n_kp=False)
        self.iq = dev(b["img_q"]); self.is_ = [dev(x) for x in b["img_s"]]; self.ts = [dev(x) for x in b["target_s"]]
        self.ms = dev(b["target_weight_s"][0].reshape(bs, -1))
        self.edges, self.off = self.eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], bs)
        self.outs = self.eng._outputs(bs)
        self.stream = torch.cuda.Stream()

    def run(self, n):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.eng.forward_resident(self.iq, self.is_, self.ts, self.ms, self.edges, self.off, self.outs)
            self.stream.synchronize()


w = [Worker(1000), Worker(2000)]
for x in w:
    x.run(5)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter(); w[0].run(n); t1 = time.perf_counter()
print(f"one engine : {bs * n / (t1 - t0):8.1f} pairs/s  ({(t1 - t0) / n * 1e3:.3f} ms/step)")
th = [threading.Thread(target=x.run, args=(n,)) for x in w]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"two engines: {2 * bs * n / (t1 - t0):8.1f} pairs/s aggregate  ({(t1 - t0) / n * 1e3:.3f} ms per step pair)")
