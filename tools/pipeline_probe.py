#!/usr/bin/env python
"""Probe (tools/, not product code): what does a TRUE two-stage software pipeline - head(i) on one stream beside backbone(i+1) on
another, backbones never beside each other - gain over the serial step, and how does the head's time scale with the batch?
Uses the existing entry points ec_backbone / ec_head of ONE handle (they share no workspace)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import _lib, synth
from edgecape_amd.engine import HipEngine

bs = int(os.environ.get("BS", 32))
MAXB = int(os.environ.get("MAXB", 64))
S, H, arch = 1, 256, "dinov2_vitb14"
sd = synth.make_weights(arch, seed=0)
eng = HipEngine(sd, arch=arch, image_size=H, max_batch=MAXB, max_shots=S, backbone_precision="fp16", head_precision="mixed")
lib = eng.lib
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()


def batch(n, seed):
    b = synth.make_pairs(n, S, H, seed=seed, fixed_n_kp=False)
    d = dict(iq=dev(b["img_q"]), is_=[dev(x) for x in b["img_s"]], ts=[dev(x) for x in b["target_s"]],
             ms=dev(b["target_weight_s"][0].reshape(n, -1)))
    d["edges"], d["off"] = eng._edges([m["sample_skeleton"][0] for m in b["img_metas"]], n)
    d["imgs"] = torch.cat([d["iq"]] + d["is_"], 0).contiguous()
    d["outs"] = eng._outputs(n)
    return d


def backbone(d, feat):
    _lib.check(lib.ec_backbone(eng.h, d["imgs"].data_ptr(), d["imgs"].shape[0], feat.data_ptr(), _lib.EC_LAYOUT_TOKENS, _lib.current_stream()))


def head(d, feat, n):
    fs = [feat[(1 + s) * n:(2 + s) * n] for s in range(S)]
    _lib.check(lib.ec_head(eng.h, feat.data_ptr(), eng._ptr_array(fs), _lib.EC_LAYOUT_TOKENS, eng._ptr_array(d["ts"]), d["ms"].data_ptr(),
                           d["edges"].ctypes.data, d["off"].ctypes.data, n, S, _lib.current_stream(), C.byref(d["outs"][1])))


def timeit(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


B = [batch(bs, 1000), batch(bs, 2000)]
feat = [torch.empty((1 + S) * bs, eng.HW, eng.C, device="cuda") for _ in range(2)]
ms_fwd = timeit(lambda: eng.forward_resident(B[0]["iq"], B[0]["is_"], B[0]["ts"], B[0]["ms"], B[0]["edges"], B[0]["off"], B[0]["outs"]))
ms_bb = timeit(lambda: backbone(B[0], feat[0]))
ms_hd = timeit(lambda: head(B[0], feat[0], bs))
ms_split = timeit(lambda: (backbone(B[0], feat[0]), head(B[0], feat[0], bs)))
print(f"bs={bs}: ec_forward {ms_fwd:.3f} ms | backbone {ms_bb:.3f} | head {ms_hd:.3f} | backbone+head one stream {ms_split:.3f}", flush=True)
ref = {k: v.clone() for k, v in B[1]["outs"][0].items()}
backbone(B[1], feat[1]); head(B[1], feat[1], bs); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in B[1]["outs"][0].items()}

# pipelined: stream A = backbones, stream B = heads
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("HEAD_PRIO", 0)))
ev_feat = [torch.cuda.Event() for _ in range(2)]
ev_head = [torch.cuda.Event() for _ in range(2)]
it = [0]


def pipe_step():
    i = it[0]; it[0] += 1
    k = i & 1
    with torch.cuda.stream(sa):
        sa.wait_event(ev_head[k])          # head(i-2) has consumed feat[k]
        backbone(B[k], feat[k])
        ev_feat[k].record(sa)
    with torch.cuda.stream(sb):
        sb.wait_event(ev_feat[k])
        head(B[k], feat[k], bs)
        ev_head[k].record(sb)


for e in ev_head:
    e.record(sb)
ms_pipe = timeit(pipe_step, n=40, warm=6)
same = all(torch.equal(ref[k], B[1]["outs"][0][k]) for k in ref)
print(f"pipelined (head(i) beside backbone(i+1)): {ms_pipe:.3f} ms/step = {bs / ms_pipe * 1e3:.0f} pairs/s vs serial {bs / ms_fwd * 1e3:.0f}  ({ms_fwd / ms_pipe - 1:+.1%}); outputs of batch 1 bit-equal to the serial run: {same}", flush=True)

# head time vs batch
for n in (8, 16, 32, 64):
    if n > MAXB:
        continue
    d = batch(n, 3000 + n)
    f = torch.empty((1 + S) * n, eng.HW, eng.C, device="cuda")
    backbone(d, f)
    t_h = timeit(lambda: head(d, f, n))
    t_b = timeit(lambda: backbone(d, f))
    print(f"bs={n}: head {t_h:.3f} ms ({t_h / n * 1e3:.1f} us/pair)  backbone {t_b:.3f} ms ({t_b / n * 1e3:.1f} us/pair)", flush=True)
