#!/usr/bin/env python
"""Time the backbone-shaped bf16 attention through the backbone entry point is noisy; instead launch ec_op_attention's
kernel repeatedly under rocprofv3 and read the per-kernel average:  rocprofv3 --kernel-trace --stats -- python tools/attn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import _lib

lib = _lib.load()
B, H, L, hd = 64, 12, 325, 64
q = torch.randn(B, L, H * hd, device="cuda")
k = torch.randn(B, L, H * hd, device="cuda")
v = torch.randn(B, L, H * hd, device="cuda")
o = torch.empty_like(q)
for _ in range(int(os.environ.get("ITERS", 20))):
    _lib.check(lib.ec_op_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), None, None, o.data_ptr(), B, H, L, L, hd, int(os.environ.get("PREC", 1)), None))
torch.cuda.synchronize()
