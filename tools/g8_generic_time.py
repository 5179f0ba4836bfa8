#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats -- python tools/g8_generic_time.py : the 8-phase GEMM with the GENERIC epilogue (fp32 output + residual)
on the shapes the head sends it (skeleton K|V / image-query projections) and on the patch embedding."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import _lib

lib = _lib.load()
for M, N, K in [(10368, 1024, 256), (10368, 512, 256), (20736, 768, 640)]:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda")
    Cd = torch.empty(M, N, device="cuda")
    for _ in range(10):
        _lib.check(lib.ec_op_linear(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, R.data_ptr(), Cd.data_ptr(), M, N, K, 0, 1, None))
torch.cuda.synchronize()
