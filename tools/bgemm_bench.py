#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats -- python tools/bgemm_bench.py : per-shape kernel time of ec_op_bgemm (head contractions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import _lib

lib = _lib.load()
for (b, M, N, K, tB) in [(32, 100, 768, 100, 0), (32, 100, 768, 324, 0), (32, 100, 100, 100, 0), (32, 100, 324, 256, 1), (32, 100, 384, 100, 0)]:
    A = torch.randn(b, M, K, device="cuda")
    B = torch.randn(b, N, K, device="cuda") if tB else torch.randn(b, K, N, device="cuda")
    C = torch.empty(b, M, N, device="cuda")
    for _ in range(10):
        _lib.check(lib.ec_op_bgemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), b, M, N, K, tB, None))
    torch.cuda.synchronize()
