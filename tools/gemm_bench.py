#!/usr/bin/env python
"""Micro-benchmark of the persistent NT GEMM (ec_op_gemm_bench) on the backbone shapes + a 4096^3 reference."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import _lib

SHAPES = [("qkv", 20800, 2304, 768), ("proj", 20800, 768, 768), ("fc1", 20800, 3072, 768), ("fc2", 20800, 768, 3072),
          ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]


def main():
    lib = _lib.load()
    if os.environ.get("SHAPES"):   # e.g. SHAPES="a:7168:2304:768,b:14336:2304:768"
        SHAPES[:] = [(n, int(m), int(nn), int(k)) for n, m, nn, k in (x.split(":") for x in os.environ["SHAPES"].split(","))]
    prec = {"bf16": 1, "fp32": 0, "bf16x3": 2}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    iters = int(os.environ.get("ITERS", 20))
    only = os.environ.get("ONLY")
    for name, M, N, K in SHAPES:
        if only and name not in only.split(","):
            continue
        dt = torch.bfloat16 if prec == 1 else torch.float32
        A = torch.randn(M, K, device="cuda").to(dt)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
        b = torch.randn(N, device="cuda")
        Cd = torch.empty(M, N, device="cuda", dtype=dt)
        ms = C.c_float()
        _lib.check(lib.ec_op_gemm_bench(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, prec, iters, None, C.byref(ms)))
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        err = float("nan")
        if not os.environ.get("NOCHECK"):
            ref = (A.float() @ W.float().T + b)
            err = (Cd.float() - ref).abs().max().item()
        print(f"{name:8s} M={M} N={N} K={K} {['fp32', 'bf16', 'bf16x3'][prec]}: {ms.value * 1e3:8.1f} us  {tf:7.1f} TFLOP/s  max|err|={err:.3g}", flush=True)


if __name__ == "__main__":
    main()
