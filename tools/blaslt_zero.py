#!/usr/bin/env python
"""hipBLASLt (torch.nn.functional.linear) on the QKV shape with random and with zero-filled fp16 / bf16 operands: the vendor library's
rate on the same box, and how much of it the chip's power management takes back on random data (compare tools/g8_lab.py ZERO=1)."""
import torch

M, N, K = 20800, 2304, 768
for dt in (torch.bfloat16, torch.float16):
    for zero in (False, True):
        A = torch.randn(M, K, device="cuda").to(dt)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
        b = torch.randn(N, device="cuda").to(dt)
        if zero:
            A.zero_(); W.zero_()
        for _ in range(5):
            torch.nn.functional.linear(A, W, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            torch.nn.functional.linear(A, W, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"hipBLASLt {str(dt)[6:]:9s} {'zero' if zero else 'random':6s}: {us:6.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s", flush=True)
