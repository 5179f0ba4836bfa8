#!/usr/bin/env python
"""Do two chains of small GEMMs on two HIP streams overlap on one MI355X?  Each thread runs ec_op_gemm_bench (a loop of
launches timed with HIP events on its own stream); ctypes releases the GIL, so the two loops are enqueued concurrently."""
import ctypes as C
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import _lib

lib = _lib.load()
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (3200, 256, 256))]
iters = 3000


def mk():
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    Cd = torch.empty(M, N, device="cuda")
    return A, W, b, Cd


def run(bufs, stream, out, i):
    A, W, b, Cd = bufs
    ms = C.c_float()
    _lib.check(lib.ec_op_gemm_bench(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, 2, iters,
                                    C.c_void_p(stream.cuda_stream), C.byref(ms)))
    out[i] = ms.value * 1e3


b0, b1 = mk(), mk()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
out = [0, 0]
run(b0, s0, out, 0)
print(f"solo     : {out[0]:.2f} us / launch")
t0 = threading.Thread(target=run, args=(b0, s0, out, 0))
t1 = threading.Thread(target=run, args=(b1, s1, out, 1))
t0.start(); t1.start(); t0.join(); t1.join()
print(f"two streams concurrently: {out[0]:.2f} / {out[1]:.2f} us / launch  (perfect overlap = solo, none = 2 x solo)")
