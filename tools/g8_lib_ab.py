#!/usr/bin/env python
"""Interleaved A/B of the shipped 8-phase GEMM instantiations between TWO lab libraries (e.g. before / after a source change):
    python tools/g8_lib_ab.py old.so new.so      (ROUNDS=10 ITERS=20 SHAPES=qkv,proj,fc1,fc2)"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import g8_sched as G

libs = []
for path in sys.argv[1:3]:
    lib = C.CDLL(os.path.abspath(path))
    vp, ci = C.c_void_p, C.c_int
    lib.ec_lab_gemm8.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    lib.ec_last_error.restype = C.c_char_p
    libs.append((os.path.basename(path), lib))
rounds, iters = int(os.environ.get("ROUNDS", 10)), int(os.environ.get("ITERS", 20))
want = os.environ.get("SHAPES", "qkv,proj,fc1,fc2").split(",")
for name, M, N, K, code in G.SHAPES:
    if name not in want:
        continue
    ops = G.operands(M, N, K)
    res = {n: [] for n, _ in libs}
    for r in range(rounds):
        for n, lib in (libs if r % 2 == 0 else libs[::-1]):
            res[n].append(G.run(lib, ops, M, N, K, code, 256, iters))
    base = statistics.median(res[libs[0][0]])
    for n, v in res.items():
        md = statistics.median(v)
        print(f"{name} {n}: median {md:.1f} us ({2.0 * M * N * K / md / 1e6:.0f} TFLOP/s, {md / base - 1:+.1%})  min {min(v):.1f}  max {max(v):.1f}", flush=True)
