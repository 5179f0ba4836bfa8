#!/usr/bin/env python
"""Interleaved A/B timing of lab configurations of the 8-phase GEMM (lab library): the box's clock wanders by several percent between
back-to-back measurements, so every configuration is measured ROUNDS times in alternation and reported as median / min.
    CONFIGS="name:code:grid,..." SHAPES=qkv,fc1 ROUNDS=10 ITERS=20 python tools/g8_ab.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import g8_sched as G

lib = G.load()
rounds, iters = int(os.environ.get("ROUNDS", 10)), int(os.environ.get("ITERS", 20))
want = os.environ.get("SHAPES", "qkv,proj,fc1,fc2").split(",")
for name, M, N, K, code in G.SHAPES:
    if name not in want:
        continue
    cfgs = [("g256", code, 256), ("g246", code, 246)]
    if os.environ.get("CONFIGS"):
        cfgs = [(a, int(b) if int(b) else code, int(c)) for a, b, c in (x.split(":") for x in os.environ["CONFIGS"].split(","))]
    ops = G.operands(M, N, K)
    res = {c[0]: [] for c in cfgs}
    for r in range(rounds):
        order = cfgs if r % 2 == 0 else cfgs[::-1]
        for cname, ccode, grid in order:
            res[cname].append(G.run(lib, ops, M, N, K, ccode, grid, iters))
    fl = 2.0 * M * N * K
    base = statistics.median(res[cfgs[0][0]])
    for cname, v in res.items():
        md = statistics.median(v)
        print(f"{name} {cname}: median {md:.1f} us ({fl / md / 1e6:.0f} TFLOP/s, {md / base - 1:+.1%})  min {min(v):.1f}  max {max(v):.1f}", flush=True)
