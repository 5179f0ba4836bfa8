#!/usr/bin/env python
"""Lab timing of the fp16x2 block GEMMs (lab library only: python -m edgecape_amd.build --lab): the shipped kernels against the SAME kernels
with the FP8 K-tiles' MFMA blocks run as fp16 MFMAs on the same bytes (EC_X2_LAB_AS16=1: wrong numbers, identical load stream, LDS traffic and
epilogue) - does an FP8 K-tile cost more than an fp16 one because of the FP8 MFMA itself?  One process per variant (the switch is read once),
interleaved rounds on one box:
    python tools/x2_lab.py            # driver: spawns the rounds, prints medians
    python tools/x2_lab.py --child    # one process: times the four cfg2 shapes, prints one JSON line
"""
import ctypes as C
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("qkv", 20800, 2304, 768, 0), ("proj", 20800, 768, 768, 1), ("fc1", 20800, 3072, 768, 2), ("fc2", 20800, 768, 3072, 1)]   # kind: 0 fp32, 1 residual, 2 gelu


def child():
    import torch
    from edgecape_amd import build
    lab = build.LIB.replace(".so", "_lab.so")
    if not os.path.exists(lab):
        build.build_lab(verbose=False)
    lib = C.CDLL(lab)
    vp, ci = C.c_void_p, C.c_int
    lib.ec_op_linear_x2.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    lib.ec_last_error.restype = C.c_char_p
    out = {}
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, kind in SHAPES:
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        gam = torch.rand(N, generator=g).cuda() if kind == 1 else None
        Cb = torch.zeros(M, N, device="cuda") if kind != 2 else None
        P = torch.empty(M * 4 * N, dtype=torch.uint8, device="cuda") if kind == 2 else None
        ms = C.c_float()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for reps in (3, 30):
            rc = lib.ec_op_linear_x2(p(A), p(W), p(b), p(gam), p(Cb), p(P), M, N, K, 2 if kind == 2 else 0, reps, None, C.byref(ms))
            assert rc == 0, lib.ec_last_error().decode()
        out[name] = ms.value * 1e3
    print(json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    rounds = int(os.environ.get("ROUNDS", 4))
    res = {"fp8": [], "as16": []}
    for r in range(rounds):
        for tag, env in (("fp8", {}), ("as16", {"EC_X2_LAB_AS16": "1"})):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=dict(os.environ, **env))
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(o.stdout[-500:], o.stderr[-1500:])
                continue
            res[tag].append(json.loads(line[-1]))
    print("fp16x2 block GEMMs at cfg2 shapes (M = 20800), us per launch, median of %d interleaved processes; FP8 K-tiles as shipped | their MFMA blocks as fp16 MFMAs" % rounds)
    for name, M, N, K, _ in SHAPES:
        a = statistics.median(x[name] for x in res["fp8"])
        b = statistics.median(x[name] for x in res["as16"])
        units = 2 * 2.0 * M * N * K
        print(f"{name:5s} shipped {a:7.1f} us ({units / a / 1e6:6.0f} TFLOP/s in 16-bit-MFMA units)   FP8 tiles as fp16 MFMAs {b:7.1f} us   shipped / as16 = {a / b:.3f}")


if __name__ == "__main__":
    main()
