#!/usr/bin/env python
"""Schedule sweep for the 8-phase GEMM (lab library, `python -m edgecape_amd.build --lab`) on the four block-GEMM shapes of the
backbone (cfg2: M = 20800): lab instantiations (GELU epilogue variants) and persistent-grid sizes of the SHIPPED kernels.

    python tools/g8_sched.py sweep  [--out gpurun_out/g8s]        time every configuration, write the table and a manifest
    python tools/g8_sched.py replay --manifest <file> [--iters 3] run exactly the manifest's configurations, `iters` launches each, in
                                                                  order (meant to run under rocprofv3 --pmc; tools/g8_sched_pmc.py maps
                                                                  the dispatch sequence back to the configurations)

Round-3 record (profiles/r03_g8_sched_sweep.txt, r03_g8_sched_pmc.csv) was taken with two more knobs in the kernel, since removed
because they lost: a per-tile rotation of the K walk (rot_n / rot_m K-tiles per tile column / row) and a start stagger over the
workgroup slots; the `krot` / `stagger` fields of the manifest are kept (0) so that the PMC mapper reads old and new manifests.
"""
import argparse
import ctypes as C
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import build

# name, M, N, K, lab code of the shipped fp16 instantiation (ec_gemm8.hip ec_lab_gemm8)
SHAPES = [("qkv", 20800, 2304, 768, 1000), ("proj", 20800, 768, 768, 2000), ("fc1", 20800, 3072, 768, 3000), ("fc2", 20800, 768, 3072, 2000)]
# extra lab instantiations timed beside the shipped one (ec_lab_gemm8 codes)
VARIANTS = {}   # e.g. {"fc1": [(code, "label")]} for lab instantiations added to ec_lab_gemm8
GRID_SIZES = (256, 248, 246, 240, 224, 192)


def load():
    path = build.LIB.replace(".so", "_lab.so")
    if not os.path.exists(path):
        build.build_lab()
    lib = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    lib.ec_lab_gemm8.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    lib.ec_last_error.restype = C.c_char_p
    return lib


def operands(M, N, K):
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    Cd = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    return A, W, b, Cd


def run(lib, ops, M, N, K, code, grid, iters):
    os.environ["EC_G8_GRID"] = str(grid)
    A, W, b, Cd = ops
    ms = C.c_float()
    rc = lib.ec_lab_gemm8(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, code, iters, None, C.byref(ms))
    assert rc == 0, lib.ec_last_error().decode()
    return ms.value * 1e3


def check(lib, ops, M, N, K, code):
    """max |C - torch reference| on the first and last 300 rows (fp16 output)."""
    A, W, b, Cd = ops
    Cd.zero_()
    run(lib, ops, M, N, K, code, 0, 1)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)]).cuda()
    ref = A[rows].float() @ W.float().t() + b
    if code == 2000:
        ref = ref * b
    if code >= 3000:
        ref = torch.nn.functional.gelu(ref)
    return (Cd[rows].float() - ref).abs().max().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["sweep", "replay"])
    ap.add_argument("--out", default="gpurun_out/g8s")
    ap.add_argument("--manifest")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--shapes", default="qkv,proj,fc1,fc2")
    args = ap.parse_args()
    lib = load()
    if args.mode == "replay":
        man = json.load(open(args.manifest))
        for e in man["configs"]:
            ops = operands(e["M"], e["N"], e["K"])
            run(lib, ops, e["M"], e["N"], e["K"], e["code"], e.get("grid", 0), args.iters - 1)   # (+1 warm-up launch inside)
            torch.cuda.synchronize()
        return
    os.makedirs(args.out, exist_ok=True)
    manifest = []
    lines = []
    for name, M, N, K, code in SHAPES:
        if name not in args.shapes.split(","):
            continue
        ops = operands(M, N, K)
        fl = 2.0 * M * N * K
        n0 = len(lines)
        base = None
        for c, label in [(code, "shipped")] + VARIANTS.get(name, []):
            v = [run(lib, ops, M, N, K, c, 0, args.iters) for _ in range(args.reps)]
            base = base or min(v)
            lines.append(f"{name} {label} (code {c}): " + " ".join(f"{x:.1f}" for x in v) + f" us  best {fl / min(v) / 1e6:.0f} TFLOP/s  ({min(v) / base - 1:+.1%})  max |err| {check(lib, ops, M, N, K, c):.3g}")
            manifest.append(dict(shape=name, M=M, N=N, K=K, code=c, krot=0, stagger=0, us=min(v)))
        for grid in GRID_SIZES:
            us = min(run(lib, ops, M, N, K, code, grid, args.iters) for _ in range(2))
            lines.append(f"{name} grid={grid}: {us:.1f} us ({us / base - 1:+.1%} vs base)")
            if grid in (248, 246) and name in ("proj", "fc2"):
                manifest.append(dict(shape=name, M=M, N=N, K=K, code=code, krot=0, stagger=0, grid=grid, us=us))
        print("\n".join(lines[n0:]), flush=True)
    with open(os.path.join(args.out, "sweep.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(args.out, "manifest.json"), "w") as f:
        json.dump(dict(configs=manifest), f, indent=1)


if __name__ == "__main__":
    main()
