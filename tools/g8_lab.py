#!/usr/bin/env python
"""Kernel lab for the 8-phase GEMM: times ablated instantiations (ec_gemm8.hip LAB bits) from libedgecape_hip_lab.so
(`python -m edgecape_amd.build --lab`, built with -DEC_G8_LAB; the shipped library has none of them).

    LAB bits: 1 no global stores | 2 every tile loads tile 0's operands | 4 no LDS-DMA in the steady state | 8 no MFMAs | 32 no epilogue
    SHAPES="name:M:N:K,..."  VARIANTS="0,1,2,..."  python tools/g8_lab.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgecape_amd import build

NAMES = {0: "full", 1: "nostore", 2: "alias", 3: "alias+nostore", 4: "noload", 5: "noload+nostore", 8: "nomfma", 9: "nomfma+nostore",
         32: "noepi", 34: "alias+noepi", 36: "noload+noepi", 40: "nomfma+noepi"}


def main():
    path = build.LIB.replace(".so", "_lab.so")
    if not os.path.exists(path):
        build.build_lab()
    lib = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    lib.ec_lab_gemm8.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    lib.ec_last_error.restype = C.c_char_p
    lib.ec_lab_gemm_nt.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    nt_cfgs = [int(v) for v in os.environ.get("NT", "").split(",") if v]
    g4_variants = [int(v) for v in os.environ.get("G4", "").split(",") if v]
    if g4_variants:   # the four-wave kernel of round 2 (git show 22a53bf:edgecape_amd/csrc/ec_gemm4.hip) is no longer in the lab library
        lib.ec_lab_gemm4.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float)]
    shapes = [("qkv", 20800, 2304, 768)]
    if os.environ.get("SHAPES"):
        shapes = [(n, int(m), int(nn), int(k)) for n, m, nn, k in (x.split(":") for x in os.environ["SHAPES"].split(","))]
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,8,32,34,36,40").split(",")]
    iters = int(os.environ.get("ITERS", 30))
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        if os.environ.get("ZERO"):          # zero-filled operands: the same instruction stream at a higher sustained clock (DVFS)
            A.zero_(); W.zero_()
        b = torch.randn(N, device="cuda")
        Cg = torch.zeros(M * N + 300 * N, device="cuda", dtype=torch.bfloat16)   # output + a guard region behind it (rows past M must never be stored)
        Cd = Cg[:M * N].view(M, N)
        for rep in range(int(os.environ.get("REPS", 2))):
            row = []
            for v in variants:
                ms = C.c_float()
                rc = lib.ec_lab_gemm8(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, v, iters, None, C.byref(ms))
                assert rc == 0, lib.ec_last_error().decode()
                row.append(f"{NAMES.get(v, v)} {ms.value * 1e3:.1f}us ({2.0 * M * N * K / (ms.value * 1e-3) / 1e12:.0f})")
            for v in g4_variants:   # the four-wave kernel (ec_gemm4.hip); variant 0 is checked against torch
                ms = C.c_float()
                if v in (0, 100, 200):
                    Cd.fill_(float("nan"))
                rc = lib.ec_lab_gemm4(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, v, iters, None, C.byref(ms))
                assert rc == 0, lib.ec_last_error().decode()
                tag = f"g4:{v} {ms.value * 1e3:.1f}us ({2.0 * M * N * K / (ms.value * 1e-3) / 1e12:.0f})"
                if v in (0, 100, 200) and rep == 0:
                    torch.cuda.synchronize()
                    rows = torch.cat([torch.arange(0, min(M, 512)), torch.arange(max(M - 700, 0), M)]).cuda()
                    ref = A[rows].float() @ W.float().t() + b
                    if v == 100:
                        ref = torch.nn.functional.gelu(ref)
                    if v == 200:
                        ref = ref * b
                    err = (Cd[rows].float() - ref).abs().max().item()
                    guard = Cg[M * N:].float().abs().max().item()
                    tag += f" err {err:.3g} guard {guard:.3g} nan {int(torch.isnan(Cd.float()).sum().item())}"
                row.append(tag)
            for c in nt_cfgs:
                ms = C.c_float()
                rc = lib.ec_lab_gemm_nt(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, c, iters, None, C.byref(ms))
                assert rc == 0, lib.ec_last_error().decode()
                row.append(f"nt{c} {ms.value * 1e3:.1f}us ({2.0 * M * N * K / (ms.value * 1e-3) / 1e12:.0f})")
            print(f"{name} M={M} N={N} K={K}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
