#!/usr/bin/env python
"""Cycle ledger of the 8-phase GEMM per output tile: one launch of the shipped fp16 instantiation with s_memtime
stamps (lab library, ec_gemm8.hip LAB 512: kernel start | per tile: K loop start, K loop end, epilogue end | kernel end; wave 0 of
each wave group, shader cycles) on the backbone's block-GEMM shapes.  Prints, per shape: prologue, K loop, epilogue (seam), tail as
mean / min / max over the workgroups, per tile index, and the launch span per XCD (stamps of different XCDs are not compared).
    python tools/g8_ledger.py [qkv fc1 proj fc2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from edgecape_amd import build

SHAPES = {"qkv": (20800, 2304, 768, 1), "fc1": (20800, 3072, 768, 3), "proj": (20800, 768, 768, 2), "fc2": (20800, 768, 3072, 2), "sq4096": (4096, 4096, 4096, 1)}


def main():
    path = os.environ.get("EC_LAB_LIB", build.LIB.replace(".so", "_lab.so"))
    lib = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    lib.ec_lab_gemm8_trace.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, C.POINTER(C.c_float)]
    lib.ec_last_error.restype = C.c_char_p
    for name in (sys.argv[1:] or ["qkv", "fc1", "proj", "fc2", "sq4096"]):
        M, N, K, kind = SHAPES[name]
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        if os.environ.get("ZERO"):
            A.zero_(); W.zero_()
        b = torch.randn(N, device="cuda")
        Cd = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        ntiles = ((M + 255) // 256) * ((N + 255) // 256)
        grid = min(ntiles, torch.cuda.get_device_properties(0).multi_processor_count)
        tr = torch.zeros(grid * 64, device="cuda", dtype=torch.int64)
        ms = C.c_float()
        rc = lib.ec_lab_gemm8_trace(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cd.data_ptr(), M, N, K, kind, tr.data_ptr(), None, C.byref(ms))
        assert rc == 0, lib.ec_last_error().decode()
        t = tr.cpu().numpy().reshape(grid, 2, 32).astype(np.int64)
        nk = K // 64
        span = float(np.mean(t[:, :, 31] - t[:, :, 0]))
        print(f"== {name}: traced launch {ms.value * 1e3:.1f} us between HIP events (incl. ~2-3 us of launch / event overhead); mean workgroup span "
              f"{span:.0f} ticks -> {span / (ms.value * 1e3):.0f} ticks per us")
        print(f"== {name}: M={M} N={N} K={K} ({ntiles} tiles of 256x256 on {grid} workgroups, {nk} K-tiles per tile; ideal MFMA time per tile "
              f"{nk * 8 * 16 * 16.1:.0f} cycles = {nk} K-tiles x 8 phase slots x 16 MFMAs x 16.1)")
        for g in range(2):
            S, E = t[:, g, 0], t[:, g, 31]
            ntile = np.array([int(sum(1 for i in range(9) if t[w, g, 1 + 3 * i] > 0)) for w in range(grid)])
            print(f" wave group {g}: tiles per workgroup {dict(zip(*np.unique(ntile, return_counts=True)))}; kernel start -> end: mean {np.mean(E - S):.0f} "
                  f"min {np.min(E - S):.0f} max {np.max(E - S):.0f} cycles")
            f = lambda x: f"{np.mean(x):8.0f} [{np.min(x):6.0f} .. {np.max(x):6.0f}]"
            full = ntile == ntile.max()
            print(f"   prologue (start -> first K loop):          {f(t[:, g, 1] - S)}")
            tot_k = tot_e = 0.0
            for i in range(int(ntile.max())):
                sel = ntile > i
                a, bq, c = t[sel, g, 1 + 3 * i], t[sel, g, 2 + 3 * i], t[sel, g, 3 + 3 * i]
                print(f"   tile {i}: K loop {f(bq - a)}   epilogue + seam barrier {f(c - bq)}   ({int(sel.sum())} workgroups)")
                tot_k += np.mean(bq - a); tot_e += np.mean(c - bq)
            last = np.array([t[w, g, 3 * ntile[w]] for w in range(grid)])
            print(f"   tail (last epilogue end -> kernel end):    {f(E - last)}")
            print(f"   sum over a {int(ntile.max())}-tile workgroup: prologue {np.mean(t[full, g, 1] - S[full]):.0f} + K loops {tot_k:.0f} + epilogues {tot_e:.0f} + tail {np.mean((E - last)[full]):.0f}"
                  f" = {np.mean(t[full, g, 1] - S[full]) + tot_k + tot_e + np.mean((E - last)[full]):.0f}")
        for x in range(min(8, grid)):
            sel = np.arange(grid) % 8 == x
            print(f"   XCD {x}: launch span (first start -> last end of its workgroups) {t[sel, :, 31].max() - t[sel, :, 0].min()} cycles, start skew {t[sel, 0, 0].max() - t[sel, 0, 0].min()}")


if __name__ == "__main__":
    main()
