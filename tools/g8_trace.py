#!/usr/bin/env python
"""Decode /tmp/g8_trace.txt (EC_G8_TRACE=1 tools/gemm_bench.py bf16): per wave, per phase of one K-tile pair:
R = phase start -> before barrier 1, B1 = barrier 1 wait, M = MFMA block, B2 = barrier 2 wait (shader cycles)."""
import sys

rows = [list(map(int, l.split())) for l in open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/g8_trace.txt") if l.strip()]
for w, r in enumerate(rows):
    out = []
    for ph in range(8):
        b = (ph // 4) * 20 + (ph % 4) * 5
        t0, t1, t2, t3, t4 = r[b:b + 5]
        d = lambda a, c: (c - a) & 0xffffffff
        seg = ""
        if len(r) >= 56:   # finer stamps inside R: after the ds_reads, after the LDS-DMA issue (then the vmcnt wait up to t1)
            a, i = r[40 + (ph // 4) * 8 + (ph % 4) * 2], r[41 + (ph // 4) * 8 + (ph % 4) * 2]
            seg = f"[ds{d(t0,a):4d} dma{d(a,i):4d} vm{d(i,t1):4d}]"
        out.append(f"R{d(t0,t1):4d}{seg} B{d(t1,t2):4d} M{d(t2,t3):4d} B{d(t3,t4):4d}")
    tot = (r[39] - r[0]) & 0xffffffff
    print(f"wave {w} (group {w >> 2}): " + " | ".join(out) + f" | total {tot}")
