#!/usr/bin/env python
"""Coefficients of the single-transcendental GELU of the 8-phase GEMM's epilogue (ec_gemm8.hip gelu_fast8, round 4):
    gelu(x) = max(x, 0) - |x| 2^p(|x|),   p(a) ~ L(a) = log2 Phi(-a) on a >= 0.
Weighted minimax fit (Lawson iteration; weight a Phi(-a) ln 2 = the sensitivity of the result to L) of a degree-d polynomial in a on
[0, A]; prints the monomial coefficients, the fit error and the error of the fp32 evaluation on |x| <= 12 against the erf form.
    python tools/gelu_fit.py [degree ...]        (CPU only: numpy + scipy)"""
import sys
from math import comb

import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf, log_ndtr


def fit(deg, A=6.0, iters=200, n=20001):
    a = np.linspace(0, A, n)
    L = log_ndtr(-a) / np.log(2.0)
    h = a * np.exp(log_ndtr(-a))
    w0 = h * np.log(2.0) + 1e-9
    V = Ch.chebvander(2 * a / A - 1, deg)
    w = w0.copy()
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V * w[:, None], L * w, rcond=None)
        e = np.abs(V @ c - L) * w0
        w = w * (1 + 3 * e / e.max())
        w /= w.max()
    ct = Ch.cheb2poly(c)
    mono = np.zeros(deg + 1)
    for k, ck in enumerate(ct):
        for j in range(k + 1):
            mono[j] += ck * comb(k, j) * (2 / A) ** j * (-1) ** (k - j)
    return mono, float(np.abs(a * np.exp2(V @ c) - h).max())


def check(mono):
    m32 = mono.astype(np.float32)
    x = np.linspace(-12, 12, 2000001).astype(np.float32)
    ax = np.abs(x)
    q = np.full_like(x, m32[-1])
    for k in range(len(m32) - 2, -1, -1):
        q = (q * ax + m32[k]).astype(np.float32)
    y = np.maximum(x, 0) - ax * np.exp2(q)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    return float(np.abs(y - ref).max())


for deg in ([int(v) for v in sys.argv[1:]] or [3, 5]):
    mono, e = fit(deg)
    print(f"degree {deg}: fit error {e:.3g}, fp32 evaluation error on |x| <= 12 {check(mono):.3g}, leading coefficient {mono[-1]:.3g} "
          f"({'runs to -inf: no clamp needed' if mono[-1] < 0 else 'POSITIVE: clamp |x|'})")
    print("   high -> low: " + ", ".join(f"{v:.9e}f" for v in mono[::-1]))
