"""Host-side (load-time) interpolation of the DINOv2 positional table.

Upstream `interpolate_pos_encoding` (facebookresearch/dinov2 vision_transformer.py):
bicubic (a = -0.75), align_corners=False, antialias=False, with an explicit
scale_factor = (g + 0.1) / M ("interpolate_offset" kludge), so the source coordinate of output i is
(i + 0.5) * M / (g + 0.1) - 0.5 — NOT the `size=` convention.  Done once per (weights, g) in numpy;
the HIP library only ever sees the already interpolated [1 + g*g, C] table (ec_set_pos_embed).
"""
import math

import numpy as np


def _cubic_coeffs(t, A=np.float32(-0.75)):
    t = t.astype(np.float32)
    one = np.float32(1)

    def c1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + one

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + one), c1(t), c1(one - t), c2(np.float32(2) - t)], 0).astype(np.float32)


def interpolate_pos_embed(pos_embed, g):
    """pos_embed [1, 1 + M*M, C] (or [1+M*M, C]) -> [1 + g*g, C] float32."""
    pe = np.asarray(pos_embed, np.float32)
    if pe.ndim == 3:
        pe = pe[0]
    N = pe.shape[0] - 1
    M = int(math.sqrt(N))
    assert M * M == N, "positional table must be 1 + M*M rows"
    if g == M:
        return pe.copy()
    C = pe.shape[1]
    grid = pe[1:].reshape(M, M, C)
    sf = float(g + 0.1) / M
    assert int(math.floor(M * sf)) == g
    inv = np.float32(1.0 / sf)  # ATen: scale = 1 / scale_factor when the factor is given
    dst = np.arange(g, dtype=np.float32)
    src = inv * (dst + np.float32(0.5)) - np.float32(0.5)   # cubic: no clamp of negative coordinates
    i0 = np.floor(src).astype(np.int64)
    t = (src - i0.astype(np.float32)).astype(np.float32)
    w = _cubic_coeffs(t)                                    # [4, g]
    idx = np.clip(i0[None, :] + np.arange(-1, 3)[:, None], 0, M - 1)   # [4, g] border-replicated taps
    # rows (y) then columns (x), accumulated tap by tap in fp32 like upsample_bicubic2d
    tmp = np.zeros((M, g, C), np.float32)                   # interpolate along x for every source row
    for k in range(4):
        tmp += grid[:, idx[k], :] * w[k][None, :, None]
    out = np.zeros((g, g, C), np.float32)
    for k in range(4):
        out += tmp[idx[k], :, :] * w[k][:, None, None]
    return np.concatenate([pe[:1], out.reshape(g * g, C)], 0).astype(np.float32)
