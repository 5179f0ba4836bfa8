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


def _axis_taps(g, M):
    """Bicubic taps of one axis: output size g from M source rows with scale_factor (g + 0.1) / M.  Returns (idx [4, g], w [4, g])."""
    sf = float(g + 0.1) / M
    assert int(math.floor(M * sf)) == g
    inv = np.float32(1.0 / sf)  # ATen: scale = 1 / scale_factor when the factor is given
    dst = np.arange(g, dtype=np.float32)
    src = inv * (dst + np.float32(0.5)) - np.float32(0.5)   # cubic: no clamp of negative coordinates
    i0 = np.floor(src).astype(np.int64)
    t = (src - i0.astype(np.float32)).astype(np.float32)
    w = _cubic_coeffs(t)                                    # [4, g]
    idx = np.clip(i0[None, :] + np.arange(-1, 3)[:, None], 0, M - 1)   # [4, g] border-replicated taps
    return idx, w


def interpolate_pos_embed(pos_embed, g):
    """pos_embed [1, 1 + M*M, C] (or [1+M*M, C]) -> [1 + gh*gw, C] float32.  g: the token grid, an int (square) or (gh, gw) =
    (rows, columns) = (H // 14, W // 14): upstream scales the table's first spatial axis by (rows + 0.1) / M and the second by
    (columns + 0.1) / M (its `w` / `h` are x.shape[2] / x.shape[3], i.e. rows / columns)."""
    gh, gw = (g, g) if isinstance(g, (int, np.integer)) else (int(g[0]), int(g[1]))
    pe = np.asarray(pos_embed, np.float32)
    if pe.ndim == 3:
        pe = pe[0]
    N = pe.shape[0] - 1
    M = int(math.sqrt(N))
    assert M * M == N, "positional table must be 1 + M*M rows"
    if gh == M and gw == M:
        return pe.copy()
    C = pe.shape[1]
    grid = pe[1:].reshape(M, M, C)
    idx_y, w_y = _axis_taps(gh, M)
    idx_x, w_x = _axis_taps(gw, M)
    # columns (x) for every source row, then rows (y), accumulated tap by tap in fp32 like upsample_bicubic2d
    tmp = np.zeros((M, gw, C), np.float32)
    for k in range(4):
        tmp += grid[:, idx_x[k], :] * w_x[k][None, :, None]
    out = np.zeros((gh, gw, C), np.float32)
    for k in range(4):
        out += tmp[idx_y[k], :, :] * w_y[k][:, None, None]
    return np.concatenate([pe[:1], out.reshape(gh * gw, C)], 0).astype(np.float32)
