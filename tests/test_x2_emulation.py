"""CPU: the emulation that priced the fp16x2 scheme before any kernel existed (oracle/x2_at_scale.py) - its quantisers against the conversion
table measured on gfx950 (tools/fp8_mfma_probe.hip, profiles/r06_fp8_mfma_probe.txt), its weight planes, and the size of the scheme's error."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_quantisers_match_the_hardware_conversion_table():
    from oracle import x2_at_scale as X
    t = lambda *v: torch.tensor(v, dtype=torch.float32)
    # v_cvt_pk_bf8_f32 (e5m2), round to nearest even: 1.125 -> 1 (tie to even), 1.375 -> 1.5, 1.1875 -> 1.25, 3.3 -> 3.5, -7.7 -> -8, 100 -> 96;
    # subnormals: 2^-16 stays, 2^-17 -> 0 (tie to even), 1.5 * 2^-17 -> 2^-16
    assert X.q5(t(1.125, 1.375, 1.1875, 3.3, -7.7, 100.0)).tolist() == [1.0, 1.5, 1.25, 3.5, -8.0, 96.0]
    assert X.q5(t(2.0 ** -16, 2.0 ** -17, 1.5 * 2.0 ** -17)).tolist() == [2.0 ** -16, 0.0, 2.0 ** -16]
    # the hardware does NOT saturate (>= 61440 -> inf): the kernels clamp first (ec_common.h pack4_e5m2), and so does the emulation
    assert X.q5(t(60000.0, 65504.0, -1.0e6)).tolist() == [57344.0, 57344.0, -57344.0]
    # v_cvt_pk_fp8_f32 (e4m3): 1.125 exact, 1.0625 -> 1 (tie to even), 1.4375 -> 1.5, 464 -> 448, 0.3 -> 0.3125; 2^-9 stays, 2^-10 -> 0
    assert X.q4(t(1.125, 1.0625, 1.4375, 464.0, 0.3, 1000.0)).tolist() == [1.125, 1.0, 1.5, 448.0, 0.3125, 448.0]
    assert X.q4(t(2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10)).tolist() == [2.0 ** -9, 0.0, 2.0 ** -9]


def test_weight_planes_and_scheme_error():
    from oracle import x2_at_scale as X
    g = torch.Generator().manual_seed(0)
    W = torch.randn(256, 384, generator=g) / 384 ** 0.5
    wh, wh8, wl8, wl = X.w_planes(W)
    assert torch.equal(wh, W.half().float()) and torch.allclose(wl, W - wh)
    # one power-of-two scale per plane puts its largest magnitude into e4m3's top binade: the plane's relative error is e4m3's 2^-4
    assert float(((wh8 - W).abs() / W.abs().clamp_min(float(W.abs().max()) * 2.0 ** -13)).max()) <= 2.0 ** -4
    assert float((wl8 - wl).abs().max()) <= float(wl.abs().max()) * 2.0 ** -4
    A = torch.randn(512, 384, generator=g) * torch.exp(torch.randn(512, 1, generator=g))
    b = torch.zeros(256)
    X._wcache.clear()
    y = X.linear_x2(A, W, b, "fp16x2", ("test", 0, "w")).double()
    exact = A.double() @ W.double().T
    scale = A.double().abs() @ W.double().abs().T
    rel = (y - exact).abs() / scale
    y16 = (A.half().float() @ W.half().float().T).double()
    rel16 = (y16 - exact).abs() / scale
    # two MFMA units per product: ~2^-14 of |a|.|w| at worst, rms 1.3e-6 - 18 x below the single fp16 product's 2.3e-5
    assert float(rel.max()) < 2.0 ** -13 and float((rel ** 2).mean().sqrt()) < 2.0 ** -17
    assert float((rel ** 2).mean().sqrt()) * 12 < float((rel16 ** 2).mean().sqrt())
    for scheme in ("fp16x2e4", "fp16x25"):
        X._wcache.clear()
        r2 = ((X.linear_x2(A, W, b, scheme, ("test", 0, "w")).double() - exact).abs() / scale)
        assert float((r2 ** 2).mean().sqrt()) <= float((rel ** 2).mean().sqrt()) * 1.05     # the costlier variants are at least as accurate
